/*
  accelerate_hip.c — the MagickCore side of the MI355X accelerate backend.

  This file REPLACES MagickCore/accelerate.c in a MagickCore build whose
  OpenCL call sites are switched on (-DMAGICKCORE_HAVE_CL_CL_H=1 =>
  MAGICKCORE_OPENCL_SUPPORT, MagickCore/studio.h:149-154).  It defines the
  fourteen Accelerate*Image() entry points MagickCore/accelerate-private.h:36-60
  declares; the operators call them first and fall back to their CPU code when
  NULL / MagickFalse comes back (e.g. BlurImage, MagickCore/effect.c:783-787;
  ResizeImage, resize.c:3818-3826; EqualizeImage, enhance.c:2072-2075).
  Callers of MagickCore / MagickWand are unchanged.

  The work itself happens behind the C ABI of libmagickhip.so
  (include/magickhip.h), loaded with dlopen on first use (opencl_hip.c): without
  the library, without a GPU, or with the enable switch off every function here
  returns NULL / MagickFalse and the CPU path runs.

  Images stay on the device between chained operators: an operator input is
  uploaded once and its device copy is remembered in CacheInfo::opencl; a result
  is produced on the device and NOT downloaded — the host block is brought up to
  date by the cache hooks in opencl_hip.c the first time the CPU accesses the
  pixels (the reference's own lazy-sync protocol, cache.c:5341-5353).

  Nothing of the reference's OpenCL implementation is used: no cl_mem, no
  kernels-as-strings, no OpenCL runtime.
*/
#include "MagickCore/studio.h"
#include "MagickCore/accelerate-private.h"
#include "MagickCore/cache.h"
#include "MagickCore/cache-private.h"
#include "MagickCore/exception.h"
#include "MagickCore/exception-private.h"
#include "MagickCore/image.h"
#include "MagickCore/image-private.h"
#include "MagickCore/pixel-accessor.h"
#include "MagickCore/resize.h"
#include "MagickCore/resize-private.h"
#include "MagickCore/semaphore.h"
#include "MagickCore/string_.h"
#include "MagickCore/artifact.h"
#include "MagickCore/colorspace.h"
#include "MagickCore/colorspace-private.h"
#include "MagickCore/composite.h"
#include "MagickCore/morphology.h"

#if defined(MAGICKCORE_OPENCL_SUPPORT)

#include <stdlib.h>
#include "MagickCore/memory_.h"
#include "MagickCore/opencl-private.h"
#include "magickhip_shim.h"

static size_t hip_accelerated_calls=0;        /* read by tests through GetMagickHipAcceleratedCalls */

/* operators run concurrently on different images: the counter is atomic */
#define CountAcceleratedCall() ((void) __atomic_fetch_add(&hip_accelerated_calls,1,__ATOMIC_RELAXED))

MagickExport size_t GetMagickHipAcceleratedCalls(void)
{
  return(__atomic_load_n(&hip_accelerated_calls,__ATOMIC_RELAXED));
}

/*
  `-debug accelerate` (AccelerateEvent, MagickCore/log.h:37-57): which path an operator took.
  Accepted calls name the operator and the image; declined calls say why — the gate that failed
  (the conditions of accelerate.c:110-170), or that the library / the device / the operator
  itself returned "not handled" — and MagickCore then runs its CPU path.
*/
static const char *DescribeGate(const Image *image)
{
  if (image == (const Image *) NULL)
    return("no image");
  if (image->storage_class != DirectClass)
    return("PseudoClass image");
  if ((GetImageArtifact(image,"convolve:bias") != (const char *) NULL) ||
      (GetImageArtifact(image,"convolve:scale") != (const char *) NULL) ||
      (GetImageArtifact(image,"morphology:compose") != (const char *) NULL) ||
      (GetImageArtifact(image,"morphology:showKernel") != (const char *) NULL))
    return("a convolve: / morphology: artifact is set (Blur, UnsharpMask and Convolve hooks decline; "
      "MorphologyApply honours bias, scale and the compose operators None, Lighten, Difference)");
  switch (GetImageVirtualPixelMethod(image))
  {
    case UndefinedVirtualPixelMethod:
    case EdgeVirtualPixelMethod:
      break;
    default:
      return("virtual pixel method other than Undefined / Edge");
  }
  if ((image->channels & (ReadMaskChannel | WriteMaskChannel | CompositeMaskChannel)) != 0)
    return("read, write or composite mask");
  if ((image->number_channels < 1) || (image->number_channels > 4))
    return("more than four channels");
  switch (image->colorspace)
  {
    case RGBColorspace:
    case sRGBColorspace:
    case GRAYColorspace:
    case LinearGRAYColorspace:
      break;
    default:
      return("colourspace outside sRGB / RGB / GRAY / LinearGRAY (or channel layout, artifacts, device, operator arguments)");
  }
  return("channel layout, artifacts, device or operator arguments outside the backend's reach");
}

static void LogAccelerated(const char *function,const Image *image)
{
  if (IsEventLogging() != MagickFalse)
    (void) LogMagickEvent(AccelerateEvent,GetMagickModule(),
      "%s: accelerated on the HIP backend (%.20gx%.20g, %.20g channels, %s)",function,
      (double) image->columns,(double) image->rows,(double) image->number_channels,
      image->filename);
}

static void LogDeclined(const char *function,const int line,const Image *image)
{
  if (IsEventLogging() != MagickFalse)
    (void) LogMagickEvent(AccelerateEvent,GetMagickModule(),
      "%s: not accelerated, the CPU path runs (shim line %d: %s)",function,line,DescribeGate(image));
}

#define HipAccepted(image) (CountAcceleratedCall(),LogAccelerated(__func__,(image)))
#define HipDeclined(image,value) (LogDeclined(__func__,__LINE__,(image)),(value))

/* ------------------------------------------------------------------ gates */
/*
  What the backend can take, the same conditions the reference's accelerate
  layer imposes (accelerate.c:110-170): DirectClass; sRGB / RGB / GRAY /
  LinearGRAY; Undefined or Edge virtual pixels; no read, write or composite
  mask; at most four channels laid out R[,G,B][,A].
*/
static MagickBooleanType IsLayoutAcceleratable(const Image *image);
static MagickBooleanType IsHistogramOperatorAcceleratable(const Image *image);

static MagickBooleanType IsImageAcceleratable(const Image *image)
{
  switch (image->colorspace)
  {
    case RGBColorspace:
    case sRGBColorspace:
    case GRAYColorspace:
    case LinearGRAYColorspace:
      break;
    default:
      return(MagickFalse);
  }
  return(IsLayoutAcceleratable(image));
}

/*
  The gate without the colourspace condition, for the colourspace transform itself and
  for the histogram operators on its result ("the colourspace gate must be
  relaxed for our Lab/linear hooks"; BASELINE config C4 is sRGB->Lab + ContrastStretch).
*/
static MagickBooleanType IsLayoutAcceleratable(const Image *image)
{
  if (image->storage_class != DirectClass)
    return(MagickFalse);
  switch (GetImageVirtualPixelMethod(image))
  {
    case UndefinedVirtualPixelMethod:
    case EdgeVirtualPixelMethod:
      break;
    default:
      return(MagickFalse);
  }
  if ((image->channels & (ReadMaskChannel | WriteMaskChannel | CompositeMaskChannel)) != 0)
    return(MagickFalse);
  if ((image->number_channels < 1) || (image->number_channels > 4))
    return(MagickFalse);
  if (GetPixelChannelOffset(image,RedPixelChannel) != 0)
    return(MagickFalse);
  if ((image->number_channels == 2) || (image->number_channels == 4))
    if (GetPixelChannelOffset(image,AlphaPixelChannel) != (ssize_t) image->number_channels-1)
      return(MagickFalse);
  if (image->number_channels >= 3)
    if ((GetPixelChannelOffset(image,GreenPixelChannel) != 1) ||
        (GetPixelChannelOffset(image,BluePixelChannel) != 2))
      return(MagickFalse);
  return(MagickTrue);
}

/*
  The pixel cache of an image as a private, materialised memory cache (what
  GetAuthenticOpenCLBuffer requires, cache.c:1259-1291).
*/
static CacheInfo *AcquireHeapCache(const Image *image,ExceptionInfo *exception)
{
  CacheInfo
    *cache_info;

  cache_info=(CacheInfo *) image->cache;
  if ((cache_info->type == UndefinedCache) || (cache_info->reference_count > 1))
    {
      if (SyncImagePixelCache((Image *) image,exception) == MagickFalse)
        return((CacheInfo *) NULL);
      cache_info=(CacheInfo *) image->cache;
    }
  if ((cache_info->type != MemoryCache) || (cache_info->mapped != MagickFalse) ||
      (cache_info->pixels == (Quantum *) NULL))
    return((CacheInfo *) NULL);
  return(cache_info);
}

/*
  The device copy of an image's pixels.  upload != 0: an operator input — reuse the
  resident copy, or allocate one and upload the host block (once).  upload == 0: an
  operator result — allocate only; the host block is stale until the cache hooks
  download it (record marked dirty).
*/
static void *AcquireDevicePixels(HipLibrary *library,const Image *image,const int upload,
  ExceptionInfo *exception)
{
  CacheInfo
    *cache_info;

  MagickCLCacheInfo
    info;

  void
    *device_pixels;

  cache_info=AcquireHeapCache(image,exception);
  if (cache_info == (CacheInfo *) NULL)
    return(NULL);
  LockSemaphoreInfo(cache_info->semaphore);
  info=cache_info->opencl;
  if (info != (MagickCLCacheInfo) NULL)
    {
      UnlockSemaphoreInfo(cache_info->semaphore);
      return((void *) info->buffer);            /* resident: no transfer */
    }
  device_pixels=NULL;
  if (library->DeviceAlloc(-1,(size_t) cache_info->length,&device_pixels) != MH_OK)
    {
      UnlockSemaphoreInfo(cache_info->semaphore);
      return(NULL);
    }
  if (upload != 0)
    {
      if (library->Upload(-1,device_pixels,cache_info->pixels,(size_t) cache_info->length,
            NULL) != MH_OK)
        {
          (void) library->DeviceFree(-1,device_pixels);
          UnlockSemaphoreInfo(cache_info->semaphore);
          return(NULL);
        }
      CountHipTransfer(1);
    }
  info=(MagickCLCacheInfo) AcquireCriticalMemory(sizeof(*info));
  (void) memset(info,0,sizeof(*info));
  info->buffer=(cl_mem) device_pixels;
  info->pixels=cache_info->pixels;
  info->length=cache_info->length;
  info->event_count=upload != 0 ? 0U : 1U;      /* dirty: the device copy is the newer one */
  cache_info->opencl=info;
  UnlockSemaphoreInfo(cache_info->semaphore);
  return(device_pixels);
}

static void MarkDeviceCopyNewer(const Image *image)
{
  CacheInfo *cache_info=(CacheInfo *) image->cache;
  /* the cache hooks read and clear the flag under the same semaphore (CopyOpenCLBuffer) */
  LockSemaphoreInfo(cache_info->semaphore);
  if (cache_info->opencl != (MagickCLCacheInfo) NULL)
    cache_info->opencl->event_count=1U;
  UnlockSemaphoreInfo(cache_info->semaphore);
}

static MagickBooleanType DescribeImage(HipLibrary *library,const Image *image,
  void *device_pixels,MhImage *description)
{
  ssize_t
    i;

#if (MAGICKCORE_QUANTUM_DEPTH != 16)
  return(MagickFalse);
#endif
  library->InitImage(description,device_pixels,image->columns,image->rows,
    (uint32_t) image->number_channels,image->alpha_trait != UndefinedPixelTrait ? 1 : 0,
#if defined(MAGICKCORE_HDRI_SUPPORT)
    MH_QUANTUM_F32,
#else
    MH_QUANTUM_U16,
#endif
    MH_MEMORY_DEVICE);
  description->device=(-1);
  description->stream=NULL;                    /* the device's null stream, like the transfers */
  for (i=0; i < (ssize_t) image->number_channels; i++)
  {
    PixelChannel channel = GetPixelChannelChannel(image,i);
    description->channel_traits[i]=(uint32_t) GetPixelChannelTraits(image,channel);
  }
  description->alpha_offset=(-1);
  if (image->alpha_trait != UndefinedPixelTrait)
    description->alpha_offset=(int32_t) GetPixelChannelOffset(image,AlphaPixelChannel);
  description->alpha_trait=(uint32_t) image->alpha_trait;
  description->colorspace=(uint32_t) image->colorspace;
  description->intensity=(uint32_t) image->intensity;
  description->channel_mask=(uint32_t) image->channel_mask;
  return(MagickTrue);
}

/* A new image of the given size whose pixels live on the device (host block allocated, stale). */
static Image *AcquireResultImage(HipLibrary *library,const Image *image,const size_t columns,
  const size_t rows,void **device_pixels,ExceptionInfo *exception)
{
  Image
    *result;

  result=CloneImage(image,columns,rows,MagickTrue,exception);
  if (result == (Image *) NULL)
    return((Image *) NULL);
  if (SetImageStorageClass(result,DirectClass,exception) == MagickFalse)
    return(HipDeclined(image,DestroyImage(result)));
  *device_pixels=AcquireDevicePixels(library,result,0,exception);
  if (*device_pixels == NULL)
    return(HipDeclined(image,DestroyImage(result)));
  return(result);
}

/*
  BlurImage and UnsharpMaskImage reach MorphologyImage on the CPU path (effect.c:1170 ->
  morphology.c:4164-4206), which honours these artifacts: convolve:bias and convolve:scale
  change the kernel, morphology:compose the way the two kernels' results combine,
  morphology:showKernel prints it.  The Accelerate* entry points sit in front of that code, so
  with any of them set they decline; the CPU BlurImage then arrives at the MorphologyApply hook
  with the scaled kernel and the bias, and is accelerated there.
*/
static MagickBooleanType HasMorphologyArtifacts(const Image *image)
{
  if ((GetImageArtifact(image,"convolve:bias") != (const char *) NULL) ||
      (GetImageArtifact(image,"convolve:scale") != (const char *) NULL) ||
      (GetImageArtifact(image,"morphology:compose") != (const char *) NULL) ||
      (GetImageArtifact(image,"morphology:showKernel") != (const char *) NULL))
    return(MagickTrue);
  return(MagickFalse);
}

/* ------------------------------------------------------------- operators */
MagickPrivate Image *AccelerateBlurImage(const Image *image,const double radius,
  const double sigma,ExceptionInfo *exception)
{
  HipLibrary
    *library;

  Image
    *blur_image;

  MhImage
    source,
    destination;

  void
    *p,
    *q;

  assert(image != NULL);
  assert(exception != (ExceptionInfo *) NULL);
  if ((IsImageAcceleratable(image) == MagickFalse) || (HasMorphologyArtifacts(image) != MagickFalse))
    return(HipDeclined(image,(Image *) NULL));
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  p=AcquireDevicePixels(library,image,1,exception);
  if (p == NULL)
    return(HipDeclined(image,(Image *) NULL));
  blur_image=AcquireResultImage(library,image,image->columns,image->rows,&q,exception);
  if (blur_image == (Image *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  if ((DescribeImage(library,image,p,&source) == MagickFalse) ||
      (DescribeImage(library,blur_image,q,&destination) == MagickFalse) ||
      (library->BlurImage(&source,&destination,radius,sigma) != MH_OK))
    return(HipDeclined(image,DestroyImage(blur_image)));
  blur_image->type=image->type;      /* as MorphologyPrimitive does, morphology.c:2800 */
  HipAccepted(image);
  return(blur_image);
}

MagickPrivate Image *AccelerateUnsharpMaskImage(const Image *image,
  const double radius,const double sigma,const double gain,const double threshold,
  ExceptionInfo *exception)
{
  HipLibrary
    *library;

  Image
    *unsharp_image;

  MhImage
    source,
    destination;

  void
    *p,
    *q;

  if ((IsImageAcceleratable(image) == MagickFalse) || (HasMorphologyArtifacts(image) != MagickFalse))
    return(HipDeclined(image,(Image *) NULL));
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  p=AcquireDevicePixels(library,image,1,exception);
  if (p == NULL)
    return(HipDeclined(image,(Image *) NULL));
  unsharp_image=AcquireResultImage(library,image,image->columns,image->rows,&q,exception);
  if (unsharp_image == (Image *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  if ((DescribeImage(library,image,p,&source) == MagickFalse) ||
      (DescribeImage(library,unsharp_image,q,&destination) == MagickFalse) ||
      (library->UnsharpMaskImage(&source,&destination,radius,sigma,gain,threshold) != MH_OK))
    return(HipDeclined(image,DestroyImage(unsharp_image)));
  unsharp_image->type=image->type;   /* effect.c:4385 */
  HipAccepted(image);
  return(unsharp_image);
}

static double ReferenceFilterWeight(void *user,double x)
{
  return(GetResizeFilterWeight((const ResizeFilter *) user,x));     /* resize.c:1690 */
}

MagickPrivate Image *AccelerateResizeImage(const Image *image,
  const size_t resizedColumns,const size_t resizedRows,
  const ResizeFilter *resizeFilter,ExceptionInfo *exception)
{
  HipLibrary
    *library;

  Image
    *resize_image;

  MhImage
    source,
    destination;

  MhResizeFilter
    *filter;

  MhStatus
    status;

  void
    *p,
    *q;

  if (IsImageAcceleratable(image) == MagickFalse)
    return(HipDeclined(image,(Image *) NULL));
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  p=AcquireDevicePixels(library,image,1,exception);
  if (p == NULL)
    return(HipDeclined(image,(Image *) NULL));
  resize_image=AcquireResultImage(library,image,resizedColumns,resizedRows,&q,exception);
  if (resize_image == (Image *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  /* the weights are the reference's own: expert filter:* artifacts included */
  filter=library->AcquireResizeFilterFromCallback(ReferenceFilterWeight,
    (void *) resizeFilter,GetResizeFilterSupport(resizeFilter));
  status=MH_BAD_ARGUMENT;
  if ((filter != (MhResizeFilter *) NULL) &&
      (DescribeImage(library,image,p,&source) != MagickFalse) &&
      (DescribeImage(library,resize_image,q,&destination) != MagickFalse))
    status=library->ResizeImageWithFilter(&source,&destination,filter);
  if (filter != (MhResizeFilter *) NULL)
    (void) library->DestroyResizeFilter(filter);
  if (status != MH_OK)
    return(HipDeclined(image,DestroyImage(resize_image)));
  resize_image->type=image->type;    /* resize.c:3872 */
  HipAccepted(image);
  return(resize_image);
}

MagickPrivate MagickBooleanType AccelerateEqualizeImage(Image *image,
  ExceptionInfo *exception)
{
  HipLibrary
    *library;

  MhImage
    description;

  void
    *q;

  if (IsHistogramOperatorAcceleratable(image) == MagickFalse)
    return(HipDeclined(image,MagickFalse));
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(HipDeclined(image,MagickFalse));
  q=AcquireDevicePixels(library,image,1,exception);
  if ((q == NULL) ||
      (DescribeImage(library,image,q,&description) == MagickFalse) ||
      (library->EqualizeImage(&description) != MH_OK))
    return(HipDeclined(image,MagickFalse));
  MarkDeviceCopyNewer(image);
  HipAccepted(image);
  return(MagickTrue);
}

/*
  ContrastStretchImage and EqualizeImage start with IdentifyImageType (enhance.c:1586-1588):
  for every sRGB-compatible colourspace an all-gray colour image is first re-laid-out as a GRAY
  image.  The library runs that scan for sRGB and RGB (and hands an all-gray image back to the
  CPU path); the other compatible colourspaces (Adobe98, DisplayP3, ProPhoto, scRGB,
  Transparent) therefore stay on the CPU.  Lab / XYZ (BASELINE config C4) are not
  sRGB-compatible: no scan, accelerated.
*/
static MagickBooleanType IsHistogramOperatorAcceleratable(const Image *image)
{
  switch (image->colorspace)
  {
    case sRGBColorspace:
    case RGBColorspace:
    case GRAYColorspace:
    case LinearGRAYColorspace:
    case LabColorspace:
    case XYZColorspace:
      break;
    default:
      return(HipDeclined(image,MagickFalse));
  }
  return(IsLayoutAcceleratable(image));
}

MagickPrivate MagickBooleanType AccelerateContrastStretchImage(Image *image,
  const double black_point,const double white_point,ExceptionInfo *exception)
{
  HipLibrary
    *library;

  int
    became_gray;

  MhImage
    description;

  void
    *q;

  if (IsHistogramOperatorAcceleratable(image) == MagickFalse)
    return(HipDeclined(image,MagickFalse));
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(HipDeclined(image,MagickFalse));
  q=AcquireDevicePixels(library,image,1,exception);
  became_gray=0;
  if ((q == NULL) ||
      (DescribeImage(library,image,q,&description) == MagickFalse) ||
      (library->ContrastStretchImage(&description,black_point,white_point,&became_gray) != MH_OK))
    return(HipDeclined(image,MagickFalse));
  /* (an all-gray colour image comes back as MH_UNSUPPORTED above: the CPU path then does the
     IdentifyImageType re-layout itself, enhance.c:1586-1588) */
  MarkDeviceCopyNewer(image);
  HipAccepted(image);
  return(MagickTrue);
}

/*
  MorphologyApply (morphology.c:3634): every caller of MorphologyImage / ConvolveImage
  (GaussianBlur, Sharpen, Edge, Emboss, -morphology ...) arrives here through the hook
  shim/patch_hooks.py adds at the top of the reference function.  The KernelInfo list is
  described in place — MhKernelInfo points at the reference's own value arrays.
*/
#define MaxAcceleratedKernels  64

MagickPrivate Image *AccelerateMorphologyApply(const Image *image,
  const MorphologyMethod method,const ssize_t iterations,const KernelInfo *kernel,
  const CompositeOperator compose,const double bias,ExceptionInfo *exception)
{
  const KernelInfo
    *k;

  HipLibrary
    *library;

  Image
    *morphology_image;

  MhImage
    source,
    destination;

  MhKernelInfo
    kernels[MaxAcceleratedKernels];

  MhMorphologyCompose
    override;

  size_t
    n;

  void
    *p,
    *q;

  /* the user's morphology:compose (morphology.c:4206): the operators the backend composes with */
  switch (compose)
  {
    case UndefinedCompositeOp: override=MH_MORPHOLOGY_COMPOSE_DEFAULT; break;
    case NoCompositeOp: override=MH_MORPHOLOGY_COMPOSE_NONE; break;
    case LightenCompositeOp: override=MH_MORPHOLOGY_COMPOSE_LIGHTEN; break;
    case DifferenceCompositeOp: override=MH_MORPHOLOGY_COMPOSE_DIFFERENCE; break;
    default: return(HipDeclined(image,(Image *) NULL));
  }
  if ((iterations == 0) || (IsImageAcceleratable(image) == MagickFalse))
    return(HipDeclined(image,(Image *) NULL));
  n=0;
  for (k=kernel; k != (const KernelInfo *) NULL; k=k->next)
  {
    if (n == MaxAcceleratedKernels)
      return(HipDeclined(image,(Image *) NULL));
    (void) memset(&kernels[n],0,sizeof(kernels[n]));
    kernels[n].type=MH_KERNEL_USERDEFINED;
    kernels[n].width=k->width;
    kernels[n].height=k->height;
    kernels[n].x=k->x;
    kernels[n].y=k->y;
    kernels[n].values=(double *) k->values;          /* MagickRealType is double */
    kernels[n].minimum=k->minimum;
    kernels[n].maximum=k->maximum;
    kernels[n].negative_range=k->negative_range;
    kernels[n].positive_range=k->positive_range;
    kernels[n].angle=k->angle;
    if (n != 0)
      kernels[n-1].next=&kernels[n];
    n++;
  }
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  p=AcquireDevicePixels(library,image,1,exception);
  if (p == NULL)
    return(HipDeclined(image,(Image *) NULL));
  morphology_image=AcquireResultImage(library,image,image->columns,image->rows,&q,exception);
  if (morphology_image == (Image *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  /* MorphologyMethod and MhMorphologyMethod share their values (morphology.h:72-98) */
  if ((DescribeImage(library,image,p,&source) == MagickFalse) ||
      (DescribeImage(library,morphology_image,q,&destination) == MagickFalse) ||
      (library->MorphologyImageCompose(&source,&destination,(MhMorphologyMethod) method,iterations,
         kernels,bias,override) != MH_OK))
    return(HipDeclined(image,DestroyImage(morphology_image)));
  morphology_image->type=image->type;                /* morphology.c:2800, :3222 */
  HipAccepted(image);
  return(morphology_image);
}

/*
  TransformImageColorspace (colorspace.c:1751) between sRGB, linear RGB, XYZ and Lab with
  the default illuminant.  The bookkeeping the CPU path does around the pixel loop is
  kept: X -> sRGB ends in SetImageColorspace(sRGB), sRGB -> Y in SetImageColorspace(Y)
  (colorspace.c:1052, :2390).
*/
static MagickBooleanType IsColorspaceAccelerated(const ColorspaceType colorspace)
{
  /* sRGB, linear RGB and the pointwise colourspaces of ConvertRGBToGeneric /
     ConvertGenericToRGB (colorspace.c:958-985, :2292-2319) */
  switch (colorspace)
  {
    case sRGBColorspace: case RGBColorspace: case XYZColorspace: case LabColorspace:
    case CMYColorspace: case HCLColorspace: case HCLpColorspace: case HSBColorspace:
    case HSIColorspace: case HSLColorspace: case HSVColorspace: case HWBColorspace:
    case LCHColorspace: case LCHabColorspace: case LCHuvColorspace: case LMSColorspace:
    case LuvColorspace: case xyYColorspace: case YCbCrColorspace: case YDbDrColorspace:
    case YIQColorspace: case YPbPrColorspace: case YUVColorspace: case JzazbzColorspace:
    case DisplayP3Colorspace: case Adobe98Colorspace: case ProPhotoColorspace:
    case OklabColorspace: case OklchColorspace: case CAT02LMSColorspace:
      return(MagickTrue);
    default:
      break;
  }
  return(MagickFalse);
}

/*
  SetImageColorspace for an image whose current pixels live on the device.  It ends in
  SyncImagePixelCache -> GetImagePixelCache, which (a) brings the host block up to date
  (CopyOpenCLBuffer, cache.c:1711) and (b) re-opens a cache whose recorded colourspace
  differs from the image's: a new host block plus a copy of the old one (cache.c:3746-3790).
  Neither is wanted — the pixels are already in the new colourspace and nobody asked for
  them on the host — so the cache is re-tagged first and the device record is set aside for
  the duration of the call.  Should the cache have been replaced all the same, the device
  copy is downloaded into the new block.
*/
static MagickBooleanType SetResidentImageColorspace(HipLibrary *library,Image *image,
  const ColorspaceType colorspace,ExceptionInfo *exception)
{
  CacheInfo
    *cache_info;

  MagickBooleanType
    status;

  MagickCLCacheInfo
    info;

  cache_info=(CacheInfo *) image->cache;
  LockSemaphoreInfo(cache_info->semaphore);
  info=cache_info->opencl;
  cache_info->opencl=(MagickCLCacheInfo) NULL;
  cache_info->colorspace=colorspace;
  UnlockSemaphoreInfo(cache_info->semaphore);
  status=SetImageColorspace(image,colorspace,exception);
  if (info == (MagickCLCacheInfo) NULL)
    return(status);
  cache_info=(CacheInfo *) image->cache;
  LockSemaphoreInfo(cache_info->semaphore);
  if ((cache_info->type == MemoryCache) && (cache_info->pixels == info->pixels) &&
      (cache_info->length == info->length) && (cache_info->opencl == (MagickCLCacheInfo) NULL))
    cache_info->opencl=info;
  else
    {
      if ((cache_info->type == MemoryCache) && (cache_info->pixels != (Quantum *) NULL) &&
          (cache_info->length == info->length))
        {
          if (library->Download(-1,cache_info->pixels,(const void *) info->buffer,
                (size_t) info->length,NULL) != MH_OK)
            status=MagickFalse;
          CountHipTransfer(0);
        }
      else
        status=MagickFalse;
      (void) library->DeviceFree(-1,(void *) info->buffer);
      info=(MagickCLCacheInfo) RelinquishMagickMemory(info);
    }
  UnlockSemaphoreInfo(cache_info->semaphore);
  return(status);
}

MagickPrivate MagickBooleanType AccelerateTransformImageColorspace(Image *image,
  const ColorspaceType colorspace,ExceptionInfo *exception)
{
  HipLibrary
    *library;

  MhImage
    description;

  void
    *q;

  if (((colorspace == GRAYColorspace) || (colorspace == LinearGRAYColorspace)) &&
      (image->colorspace == sRGBColorspace) && (image->number_channels >= 3) &&
      (IsLayoutAcceleratable(image) != MagickFalse))
    {
      /*
        sRGB -> GRAY / LinearGRAY (colorspace.c:843-957): the gray value into the first channel
        on the device; SetImageColorspace then re-lays the pixel cache out as one gray channel
        (it finds the device copy newer and fetches it first: CopyOpenCLBuffer, cache.c:1711) —
        the same hand-over as after AccelerateGrayscaleImage (enhance.c:2500-2510).
      */
      library=AcquireHipLibrary();
      if (library == (HipLibrary *) NULL)
        return(HipDeclined(image,MagickFalse));
      q=AcquireDevicePixels(library,image,1,exception);
      if ((q == NULL) || (DescribeImage(library,image,q,&description) == MagickFalse) ||
          (library->TransformImageColorspace(&description,(MhColorspace) colorspace) != MH_OK))
        return(HipDeclined(image,MagickFalse));
      MarkDeviceCopyNewer(image);
      HipAccepted(image);
      if (SetImageColorspace(image,colorspace,exception) == MagickFalse)
        return(MagickFalse);
      image->type=GrayscaleType;
      return(MagickTrue);
    }
  if ((IsColorspaceAccelerated(image->colorspace) == MagickFalse) ||
      (IsColorspaceAccelerated(colorspace) == MagickFalse) ||
      (image->colorspace == colorspace) || (image->number_channels < 3) ||
      (IsLayoutAcceleratable(image) == MagickFalse) ||
      (GetImageArtifact(image,"color:illuminant") != (const char *) NULL) ||
      (GetImageProperty(image,"white-luminance",exception) != (const char *) NULL))
    return(HipDeclined(image,MagickFalse));          /* D65 and the default Jzazbz white luminance only (colorspace.c:993-995) */
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(HipDeclined(image,MagickFalse));
  q=AcquireDevicePixels(library,image,1,exception);
  /* ColorspaceType and MhColorspace share their values (colorspace.h:27-66) */
  if ((q == NULL) ||
      (DescribeImage(library,image,q,&description) == MagickFalse) ||
      (library->TransformImageColorspace(&description,(MhColorspace) colorspace) != MH_OK))
    return(HipDeclined(image,MagickFalse));
  MarkDeviceCopyNewer(image);
  HipAccepted(image);
  return(SetResidentImageColorspace(library,image,colorspace,exception));
}

/* ---- operators outside the hot path: always "not handled", the CPU code runs ---- */
/* DespeckleImage's call site: effect.c:1342-1346 */
MagickPrivate Image *AccelerateDespeckleImage(const Image *image,ExceptionInfo *exception)
{
  HipLibrary
    *library;

  Image
    *despeckle_image;

  MhImage
    source,
    destination;

  void
    *p,
    *q;

  if (IsImageAcceleratable(image) == MagickFalse)
    return(HipDeclined(image,(Image *) NULL));
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  p=AcquireDevicePixels(library,image,1,exception);
  if (p == NULL)
    return(HipDeclined(image,(Image *) NULL));
  despeckle_image=AcquireResultImage(library,image,image->columns,image->rows,&q,exception);
  if (despeckle_image == (Image *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  if ((DescribeImage(library,image,p,&source) == MagickFalse) ||
      (DescribeImage(library,despeckle_image,q,&destination) == MagickFalse) ||
      (library->DespeckleImage(&source,&destination) != MH_OK))
    return(HipDeclined(image,DestroyImage(despeckle_image)));
  despeckle_image->type=image->type;       /* effect.c:1486 */
  HipAccepted(image);
  return(despeckle_image);
}

/* LocalContrastImage's call site: effect.c:1794-1798 */
MagickPrivate Image *AccelerateLocalContrastImage(const Image *image,const double radius,
  const double strength,ExceptionInfo *exception)
{
  HipLibrary
    *library;

  Image
    *contrast_image;

  MhImage
    source,
    destination;

  void
    *p,
    *q;

  if (IsImageAcceleratable(image) == MagickFalse)
    return(HipDeclined(image,(Image *) NULL));
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  p=AcquireDevicePixels(library,image,1,exception);
  if (p == NULL)
    return(HipDeclined(image,(Image *) NULL));
  contrast_image=AcquireResultImage(library,image,image->columns,image->rows,&q,exception);
  if (contrast_image == (Image *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  if ((DescribeImage(library,image,p,&source) == MagickFalse) ||
      (DescribeImage(library,contrast_image,q,&destination) == MagickFalse) ||
      (library->LocalContrastImage(&source,&destination,radius,strength) != MH_OK))
    return(HipDeclined(image,DestroyImage(contrast_image)));
  HipAccepted(image);
  return(contrast_image);
}

/* MotionBlurImage hands its kernel and offsets to the hook (effect.c:2397-2404) */
MagickPrivate Image *AccelerateMotionBlurImage(const Image *image,const double *kernel,
  const size_t width,const OffsetInfo *offset,ExceptionInfo *exception)
{
  HipLibrary
    *library;

  Image
    *blur_image;

  MhImage
    source,
    destination;

  MhStatus
    status;

  ptrdiff_t
    *offsets;

  size_t
    i;

  void
    *p,
    *q;

  if ((IsImageAcceleratable(image) == MagickFalse) || (width == 0))
    return(HipDeclined(image,(Image *) NULL));
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  offsets=(ptrdiff_t *) AcquireQuantumMemory(width,2*sizeof(*offsets));
  if (offsets == (ptrdiff_t *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  for (i=0; i < width; i++)
  {
    offsets[2*i]=(ptrdiff_t) offset[i].x;
    offsets[2*i+1]=(ptrdiff_t) offset[i].y;
  }
  blur_image=(Image *) NULL;
  status=MH_BAD_ARGUMENT;
  p=AcquireDevicePixels(library,image,1,exception);
  if (p != NULL)
    blur_image=AcquireResultImage(library,image,image->columns,image->rows,&q,exception);
  if ((blur_image != (Image *) NULL) &&
      (DescribeImage(library,image,p,&source) != MagickFalse) &&
      (DescribeImage(library,blur_image,q,&destination) != MagickFalse))
    status=library->MotionBlurImageWithKernel(&source,&destination,kernel,width,offsets);
  offsets=(ptrdiff_t *) RelinquishMagickMemory(offsets);
  if (status != MH_OK)
    {
      if (blur_image != (Image *) NULL)
        blur_image=DestroyImage(blur_image);
      return(HipDeclined(image,(Image *) NULL));
    }
  HipAccepted(image);
  return(blur_image);
}

/* RotationalBlurImage's call site: effect.c:3241-3245 */
MagickPrivate Image *AccelerateRotationalBlurImage(const Image *image,const double angle,
  ExceptionInfo *exception)
{
  HipLibrary
    *library;

  Image
    *blur_image;

  MhImage
    source,
    destination;

  void
    *p,
    *q;

  if (IsImageAcceleratable(image) == MagickFalse)
    return(HipDeclined(image,(Image *) NULL));
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  p=AcquireDevicePixels(library,image,1,exception);
  if (p == NULL)
    return(HipDeclined(image,(Image *) NULL));
  blur_image=AcquireResultImage(library,image,image->columns,image->rows,&q,exception);
  if (blur_image == (Image *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  if ((DescribeImage(library,image,p,&source) == MagickFalse) ||
      (DescribeImage(library,blur_image,q,&destination) == MagickFalse) ||
      (library->RotationalBlurImage(&source,&destination,angle) != MH_OK))
    return(HipDeclined(image,DestroyImage(blur_image)));
  HipAccepted(image);
  return(blur_image);
}

/*
  WaveletDenoiseImage's own hook (visual-effects.c:3552-3556) does not carry `softness`, on
  which the CPU result depends, so it cannot be honoured; shim/patch_hooks.py redirects the
  call site to the variant below.
*/
MagickPrivate Image *AccelerateWaveletDenoiseImage(const Image *magick_unused(image),
  const double magick_unused(threshold),ExceptionInfo *magick_unused(exception))
{
  return((Image *) NULL);
}

MagickPrivate Image *AccelerateWaveletDenoiseImageSoft(const Image *image,
  const double threshold,const double softness,ExceptionInfo *exception)
{
  HipLibrary
    *library;

  Image
    *noise_image;

  MhImage
    source,
    destination;

  void
    *p,
    *q;

  if (IsImageAcceleratable(image) == MagickFalse)
    return(HipDeclined(image,(Image *) NULL));
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  p=AcquireDevicePixels(library,image,1,exception);
  if (p == NULL)
    return(HipDeclined(image,(Image *) NULL));
  noise_image=AcquireResultImage(library,image,image->columns,image->rows,&q,exception);
  if (noise_image == (Image *) NULL)
    return(HipDeclined(image,(Image *) NULL));
  if ((DescribeImage(library,image,p,&source) == MagickFalse) ||
      (DescribeImage(library,noise_image,q,&destination) == MagickFalse) ||
      (library->WaveletDenoiseImage(&source,&destination,threshold,softness) != MH_OK))
    return(HipDeclined(image,DestroyImage(noise_image)));
  HipAccepted(image);
  return(noise_image);
}

/* In-place operator on the device copy of `image`; marks that copy as the newer one. */
static MagickBooleanType AcquireInPlace(const Image *image,HipLibrary **library,
  MhImage *description,ExceptionInfo *exception)
{
  void
    *q;

  if (IsImageAcceleratable(image) == MagickFalse)
    return(HipDeclined(image,MagickFalse));
  *library=AcquireHipLibrary();
  if (*library == (HipLibrary *) NULL)
    return(HipDeclined(image,MagickFalse));
  q=AcquireDevicePixels(*library,image,1,exception);
  if (q == NULL)
    return(HipDeclined(image,MagickFalse));
  return(DescribeImage(*library,image,q,description));
}

MagickPrivate MagickBooleanType AccelerateFunctionImage(Image *image,
  const MagickFunction function,const size_t number_parameters,
  const double *parameters,ExceptionInfo *exception)
{
  HipLibrary
    *library;

  MhImage
    description;

  if ((image->storage_class != DirectClass) ||
      (AcquireInPlace(image,&library,&description,exception) == MagickFalse))
    return(HipDeclined(image,MagickFalse));
  /* MagickFunction and MhFunction share their values (statistic.h:129-136) */
  if (library->FunctionImage(&description,(MhFunction) function,number_parameters,
        parameters) != MH_OK)
    return(HipDeclined(image,MagickFalse));
  MarkDeviceCopyNewer(image);
  HipAccepted(image);
  return(MagickTrue);
}

MagickPrivate MagickBooleanType AccelerateGrayscaleImage(Image *image,
  const PixelIntensityMethod method,ExceptionInfo *exception)
{
  HipLibrary
    *library;

  MhImage
    description;

  /* only layouts whose first three channels are R,G,B (GrayscaleImage reads all three) */
  if ((image->number_channels < 3) ||
      (AcquireInPlace(image,&library,&description,exception) == MagickFalse))
    return(HipDeclined(image,MagickFalse));
  /* PixelIntensityMethod and MhIntensityMethod share their values (pixel.h) */
  if (library->GrayscaleImage(&description,(MhIntensityMethod) method) != MH_OK)
    return(HipDeclined(image,MagickFalse));
  MarkDeviceCopyNewer(image);
  HipAccepted(image);
  return(MagickTrue);       /* the caller sets intensity, type and the GRAY colourspace */
}

/* ContrastImage's call site is live in the reference (enhance.c:1412-1415) */
MagickPrivate MagickBooleanType AccelerateContrastImage(Image *image,
  const MagickBooleanType sharpen,ExceptionInfo *exception)
{
  HipLibrary
    *library;

  MhImage
    description;

  if ((image->number_channels < 3) ||
      (AcquireInPlace(image,&library,&description,exception) == MagickFalse))
    return(HipDeclined(image,MagickFalse));
  if (library->ContrastImage(&description,sharpen != MagickFalse ? 1 : 0) != MH_OK)
    return(HipDeclined(image,MagickFalse));
  MarkDeviceCopyNewer(image);
  HipAccepted(image);
  return(MagickTrue);
}

/*
  ModulateImage's call site (enhance.c:3770-3774) passes the parsed percentages and the
  modulate:colorspace model: all nine models of the reference are taken (HSL is also the
  default for every other value); a color:illuminant artifact (which resets the model and the
  illuminant) is left to the CPU.
*/
MagickPrivate MagickBooleanType AccelerateModulateImage(Image *image,
  const double percent_brightness,const double percent_hue,
  const double percent_saturation,const ColorspaceType colorspace,
  ExceptionInfo *exception)
{
  HipLibrary
    *library;

  MhImage
    description;

  ColorspaceType
    model;

  /* the nine models of enhance.c:3826-3890; every other value takes ModulateHSL's `default:` */
  switch (colorspace)
  {
    case HCLColorspace: case HCLpColorspace: case HSBColorspace: case HSIColorspace:
    case HSVColorspace: case HWBColorspace: case LCHColorspace: case LCHabColorspace:
    case LCHuvColorspace:
      model=colorspace;
      break;
    default:
      model=HSLColorspace;
      break;
  }
  if ((image->number_channels < 3) ||
      (GetImageArtifact(image,"color:illuminant") != (const char *) NULL) ||
      (AcquireInPlace(image,&library,&description,exception) == MagickFalse))
    return(HipDeclined(image,MagickFalse));
  /* ColorspaceType and MhColorspace share their values (colorspace.h:27-66) */
  if (library->ModulateImage(&description,percent_brightness,percent_saturation,percent_hue,
        (int) model) != MH_OK)
    return(HipDeclined(image,MagickFalse));
  MarkDeviceCopyNewer(image);
  HipAccepted(image);
  return(MagickTrue);
}

#endif /* MAGICKCORE_OPENCL_SUPPORT */
