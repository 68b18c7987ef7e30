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

#if defined(MAGICKCORE_OPENCL_SUPPORT)

#include <stdlib.h>
#include "MagickCore/memory_.h"
#include "MagickCore/opencl-private.h"
#include "magickhip_shim.h"

static size_t hip_accelerated_calls=0;        /* read by tests through GetMagickHipAcceleratedCalls */

MagickExport size_t GetMagickHipAcceleratedCalls(void)
{
  return(hip_accelerated_calls);
}

/* ------------------------------------------------------------------ gates */
/*
  What the backend can take, the same conditions the reference's accelerate
  layer imposes (accelerate.c:110-170): DirectClass; sRGB / RGB / GRAY /
  LinearGRAY; Undefined or Edge virtual pixels; no read, write or composite
  mask; at most four channels laid out R[,G,B][,A].
*/
static MagickBooleanType IsImageAcceleratable(const Image *image)
{
  if (image->storage_class != DirectClass)
    return(MagickFalse);
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
  if (cache_info->opencl != (MagickCLCacheInfo) NULL)
    cache_info->opencl->event_count=1U;
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
    return(DestroyImage(result));
  *device_pixels=AcquireDevicePixels(library,result,0,exception);
  if (*device_pixels == NULL)
    return(DestroyImage(result));
  return(result);
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
  if (IsImageAcceleratable(image) == MagickFalse)
    return((Image *) NULL);
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return((Image *) NULL);
  p=AcquireDevicePixels(library,image,1,exception);
  if (p == NULL)
    return((Image *) NULL);
  blur_image=AcquireResultImage(library,image,image->columns,image->rows,&q,exception);
  if (blur_image == (Image *) NULL)
    return((Image *) NULL);
  if ((DescribeImage(library,image,p,&source) == MagickFalse) ||
      (DescribeImage(library,blur_image,q,&destination) == MagickFalse) ||
      (library->BlurImage(&source,&destination,radius,sigma) != MH_OK))
    return(DestroyImage(blur_image));
  blur_image->type=image->type;      /* as MorphologyPrimitive does, morphology.c:2800 */
  hip_accelerated_calls++;
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

  if (IsImageAcceleratable(image) == MagickFalse)
    return((Image *) NULL);
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return((Image *) NULL);
  p=AcquireDevicePixels(library,image,1,exception);
  if (p == NULL)
    return((Image *) NULL);
  unsharp_image=AcquireResultImage(library,image,image->columns,image->rows,&q,exception);
  if (unsharp_image == (Image *) NULL)
    return((Image *) NULL);
  if ((DescribeImage(library,image,p,&source) == MagickFalse) ||
      (DescribeImage(library,unsharp_image,q,&destination) == MagickFalse) ||
      (library->UnsharpMaskImage(&source,&destination,radius,sigma,gain,threshold) != MH_OK))
    return(DestroyImage(unsharp_image));
  unsharp_image->type=image->type;   /* effect.c:4385 */
  hip_accelerated_calls++;
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
    return((Image *) NULL);
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return((Image *) NULL);
  p=AcquireDevicePixels(library,image,1,exception);
  if (p == NULL)
    return((Image *) NULL);
  resize_image=AcquireResultImage(library,image,resizedColumns,resizedRows,&q,exception);
  if (resize_image == (Image *) NULL)
    return((Image *) NULL);
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
    return(DestroyImage(resize_image));
  resize_image->type=image->type;    /* resize.c:3872 */
  hip_accelerated_calls++;
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

  if (IsImageAcceleratable(image) == MagickFalse)
    return(MagickFalse);
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(MagickFalse);
  q=AcquireDevicePixels(library,image,1,exception);
  if ((q == NULL) ||
      (DescribeImage(library,image,q,&description) == MagickFalse) ||
      (library->EqualizeImage(&description) != MH_OK))
    return(MagickFalse);
  MarkDeviceCopyNewer(image);
  hip_accelerated_calls++;
  return(MagickTrue);
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

  if (IsImageAcceleratable(image) == MagickFalse)
    return(MagickFalse);
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(MagickFalse);
  q=AcquireDevicePixels(library,image,1,exception);
  became_gray=0;
  if ((q == NULL) ||
      (DescribeImage(library,image,q,&description) == MagickFalse) ||
      (library->ContrastStretchImage(&description,black_point,white_point,&became_gray) != MH_OK))
    return(MagickFalse);
  MarkDeviceCopyNewer(image);
  if (became_gray != 0)              /* IdentifyImageType side effect, enhance.c:1586-1588 */
    (void) SetImageColorspace(image,GRAYColorspace,exception);
  hip_accelerated_calls++;
  return(MagickTrue);
}

/* ---- operators outside the hot path: always "not handled", the CPU code runs ---- */
MagickPrivate Image *AccelerateDespeckleImage(const Image *magick_unused(image),
  ExceptionInfo *magick_unused(exception))
{
  return((Image *) NULL);
}

MagickPrivate Image *AccelerateLocalContrastImage(const Image *magick_unused(image),
  const double magick_unused(radius),const double magick_unused(strength),
  ExceptionInfo *magick_unused(exception))
{
  return((Image *) NULL);
}

MagickPrivate Image *AccelerateMotionBlurImage(const Image *magick_unused(image),
  const double *magick_unused(kernel),const size_t magick_unused(width),
  const OffsetInfo *magick_unused(offset),ExceptionInfo *magick_unused(exception))
{
  return((Image *) NULL);
}

MagickPrivate Image *AccelerateRotationalBlurImage(const Image *magick_unused(image),
  const double magick_unused(angle),ExceptionInfo *magick_unused(exception))
{
  return((Image *) NULL);
}

MagickPrivate Image *AccelerateWaveletDenoiseImage(const Image *magick_unused(image),
  const double magick_unused(threshold),ExceptionInfo *magick_unused(exception))
{
  return((Image *) NULL);
}

MagickPrivate MagickBooleanType AccelerateContrastImage(Image *magick_unused(image),
  const MagickBooleanType magick_unused(sharpen),ExceptionInfo *magick_unused(exception))
{
  return(MagickFalse);
}

/* In-place operator on the device copy of `image`; marks that copy as the newer one. */
static MagickBooleanType AcquireInPlace(const Image *image,HipLibrary **library,
  MhImage *description,ExceptionInfo *exception)
{
  void
    *q;

  if (IsImageAcceleratable(image) == MagickFalse)
    return(MagickFalse);
  *library=AcquireHipLibrary();
  if (*library == (HipLibrary *) NULL)
    return(MagickFalse);
  q=AcquireDevicePixels(*library,image,1,exception);
  if (q == NULL)
    return(MagickFalse);
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
    return(MagickFalse);
  /* MagickFunction and MhFunction share their values (statistic.h:129-136) */
  if (library->FunctionImage(&description,(MhFunction) function,number_parameters,
        parameters) != MH_OK)
    return(MagickFalse);
  MarkDeviceCopyNewer(image);
  hip_accelerated_calls++;
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
    return(MagickFalse);
  /* PixelIntensityMethod and MhIntensityMethod share their values (pixel.h) */
  if (library->GrayscaleImage(&description,(MhIntensityMethod) method) != MH_OK)
    return(MagickFalse);
  MarkDeviceCopyNewer(image);
  hip_accelerated_calls++;
  return(MagickTrue);       /* the caller sets intensity, type and the GRAY colourspace */
}

MagickPrivate MagickBooleanType AccelerateModulateImage(Image *magick_unused(image),
  const double magick_unused(percent_brightness),const double magick_unused(percent_hue),
  const double magick_unused(percent_saturation),
  const ColorspaceType magick_unused(colorspace),ExceptionInfo *magick_unused(exception))
{
  return(MagickFalse);
}

#endif /* MAGICKCORE_OPENCL_SUPPORT */
