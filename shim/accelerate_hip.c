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
  (include/magickhip.h), loaded with dlopen on first use: without the library,
  without a GPU, or with MAGICK_HIP_DEVICE=off every function here returns
  NULL / MagickFalse and the CPU path runs.

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

#include <dlfcn.h>
#include <stdlib.h>
#include "magickhip.h"

/* ------------------------------------------------------------ the library */
typedef struct _HipLibrary
{
  void *handle;
  MhStatus (*Initialize)(void);
  void (*InitImage)(MhImage *,void *,size_t,size_t,uint32_t,int,MhQuantumKind,MhMemoryKind);
  MhStatus (*BlurImage)(const MhImage *,MhImage *,double,double);
  MhStatus (*UnsharpMaskImage)(const MhImage *,MhImage *,double,double,double,double);
  MhStatus (*ResizeImageWithFilter)(const MhImage *,MhImage *,const MhResizeFilter *);
  MhResizeFilter *(*AcquireResizeFilterFromCallback)(MhResizeWeightFunction,void *,double);
  MhResizeFilter *(*DestroyResizeFilter)(MhResizeFilter *);
  MhStatus (*ContrastStretchImage)(MhImage *,double,double,int *);
  MhStatus (*EqualizeImage)(MhImage *);
  int (*GetEnabled)(void);
} HipLibrary;

static HipLibrary hip_library;
static volatile int hip_library_state=0;      /* 0 = untried, 1 = ready, -1 = unavailable */
static size_t hip_accelerated_calls=0;        /* read by tests through GetMagickHipAcceleratedCalls */

static void *Resolve(void *handle,const char *name,int *missing)
{
  void *symbol=dlsym(handle,name);
  if (symbol == NULL)
    (*missing)++;
  return(symbol);
}

static HipLibrary *AcquireHipLibrary(void)
{
  const char
    *path;

  int
    missing;

  if (hip_library_state > 0)
    return(hip_library.GetEnabled() != 0 ? &hip_library : (HipLibrary *) NULL);
  if (hip_library_state < 0)
    return((HipLibrary *) NULL);
  path=getenv("MAGICK_HIP_LIBRARY");
  if (path == (const char *) NULL)
    path="libmagickhip.so";
  hip_library.handle=dlopen(path,RTLD_NOW | RTLD_LOCAL);
  if (hip_library.handle == NULL)
    {
      hip_library_state=(-1);
      return((HipLibrary *) NULL);
    }
  missing=0;
  *(void **) &hip_library.Initialize=Resolve(hip_library.handle,"MhInitialize",&missing);
  *(void **) &hip_library.InitImage=Resolve(hip_library.handle,"MhInitImage",&missing);
  *(void **) &hip_library.BlurImage=Resolve(hip_library.handle,"MagickHipBlurImage",&missing);
  *(void **) &hip_library.UnsharpMaskImage=Resolve(hip_library.handle,
    "MagickHipUnsharpMaskImage",&missing);
  *(void **) &hip_library.ResizeImageWithFilter=Resolve(hip_library.handle,
    "MagickHipResizeImageWithFilter",&missing);
  *(void **) &hip_library.AcquireResizeFilterFromCallback=Resolve(hip_library.handle,
    "MhAcquireResizeFilterFromCallback",&missing);
  *(void **) &hip_library.DestroyResizeFilter=Resolve(hip_library.handle,
    "MhDestroyResizeFilter",&missing);
  *(void **) &hip_library.ContrastStretchImage=Resolve(hip_library.handle,
    "MagickHipContrastStretchImage",&missing);
  *(void **) &hip_library.EqualizeImage=Resolve(hip_library.handle,"MagickHipEqualizeImage",
    &missing);
  *(void **) &hip_library.GetEnabled=Resolve(hip_library.handle,"MhGetEnabled",&missing);
  if ((missing != 0) || (hip_library.Initialize() != MH_OK))
    {
      hip_library_state=(-1);
      return((HipLibrary *) NULL);
    }
  hip_library_state=1;
  return(hip_library.GetEnabled() != 0 ? &hip_library : (HipLibrary *) NULL);
}

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
  The pixel-cache heap block of an image (what GetAuthenticOpenCLBuffer wraps
  in a cl_mem, cache.c:1259-1291): a private, materialised memory cache.
*/
static Quantum *AcquireHeapPixels(const Image *image,ExceptionInfo *exception)
{
  CacheInfo
    *cache_info;

  cache_info=(CacheInfo *) image->cache;
  if ((cache_info->type == UndefinedCache) || (cache_info->reference_count > 1))
    {
      if (SyncImagePixelCache((Image *) image,exception) == MagickFalse)
        return((Quantum *) NULL);
      cache_info=(CacheInfo *) image->cache;
    }
  if ((cache_info->type != MemoryCache) || (cache_info->mapped != MagickFalse))
    return((Quantum *) NULL);
  return(cache_info->pixels);
}

static MagickBooleanType DescribeImage(HipLibrary *library,const Image *image,
  Quantum *pixels,MhImage *description)
{
  ssize_t
    i;

  library->InitImage(description,pixels,image->columns,image->rows,
    (uint32_t) image->number_channels,image->alpha_trait != UndefinedPixelTrait ? 1 : 0,
#if defined(MAGICKCORE_HDRI_SUPPORT)
    MH_QUANTUM_F32,
#else
    MH_QUANTUM_U16,
#endif
    MH_MEMORY_HOST);
#if (MAGICKCORE_QUANTUM_DEPTH != 16)
  return(MagickFalse);
#endif
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

/* A new image of the given size whose cache is a fresh heap block. */
static Image *AcquireResultImage(const Image *image,const size_t columns,
  const size_t rows,Quantum **pixels,ExceptionInfo *exception)
{
  Image
    *result;

  result=CloneImage(image,columns,rows,MagickTrue,exception);
  if (result == (Image *) NULL)
    return((Image *) NULL);
  if (SetImageStorageClass(result,DirectClass,exception) == MagickFalse)
    return(DestroyImage(result));
  *pixels=AcquireHeapPixels(result,exception);
  if (*pixels == (Quantum *) NULL)
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

  Quantum
    *p,
    *q;

  assert(image != NULL);
  assert(exception != (ExceptionInfo *) NULL);
  if (IsImageAcceleratable(image) == MagickFalse)
    return((Image *) NULL);
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return((Image *) NULL);
  p=AcquireHeapPixels(image,exception);
  if (p == (Quantum *) NULL)
    return((Image *) NULL);
  blur_image=AcquireResultImage(image,image->columns,image->rows,&q,exception);
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

  Quantum
    *p,
    *q;

  if (IsImageAcceleratable(image) == MagickFalse)
    return((Image *) NULL);
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return((Image *) NULL);
  p=AcquireHeapPixels(image,exception);
  if (p == (Quantum *) NULL)
    return((Image *) NULL);
  unsharp_image=AcquireResultImage(image,image->columns,image->rows,&q,exception);
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

  Quantum
    *p,
    *q;

  if (IsImageAcceleratable(image) == MagickFalse)
    return((Image *) NULL);
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return((Image *) NULL);
  p=AcquireHeapPixels(image,exception);
  if (p == (Quantum *) NULL)
    return((Image *) NULL);
  resize_image=AcquireResultImage(image,resizedColumns,resizedRows,&q,exception);
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

  Quantum
    *q;

  if (IsImageAcceleratable(image) == MagickFalse)
    return(MagickFalse);
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(MagickFalse);
  q=AcquireHeapPixels(image,exception);
  if ((q == (Quantum *) NULL) ||
      (DescribeImage(library,image,q,&description) == MagickFalse) ||
      (library->EqualizeImage(&description) != MH_OK))
    return(MagickFalse);
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

  Quantum
    *q;

  if (IsImageAcceleratable(image) == MagickFalse)
    return(MagickFalse);
  library=AcquireHipLibrary();
  if (library == (HipLibrary *) NULL)
    return(MagickFalse);
  q=AcquireHeapPixels(image,exception);
  became_gray=0;
  if ((q == (Quantum *) NULL) ||
      (DescribeImage(library,image,q,&description) == MagickFalse) ||
      (library->ContrastStretchImage(&description,black_point,white_point,&became_gray) != MH_OK))
    return(MagickFalse);
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

MagickPrivate MagickBooleanType AccelerateFunctionImage(Image *magick_unused(image),
  const MagickFunction magick_unused(function),const size_t magick_unused(number_parameters),
  const double *magick_unused(parameters),ExceptionInfo *magick_unused(exception))
{
  return(MagickFalse);
}

MagickPrivate MagickBooleanType AccelerateGrayscaleImage(Image *magick_unused(image),
  const PixelIntensityMethod magick_unused(method),ExceptionInfo *magick_unused(exception))
{
  return(MagickFalse);
}

MagickPrivate MagickBooleanType AccelerateModulateImage(Image *magick_unused(image),
  const double magick_unused(percent_brightness),const double magick_unused(percent_hue),
  const double magick_unused(percent_saturation),
  const ColorspaceType magick_unused(colorspace),ExceptionInfo *magick_unused(exception))
{
  return(MagickFalse);
}

#endif /* MAGICKCORE_OPENCL_SUPPORT */
