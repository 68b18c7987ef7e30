/*
  TEST INFRASTRUCTURE — driver for the compiled reference oracle (oracle/_ref).

  This file is the only first-party code linked into oracle/_ref/libmagickref_*.so;
  everything else in that library is the reference's own MagickCore, compiled
  from /root/reference where it lies (see Makefile in this directory).  The
  driver moves raw Quantum buffers in and out of the reference's pixel cache
  and calls the reference's CPU/OpenMP operators, so tests (and bench.py's
  cpu_baseline leg) can compare the HIP path with the real thing.

  It is never linked into, imported by or called from the product library.

  Raw layout (both directions): row-major, channel-interleaved
  Quantum[rows][columns][number_channels] exactly as the pixel cache stores it
  (MagickCore/cache.c, MagickCore/pixel.c:6132-6205).  Quantum is
  unsigned short (Q16) or float (Q16-HDRI) depending on the build.
*/
#include "MagickCore/studio.h"
#include "MagickCore/MagickCore.h"
#include "MagickCore/resize-private.h"
#include "MagickCore/thread-private.h"
#include <omp.h>

#define REF_API __attribute__((visibility("default")))

static ExceptionInfo *ref_exception = (ExceptionInfo *) NULL;

REF_API int ref_init(void)
{
  if (ref_exception == (ExceptionInfo *) NULL)
    {
      MagickCoreGenesis("magickref",MagickFalse);
      ref_exception=AcquireExceptionInfo();
    }
  return(0);
}

REF_API int ref_quantum_is_float(void)
{
#if defined(MAGICKCORE_HDRI_SUPPORT)
  return(1);
#else
  return(0);
#endif
}

REF_API size_t ref_quantum_size(void)
{
  return(sizeof(Quantum));
}

REF_API int ref_thread_limit(void)
{
  return((int) GetMagickResourceLimit(ThreadResource));
}

REF_API void ref_set_thread_limit(int n)
{
  (void) SetMagickResourceLimit(ThreadResource,(MagickSizeType) n);
}

REF_API const char *ref_last_error(void)
{
  if ((ref_exception != (ExceptionInfo *) NULL) &&
      (ref_exception->severity != UndefinedException))
    return(ref_exception->reason != (char *) NULL ? ref_exception->reason :
      "unknown");
  return("");
}

REF_API void ref_clear_error(void)
{
  if (ref_exception != (ExceptionInfo *) NULL)
    ClearMagickException(ref_exception);
}

/*
  Image construction.  map: "RGBA" | "RGB" | "GRAY" | "GRAYA".
  colorspace: a -colorspace name ("sRGB","RGB","Lab","Gray",...).
*/
REF_API void *ref_image_new(size_t columns,size_t rows,const char *map,
  const char *colorspace,const void *pixels)
{
  Image *image;
  ImageInfo *info;
  Quantum *q;
  ssize_t cs,y;
  size_t row_bytes;

  ref_init();
  info=AcquireImageInfo();
  image=AcquireImage(info,ref_exception);
  info=DestroyImageInfo(info);
  if (image == (Image *) NULL)
    return(NULL);
  cs=ParseCommandOption(MagickColorspaceOptions,MagickFalse,colorspace);
  if (cs < 0)
    {
      image=DestroyImage(image);
      return(NULL);
    }
  image->colorspace=(ColorspaceType) cs;
  if ((LocaleCompare(map,"GRAY") == 0) || (LocaleCompare(map,"GRAYA") == 0))
    if ((cs != GRAYColorspace) && (cs != LinearGRAYColorspace))
      image->colorspace=GRAYColorspace;
  image->alpha_trait=UndefinedPixelTrait;
  if ((LocaleCompare(map,"RGBA") == 0) || (LocaleCompare(map,"GRAYA") == 0))
    image->alpha_trait=BlendPixelTrait;
  image->depth=MAGICKCORE_QUANTUM_DEPTH;
  if (SetImageExtent(image,columns,rows,ref_exception) == MagickFalse)
    {
      image=DestroyImage(image);
      return(NULL);
    }
  if (image->colorspace == sRGBColorspace)
    image->gamma=1.0/2.2;
  else if ((image->colorspace == RGBColorspace) ||
           (image->colorspace == LinearGRAYColorspace))
    image->gamma=1.0;
  row_bytes=columns*GetPixelChannels(image)*sizeof(Quantum);
  for (y=0; y < (ssize_t) rows; y++)
  {
    q=GetAuthenticPixels(image,0,y,columns,1,ref_exception);
    if (q == (Quantum *) NULL)
      {
        image=DestroyImage(image);
        return(NULL);
      }
    if (pixels != NULL)
      (void) memcpy(q,(const char *) pixels+(size_t) y*row_bytes,row_bytes);
    else
      (void) memset(q,0,row_bytes);
    (void) SyncAuthenticPixels(image,ref_exception);
  }
  return((void *) image);
}

REF_API void ref_image_free(void *handle)
{
  if (handle != NULL)
    (void) DestroyImage((Image *) handle);
}

REF_API int ref_image_info(const void *handle,size_t *columns,size_t *rows,
  size_t *channels,int *colorspace,int *alpha_trait,int *type)
{
  const Image *image=(const Image *) handle;
  if (image == (const Image *) NULL)
    return(-1);
  if (columns) *columns=image->columns;
  if (rows) *rows=image->rows;
  if (channels) *channels=GetPixelChannels(image);
  if (colorspace) *colorspace=(int) image->colorspace;
  if (alpha_trait) *alpha_trait=(int) image->alpha_trait;
  if (type) *type=(int) image->type;
  return(0);
}

REF_API const char *ref_colorspace_name(int colorspace)
{
  return(CommandOptionToMnemonic(MagickColorspaceOptions,(ssize_t) colorspace));
}

REF_API int ref_image_get(const void *handle,void *pixels)
{
  const Image *image=(const Image *) handle;
  const Quantum *p;
  size_t row_bytes;
  ssize_t y;

  if (image == (const Image *) NULL)
    return(-1);
  row_bytes=image->columns*GetPixelChannels(image)*sizeof(Quantum);
  for (y=0; y < (ssize_t) image->rows; y++)
  {
    p=GetVirtualPixels(image,0,y,image->columns,1,ref_exception);
    if (p == (const Quantum *) NULL)
      return(-1);
    (void) memcpy((char *) pixels+(size_t) y*row_bytes,p,row_bytes);
  }
  return(0);
}

/* rows [y0, y0+rows) only: a 17 GB result is compared a band at a time */
REF_API int ref_image_get_rows(const void *handle,ssize_t y0,size_t rows,void *pixels)
{
  const Image *image=(const Image *) handle;
  const Quantum *p;
  size_t row_bytes;
  ssize_t y;

  if ((image == (const Image *) NULL) || (y0 < 0) || ((size_t) y0+rows > image->rows))
    return(-1);
  row_bytes=image->columns*GetPixelChannels(image)*sizeof(Quantum);
  for (y=0; y < (ssize_t) rows; y++)
  {
    p=GetVirtualPixels(image,0,y0+y,image->columns,1,ref_exception);
    if (p == (const Quantum *) NULL)
      return(-1);
    (void) memcpy((char *) pixels+(size_t) y*row_bytes,p,row_bytes);
  }
  return(0);
}

/* -channel style mask, e.g. "RGB", "R", "A", "All" ... returns previous mask */
REF_API int ref_image_set_channel_mask(void *handle,const char *channels)
{
  Image *image=(Image *) handle;
  ssize_t mask=ParseChannelOption(channels);
  if (mask < 0)
    return(-1);
  return((int) SetImageChannelMask(image,(ChannelType) mask));
}

REF_API int ref_image_set_artifact(void *handle,const char *key,
  const char *value)
{
  Image *image=(Image *) handle;
  if (value == NULL)
    return(DeleteImageArtifact(image,key) != MagickFalse ? 0 : 1);
  return(SetImageArtifact(image,key,value) != MagickFalse ? 0 : -1);
}

REF_API const char *ref_image_get_property(void *handle,const char *key)
{
  return(GetImageProperty((Image *) handle,key,ref_exception));
}

/*
  Operators.  New-image operators return a new handle (NULL on failure),
  in-place operators return 0 on success.  *seconds (optional) receives the
  wall time of the operator call alone.
*/
#define TIMED_BEGIN  double t0_=omp_get_wtime()
#define TIMED_END    if (seconds != NULL) *seconds=omp_get_wtime()-t0_

REF_API void *ref_blur(const void *handle,double radius,double sigma,
  double *seconds)
{
  Image *out;
  TIMED_BEGIN;
  out=BlurImage((const Image *) handle,radius,sigma,ref_exception);
  TIMED_END;
  return((void *) out);
}

REF_API void *ref_sharpen(const void *handle,double radius,double sigma,
  double *seconds)
{
  Image *out;
  TIMED_BEGIN;
  out=SharpenImage((const Image *) handle,radius,sigma,ref_exception);
  TIMED_END;
  return((void *) out);
}

REF_API void *ref_emboss(const void *handle,double radius,double sigma,
  double *seconds)
{
  Image *out;
  TIMED_BEGIN;
  out=EmbossImage((const Image *) handle,radius,sigma,ref_exception);
  TIMED_END;
  return((void *) out);
}

REF_API void *ref_edge(const void *handle,double radius,double *seconds)
{
  Image *out;
  TIMED_BEGIN;
  out=EdgeImage((const Image *) handle,radius,ref_exception);
  TIMED_END;
  return((void *) out);
}

REF_API void *ref_wavelet_denoise(const void *handle,double threshold,double softness,
  double *seconds)
{
  Image *out;
  TIMED_BEGIN;
  out=WaveletDenoiseImage((const Image *) handle,threshold,softness,ref_exception);
  TIMED_END;
  return((void *) out);
}

REF_API void *ref_despeckle(const void *handle,double *seconds)
{
  Image *out;
  TIMED_BEGIN;
  out=DespeckleImage((const Image *) handle,ref_exception);
  TIMED_END;
  return((void *) out);
}

REF_API void *ref_local_contrast(const void *handle,double radius,double strength,double *seconds)
{
  Image *out;
  TIMED_BEGIN;
  out=LocalContrastImage((const Image *) handle,radius,strength,ref_exception);
  TIMED_END;
  return((void *) out);
}

REF_API void *ref_rotational_blur(const void *handle,double angle,double *seconds)
{
  Image *out;
  TIMED_BEGIN;
  out=RotationalBlurImage((const Image *) handle,angle,ref_exception);
  TIMED_END;
  return((void *) out);
}

REF_API void *ref_motion_blur(const void *handle,double radius,double sigma,double angle,
  double *seconds)
{
  Image *out;
  TIMED_BEGIN;
  out=MotionBlurImage((const Image *) handle,radius,sigma,angle,ref_exception);
  TIMED_END;
  return((void *) out);
}

REF_API void *ref_gaussian_blur(const void *handle,double radius,double sigma,
  double *seconds)
{
  Image *out;
  TIMED_BEGIN;
  out=GaussianBlurImage((const Image *) handle,radius,sigma,ref_exception);
  TIMED_END;
  return((void *) out);
}

REF_API void *ref_unsharp(const void *handle,double radius,double sigma,
  double gain,double threshold,double *seconds)
{
  Image *out;
  TIMED_BEGIN;
  out=UnsharpMaskImage((const Image *) handle,radius,sigma,gain,threshold,
    ref_exception);
  TIMED_END;
  return((void *) out);
}

REF_API void *ref_convolve(const void *handle,const char *kernel,
  double *seconds)
{
  Image *out;
  KernelInfo *kernel_info=AcquireKernelInfo(kernel,ref_exception);
  if (kernel_info == (KernelInfo *) NULL)
    return(NULL);
  {
    TIMED_BEGIN;
    out=ConvolveImage((const Image *) handle,kernel_info,ref_exception);
    TIMED_END;
  }
  kernel_info=DestroyKernelInfo(kernel_info);
  return((void *) out);
}

REF_API void *ref_morphology(const void *handle,const char *method,
  ssize_t iterations,const char *kernel,double *seconds)
{
  Image *out;
  KernelInfo *kernel_info;
  ssize_t m=ParseCommandOption(MagickMorphologyOptions,MagickFalse,method);
  if (m < 0)
    return(NULL);
  kernel_info=AcquireKernelInfo(kernel,ref_exception);
  if (kernel_info == (KernelInfo *) NULL)
    return(NULL);
  {
    TIMED_BEGIN;
    out=MorphologyImage((const Image *) handle,(MorphologyMethod) m,iterations,
      kernel_info,ref_exception);
    TIMED_END;
  }
  kernel_info=DestroyKernelInfo(kernel_info);
  return((void *) out);
}

REF_API void *ref_resize(const void *handle,size_t columns,size_t rows,
  const char *filter,double *seconds)
{
  Image *out;
  ssize_t f=ParseCommandOption(MagickFilterOptions,MagickFalse,filter);
  if (f < 0)
    return(NULL);
  {
    TIMED_BEGIN;
    out=ResizeImage((const Image *) handle,columns,rows,(FilterType) f,
      ref_exception);
    TIMED_END;
  }
  return((void *) out);
}

REF_API int ref_contrast_stretch(void *handle,double black_point,
  double white_point,double *seconds)
{
  MagickBooleanType status;
  TIMED_BEGIN;
  status=ContrastStretchImage((Image *) handle,black_point,white_point,
    ref_exception);
  TIMED_END;
  return(status != MagickFalse ? 0 : -1);
}

REF_API int ref_equalize(void *handle,double *seconds)
{
  MagickBooleanType status;
  TIMED_BEGIN;
  status=EqualizeImage((Image *) handle,ref_exception);
  TIMED_END;
  return(status != MagickFalse ? 0 : -1);
}

REF_API int ref_colorspace(void *handle,const char *colorspace,double *seconds)
{
  MagickBooleanType status;
  ssize_t cs=ParseCommandOption(MagickColorspaceOptions,MagickFalse,colorspace);
  if (cs < 0)
    return(-1);
  {
    TIMED_BEGIN;
    status=TransformImageColorspace((Image *) handle,(ColorspaceType) cs,
      ref_exception);
    TIMED_END;
  }
  return(status != MagickFalse ? 0 : -1);
}

/* ImportImagePixels / ExportImagePixels with the reference's StorageType numbering */
REF_API int ref_import_pixels(void *handle,ssize_t x,ssize_t y,size_t width,size_t height,
  const char *map,int storage,const void *pixels)
{
  return(ImportImagePixels((Image *) handle,x,y,width,height,map,(StorageType) storage,pixels,
    ref_exception) != MagickFalse ? 0 : -1);
}

REF_API int ref_export_pixels(const void *handle,ssize_t x,ssize_t y,size_t width,size_t height,
  const char *map,int storage,void *pixels)
{
  return(ExportImagePixels((const Image *) handle,x,y,width,height,map,(StorageType) storage,
    pixels,ref_exception) != MagickFalse ? 0 : -1);
}

REF_API int ref_contrast(void *handle,int sharpen,double *seconds)
{
  MagickBooleanType status;
  TIMED_BEGIN;
  status=ContrastImage((Image *) handle,sharpen != 0 ? MagickTrue : MagickFalse,ref_exception);
  TIMED_END;
  return(status != MagickFalse ? 0 : -1);
}

REF_API int ref_modulate(void *handle,const char *modulate,double *seconds)
{
  MagickBooleanType status;
  TIMED_BEGIN;
  status=ModulateImage((Image *) handle,modulate,ref_exception);
  TIMED_END;
  return(status != MagickFalse ? 0 : -1);
}

REF_API int ref_grayscale(void *handle,const char *method,double *seconds)
{
  MagickBooleanType status;
  ssize_t m=ParseCommandOption(MagickPixelIntensityOptions,MagickFalse,method);
  if (m < 0)
    return(-1);
  {
    TIMED_BEGIN;
    status=GrayscaleImage((Image *) handle,(PixelIntensityMethod) m,ref_exception);
    TIMED_END;
  }
  return(status != MagickFalse ? 0 : -1);
}

REF_API int ref_function(void *handle,const char *function,size_t count,
  const double *parameters,double *seconds)
{
  MagickBooleanType status;
  ssize_t f=ParseCommandOption(MagickFunctionOptions,MagickFalse,function);
  if (f < 0)
    return(-1);
  {
    TIMED_BEGIN;
    status=FunctionImage((Image *) handle,(MagickFunction) f,count,parameters,ref_exception);
    TIMED_END;
  }
  return(status != MagickFalse ? 0 : -1);
}

/*
  Host-side builders, exposed so the product's restated builders can be
  checked value-for-value.
*/

/* Parse a (possibly multi-) kernel string; copy kernel number `index`.
   values may be NULL to query the geometry only.  Returns the number of
   kernels in the list, or -1. */
REF_API int ref_kernel(const char *kernel,int index,size_t *width,
  size_t *height,ssize_t *x,ssize_t *y,double *values,double *range4)
{
  KernelInfo *list,*k;
  int count=0,i;

  ref_init();
  list=AcquireKernelInfo(kernel,ref_exception);
  if (list == (KernelInfo *) NULL)
    return(-1);
  for (k=list; k != (KernelInfo *) NULL; k=k->next)
    count++;
  k=list;
  for (i=0; (i < index) && (k != (KernelInfo *) NULL); i++)
    k=k->next;
  if (k != (KernelInfo *) NULL)
    {
      if (width) *width=k->width;
      if (height) *height=k->height;
      if (x) *x=k->x;
      if (y) *y=k->y;
      if (values)
        (void) memcpy(values,k->values,k->width*k->height*sizeof(double));
      if (range4)
        {
          range4[0]=k->minimum; range4[1]=k->maximum;
          range4[2]=k->negative_range; range4[3]=k->positive_range;
        }
    }
  list=DestroyKernelInfo(list);
  return(count);
}

/* Filter weight as ResizeImage would evaluate it for `image` (artifacts
   honoured): weight(x) and the filter support. */
REF_API int ref_resize_filter_weights(const void *handle,const char *filter,
  const double *x,size_t n,double *weights,double *support)
{
  ResizeFilter *resize_filter;
  size_t i;
  ssize_t f=ParseCommandOption(MagickFilterOptions,MagickFalse,filter);
  if (f < 0)
    return(-1);
  resize_filter=AcquireResizeFilter((const Image *) handle,(FilterType) f,
    MagickFalse,ref_exception);
  if (resize_filter == (ResizeFilter *) NULL)
    return(-1);
  for (i=0; i < n; i++)
    weights[i]=GetResizeFilterWeight(resize_filter,x[i]);
  if (support)
    *support=GetResizeFilterSupport(resize_filter);
  resize_filter=DestroyResizeFilter(resize_filter);
  return(0);
}

REF_API double ref_pixel_intensity(const void *handle,const void *pixel)
{
  return((double) GetPixelIntensity((const Image *) handle,
    (const Quantum *) pixel));
}

REF_API double ref_decode_gamma(double pixel)
{
  return((double) DecodePixelGamma((MagickRealType) pixel));
}

REF_API double ref_encode_gamma(double pixel)
{
  return((double) EncodePixelGamma((MagickRealType) pixel));
}
