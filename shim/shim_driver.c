/*
  shim_driver.c — a few ctypes-able entry points linked into the HIP-backed MagickCore build
  (shim/_build/libMagickCore-hip-*.so) beside the operator driver, for bench.py and the shim
  tests: what an application would do with MagickCore's own API around the accelerated
  operators.  Everything here is plain MagickCore (cache.h / cache-view.h); nothing calls the
  library directly.
*/
#include "MagickCore/studio.h"
#include "MagickCore/MagickCore.h"

#define SHIM_API __attribute__((visibility("default")))

/*
  The CPU looks at the pixels: GetVirtualPixels on one pixel makes the cache current —
  CopyOpenCLBuffer (cache.c:2772) downloads the device copy if it is the newer one.  What
  WriteImage or GetPixelCachePixels would trigger; returns 0 on success.
*/
SHIM_API int shim_image_sync(void *handle)
{
  ExceptionInfo
    *exception;

  const Quantum
    *p;

  exception=AcquireExceptionInfo();
  p=GetVirtualPixels((Image *) handle,0,0,1,1,exception);
  exception=DestroyExceptionInfo(exception);
  return(p != (const Quantum *) NULL ? 0 : 1);
}

/*
  The CPU writes the pixels: an authentic access of one pixel (cache.c:1711 brings the host block
  up to date and drops the device copy, the CPU may write).  The next accelerated operator uploads
  the image again — how bench.py times a call that is NOT part of a device-resident chain.
*/
SHIM_API int shim_image_touch(void *handle)
{
  ExceptionInfo
    *exception;

  MagickBooleanType
    status;

  Quantum
    *q;

  exception=AcquireExceptionInfo();
  status=MagickFalse;
  q=GetAuthenticPixels((Image *) handle,0,0,1,1,exception);
  if (q != (Quantum *) NULL)
    status=SyncAuthenticPixels((Image *) handle,exception);
  exception=DestroyExceptionInfo(exception);
  return(status != MagickFalse ? 0 : 1);
}

/* the host block of the pixel cache (NULL when it is not a heap cache) and its length in bytes */
SHIM_API void *shim_image_pixels(void *handle,size_t *length)
{
  ExceptionInfo
    *exception;

  MagickSizeType
    extent;

  void
    *pixels;

  exception=AcquireExceptionInfo();
  extent=0;
  pixels=GetPixelCachePixels((Image *) handle,&extent,exception);
  exception=DestroyExceptionInfo(exception);
  if (length != (size_t *) NULL)
    *length=(size_t) extent;
  return(pixels);
}
