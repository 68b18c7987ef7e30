/*
  opencl_hip.c — REPLACES MagickCore/opencl.c in the HIP-backed MagickCore build
  (see accelerate_hip.c).  With MAGICKCORE_OPENCL_SUPPORT on, the rest of
  MagickCore needs a handful of symbols from opencl.c:

    cache.c           AcquireMagickCLCacheInfo, CopyMagickCLCacheInfo,
                      RelinquishMagickCLCacheInfo, RetainOpenCLMemObject
                      (cache.c:979-982, :1276-1286, :5341-5353)
    magick.c          OpenCLTerminus                         (magick.c:1649-1651)
    public API        GetOpenCLEnabled / SetOpenCLEnabled / GetOpenCLDevices ...
                      (MagickCore/opencl.h:46-72)

  The HIP backend keeps no device-side state inside the pixel cache (every
  Accelerate*Image() call stages the heap block and synchronises before it
  returns), so CacheInfo::opencl is never set and the four cache hooks are
  unreachable stubs.  The enable switch is forwarded to libmagickhip.so; the
  device list is empty (devices are chosen inside the library).
*/
#include "MagickCore/studio.h"
#include "MagickCore/exception.h"
#include "MagickCore/opencl.h"
#include "MagickCore/opencl-private.h"

#if defined(MAGICKCORE_OPENCL_SUPPORT)

#include <dlfcn.h>
#include <stdlib.h>

static MagickBooleanType hip_enabled = MagickTrue;

static void ForwardEnabled(const MagickBooleanType value)
{
  int (*set_enabled)(int);
  void *handle;
  const char *path=getenv("MAGICK_HIP_LIBRARY");

  handle=dlopen(path != (const char *) NULL ? path : "libmagickhip.so",RTLD_NOW | RTLD_NOLOAD);
  if (handle == NULL)
    return;
  *(void **) &set_enabled=dlsym(handle,"MhSetEnabled");
  if (set_enabled != NULL)
    (void) set_enabled(value != MagickFalse ? 1 : 0);
  (void) dlclose(handle);
}

/* ------------------------------------------------------------- cache hooks */
MagickPrivate MagickCLCacheInfo AcquireMagickCLCacheInfo(
  MagickCLDevice magick_unused(device),Quantum *magick_unused(pixels),
  const MagickSizeType magick_unused(length))
{
  return((MagickCLCacheInfo) NULL);
}

MagickPrivate MagickCLCacheInfo CopyMagickCLCacheInfo(MagickCLCacheInfo info)
{
  return(info);
}

MagickPrivate MagickCLCacheInfo RelinquishMagickCLCacheInfo(
  MagickCLCacheInfo magick_unused(info),const MagickBooleanType magick_unused(relinquish_pixels))
{
  return((MagickCLCacheInfo) NULL);
}

MagickPrivate void RetainOpenCLMemObject(cl_mem magick_unused(memobj))
{
}

MagickPrivate void OpenCLTerminus(void)
{
  void (*terminus)(void);
  void *handle;
  const char *path=getenv("MAGICK_HIP_LIBRARY");

  handle=dlopen(path != (const char *) NULL ? path : "libmagickhip.so",RTLD_NOW | RTLD_NOLOAD);
  if (handle == NULL)
    return;
  *(void **) &terminus=dlsym(handle,"MhTerminus");
  if (terminus != NULL)
    terminus();
  (void) dlclose(handle);
}

/* -------------------------------------------------------------- public API */
MagickExport MagickBooleanType GetOpenCLEnabled(void)
{
  return(hip_enabled);
}

MagickExport MagickBooleanType SetOpenCLEnabled(const MagickBooleanType value)
{
  hip_enabled=value;
  ForwardEnabled(value);
  return(hip_enabled);
}

MagickExport MagickCLDevice *GetOpenCLDevices(size_t *length,
  ExceptionInfo *magick_unused(exception))
{
  if (length != (size_t *) NULL)
    *length=0;
  return((MagickCLDevice *) NULL);
}

MagickExport const char *GetOpenCLDeviceName(const MagickCLDevice magick_unused(device))
{
  return((const char *) NULL);
}

MagickExport const char *GetOpenCLDeviceVendorName(const MagickCLDevice magick_unused(device))
{
  return((const char *) NULL);
}

MagickExport const char *GetOpenCLDeviceVersion(const MagickCLDevice magick_unused(device))
{
  return((const char *) NULL);
}

MagickExport const KernelProfileRecord *GetOpenCLKernelProfileRecords(
  const MagickCLDevice magick_unused(device),size_t *length)
{
  if (length != (size_t *) NULL)
    *length=0;
  return((const KernelProfileRecord *) NULL);
}

MagickExport double GetOpenCLDeviceBenchmarkScore(const MagickCLDevice magick_unused(device))
{
  return(MAGICKCORE_OPENCL_UNDEFINED_SCORE);
}

MagickExport MagickCLDeviceType GetOpenCLDeviceType(const MagickCLDevice magick_unused(device))
{
  return(UndefinedCLDeviceType);
}

MagickExport MagickBooleanType GetOpenCLDeviceEnabled(const MagickCLDevice magick_unused(device))
{
  return(MagickFalse);
}

MagickExport void SetOpenCLDeviceEnabled(MagickCLDevice magick_unused(device),
  const MagickBooleanType magick_unused(value))
{
}

MagickExport void SetOpenCLKernelProfileEnabled(MagickCLDevice magick_unused(device),
  const MagickBooleanType magick_unused(value))
{
}

#endif /* MAGICKCORE_OPENCL_SUPPORT */
