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

  Device residency.  The reference keeps an image on the device across chained
  operators by hanging a MagickCLCacheInfo off CacheInfo::opencl and calling
  CopyMagickCLCacheInfo() from every CPU access path (CopyOpenCLBuffer,
  cache.c:5341; call sites cache.c:1711, :2772, :4080, cache-view.c:160).  This
  file re-creates that protocol for HIP: accelerate_hip.c attaches a record
  { device pointer, host pixels, length, dirty } to the pixel cache of operator
  inputs (uploaded once) and results (NOT downloaded), and the hooks below bring
  the host block up to date the first time the CPU looks at the pixels:

    CopyMagickCLCacheInfo        dirty ? download : nothing; free the device copy
    RelinquishMagickCLCacheInfo  free the device copy (and the host block when
                                 asked: the cache is being destroyed)

  so  BlurImage -> ResizeImage -> EqualizeImage  pays PCIe once each way.
  The record re-uses the reference's struct (opencl-private.h:42-64): `buffer`
  holds the device pointer, `event_count` the dirty flag; no cl_* call is made.
*/
#include "MagickCore/studio.h"
#include "MagickCore/exception.h"
#include "MagickCore/memory_.h"
#include "MagickCore/memory-private.h"
#include "MagickCore/opencl.h"
#include "MagickCore/opencl-private.h"
#include "MagickCore/semaphore.h"

#if defined(MAGICKCORE_OPENCL_SUPPORT)

#include <dlfcn.h>
#include <stdlib.h>
#include "magickhip_shim.h"

static HipLibrary hip_library;
static volatile int hip_library_state=0;      /* 0 = untried, 1 = ready, -1 = unavailable */
static MagickBooleanType hip_enabled = MagickTrue;
static size_t hip_uploads=0,hip_downloads=0;
static SemaphoreInfo *hip_library_semaphore=(SemaphoreInfo *) NULL;

static void *Resolve(void *handle,const char *name,int *missing)
{
  void *symbol=dlsym(handle,name);
  if (symbol == NULL)
    (*missing)++;
  return(symbol);
}

static void LoadHipLibrary(void);

MagickPrivate HipLibrary *AcquireHipLibrary(void)
{
  if (hip_enabled == MagickFalse)
    return((HipLibrary *) NULL);
  if (__atomic_load_n(&hip_library_state,__ATOMIC_ACQUIRE) > 0)
    return(hip_library.GetEnabled() != 0 ? &hip_library : (HipLibrary *) NULL);
  if (__atomic_load_n(&hip_library_state,__ATOMIC_ACQUIRE) < 0)
    return((HipLibrary *) NULL);
  /*
    First use: one thread loads the library and fills the function table, concurrent first
    callers wait for it (operators run concurrently on different images).
  */
  if (hip_library_semaphore == (SemaphoreInfo *) NULL)
    ActivateSemaphoreInfo(&hip_library_semaphore);
  LockSemaphoreInfo(hip_library_semaphore);
  if (hip_library_state == 0)
    LoadHipLibrary();
  UnlockSemaphoreInfo(hip_library_semaphore);
  if (hip_library_state > 0)
    return(hip_library.GetEnabled() != 0 ? &hip_library : (HipLibrary *) NULL);
  return((HipLibrary *) NULL);
}

static void LoadHipLibrary(void)
{
  const char
    *path;

  int
    missing;

  path=getenv("MAGICK_HIP_LIBRARY");
  if (path == (const char *) NULL)
    path="libmagickhip.so";
  hip_library.handle=dlopen(path,RTLD_NOW | RTLD_LOCAL);
  if (hip_library.handle == NULL)
    {
      __atomic_store_n(&hip_library_state,-1,__ATOMIC_RELEASE);
      return;
    }
  missing=0;
#define MH_RESOLVE(field,name) *(void **) &hip_library.field=Resolve(hip_library.handle,name,&missing)
  MH_RESOLVE(Initialize,"MhInitialize");
  MH_RESOLVE(Terminus,"MhTerminus");
  MH_RESOLVE(GetEnabled,"MhGetEnabled");
  MH_RESOLVE(SetEnabled,"MhSetEnabled");
  MH_RESOLVE(InitImage,"MhInitImage");
  MH_RESOLVE(DeviceAlloc,"MhDeviceAlloc");
  MH_RESOLVE(DeviceFree,"MhDeviceFree");
  MH_RESOLVE(Upload,"MhUpload");
  MH_RESOLVE(Download,"MhDownload");
  MH_RESOLVE(Synchronize,"MhSynchronize");
  MH_RESOLVE(HostAlloc,"MhHostAlloc");
  MH_RESOLVE(HostFree,"MhHostFree");
  MH_RESOLVE(HostAllocatedBytes,"MhHostAllocatedBytes");
  MH_RESOLVE(BlurImage,"MagickHipBlurImage");
  MH_RESOLVE(UnsharpMaskImage,"MagickHipUnsharpMaskImage");
  MH_RESOLVE(ResizeImageWithFilter,"MagickHipResizeImageWithFilter");
  MH_RESOLVE(AcquireResizeFilterFromCallback,"MhAcquireResizeFilterFromCallback");
  MH_RESOLVE(DestroyResizeFilter,"MhDestroyResizeFilter");
  MH_RESOLVE(ContrastStretchImage,"MagickHipContrastStretchImage");
  MH_RESOLVE(EqualizeImage,"MagickHipEqualizeImage");
  MH_RESOLVE(GrayscaleImage,"MagickHipGrayscaleImage");
  MH_RESOLVE(FunctionImage,"MagickHipFunctionImage");
  MH_RESOLVE(MotionBlurImageWithKernel,"MagickHipMotionBlurImageWithKernel");
  MH_RESOLVE(WaveletDenoiseImage,"MagickHipWaveletDenoiseImage");
  MH_RESOLVE(DespeckleImage,"MagickHipDespeckleImage");
  MH_RESOLVE(LocalContrastImage,"MagickHipLocalContrastImage");
  MH_RESOLVE(RotationalBlurImage,"MagickHipRotationalBlurImage");
  MH_RESOLVE(ContrastImage,"MagickHipContrastImage");
  MH_RESOLVE(ModulateImage,"MagickHipModulateImage");
  MH_RESOLVE(MorphologyImage,"MagickHipMorphologyImage");
  MH_RESOLVE(MorphologyImageCompose,"MagickHipMorphologyImageCompose");
  MH_RESOLVE(TransformImageColorspace,"MagickHipTransformImageColorspace");
#undef MH_RESOLVE
  if ((missing != 0) || (hip_library.Initialize() != MH_OK))
    {
      __atomic_store_n(&hip_library_state,-1,__ATOMIC_RELEASE);
      return;
    }
  __atomic_store_n(&hip_library_state,1,__ATOMIC_RELEASE);
}

MagickPrivate void CountHipTransfer(int upload)
{
  if (upload != 0)
    (void) __atomic_fetch_add(&hip_uploads,1,__ATOMIC_RELAXED);
  else
    (void) __atomic_fetch_add(&hip_downloads,1,__ATOMIC_RELAXED);
}

MagickExport void GetMagickHipTransfers(size_t *uploads,size_t *downloads)
{
  *uploads=__atomic_load_n(&hip_uploads,__ATOMIC_RELAXED);
  *downloads=__atomic_load_n(&hip_downloads,__ATOMIC_RELAXED);
}

/* ------------------------------------------------------------- cache hooks */
static void ReleaseDeviceCopy(MagickCLCacheInfo info)
{
  if ((info->buffer != (cl_mem) NULL) && (hip_library_state > 0))
    (void) hip_library.DeviceFree(-1,(void *) info->buffer);
  info->buffer=(cl_mem) NULL;
}

/* only GetAuthenticOpenCLBuffer (cache.c:1259) calls this, and nothing in this build calls that */
MagickPrivate MagickCLCacheInfo AcquireMagickCLCacheInfo(
  MagickCLDevice magick_unused(device),Quantum *magick_unused(pixels),
  const MagickSizeType magick_unused(length))
{
  return((MagickCLCacheInfo) NULL);
}

/*
  The CPU is about to touch the pixels (CopyOpenCLBuffer, cache.c:5341-5353):
  make the host block current, then drop the device copy — the CPU may write.
*/
MagickPrivate MagickCLCacheInfo CopyMagickCLCacheInfo(MagickCLCacheInfo info)
{
  if (info == (MagickCLCacheInfo) NULL)
    return((MagickCLCacheInfo) NULL);
  if ((info->event_count != 0) && (info->buffer != (cl_mem) NULL) && (hip_library_state > 0))
    {
      (void) hip_library.Download(-1,(void *) info->pixels,(const void *) info->buffer,
        (size_t) info->length,NULL);
      (void) hip_library.Synchronize(-1,NULL);
      CountHipTransfer(0);
    }
  return(RelinquishMagickCLCacheInfo(info,MagickFalse));
}

MagickPrivate MagickCLCacheInfo RelinquishMagickCLCacheInfo(MagickCLCacheInfo info,
  const MagickBooleanType relinquish_pixels)
{
  if (info == (MagickCLCacheInfo) NULL)
    return((MagickCLCacheInfo) NULL);
  ReleaseDeviceCopy(info);
  if (relinquish_pixels != MagickFalse)
    info->pixels=(Quantum *) RelinquishAlignedMemory(info->pixels);   /* as cache.c:985-987 would */
  (void) RelinquishMagickMemory(info);
  return((MagickCLCacheInfo) NULL);
}

MagickPrivate void RetainOpenCLMemObject(cl_mem magick_unused(memobj))
{
}

MagickPrivate void OpenCLTerminus(void)
{
  if (hip_library_state > 0)
    hip_library.Terminus();
}

/* -------------------------------------------------------------- public API */
MagickExport MagickBooleanType GetOpenCLEnabled(void)
{
  return(hip_enabled);
}

/*
  Page-locked pixel caches.  cache.c:3754-3758 takes a memory cache's block from
  AcquireAlignedMemory; with these two installed behind it (SetMagickAlignedMemoryMethods,
  memory.c:1541) a block of 4 MiB or more is page-locked by the library (hipHostMalloc) and the
  cache moves to and from the device with one DMA transfer per direction instead of through the
  library's staging threads.  Installed when acceleration is switched on, so only images created
  afterwards are affected; blocks that predate the installation came from posix_memalign and are
  released with free(), like the smaller ones.  MAGICK_HIP_PINNED_CACHES=0 leaves the allocator
  alone.
*/
#define HipPinnedCacheExtent  ((size_t) 4 << 20)

static void *AcquireHipAlignedMemory(const size_t size,const size_t alignment)
{
  void
    *memory;

  if (size >= HipPinnedCacheExtent)
    {
      HipLibrary
        *library;

      library=AcquireHipLibrary();
      if (library != (HipLibrary *) NULL)
        {
          memory=library->HostAlloc(size);
          if (memory != NULL)
            return(memory);
        }
    }
  memory=NULL;
  if (posix_memalign(&memory,alignment < sizeof(void *) ? sizeof(void *) : alignment,
        size == 0 ? 1 : size) != 0)
    return(NULL);
  return(memory);
}

static void RelinquishHipAlignedMemory(void *memory)
{
  if (memory == NULL)
    return;
  if ((hip_library_state > 0) && (hip_library.HostFree(memory) != 0))
    return;
  free(memory);
}

MagickExport size_t GetMagickHipPinnedCacheExtent(void)
{
  return(hip_library_state > 0 ? hip_library.HostAllocatedBytes() : 0);
}

MagickExport MagickBooleanType SetOpenCLEnabled(const MagickBooleanType value)
{
  const char
    *pinned;

  pinned=getenv("MAGICK_HIP_PINNED_CACHES");
  if ((value != MagickFalse) && ((pinned == (const char *) NULL) || (*pinned != '0')))
    SetMagickAlignedMemoryMethods(AcquireHipAlignedMemory,RelinquishHipAlignedMemory);
  hip_enabled=value;
  if (hip_library_state > 0)
    (void) hip_library.SetEnabled(value != MagickFalse ? 1 : 0);
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
