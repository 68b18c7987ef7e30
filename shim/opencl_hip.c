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
  holds the device pointer, `event_count` the dirty flag, `device` the device
  the copy lives on and `events` the stream it was last used on; no cl_* call
  is made.

  Devices and queues.  The reference picks a device per operator call — the
  enabled device with the lowest score x (1 + calls in flight), RequestOpenCLDevice,
  opencl.c:3056-3102 — and hands out one of 16 in-order queues per device
  (AcquireOpenCLCommandQueue, opencl.c:656).  The same here, over the GPUs of the
  node: one MagickCLDevice record per HIP device (the reference's own struct,
  opencl-private.h:300-352: `command_queues` holds the device's HIP streams,
  `deviceID` nothing), AcquireHipQueue() = least-busy enabled device + the next
  of its streams round-robin.  An image that is resident on a device keeps that
  device AND its stream for every later operator (and so does each result
  derived from it), so a chain needs no cross-stream events; independent images
  from different host threads land on different devices and streams.  The
  public device API of MagickCore/opencl.h (GetOpenCLDevices, the getters,
  SetOpenCLDeviceEnabled, the kernel profile records) works on these records.
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
#include <unistd.h>
#include "MagickCore/string_.h"
#include "magickhip_shim.h"

#define HipMaxDevices  64
#define HipStreamsPerDevice  MAGICKCORE_OPENCL_COMMAND_QUEUES      /* 16, opencl-private.h:70 */

static HipLibrary hip_library;
static volatile int hip_library_state=0;      /* 0 = untried, 1 = ready, -1 = unavailable */
static MagickBooleanType hip_enabled = MagickTrue;
static size_t hip_uploads=0,hip_downloads=0;
static SemaphoreInfo *hip_library_semaphore=(SemaphoreInfo *) NULL;

/* the devices the calls are arbitrated over; NULL-terminated for GetOpenCLDevices */
static MagickCLDevice hip_devices[HipMaxDevices+1];
static int hip_device_physical[HipMaxDevices];
static size_t hip_device_calls[HipMaxDevices];      /* operator calls that ran on the device */
static size_t hip_number_devices=0;
static size_t hip_last_device=0;                    /* ties go round the devices */
static SemaphoreInfo *hip_devices_semaphore=(SemaphoreInfo *) NULL;   /* openCL_lock, opencl.c:3081 */
static size_t hip_pinned_budget=0;                  /* bytes of page-locked pixel caches at most */

static void InitializeHipDevices(void);

static void *Resolve(void *handle,const char *name,int *missing)
{
  void *symbol=dlsym(handle,name);
  if (symbol == NULL)
    (*missing)++;
  return(symbol);
}

static void LoadHipLibrary(void);

/* the library, loaded on first use, whatever the enable switch says (the device API needs it) */
static HipLibrary *LoadedHipLibrary(void)
{
  if (__atomic_load_n(&hip_library_state,__ATOMIC_ACQUIRE) > 0)
    return(&hip_library);
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
  return(hip_library_state > 0 ? &hip_library : (HipLibrary *) NULL);
}

MagickPrivate HipLibrary *AcquireHipLibrary(void)
{
  HipLibrary
    *library;

  if (hip_enabled == MagickFalse)
    return((HipLibrary *) NULL);
  library=LoadedHipLibrary();
  if ((library == (HipLibrary *) NULL) || (library->GetEnabled() == 0))
    return((HipLibrary *) NULL);
  return(library);
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
  MH_RESOLVE(DeviceCount,"MhDeviceCount");
  MH_RESOLVE(LogicalDeviceCount,"MhLogicalDeviceCount");
  MH_RESOLVE(GetDeviceInfo,"MhGetDeviceInfo");
  MH_RESOLVE(StreamCreate,"MhStreamCreate");
  MH_RESOLVE(DeviceAllocAsync,"MhDeviceAllocAsync");
  MH_RESOLVE(DeviceFreeAsync,"MhDeviceFreeAsync");
  MH_RESOLVE(SetProfileEnabled,"MhSetProfileEnabled");
  MH_RESOLVE(GetDeviceProfileRecords,"MhGetDeviceProfileRecords");
  MH_RESOLVE(Upload,"MhUpload");
  MH_RESOLVE(Download,"MhDownload");
  MH_RESOLVE(Synchronize,"MhSynchronize");
  MH_RESOLVE(HostAlloc,"MhHostAlloc");
  MH_RESOLVE(HostFree,"MhHostFree");
  MH_RESOLVE(HostAllocatedBytes,"MhHostAllocatedBytes");
  MH_RESOLVE(HostPinnedBytes,"MhHostPinnedBytes");
  MH_RESOLVE(BlurImage,"MagickHipBlurImage");
  MH_RESOLVE(UnsharpMaskImage,"MagickHipUnsharpMaskImage");
  MH_RESOLVE(ResizeImageWithFilter,"MagickHipResizeImageWithFilter");
  MH_RESOLVE(AcquireResizeFilterFromCallback,"MhAcquireResizeFilterFromCallback");
  MH_RESOLVE(DestroyResizeFilter,"MhDestroyResizeFilter");
  MH_RESOLVE(ContrastStretchImage,"MagickHipContrastStretchImage");
  MH_RESOLVE(EqualizeImage,"MagickHipEqualizeImage");
  MH_RESOLVE(ShardedImage,"MagickHipShardedImage");
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
  InitializeHipDevices();
  if (hip_number_devices == 0)
    {
      __atomic_store_n(&hip_library_state,-1,__ATOMIC_RELEASE);
      return;
    }
  __atomic_store_n(&hip_library_state,1,__ATOMIC_RELEASE);
}

/* ------------------------------------------------------- devices and queues */
/*
  One record per device the calls are arbitrated over: the GPUs of the node, or
  MAGICKHIP_LOGICAL_DEVICES of them mapped round-robin onto the GPUs present (how a one-GPU box
  rehearses the multi-GPU arbitration).  Runs once, under hip_library_semaphore.
*/
static void InitializeHipDevices(void)
{
  int
    logical,
    physical;

  ssize_t
    i;

  physical=hip_library.DeviceCount();
  logical=hip_library.LogicalDeviceCount();
  if ((physical <= 0) || (logical <= 0))
    return;
  if (logical > HipMaxDevices)
    logical=HipMaxDevices;
  ActivateSemaphoreInfo(&hip_devices_semaphore);
  for (i=0; i < (ssize_t) logical; i++)
  {
    char
      version[MagickPathExtent];

    MagickCLDevice
      device;

    MhDeviceInfo
      info;

    if (hip_library.GetDeviceInfo((int) (i % physical),&info) != MH_OK)
      break;
    device=(MagickCLDevice) AcquireCriticalMemory(sizeof(*device));
    (void) memset(device,0,sizeof(*device));
    device->name=ConstantString(info.name);
    device->platform_name=ConstantString("HIP");
    device->vendor_name=ConstantString("Advanced Micro Devices, Inc.");
    (void) FormatLocaleString(version,MagickPathExtent,"HIP %s (device %d)",info.architecture,
      (int) (i % physical));
    device->version=ConstantString(version);
    device->type=CL_DEVICE_TYPE_GPU;
    device->max_clock_frequency=(cl_uint) info.clock_mhz;
    device->max_compute_units=(cl_uint) info.compute_units;
    device->local_memory_size=(cl_ulong) info.local_memory;
    /* the reference's score is a benchmark time, lower = faster (opencl.c:1280-1310); a nominal
       one here: identical GPUs tie and the arbitration falls back to the calls in flight */
    device->score=1.0e6/(((double) info.compute_units*(double) info.clock_mhz)+1.0);
    device->enabled=MagickTrue;
    device->command_queues_index=0;                  /* the next stream to hand out */
    device->lock=AcquireSemaphoreInfo();
    hip_devices[i]=device;
    hip_device_physical[i]=(int) (i % physical);
    hip_device_calls[i]=0;
  }
  hip_number_devices=(size_t) i;
  hip_devices[hip_number_devices]=(MagickCLDevice) NULL;
}

static ssize_t HipDeviceIndex(const MagickCLDevice device)
{
  ssize_t
    i;

  for (i=0; i < (ssize_t) hip_number_devices; i++)
    if (hip_devices[i] == device)
      return(i);
  return(-1);
}

MagickPrivate int GetHipDevicePhysical(const MagickCLDevice device)
{
  ssize_t index=HipDeviceIndex(device);
  return(index < 0 ? -1 : hip_device_physical[index]);
}

/*
  RequestOpenCLDevice (opencl.c:3056-3102): the enabled device with the lowest
  score+score*requested; the scan starts behind the device chosen last, so equal devices with
  nothing in flight are taken in turn (a single thread that enqueues asynchronous calls one after
  the other has none "in flight" when it asks).  Then AcquireOpenCLCommandQueue (opencl.c:656):
  the next of the device's streams, created on first use.
*/
MagickPrivate MagickBooleanType AcquireHipQueue(HipLibrary *library,HipQueue *queue)
{
  double
    best_score,
    score;

  MagickCLDevice
    device;

  size_t
    k;

  ssize_t
    best,
    slot;

  queue->device=(MagickCLDevice) NULL;
  queue->physical=(-1);
  queue->stream=NULL;
  if (hip_number_devices == 0)
    return(MagickFalse);
  best=(-1);
  best_score=0.0;
  LockSemaphoreInfo(hip_devices_semaphore);
  for (k=0; k < hip_number_devices; k++)
  {
    size_t i=(hip_last_device+1+k) % hip_number_devices;
    if (hip_devices[i]->enabled == MagickFalse)
      continue;
    score=hip_devices[i]->score+hip_devices[i]->score*(double) hip_devices[i]->requested;
    if ((best < 0) || (score < best_score))
      {
        best=(ssize_t) i;
        best_score=score;
      }
  }
  if (best >= 0)
    {
      hip_devices[best]->requested++;
      hip_device_calls[best]++;
      hip_last_device=(size_t) best;
    }
  UnlockSemaphoreInfo(hip_devices_semaphore);
  if (best < 0)
    return(MagickFalse);
  device=hip_devices[best];
  LockSemaphoreInfo(device->lock);
  slot=device->command_queues_index;
  device->command_queues_index=(slot+1) % HipStreamsPerDevice;
  if (device->command_queues[slot] == (cl_command_queue) NULL)
    {
      void *stream=NULL;
      if (library->StreamCreate(hip_device_physical[best],&stream) == MH_OK)
        device->command_queues[slot]=(cl_command_queue) stream;
    }
  queue->stream=(void *) device->command_queues[slot];
  UnlockSemaphoreInfo(device->lock);
  queue->device=device;
  queue->physical=hip_device_physical[best];
  if (queue->stream == NULL)
    {
      ReleaseHipQueue(queue);              /* no stream: the CPU path runs */
      return(MagickFalse);
    }
  return(MagickTrue);
}

MagickPrivate void RetainHipQueue(MagickCLDevice device,void *stream,HipQueue *queue)
{
  ssize_t index=HipDeviceIndex(device);
  queue->device=device;
  queue->physical=index < 0 ? -1 : hip_device_physical[index];
  queue->stream=stream;
  if (index < 0)
    return;
  LockSemaphoreInfo(hip_devices_semaphore);
  device->requested++;
  hip_device_calls[index]++;
  UnlockSemaphoreInfo(hip_devices_semaphore);
}

/* ReleaseOpenCLDevice, opencl.c:2863-2869 */
MagickPrivate void ReleaseHipQueue(HipQueue *queue)
{
  if (queue->device == (MagickCLDevice) NULL)
    return;
  LockSemaphoreInfo(hip_devices_semaphore);
  if (queue->device->requested > 0)
    queue->device->requested--;
  UnlockSemaphoreInfo(hip_devices_semaphore);
  queue->device=(MagickCLDevice) NULL;
}

/* for tests and bench.py: how many operator calls a device took and how many of its streams
   have been handed out */
MagickExport size_t GetMagickHipDeviceStatistics(const size_t index,size_t *calls,size_t *streams)
{
  size_t
    n;

  ssize_t
    i;

  if (index >= hip_number_devices)
    return(hip_number_devices);
  LockSemaphoreInfo(hip_devices_semaphore);
  *calls=hip_device_calls[index];
  UnlockSemaphoreInfo(hip_devices_semaphore);
  n=0;
  LockSemaphoreInfo(hip_devices[index]->lock);
  for (i=0; i < HipStreamsPerDevice; i++)
    if (hip_devices[index]->command_queues[i] != (cl_command_queue) NULL)
      n++;
  UnlockSemaphoreInfo(hip_devices[index]->lock);
  *streams=n;
  return(hip_number_devices);
}

MagickPrivate size_t GetHipSpreadDevices(void)
{
  size_t
    i,
    n;

  if ((hip_library_state <= 0) || (hip_number_devices < 2))
    return(0);
  n=hip_number_devices;
  LockSemaphoreInfo(hip_devices_semaphore);
  for (i=0; i < hip_number_devices; i++)
    if (hip_devices[i]->enabled == MagickFalse)
      n=0;
  UnlockSemaphoreInfo(hip_devices_semaphore);
  return(n);
}

MagickPrivate void CountHipSpreadCall(size_t devices)
{
  size_t
    i;

  LockSemaphoreInfo(hip_devices_semaphore);
  for (i=0; (i < devices) && (i < hip_number_devices); i++)
    hip_device_calls[i]++;
  UnlockSemaphoreInfo(hip_devices_semaphore);
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
  /* back to the library's pool, behind whatever is enqueued on the copy's stream: no hipFree
     (a device-wide synchronisation) while other threads' kernels run */
  if ((info->buffer != (cl_mem) NULL) && (hip_library_state > 0))
    (void) hip_library.DeviceFreeAsync(GetHipDevicePhysical(info->device),(void *) info->buffer,
      (void *) info->events);
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
      /* behind the kernels that produced it: the copy's own stream */
      (void) hip_library.Download(GetHipDevicePhysical(info->device),(void *) info->pixels,
        (const void *) info->buffer,(size_t) info->length,(void *) info->events);
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
  size_t
    i;

  ssize_t
    j;

  if (hip_library_state <= 0)
    return;
  /*
    MhTerminus synchronises and destroys every stream MhStreamCreate handed out: forget the
    handles first, so that a later MagickCoreGenesis (or an operator call after Terminus) creates
    its streams afresh in AcquireHipQueue instead of launching on destroyed ones.
  */
  for (i=0; i < hip_number_devices; i++)
  {
    LockSemaphoreInfo(hip_devices[i]->lock);
    for (j=0; j < HipStreamsPerDevice; j++)
      hip_devices[i]->command_queues[j]=(cl_command_queue) NULL;
    hip_devices[i]->command_queues_index=0;
    UnlockSemaphoreInfo(hip_devices[i]->lock);
    LockSemaphoreInfo(hip_devices_semaphore);
    hip_devices[i]->requested=0;
    UnlockSemaphoreInfo(hip_devices_semaphore);
  }
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
  alone; MAGICK_HIP_PINNED_BUDGET=<bytes> caps the page-locked total (default: half the physical
  memory); beyond it, and for every block while the library is not loaded, posix_memalign.
*/
#define HipPinnedCacheExtent  ((size_t) 4 << 20)

static void *AcquireHipAlignedMemory(const size_t size,const size_t alignment)
{
  void
    *memory;

  /*
    Only with the library already up (SetOpenCLEnabled loads it: no dlopen / HIP start-up from
    inside an allocation), and only within the budget: every AcquireAlignedMemory /
    AcquireVirtualMemory block of this size comes here, pixel cache or not, and page-locked
    memory is a resource of the whole machine.
  */
  if ((size >= HipPinnedCacheExtent) && (hip_enabled != MagickFalse) &&
      (__atomic_load_n(&hip_library_state,__ATOMIC_ACQUIRE) > 0) &&
      (hip_library.HostPinnedBytes()+size <= hip_pinned_budget))
    {
      memory=hip_library.HostAlloc(size);
      if (memory != NULL)
        return(memory);
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
    *budget,
    *pinned;

  hip_enabled=value;
  pinned=getenv("MAGICK_HIP_PINNED_CACHES");
  if ((value != MagickFalse) && ((pinned == (const char *) NULL) || (*pinned != '0')))
    {
      /* load the library now, outside any allocation; the allocator only uses it once it is up */
      if (hip_pinned_budget == 0)
        {
          long pages=sysconf(_SC_PHYS_PAGES),page_size=sysconf(_SC_PAGESIZE);
          hip_pinned_budget=((pages > 0) && (page_size > 0)) ?
            (size_t) pages/2*(size_t) page_size : (size_t) 8 << 30;
          budget=getenv("MAGICK_HIP_PINNED_BUDGET");
          if (budget != (const char *) NULL)
            hip_pinned_budget=(size_t) strtoull(budget,(char **) NULL,10);
        }
      (void) LoadedHipLibrary();
      SetMagickAlignedMemoryMethods(AcquireHipAlignedMemory,RelinquishHipAlignedMemory);
    }
  if (hip_library_state > 0)
    (void) hip_library.SetEnabled(value != MagickFalse ? 1 : 0);
  return(hip_enabled);
}

/*
  The devices of the node (GetOpenCLDevices, opencl.c:1946-1964): one MagickCLDevice per GPU;
  the array is NULL-terminated and owned by MagickCore.
*/
MagickExport MagickCLDevice *GetOpenCLDevices(size_t *length,
  ExceptionInfo *magick_unused(exception))
{
  if (LoadedHipLibrary() == (HipLibrary *) NULL)
    {
      if (length != (size_t *) NULL)
        *length=0;
      return((MagickCLDevice *) NULL);
    }
  if (length != (size_t *) NULL)
    *length=hip_number_devices;
  return(hip_devices);
}

MagickExport const char *GetOpenCLDeviceName(const MagickCLDevice device)
{
  if (device == (MagickCLDevice) NULL)
    return((const char *) NULL);
  return(device->name);
}

MagickExport const char *GetOpenCLDeviceVendorName(const MagickCLDevice device)
{
  if (device == (MagickCLDevice) NULL)
    return((const char *) NULL);
  return(device->vendor_name);
}

MagickExport const char *GetOpenCLDeviceVersion(const MagickCLDevice device)
{
  if (device == (MagickCLDevice) NULL)
    return((const char *) NULL);
  return(device->version);
}

/*
  GetOpenCLKernelProfileRecords (opencl.c:2081-2100): the device's records, one per kernel, times
  in microseconds as RecordProfileData keeps them (opencl.c:2749-2751) — here the library's
  hipEvent records of that GPU (MhGetDeviceProfileRecords), gathered while
  SetOpenCLKernelProfileEnabled was on.  The array is rebuilt by every call and stays valid until
  the next call for the same device.
*/
static void RelinquishProfileRecords(MagickCLDevice device)
{
  size_t
    i;

  if (device->profile_records == (KernelProfileRecord *) NULL)
    return;
  for (i=0; device->profile_records[i] != (KernelProfileRecord) NULL; i++)
  {
    device->profile_records[i]->kernel_name=DestroyString(device->profile_records[i]->kernel_name);
    device->profile_records[i]=(KernelProfileRecord) RelinquishMagickMemory(
      device->profile_records[i]);
  }
  device->profile_records=(KernelProfileRecord *) RelinquishMagickMemory(device->profile_records);
}

MagickExport const KernelProfileRecord *GetOpenCLKernelProfileRecords(
  const MagickCLDevice device,size_t *length)
{
  MhKernelProfileRecord
    *records;

  size_t
    i,
    n;

  if (length != (size_t *) NULL)
    *length=0;
  if ((device == (MagickCLDevice) NULL) || (hip_library_state <= 0) ||
      (HipDeviceIndex(device) < 0))
    return((const KernelProfileRecord *) NULL);
  n=hip_library.GetDeviceProfileRecords(GetHipDevicePhysical(device),
    (MhKernelProfileRecord *) NULL,0);
  if (n == 0)
    return((const KernelProfileRecord *) NULL);
  records=(MhKernelProfileRecord *) AcquireQuantumMemory(n,sizeof(*records));
  if (records == (MhKernelProfileRecord *) NULL)
    return((const KernelProfileRecord *) NULL);
  i=hip_library.GetDeviceProfileRecords(GetHipDevicePhysical(device),records,n);
  if (i < n)
    n=i;
  LockSemaphoreInfo(device->lock);
  RelinquishProfileRecords(device);
  device->profile_records=(KernelProfileRecord *) AcquireQuantumMemory(n+1,
    sizeof(*device->profile_records));
  if (device->profile_records != (KernelProfileRecord *) NULL)
    {
      for (i=0; i < n; i++)
      {
        KernelProfileRecord record=(KernelProfileRecord) AcquireCriticalMemory(sizeof(*record));
        record->kernel_name=ConstantString(records[i].kernel_name);
        record->count=records[i].count;
        record->min=(unsigned long) (1000.0*records[i].min_ms+0.5);
        record->max=(unsigned long) (1000.0*records[i].max_ms+0.5);
        record->total=(unsigned long) (1000.0*records[i].total_ms+0.5);
        device->profile_records[i]=record;
      }
      device->profile_records[n]=(KernelProfileRecord) NULL;
      if (length != (size_t *) NULL)
        *length=n;
    }
  UnlockSemaphoreInfo(device->lock);
  records=(MhKernelProfileRecord *) RelinquishMagickMemory(records);
  return((const KernelProfileRecord *) device->profile_records);
}

MagickExport double GetOpenCLDeviceBenchmarkScore(const MagickCLDevice device)
{
  if (device == (MagickCLDevice) NULL)
    return(MAGICKCORE_OPENCL_UNDEFINED_SCORE);
  return(device->score);
}

MagickExport MagickCLDeviceType GetOpenCLDeviceType(const MagickCLDevice device)
{
  if (device == (MagickCLDevice) NULL)
    return(UndefinedCLDeviceType);
  if (device->type == CL_DEVICE_TYPE_GPU)
    return(GpuCLDeviceType);
  if (device->type == CL_DEVICE_TYPE_CPU)
    return(CpuCLDeviceType);
  return(UndefinedCLDeviceType);
}

MagickExport MagickBooleanType GetOpenCLDeviceEnabled(const MagickCLDevice device)
{
  if (device == (MagickCLDevice) NULL)
    return(MagickFalse);
  return(device->enabled);
}

/*
  SetOpenCLDeviceEnabled (opencl.c:3127-3134): the device mask of the arbitration.  An image that
  is resident on a device when it is switched off is brought back to the host by the next operator
  that wants it (accelerate_hip.c), and goes to another device from there.
*/
MagickExport void SetOpenCLDeviceEnabled(MagickCLDevice device,
  const MagickBooleanType value)
{
  if ((device == (MagickCLDevice) NULL) || (HipDeviceIndex(device) < 0))
    return;
  LockSemaphoreInfo(hip_devices_semaphore);
  device->enabled=value;
  UnlockSemaphoreInfo(hip_devices_semaphore);
}

/* SetOpenCLKernelProfileEnabled (opencl.c:3162-3168): hipEvent records around every kernel of
   the library while any device has it on */
MagickExport void SetOpenCLKernelProfileEnabled(MagickCLDevice device,
  const MagickBooleanType value)
{
  MagickBooleanType
    any;

  size_t
    i;

  if ((device == (MagickCLDevice) NULL) || (HipDeviceIndex(device) < 0))
    return;
  LockSemaphoreInfo(hip_devices_semaphore);
  device->profile_kernels=value;
  any=MagickFalse;
  for (i=0; i < hip_number_devices; i++)
    if (hip_devices[i]->profile_kernels != MagickFalse)
      any=MagickTrue;
  UnlockSemaphoreInfo(hip_devices_semaphore);
  if (hip_library_state > 0)
    (void) hip_library.SetProfileEnabled(any != MagickFalse ? 1 : 0);
}

#endif /* MAGICKCORE_OPENCL_SUPPORT */
