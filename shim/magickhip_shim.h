/*
  Shared between shim/accelerate_hip.c and shim/opencl_hip.c: the dlopen'ed
  libmagickhip.so entry points and the device-residency record kept in
  CacheInfo::opencl.
*/
#ifndef MAGICKHIP_SHIM_H
#define MAGICKHIP_SHIM_H

#include "magickhip.h"

typedef struct _HipLibrary
{
  void *handle;
  MhStatus (*Initialize)(void);
  void (*Terminus)(void);
  int (*GetEnabled)(void);
  int (*SetEnabled)(int);
  void (*InitImage)(MhImage *,void *,size_t,size_t,uint32_t,int,MhQuantumKind,MhMemoryKind);
  int (*DeviceCount)(void);
  int (*LogicalDeviceCount)(void);
  MhStatus (*GetDeviceInfo)(int,MhDeviceInfo *);
  MhStatus (*StreamCreate)(int,void **);
  MhStatus (*DeviceAllocAsync)(int,size_t,void *,void **);
  MhStatus (*DeviceFreeAsync)(int,void *,void *);
  int (*SetProfileEnabled)(int);
  size_t (*GetDeviceProfileRecords)(int,MhKernelProfileRecord *,size_t);
  MhStatus (*Upload)(int,void *,const void *,size_t,void *);
  MhStatus (*Download)(int,void *,const void *,size_t,void *);
  MhStatus (*Synchronize)(int,void *);
  void *(*HostAlloc)(size_t);
  int (*HostFree)(void *);
  size_t (*HostAllocatedBytes)(void);
  size_t (*HostPinnedBytes)(void);     /* + the spare blocks the library keeps page-locked */
  MhStatus (*BlurImage)(const MhImage *,MhImage *,double,double);
  MhStatus (*UnsharpMaskImage)(const MhImage *,MhImage *,double,double,double,double);
  MhStatus (*ResizeImageWithFilter)(const MhImage *,MhImage *,const MhResizeFilter *);
  MhResizeFilter *(*AcquireResizeFilterFromCallback)(MhResizeWeightFunction,void *,double);
  MhResizeFilter *(*DestroyResizeFilter)(MhResizeFilter *);
  MhStatus (*ContrastStretchImage)(MhImage *,double,double,int *);
  MhStatus (*EqualizeImage)(MhImage *);
  MhStatus (*ShardedImage)(const MhOperator *,size_t,const MhImage *,MhImage *,int,MhBatchReport *);
  MhStatus (*GrayscaleImage)(MhImage *,MhIntensityMethod);
  MhStatus (*FunctionImage)(MhImage *,MhFunction,size_t,const double *);
  MhStatus (*MotionBlurImageWithKernel)(const MhImage *,MhImage *,const double *,size_t,
    const ptrdiff_t *);
  MhStatus (*WaveletDenoiseImage)(const MhImage *,MhImage *,double,double);
  MhStatus (*DespeckleImage)(const MhImage *,MhImage *);
  MhStatus (*LocalContrastImage)(const MhImage *,MhImage *,double,double);
  MhStatus (*RotationalBlurImage)(const MhImage *,MhImage *,double);
  MhStatus (*ContrastImage)(MhImage *,int);
  MhStatus (*ModulateImage)(MhImage *,double,double,double,int);
  MhStatus (*MorphologyImage)(const MhImage *,MhImage *,MhMorphologyMethod,ptrdiff_t,
    const MhKernelInfo *,double);
  MhStatus (*MorphologyImageCompose)(const MhImage *,MhImage *,MhMorphologyMethod,ptrdiff_t,
    const MhKernelInfo *,double,MhMorphologyCompose);
  MhStatus (*TransformImageColorspace)(MhImage *,MhColorspace);
} HipLibrary;

/* NULL when the library, a GPU or the enable switch is missing: the caller runs the CPU path */
extern MagickPrivate HipLibrary *AcquireHipLibrary(void);

/*
  What one operator call runs on: a device picked by RequestHipDevice (the reference:
  RequestOpenCLDevice, opencl.c:3056-3102) and one of its streams (AcquireOpenCLCommandQueue,
  opencl.c:656).  `physical` is the HIP ordinal the library is called with; several logical
  devices may share one (MAGICKHIP_LOGICAL_DEVICES).
*/
typedef struct _HipQueue
{
  MagickCLDevice device;
  int physical;
  void *stream;
} HipQueue;

/* the enabled device with the least outstanding work + the next of its streams; MagickFalse when
   no device is enabled.  Every successful call is paired with ReleaseHipQueue. */
extern MagickPrivate MagickBooleanType AcquireHipQueue(HipLibrary *,HipQueue *);
/* the queue an image that is already resident on `device` keeps using (its own stream) */
extern MagickPrivate void RetainHipQueue(MagickCLDevice device,void *stream,HipQueue *);
extern MagickPrivate void ReleaseHipQueue(HipQueue *);
extern MagickPrivate int GetHipDevicePhysical(const MagickCLDevice device);

/* transfer counters, for tests (uploads / downloads of whole pixel caches) */
extern MagickPrivate void CountHipTransfer(int upload);
/* one big host-resident image over every device (row bands): the number of devices to use — all of
   them, when at least two exist and none has been switched off —, else 0; and the bookkeeping of
   such a call (every device's call counter) */
extern MagickPrivate size_t GetHipSpreadDevices(void);
extern MagickPrivate void CountHipSpreadCall(size_t devices);

#endif
