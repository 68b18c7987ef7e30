// Host runtime of libmagickhip.so: device discovery, enable/precision state,
// streams, the workspace pool, host<->HBM staging of pixel-cache buffers and
// hipEvent kernel profiling.  This replaces what MagickCore/opencl.c does for
// the reference's OpenCL path (device pick opencl.c:2289-2430, queues
// opencl.c:656, profiling opencl.c:2704) with a HIP-native equivalent; none of
// that file's structure is reused.
#include "mh_internal.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <strings.h>
#include <thread>
#include <unistd.h>

extern char **environ;

namespace mh {

// ------------------------------------------------------------------ errors
static thread_local char g_error[512] = "";

void set_error(const char *fmt,...)
{
  va_list ap;
  va_start(ap,fmt);
  vsnprintf(g_error,sizeof(g_error),fmt,ap);
  va_end(ap);
}

MhStatus fail(MhStatus status,const char *fmt,...)
{
  va_list ap;
  va_start(ap,fmt);
  vsnprintf(g_error,sizeof(g_error),fmt,ap);
  va_end(ap);
  return status;
}

// ------------------------------------------------------------------- state
struct PoolBlock { void *ptr; size_t bytes; hipStream_t stream; };

// Page-locked staging block for host->device table uploads; `ready` is recorded behind the
// copy that last read it, so the block is reusable once the event has completed.
struct StagingBlock { void *host; size_t bytes; hipEvent_t ready; };

struct DeviceState
{
  hipStream_t stream=nullptr;
  int compute_units=0;
  int lds_bytes=0;
  std::vector<StagingBlock> staging;
  std::multimap<size_t,PoolBlock> free_blocks;   // by capacity
  std::map<void *,size_t> live;                  // ptr -> capacity
  size_t cached_bytes=0;
};

struct Runtime
{
  std::once_flag once;
  MhStatus init_status=MH_NO_DEVICE;
  int ndevices=0;
  int default_device=0;
  int enabled=1;
  MhPrecision precision=MH_PRECISION_FAST;     // the drop-in's default (MAGICK_HIP_PRECISION=exact: bit-identical)
  std::vector<DeviceState> devices;
  std::mutex lock;
  // profiling
  int profiling=0;
  struct Pending { const char *name; int device; hipEvent_t start,stop; };
  std::vector<Pending> pending;
  struct Rec { unsigned long count=0; double min_ms=1e300,max_ms=0,total_ms=0; };
  std::map<std::pair<int,std::string>,Rec> records;     // (device, kernel)
  std::map<std::string,const char *> record_names;       // stable storage for returned names
  int logical_devices=0;
  // streams handed out by MhStreamCreate (destroyed by MhStreamDestroy or MhTerminus)
  std::vector<std::pair<int,hipStream_t>> caller_streams;
};

static Runtime &rt()
{
  static Runtime *r=new Runtime();   // intentionally leaked: safe at exit
  return *r;
}

// ----------------------------------------------------------------- options
// The environment is read ONCE, here: every MAGICKHIP_* / MAGICK_HIP_* variable goes into a
// table that option() reads from then on (getenv racing a setenv is undefined, and a library
// that re-reads the environment per call follows whatever the host process does to it).  Values
// are interned and never freed, so a pointer option() returned stays valid.
struct Options
{
  std::shared_mutex lock;
  std::map<std::string,const char *> values;
};
static Options &options() { static Options &o=*new Options; return o; }

// What a USER may set from the environment: MAGICK_HIP_* (the binding's variables: LIBRARY, DEVICE, PRECISION,
// PINNED_CACHES, ...) and the deployment knobs below.  Every other MAGICKHIP_* name in the sources selects between
// kernels for a test or an A/B measurement: those are reachable through MhSetOption (what tests/ and tools/ use)
// and, from the environment, only in a build with -DMH_DIAGNOSTIC.
static bool user_option(const std::string &name)
{
  static const char *const deployment[]={
    "MAGICKHIP_LOGICAL_DEVICES",      // logical devices mapped onto the node's GPUs (runtime.cpp)
    "MAGICKHIP_BANDED_MIN_BYTES",     // host frames from this size on go through the band pipeline (batch.cpp)
    "MAGICKHIP_BANDED_WORKERS",       // its threads per device
    "MAGICKHIP_NO_BANDED",            // ... or not at all
    "MAGICKHIP_PINNED_SPARE_BYTES",   // page-locked blocks kept for reuse
    "MAGICKHIP_TRANSFER_THREADS",     // staging threads of MhUpload / MhDownload
    "MAGICKHIP_HOST_COPY",            // how unpinned host memory moves
    "MAGICKHIP_RCCL"};                // 0: the histogram all-reduce through the host instead of RCCL
  if (name.compare(0,11,"MAGICK_HIP_") == 0)
    return true;
  for (const char *known : deployment)
    if (name == known)
      return true;
  return false;
}

static void load_options()
{
  Options &o=options();
  std::unique_lock<std::shared_mutex> guard(o.lock);
  for (char **e=environ; (e != nullptr) && (*e != nullptr); e++)
    {
      if ((strncmp(*e,"MAGICKHIP_",10) != 0) && (strncmp(*e,"MAGICK_HIP_",11) != 0))
        continue;
      const char *eq=strchr(*e,'=');
      if (eq == nullptr)
        continue;
      const std::string name(*e,(size_t) (eq-*e));
#ifndef MH_DIAGNOSTIC
      if (!user_option(name))
        continue;
#endif
      o.values[name]=strdup(eq+1);
    }
}

static const char *lookup_option(const char *name)
{
  Options &o=options();
  std::shared_lock<std::shared_mutex> guard(o.lock);
  auto it=o.values.find(name);
  return it == o.values.end() ? nullptr : it->second;
}

static void do_init();

const char *option(const char *name)
{
  std::call_once(rt().once,do_init);
  return lookup_option(name);
}

long option_long(const char *name,long fallback)
{
  const char *value=option(name);
  return value != nullptr ? atol(value) : fallback;
}

// the precision of the operator call this thread is in: MhImage::precision of the image the
// entry point was gated with, -1 = the library default
static thread_local int t_call_precision=-1;

void set_call_precision(const MhImage *image)
{
  t_call_precision=((image != nullptr) && (image->precision != 0)) ?
    (image->precision == MH_IMAGE_PRECISION(MH_PRECISION_FAST) ? (int) MH_PRECISION_FAST : (int) MH_PRECISION_EXACT) : -1;
}

static void do_init()
{
  Runtime &r=rt();
  load_options();
  const char *env=lookup_option("MAGICK_HIP_DEVICE");
  if ((env != nullptr) && ((strcasecmp(env,"off") == 0) ||
      (strcasecmp(env,"false") == 0) || (strcasecmp(env,"cpu") == 0)))
    r.enabled=0;
  env=lookup_option("MAGICK_HIP_PRECISION");
  if ((env != nullptr) && (strcasecmp(env,"fast") == 0))
    r.precision=MH_PRECISION_FAST;
  if ((env != nullptr) && (strcasecmp(env,"exact") == 0))
    r.precision=MH_PRECISION_EXACT;
  int n=0;
  hipError_t err=hipGetDeviceCount(&n);
  if ((err != hipSuccess) || (n <= 0))
    {
      r.ndevices=0;
      r.init_status=MH_NO_DEVICE;
      set_error("no HIP device: %s",err != hipSuccess ? hipGetErrorString(err) :
        "device count is 0");
      (void) hipGetLastError();
      return;
    }
  r.ndevices=n;
  r.devices.resize((size_t) n);
  env=lookup_option("MAGICK_HIP_DEVICE");
  if ((env != nullptr) && (env[0] >= '0') && (env[0] <= '9'))
    {
      int d=atoi(env);
      if (d < n)
        r.default_device=d;
    }
  r.logical_devices=n;
  env=lookup_option("MAGICKHIP_LOGICAL_DEVICES");
  if ((env != nullptr) && (atoi(env) > n))
    r.logical_devices=atoi(env) > 64 ? 64 : atoi(env);
  r.init_status=MH_OK;
}

MhStatus runtime_ready()
{
  Runtime &r=rt();
  std::call_once(r.once,do_init);
  if (r.enabled == 0)
    return fail(MH_DISABLED,"magickhip is disabled");
  if (r.init_status != MH_OK)
    return fail(r.init_status,"no usable HIP device");
  return MH_OK;
}

int default_device() { return rt().default_device; }
int device_count() { Runtime &r=rt(); std::call_once(r.once,do_init); return r.ndevices; }
int logical_device_count() { Runtime &r=rt(); std::call_once(r.once,do_init); return r.logical_devices; }
MhPrecision precision()
{
  return t_call_precision >= 0 ? (MhPrecision) t_call_precision : rt().precision;
}

int compute_units(int device)
{
  Runtime &r=rt();
  std::call_once(r.once,do_init);
  if ((device < 0) || (device >= r.ndevices))
    return 256;
  std::lock_guard<std::mutex> guard(r.lock);
  DeviceState &d=r.devices[(size_t) device];
  if (d.compute_units == 0)
    {
      int n=0;
      if ((hipDeviceGetAttribute(&n,hipDeviceAttributeMultiprocessorCount,device) != hipSuccess) ||
          (n <= 0))
        {
          (void) hipGetLastError();
          n=256;
        }
      d.compute_units=n;
    }
  return d.compute_units;
}

// the LDS a workgroup may ask for (hipDeviceAttributeMaxSharedMemoryPerBlock; 160 KiB on gfx950): the
// one-launch resize forms size their workgroups-per-CU on it and decline on a smaller part
int lds_bytes_per_workgroup(int device)
{
  Runtime &r=rt();
  std::call_once(r.once,do_init);
  if ((device < 0) || (device >= r.ndevices))
    return 65536;
  std::lock_guard<std::mutex> guard(r.lock);
  DeviceState &d=r.devices[(size_t) device];
  if (d.lds_bytes == 0)
    {
      int n=0;
      if ((hipDeviceGetAttribute(&n,hipDeviceAttributeMaxSharedMemoryPerBlock,device) != hipSuccess) || (n <= 0))
        {
          (void) hipGetLastError();
          n=65536;
        }
      d.lds_bytes=n;
    }
  return d.lds_bytes;
}

hipStream_t library_stream(int device)
{
  Runtime &r=rt();
  std::lock_guard<std::mutex> guard(r.lock);
  DeviceState &d=r.devices[(size_t) device];
  if (d.stream == nullptr)
    {
      int prev=0;
      (void) hipGetDevice(&prev);
      (void) hipSetDevice(device);
      if (hipStreamCreateWithFlags(&d.stream,hipStreamNonBlocking) != hipSuccess)
        d.stream=nullptr;
      (void) hipSetDevice(prev);
    }
  return d.stream;
}

// -------------------------------------------------------------------- pool
static constexpr size_t kPoolGranule = 2u<<20;       // 2 MiB
static constexpr size_t kPoolSmall = 256;            // alignment of tiny tables

MhStatus pool_alloc(int device,size_t bytes,hipStream_t stream,void **ptr)
{
  Runtime &r=rt();
  if (bytes == 0)
    bytes=kPoolSmall;
  size_t capacity=bytes <= (64u<<10) ? ((bytes+kPoolSmall-1)/kPoolSmall)*kPoolSmall :
    ((bytes+kPoolGranule-1)/kPoolGranule)*kPoolGranule;
  bool reused=false;
  hipStream_t reused_from=nullptr;
  {
    std::lock_guard<std::mutex> guard(r.lock);
    DeviceState &d=r.devices[(size_t) device];
    auto it=d.free_blocks.lower_bound(capacity);
    // accept a cached block up to 25% larger than needed
    if ((it != d.free_blocks.end()) && (it->first <= capacity+capacity/4+kPoolGranule))
      {
        PoolBlock b=it->second;
        d.free_blocks.erase(it);
        d.cached_bytes-=b.bytes;
        d.live[b.ptr]=b.bytes;
        *ptr=b.ptr;
        reused_from=b.stream;
        reused=true;
      }
  }
  if (reused)
    {
      // last used on another stream: order behind it — outside the runtime lock, so that the
      // workers of the other devices and streams keep allocating meanwhile
      if (reused_from != stream)
        if (hipStreamSynchronize(reused_from) != hipSuccess)
          (void) hipGetLastError();      // never leave a sticky error for the next launch check
      return MH_OK;
    }
  int prev=0;
  (void) hipGetDevice(&prev);
  if (prev != device)
    (void) hipSetDevice(device);
  void *p=nullptr;
  hipError_t err=hipMalloc(&p,capacity);
  if (err != hipSuccess)
    {
      (void) hipGetLastError();
      pool_trim();
      err=hipMalloc(&p,capacity);
    }
  if (prev != device)
    (void) hipSetDevice(prev);
  if (err != hipSuccess)
    {
      (void) hipGetLastError();
      return fail(MH_OUT_OF_MEMORY,"hipMalloc(%zu) failed: %s",capacity,
        hipGetErrorString(err));
    }
  {
    std::lock_guard<std::mutex> guard(r.lock);
    r.devices[(size_t) device].live[p]=capacity;
  }
  *ptr=p;
  return MH_OK;
}

void pool_free(int device,void *ptr,hipStream_t stream)
{
  if (ptr == nullptr)
    return;
  Runtime &r=rt();
  std::lock_guard<std::mutex> guard(r.lock);
  DeviceState &d=r.devices[(size_t) device];
  auto it=d.live.find(ptr);
  if (it == d.live.end())
    return;
  PoolBlock b{ptr,it->second,stream};
  d.live.erase(it);
  d.free_blocks.emplace(b.bytes,b);
  d.cached_bytes+=b.bytes;
}

void pool_trim()
{
  Runtime &r=rt();
  std::vector<std::pair<int,void *>> victims;
  {
    std::lock_guard<std::mutex> guard(r.lock);
    for (size_t i=0; i < r.devices.size(); i++)
      {
        for (auto &kv : r.devices[i].free_blocks)
          victims.emplace_back((int) i,kv.second.ptr);
        r.devices[i].free_blocks.clear();
        r.devices[i].cached_bytes=0;
      }
  }
  if (victims.empty())
    return;
  (void) hipDeviceSynchronize();
  for (auto &v : victims)
    (void) hipFree(v.second);
}

// Tables above this size go through page-locked staging: a hipMemcpyAsync from pageable
// memory makes the runtime wait for the stream before it returns (measured: a resize call
// blocked for the whole GPU time of the previous one), which serialises host and device.
static constexpr size_t kStageDirect = 2048;
static constexpr size_t kStageGranule = 64u<<10;

static bool staging_acquire(int device,size_t bytes,StagingBlock *out)
{
  Runtime &r=rt();
  {
    std::lock_guard<std::mutex> guard(r.lock);
    std::vector<StagingBlock> &list=r.devices[(size_t) device].staging;
    for (size_t i=0; i < list.size(); i++)
      if ((list[i].bytes >= bytes) && (list[i].bytes <= 4*bytes+kStageGranule) &&
          (hipEventQuery(list[i].ready) == hipSuccess))
        {
          *out=list[i];
          list.erase(list.begin()+(ptrdiff_t) i);
          return true;
        }
    (void) hipGetLastError();            // hipErrorNotReady from the queries
  }
  StagingBlock b{nullptr,((bytes+kStageGranule-1)/kStageGranule)*kStageGranule,nullptr};
  if (hipHostMalloc(&b.host,b.bytes,hipHostMallocDefault) != hipSuccess)
    {
      (void) hipGetLastError();
      return false;
    }
  if (hipEventCreateWithFlags(&b.ready,hipEventDisableTiming) != hipSuccess)
    {
      (void) hipGetLastError();
      (void) hipHostFree(b.host);
      return false;
    }
  *out=b;
  return true;
}

static void staging_release(int device,const StagingBlock &b)
{
  Runtime &r=rt();
  std::lock_guard<std::mutex> guard(r.lock);
  r.devices[(size_t) device].staging.push_back(b);
}

static void staging_trim()
{
  Runtime &r=rt();
  std::vector<StagingBlock> victims;
  {
    std::lock_guard<std::mutex> guard(r.lock);
    for (DeviceState &d : r.devices)
      {
        victims.insert(victims.end(),d.staging.begin(),d.staging.end());
        d.staging.clear();
      }
  }
  for (StagingBlock &b : victims)
    {
      (void) hipEventSynchronize(b.ready);
      (void) hipEventDestroy(b.ready);
      (void) hipHostFree(b.host);
    }
}

MhStatus upload_table(Temp &dst,int device,hipStream_t stream,const void *host,size_t bytes)
{
  MH_TRY(dst.alloc(device,bytes,stream));
  StagingBlock block;
  if ((bytes > kStageDirect) && staging_acquire(device,bytes,&block))
    {
      memcpy(block.host,host,bytes);
      hipError_t err=hipMemcpyAsync(dst.ptr,block.host,bytes,hipMemcpyHostToDevice,stream);
      if (err == hipSuccess)
        err=hipEventRecord(block.ready,stream);
      staging_release(device,block);
      MH_HIP(err);
      return MH_OK;
    }
  // small pageable source: the runtime stages it before returning, so `host` may be
  // released by the caller right away.
  MH_HIP(hipMemcpyAsync(dst.ptr,host,bytes,hipMemcpyHostToDevice,stream));
  return MH_OK;
}

namespace {
// the device block of a shared table: freed when the cache AND every caller that is about to
// launch a kernel on it have let go (hipFree then waits for whatever is enqueued)
struct SharedBlock
{
  void *ptr=nullptr;
  hipEvent_t ready=nullptr;
  ~SharedBlock()
  {
    if (ptr != nullptr)
      (void) hipFree(ptr);
    if (ready != nullptr)
      (void) hipEventDestroy(ready);
  }
};
struct SharedTable
{
  int device;
  std::vector<unsigned char> content;
  std::shared_ptr<SharedBlock> block;
  hipStream_t stream;
};
// (never destroyed: the blocks belong to a runtime that may be gone at process exit)
std::mutex &shared_tables_lock() { static std::mutex &m=*new std::mutex; return m; }
std::vector<SharedTable> &shared_tables() { static std::vector<SharedTable> &v=*new std::vector<SharedTable>; return v; }
}

void release_shared_tables()
{
  std::lock_guard<std::mutex> guard(shared_tables_lock());
  shared_tables().clear();
}

MhStatus shared_table(int device,hipStream_t stream,const void *host,size_t bytes,const void **device_ptr,
  std::shared_ptr<void> *keep)
{
  typedef SharedTable Entry;
  constexpr size_t kEntries=32;
  std::vector<Entry> &entries=shared_tables();
  std::lock_guard<std::mutex> guard(shared_tables_lock());
  for (size_t i=0; i < entries.size(); i++)
    if ((entries[i].device == device) && (entries[i].content.size() == bytes) &&
        (memcmp(entries[i].content.data(),host,bytes) == 0))
      {
        Entry hit=std::move(entries[i]);
        entries.erase(entries.begin()+(ptrdiff_t) i);
        entries.insert(entries.begin(),std::move(hit));
        if (entries[0].stream != stream)
          MH_HIP(hipStreamWaitEvent(stream,entries[0].block->ready,0));
        *device_ptr=entries[0].block->ptr;
        if (keep != nullptr)
          *keep=entries[0].block;
        return MH_OK;
      }
  DeviceGuard device_guard;
  MH_HIP(device_guard.enter(device));
  Entry e;
  e.device=device;
  e.content.assign(static_cast<const unsigned char *>(host),static_cast<const unsigned char *>(host)+bytes);
  e.block=std::make_shared<SharedBlock>();
  e.stream=stream;
  MH_HIP(hipMalloc(&e.block->ptr,bytes < 256 ? 256 : bytes));
  // the entry owns the bytes: the pageable copy may be staged whenever the runtime likes
  MH_HIP(hipMemcpyAsync(e.block->ptr,e.content.data(),bytes,hipMemcpyHostToDevice,stream));
  MH_HIP(hipEventCreateWithFlags(&e.block->ready,hipEventDisableTiming));
  MH_HIP(hipEventRecord(e.block->ready,stream));
  *device_ptr=e.block->ptr;
  if (keep != nullptr)
    *keep=e.block;
  entries.insert(entries.begin(),std::move(e));
  if (entries.size() > kEntries)
    entries.pop_back();            // the block goes when its last holder lets go (SharedBlock)
  return MH_OK;
}

MhStatus TableBundle::upload(int device,hipStream_t stream)
{
  if (total_ == 0)
    return MH_OK;
  MH_TRY(block_.alloc(device,total_,stream));
  StagingBlock block;
  if (staging_acquire(device,total_,&block))
    {
      for (const Part &part : parts_)
        memcpy(static_cast<char *>(block.host)+part.offset,part.host,part.bytes);
      hipError_t err=hipMemcpyAsync(block_.ptr,block.host,total_,hipMemcpyHostToDevice,stream);
      if (err == hipSuccess)
        err=hipEventRecord(block.ready,stream);
      staging_release(device,block);
      MH_HIP(err);
      return MH_OK;
    }
  for (const Part &part : parts_)
    MH_HIP(hipMemcpyAsync(static_cast<char *>(block_.ptr)+part.offset,part.host,part.bytes,
      hipMemcpyHostToDevice,stream));
  return MH_OK;
}

// Whole-image transfers between a pageable host block (the pixel cache) and device memory.
// Page-locking the block in place (hipHostRegister) costs more than the transfer itself for a
// one-shot call (measured: 8192^2 RGBA Q16 BlurImage on host buffers 58.8 ms against 44.1 ms), so the
// block is moved in 4 MiB pieces through page-locked staging buffers: a few threads copy
// pieces between the block and their two staging buffers while the DMA engine moves the
// previous ones.  Everything is enqueued on `stream`, so uploads are ordered before the
// kernels that follow and downloads after the kernels that precede.
static constexpr size_t kPiece = 4u<<20;

static MhStatus transfer_slice(int device,hipStream_t stream,char *dev,char *host,size_t bytes,
  size_t first,size_t stride,bool upload)
{
  DeviceGuard guard;        // slice 0 runs on the caller's own thread
  if (guard.enter(device) != hipSuccess)
    return fail(MH_DEVICE_ERROR,"hipSetDevice(%d) failed",device);
  StagingBlock block[2];
  int have=0;
  for (; have < 2; have++)
    if (!staging_acquire(device,kPiece,&block[have]))
      break;
  MhStatus status=MH_OK;
  if (have < 2)
    status=fail(MH_DEVICE_ERROR,"cannot allocate page-locked staging memory");
  size_t pending_off=0,pending_len=0;
  int pending=-1,turn=0;
  for (size_t off=first*kPiece; (status == MH_OK) && (off < bytes); off+=stride*kPiece)
    {
      const size_t len=bytes-off < kPiece ? bytes-off : kPiece;
      StagingBlock &b=block[turn];
      hipError_t err=hipEventSynchronize(b.ready);          // the buffer's last transfer is done
      if ((err == hipSuccess) && upload)
        {
          memcpy(b.host,host+off,len);
          err=hipMemcpyAsync(dev+off,b.host,len,hipMemcpyHostToDevice,stream);
        }
      else if (err == hipSuccess)
        err=hipMemcpyAsync(b.host,dev+off,len,hipMemcpyDeviceToHost,stream);
      if (err == hipSuccess)
        err=hipEventRecord(b.ready,stream);
      if ((err == hipSuccess) && !upload && (pending >= 0))
        {
          // while this piece is in flight, hand the previous one to the caller's block
          err=hipEventSynchronize(block[pending].ready);
          if (err == hipSuccess)
            memcpy(host+pending_off,block[pending].host,pending_len);
        }
      if (err != hipSuccess)
        status=fail(MH_DEVICE_ERROR,"host transfer: %s",hipGetErrorString(err));
      pending=turn;
      pending_off=off;
      pending_len=len;
      turn^=1;
    }
  if ((status == MH_OK) && !upload && (pending >= 0))
    {
      if (hipEventSynchronize(block[pending].ready) != hipSuccess)
        status=fail(MH_DEVICE_ERROR,"host transfer: download failed");
      else
        memcpy(host+pending_off,block[pending].host,pending_len);
    }
  for (int i=0; i < have; i++)
    staging_release(device,block[i]);
  return status;
}

// upload: returns once every piece is enqueued (the host block may be reused right away);
// download: returns once the host block holds the data
MhStatus transfer_image(int device,hipStream_t stream,void *dev,void *host,size_t bytes,bool upload)
{
  const size_t pieces=(bytes+kPiece-1)/kPiece;
  size_t workers=std::thread::hardware_concurrency()/4;
  workers=workers > 4 ? 4 : (workers < 1 ? 1 : workers);      // 1/2/4/8/16 threads: 75/54/44/47/48 ms
  workers=workers > pieces ? pieces : workers;
  if (const char *e=option("MAGICKHIP_TRANSFER_THREADS"))
    {
      const long n=atol(e);
      workers=n < 1 ? 1 : (n > 32 ? 32 : (size_t) n);
    }
  if (workers <= 1)
    return transfer_slice(device,stream,static_cast<char *>(dev),static_cast<char *>(host),bytes,0,1,upload);
  std::vector<MhStatus> status(workers,MH_OK);
  std::vector<std::thread> pool;
  for (size_t t=1; t < workers; t++)
    pool.emplace_back([&,t]()
    {
      status[t]=transfer_slice(device,stream,static_cast<char *>(dev),static_cast<char *>(host),bytes,
        t,workers,upload);
    });
  status[0]=transfer_slice(device,stream,static_cast<char *>(dev),static_cast<char *>(host),bytes,0,
    workers,upload);
  for (std::thread &t : pool)
    t.join();
  // error text is per thread: restate a worker's failure for the caller's thread
  for (size_t t=0; t < workers; t++)
    if (status[t] != MH_OK)
      return t == 0 ? status[t] : fail(status[t],"host transfer: a staging thread failed (%s)",
        upload ? "upload" : "download");
  return MH_OK;
}

// ------------------------------------------------------------ image checks
MhStatus validate_image(const MhImage *image,const char *what)
{
  if (image == nullptr)
    return fail(MH_BAD_ARGUMENT,"%s: null image",what);
  if ((image->pixels == nullptr) || (image->columns == 0) || (image->rows == 0))
    return fail(MH_BAD_ARGUMENT,"%s: empty image",what);
  if ((image->number_channels == 0) || (image->number_channels > MH_MAX_CHANNELS))
    return fail(MH_UNSUPPORTED,"%s: %u channels (gate admits 1..%d)",what,
      image->number_channels,MH_MAX_CHANNELS);
  if ((image->quantum != MH_QUANTUM_U16) && (image->quantum != MH_QUANTUM_F32))
    return fail(MH_BAD_ARGUMENT,"%s: unknown quantum kind %u",what,image->quantum);
  if ((image->memory != MH_MEMORY_HOST) && (image->memory != MH_MEMORY_DEVICE))
    return fail(MH_BAD_ARGUMENT,"%s: unknown memory kind %u",what,image->memory);
  if ((image->alpha_offset >= (int32_t) image->number_channels))
    return fail(MH_BAD_ARGUMENT,"%s: alpha offset out of range",what);
  if ((image->columns > 0x7fffffffu) || (image->rows > 0x7fffffffu))
    return fail(MH_UNSUPPORTED,"%s: geometry exceeds 2^31",what);
  if (image->precision > MH_IMAGE_PRECISION(MH_PRECISION_FAST))
    return fail(MH_BAD_ARGUMENT,"%s: MhImage::precision %u (0, or MH_IMAGE_PRECISION of a mode: fill the descriptor with MhInitImage)",
      what,image->precision);
  return MH_OK;
}

int resolve_device(const MhImage *image)
{
  if ((image != nullptr) && (image->device >= 0) && (image->device < device_count()))
    return image->device;
  return default_device();
}

hipStream_t resolve_stream(const MhImage *image,int device)
{
  if ((image != nullptr) && (image->memory == MH_MEMORY_DEVICE))
    return (hipStream_t) image->stream;     // NULL => the device's null stream
  return library_stream(device);
}

Roles channel_roles(const MhImage *src,const MhImage *dst)
{
  Roles roles;
  const MhImage *t=dst != nullptr ? dst : src;
  roles.alpha=src->alpha_offset;
  for (uint32_t c=0; c < src->number_channels; c++)
    {
      uint32_t st=src->channel_traits[c];
      uint32_t dt=t->channel_traits[c];
      if ((st == MH_TRAIT_UNDEFINED) || (dt == MH_TRAIT_UNDEFINED) ||
          ((st & MH_TRAIT_COPY) != 0))
        roles.copy_mask|=1u<<c;
      else
        roles.update_mask|=1u<<c;
    }
  // morphology.c:2743-2744 / :2929-2930: alpha weighting needs the image to
  // have an active alpha and the destination channel to carry Blend.
  bool any_blend=false;
  for (uint32_t c=0; c < src->number_channels; c++)
    if ((t->channel_traits[c] & MH_TRAIT_BLEND) != 0)
      any_blend=true;
  roles.blend=((src->alpha_trait & MH_TRAIT_BLEND) != 0) && any_blend &&
    (roles.alpha >= 0);
  return roles;
}

// ---------------------------------------------------------------- Resident
Resident::~Resident()
{
  if (registered_ && (image_ != nullptr))
    (void) hipHostUnregister(image_->pixels);
}

MhStatus Resident::open(const MhImage *image,int mode,hipStream_t stream_hint,int device_hint)
{
  image_=image;
  mode_=mode;
  view.columns=image->columns;
  view.rows=image->rows;
  view.channels=(int) image->number_channels;
  view.quantum=(MhQuantumKind) image->quantum;
  if (image->memory == MH_MEMORY_DEVICE)
    {
      view.device=resolve_device(image);
      view.stream=(hipStream_t) image->stream;
      view.pixels=image->pixels;
      return MH_OK;
    }
  view.device=device_hint >= 0 ? device_hint : resolve_device(image);
  view.stream=stream_hint != nullptr ? stream_hint : library_stream(view.device);
  MH_TRY(temp_.alloc(view.device,view.bytes(),view.stream));
  view.pixels=temp_.ptr;
  staged_=true;
  // MAGICKHIP_HOST_COPY=register: page-lock the pixel-cache block in place instead (cache.c:
  // 3754-3758 allocates it with AcquireAlignedMemory, so that is legal) — slower for one call
  const char *how=option("MAGICKHIP_HOST_COPY");
  if ((how != nullptr) && (strcasecmp(how,"register") == 0))
    {
      if (hipHostRegister(image->pixels,view.bytes(),hipHostRegisterDefault) == hipSuccess)
        registered_=true;
      else
        (void) hipGetLastError();
      if ((mode == 0) || (mode == 2))
        MH_HIP(hipMemcpyAsync(view.pixels,image->pixels,view.bytes(),
          hipMemcpyHostToDevice,view.stream));
      return MH_OK;
    }
  // a pixel cache that is page-locked already (MhHostAlloc): one DMA transfer, no staging
  if (host_block_is_pinned(image->pixels,view.bytes()))
    {
      if ((mode == 0) || (mode == 2))
        MH_HIP(hipMemcpyAsync(view.pixels,image->pixels,view.bytes(),hipMemcpyHostToDevice,view.stream));
      return MH_OK;
    }
  pipelined_=true;
  if ((mode == 0) || (mode == 2))
    MH_TRY(transfer_image(view.device,view.stream,view.pixels,image->pixels,view.bytes(),true));
  return MH_OK;
}

MhStatus Resident::commit()
{
  if (!staged_)
    return MH_OK;
  if ((mode_ == 1) || (mode_ == 2))
    {
      if (pipelined_)
        MH_TRY(transfer_image(view.device,view.stream,view.pixels,image_->pixels,view.bytes(),false));
      else
        MH_HIP(hipMemcpyAsync(image_->pixels,view.pixels,view.bytes(),
          hipMemcpyDeviceToHost,view.stream));
    }
  MH_HIP(hipStreamSynchronize(view.stream));
  return MH_OK;
}

// --------------------------------------------------------------- profiling
ProfileScope::ProfileScope(const char *n,hipStream_t s) : name(n), stream(s)
{
  Runtime &r=rt();
  if (r.profiling == 0)
    return;
  if ((hipEventCreate(&start) != hipSuccess) || (hipEventCreate(&stop) != hipSuccess))
    return;
  on=true;
  if (hipGetDevice(&device) != hipSuccess)
    device=0;
  (void) hipEventRecord(start,stream);
}

ProfileScope::~ProfileScope()
{
  if (!on)
    return;
  (void) hipEventRecord(stop,stream);
  Runtime &r=rt();
  std::lock_guard<std::mutex> guard(r.lock);
  r.pending.push_back({name,device,start,stop});
}

static void drain_profile()
{
  Runtime &r=rt();
  std::vector<Runtime::Pending> pending;
  {
    std::lock_guard<std::mutex> guard(r.lock);
    pending.swap(r.pending);
  }
  for (auto &p : pending)
    {
      float ms=0.0f;
      if ((hipEventSynchronize(p.stop) == hipSuccess) &&
          (hipEventElapsedTime(&ms,p.start,p.stop) == hipSuccess))
        {
          std::lock_guard<std::mutex> guard(r.lock);
          Runtime::Rec &rec=r.records[std::make_pair(p.device,std::string(p.name))];
          rec.count++;
          rec.total_ms+=ms;
          if (ms < rec.min_ms) rec.min_ms=ms;
          if (ms > rec.max_ms) rec.max_ms=ms;
        }
      (void) hipEventDestroy(p.start);
      (void) hipEventDestroy(p.stop);
    }
}

} // namespace mh

// ===================================================================== C ABI
using namespace mh;

// ------------------------------------------------------------ page-locked pixel caches
namespace mh {
struct PinnedBlocks
{
  std::mutex lock;
  std::map<const char *,size_t> blocks;         // base -> bytes (handed out)
  size_t total=0;
  // released blocks kept for the next pixel cache of that size: page-locking 537 MB costs ~25 ms,
  // more than the BlurImage it is allocated for (CloneImage -> AcquireAlignedMemory per result)
  std::multimap<size_t,void *> spare;           // bytes -> base
  size_t spare_total=0;
};
static PinnedBlocks &pinned_blocks() { static PinnedBlocks &p=*new PinnedBlocks; return p; }

// [block, block+bytes) lies inside a block of MhHostAlloc's
bool host_block_is_pinned(const void *block,size_t bytes)
{
  PinnedBlocks &p=pinned_blocks();
  std::lock_guard<std::mutex> guard(p.lock);
  if (p.blocks.empty())
    return false;
  const char *at=static_cast<const char *>(block);
  auto it=p.blocks.upper_bound(at);
  if (it == p.blocks.begin())
    return false;
  --it;
  return (at >= it->first) && (at+bytes <= it->first+it->second);
}
} // namespace mh

static void release_spare_pinned_blocks(size_t keep_bytes);

extern "C" {

MH_API MhStatus MhInitialize(void)
{
  Runtime &r=rt();
  std::call_once(r.once,do_init);
  return r.init_status;
}

MH_API void MhTerminus(void)
{
  Runtime &r=rt();
  if (r.init_status != MH_OK)
    return;
  drain_profile();
  release_rccl_communicators();
  release_shared_tables();
  release_resize_tables();
  release_resize_mfma_plans();
  release_resize_stream_plans();
  pool_trim();
  staging_trim();
  release_color_tables();
  release_batch_streams();
  release_spare_pinned_blocks(0);
  std::vector<std::pair<int,hipStream_t>> streams;
  {
    std::lock_guard<std::mutex> guard(r.lock);
    streams.swap(r.caller_streams);
  }
  for (auto &entry : streams)
    {
      DeviceGuard device;
      if (device.enter(entry.first) == hipSuccess)
        {
          (void) hipStreamSynchronize(entry.second);
          (void) hipStreamDestroy(entry.second);
        }
    }
}

MH_API int MhDeviceCount(void) { return device_count(); }

MH_API MhStatus MhSetDevice(int device)
{
  if ((device < 0) || (device >= device_count()))
    return fail(MH_BAD_ARGUMENT,"device %d out of range",device);
  rt().default_device=device;
  return MH_OK;
}

MH_API int MhGetEnabled(void)
{
  Runtime &r=rt();
  std::call_once(r.once,do_init);
  return r.enabled;
}

MH_API int MhSetEnabled(int enabled)
{
  Runtime &r=rt();
  std::call_once(r.once,do_init);
  r.enabled=enabled != 0 ? 1 : 0;
  return r.enabled;
}

MH_API const char *MhGetLastError(void) { return g_error; }

MH_API const char *MhGetVersion(void) { return "magickhip 0.1 (gfx950)"; }

MH_API MhPrecision MhGetPrecision(void)
{
  Runtime &r=rt();
  std::call_once(r.once,do_init);
  return r.precision;
}

MH_API MhPrecision MhSetPrecision(MhPrecision p)
{
  Runtime &r=rt();
  std::call_once(r.once,do_init);
  r.precision=(p == MH_PRECISION_FAST) ? MH_PRECISION_FAST : MH_PRECISION_EXACT;
  return r.precision;
}

MH_API MhStatus MhSetOption(const char *name,const char *value)
{
  if ((name == nullptr) || ((strncmp(name,"MAGICKHIP_",10) != 0) && (strncmp(name,"MAGICK_HIP_",11) != 0)))
    return fail(MH_BAD_ARGUMENT,"MhSetOption: option names begin with MAGICKHIP_ or MAGICK_HIP_");
  Runtime &r=rt();
  std::call_once(r.once,do_init);
  Options &o=options();
  std::unique_lock<std::shared_mutex> guard(o.lock);
  if (value == nullptr)
    o.values.erase(name);
  else
    o.values[name]=strdup(value);      // interned (a few bytes per change of a diagnostic switch)
  return MH_OK;
}

MH_API const char *MhGetOption(const char *name)
{
  return name != nullptr ? option(name) : nullptr;
}

MH_API int MhLogicalDeviceCount(void) { return logical_device_count(); }

MH_API MhStatus MhGetDeviceInfo(int device,MhDeviceInfo *info)
{
  MH_TRY(runtime_ready());
  if ((info == nullptr) || (device < 0) || (device >= device_count()))
    return fail(MH_BAD_ARGUMENT,"MhGetDeviceInfo: device %d out of range",device);
  hipDeviceProp_t prop;
  MH_HIP(hipGetDeviceProperties(&prop,device));
  memset(info,0,sizeof(*info));
  snprintf(info->name,sizeof(info->name),"%s",prop.name);
  snprintf(info->architecture,sizeof(info->architecture),"%s",prop.gcnArchName);
  info->compute_units=prop.multiProcessorCount;
  info->clock_mhz=prop.clockRate/1000;
  info->global_memory=prop.totalGlobalMem;
  info->local_memory=prop.sharedMemPerBlock;
  return MH_OK;
}

MH_API MhStatus MhStreamCreate(int device,void **stream)
{
  MH_TRY(runtime_ready());
  if ((stream == nullptr) || (device < 0) || (device >= device_count()))
    return fail(MH_BAD_ARGUMENT,"MhStreamCreate: device %d out of range",device);
  DeviceGuard guard;
  MH_HIP(guard.enter(device));
  hipStream_t s=nullptr;
  MH_HIP(hipStreamCreateWithFlags(&s,hipStreamNonBlocking));
  Runtime &r=rt();
  std::lock_guard<std::mutex> lock(r.lock);
  r.caller_streams.emplace_back(device,s);
  *stream=s;
  return MH_OK;
}

MH_API MhStatus MhStreamDestroy(int device,void *stream)
{
  MH_TRY(runtime_ready());
  Runtime &r=rt();
  {
    std::lock_guard<std::mutex> lock(r.lock);
    bool known=false;
    for (size_t i=0; i < r.caller_streams.size(); i++)
      if (r.caller_streams[i].second == (hipStream_t) stream)
        {
          device=r.caller_streams[i].first;
          r.caller_streams.erase(r.caller_streams.begin()+(ptrdiff_t) i);
          known=true;
          break;
        }
    if (!known)
      return fail(MH_BAD_ARGUMENT,"MhStreamDestroy: not a stream of MhStreamCreate");
  }
  DeviceGuard guard;
  MH_HIP(guard.enter(device));
  MH_HIP(hipStreamSynchronize((hipStream_t) stream));
  {
    // cached pool blocks last used on this stream are complete now: retag them to the device's
    // null stream (always a valid handle), so that pool_alloc never waits on a destroyed or
    // recycled one
    std::lock_guard<std::mutex> lock(r.lock);
    for (DeviceState &d : r.devices)
      for (auto &entry : d.free_blocks)
        if (entry.second.stream == (hipStream_t) stream)
          entry.second.stream=nullptr;
  }
  MH_HIP(hipStreamDestroy((hipStream_t) stream));
  return MH_OK;
}

MH_API MhStatus MhDeviceAllocAsync(int device,size_t bytes,void *stream,void **ptr)
{
  MH_TRY(runtime_ready());
  if (ptr == nullptr)
    return fail(MH_BAD_ARGUMENT,"null ptr");
  if (device < 0) device=default_device();
  if (device >= device_count())
    return fail(MH_BAD_ARGUMENT,"device %d out of range",device);
  return pool_alloc(device,bytes,(hipStream_t) stream,ptr);
}

MH_API MhStatus MhDeviceFreeAsync(int device,void *ptr,void *stream)
{
  MH_TRY(runtime_ready());
  if (device < 0) device=default_device();
  if (device >= device_count())
    return fail(MH_BAD_ARGUMENT,"device %d out of range",device);
  pool_free(device,ptr,(hipStream_t) stream);
  return MH_OK;
}

static void release_spare_pinned_blocks(size_t keep_bytes);

MH_API void *MhHostAlloc(size_t bytes)
{
  if ((runtime_ready() != MH_OK) || (bytes == 0))
    return nullptr;
  void *block=nullptr;
  PinnedBlocks &p=pinned_blocks();
  {
    std::lock_guard<std::mutex> guard(p.lock);
    auto it=p.spare.lower_bound(bytes);
    if ((it != p.spare.end()) && (it->first <= bytes+bytes/8))
      {
        block=it->second;
        p.spare_total-=it->first;
        p.blocks[static_cast<const char *>(block)]=it->first;
        p.total+=it->first;
        p.spare.erase(it);
        return block;
      }
  }
  if (hipHostMalloc(&block,bytes,hipHostMallocPortable) != hipSuccess)
    {
      (void) hipGetLastError();
      release_spare_pinned_blocks(0);    // make room out of the spare list and try once more
      if (hipHostMalloc(&block,bytes,hipHostMallocPortable) != hipSuccess)
        {
          (void) hipGetLastError();
          return nullptr;
        }
    }
  std::lock_guard<std::mutex> guard(p.lock);
  p.blocks[static_cast<const char *>(block)]=bytes;
  p.total+=bytes;
  return block;
}

// the spare page-locked blocks go back to the system (MhTerminus, or to make room)
static void release_spare_pinned_blocks(size_t keep_bytes)
{
  PinnedBlocks &p=pinned_blocks();
  std::vector<void *> victims;
  {
    std::lock_guard<std::mutex> guard(p.lock);
    while ((p.spare_total > keep_bytes) && !p.spare.empty())
      {
        auto it=p.spare.begin();
        p.spare_total-=it->first;
        victims.push_back(it->second);
        p.spare.erase(it);
      }
  }
  for (void *block : victims)
    (void) hipHostFree(block);
}

MH_API int MhHostFree(void *block)
{
  if (block == nullptr)
    return 0;
  PinnedBlocks &p=pinned_blocks();
  // MAGICKHIP_PINNED_SPARE_BYTES: how much released page-locked memory is kept for reuse (default 1 GiB)
  const size_t spare_limit=(size_t) option_long("MAGICKHIP_PINNED_SPARE_BYTES",(long) 1 << 30);
  {
    std::lock_guard<std::mutex> guard(p.lock);
    auto it=p.blocks.find(static_cast<const char *>(block));
    if (it == p.blocks.end())
      return 0;
    const size_t bytes=it->second;
    p.total-=bytes;
    p.blocks.erase(it);
    if (bytes <= spare_limit)
      {
        p.spare.emplace(bytes,block);
        p.spare_total+=bytes;
        block=nullptr;
      }
  }
  if (block != nullptr)
    (void) hipHostFree(block);
  else
    release_spare_pinned_blocks(spare_limit);
  return 1;
}

MH_API size_t MhHostAllocatedBytes(void)
{
  PinnedBlocks &p=pinned_blocks();
  std::lock_guard<std::mutex> guard(p.lock);
  return p.total;
}

MH_API size_t MhHostPinnedBytes(void)
{
  PinnedBlocks &p=pinned_blocks();
  std::lock_guard<std::mutex> guard(p.lock);
  return p.total+p.spare_total;          // spare blocks are still page-locked: they count against a budget
}

MH_API MhStatus MhDeviceAlloc(int device,size_t bytes,void **ptr)
{
  MH_TRY(runtime_ready());
  if (ptr == nullptr)
    return fail(MH_BAD_ARGUMENT,"null ptr");
  if (device < 0) device=default_device();
  DeviceGuard guard;
  MH_HIP(guard.enter(device));
  hipError_t err=hipMalloc(ptr,bytes);
  if (err != hipSuccess)
    {
      // the workspace pool may be holding the memory: give its cached blocks back and retry
      // (as pool_alloc does), so a caller's image allocation does not fail — and the MagickCore
      // shim fall back to the CPU — because of scratch space nobody is using
      (void) hipGetLastError();
      pool_trim();
      err=hipMalloc(ptr,bytes);
    }
  if (err != hipSuccess)
    return fail(MH_OUT_OF_MEMORY,"hipMalloc(%zu): %s",bytes,hipGetErrorString(err));
  return MH_OK;
}

MH_API MhStatus MhDeviceFree(int device,void *ptr)
{
  (void) device;
  MH_TRY(runtime_ready());
  MH_HIP(hipFree(ptr));
  return MH_OK;
}

MH_API MhStatus MhUpload(int device,void *dst,const void *src,size_t bytes,void *stream)
{
  MH_TRY(runtime_ready());
  DeviceGuard guard;
  if ((device >= 0) && (device < device_count()))
    MH_HIP(guard.enter(device));
  if ((bytes >= 2*kPiece) && (device >= 0) && (device < device_count()) && !host_block_is_pinned(src,bytes))
    return transfer_image(device,(hipStream_t) stream,dst,const_cast<void *>(src),bytes,true);
  MH_HIP(hipMemcpyAsync(dst,src,bytes,hipMemcpyHostToDevice,(hipStream_t) stream));
  return MH_OK;
}

MH_API MhStatus MhDownload(int device,void *dst,const void *src,size_t bytes,void *stream)
{
  MH_TRY(runtime_ready());
  DeviceGuard guard;
  if ((device >= 0) && (device < device_count()))
    MH_HIP(guard.enter(device));
  if ((bytes >= 2*kPiece) && (device >= 0) && (device < device_count()) && !host_block_is_pinned(dst,bytes))
    MH_TRY(transfer_image(device,(hipStream_t) stream,const_cast<void *>(src),dst,bytes,false));
  else
    MH_HIP(hipMemcpyAsync(dst,src,bytes,hipMemcpyDeviceToHost,(hipStream_t) stream));
  MH_HIP(hipStreamSynchronize((hipStream_t) stream));
  return MH_OK;
}

MH_API MhStatus MhSynchronize(int device,void *stream)
{
  MH_TRY(runtime_ready());
  DeviceGuard guard;
  if ((device >= 0) && (device < device_count()))
    MH_HIP(guard.enter(device));
  MH_HIP(hipStreamSynchronize((hipStream_t) stream));
  return MH_OK;
}

MH_API int MhSetProfileEnabled(int enabled)
{
  Runtime &r=rt();
  std::call_once(r.once,do_init);
  r.profiling=enabled != 0 ? 1 : 0;
  return r.profiling;
}

static size_t profile_records(int device,MhKernelProfileRecord *records,size_t capacity)
{
  Runtime &r=rt();
  drain_profile();
  std::lock_guard<std::mutex> guard(r.lock);
  // device < 0: every device, a kernel's records summed over the devices
  std::map<std::string,Runtime::Rec> merged;
  for (auto &kv : r.records)
    {
      if ((device >= 0) && (kv.first.first != device))
        continue;
      Runtime::Rec &m=merged[kv.first.second];
      m.count+=kv.second.count;
      m.total_ms+=kv.second.total_ms;
      if (kv.second.min_ms < m.min_ms) m.min_ms=kv.second.min_ms;
      if (kv.second.max_ms > m.max_ms) m.max_ms=kv.second.max_ms;
    }
  size_t n=0;
  for (auto &kv : merged)
    {
      if ((records != nullptr) && (n < capacity))
        {
          const char *&name=r.record_names[kv.first];
          if (name == nullptr)
            name=strdup(kv.first.c_str());       // interned: the pointer outlives the call
          records[n].kernel_name=name;
          records[n].count=kv.second.count;
          records[n].min_ms=kv.second.min_ms;
          records[n].max_ms=kv.second.max_ms;
          records[n].total_ms=kv.second.total_ms;
        }
      n++;
    }
  return n;
}

MH_API size_t MhGetProfileRecords(MhKernelProfileRecord *records,size_t capacity)
{
  return profile_records(-1,records,capacity);
}

MH_API size_t MhGetDeviceProfileRecords(int device,MhKernelProfileRecord *records,size_t capacity)
{
  if ((device < 0) || (device >= device_count()))
    return 0;
  return profile_records(device,records,capacity);
}

MH_API void MhResetProfileRecords(void)
{
  Runtime &r=rt();
  drain_profile();
  std::lock_guard<std::mutex> guard(r.lock);
  r.records.clear();
}

MH_API void MhInitImage(MhImage *image,void *pixels,size_t columns,size_t rows,
  uint32_t number_channels,int has_alpha,MhQuantumKind quantum,MhMemoryKind memory)
{
  memset(image,0,sizeof(*image));
  image->pixels=pixels;
  image->columns=columns;
  image->rows=rows;
  image->number_channels=number_channels;
  image->quantum=(uint32_t) quantum;
  image->memory=(uint32_t) memory;
  image->device=-1;
  image->alpha_offset=-1;
  image->alpha_trait=MH_TRAIT_UNDEFINED;
  if ((has_alpha != 0) && (number_channels >= 2))
    {
      image->alpha_offset=(int32_t) number_channels-1;
      image->alpha_trait=MH_TRAIT_BLEND;
    }
  for (uint32_t c=0; c < number_channels && c < MH_MAX_CHANNELS; c++)
    {
      // pixel.c:6338-6383: selected channels are Update; colour channels also
      // carry Blend when the image has an active alpha channel.
      uint32_t traits=MH_TRAIT_UPDATE;
      if ((image->alpha_offset >= 0) && ((int32_t) c != image->alpha_offset))
        traits|=MH_TRAIT_BLEND;
      image->channel_traits[c]=traits;
    }
  image->colorspace=(number_channels-(has_alpha != 0 ? 1u : 0u)) == 1u ?
    MH_COLORSPACE_GRAY : MH_COLORSPACE_SRGB;
  image->intensity=MH_INTENSITY_UNDEFINED;
  image->channel_mask=MH_ALL_CHANNELS;
  image->stream=nullptr;
}

} // extern "C"
