// Operator chains over many images and over several GPUs, from plain C:
//
//   MagickHipBatchImages   independent images: a work queue over devices x streams host threads,
//                          one HIP stream each — upload, kernels and download of different
//                          images overlap on every device; no collective.
//   MagickHipShardedImage  one image in row bands, one per device: halo rows move between
//                          neighbouring bands with hipMemcpyPeerAsync before every stencil pass;
//                          the histogram operators all-reduce their 65536 x channels table
//                          (RCCL when the bands sit on distinct GPUs, peer copies + an add
//                          kernel otherwise) and build and apply the identical LUT everywhere.
//
// The reference arbitrates devices and queues per call (RequestOpenCLDevice,
// MagickCore/opencl.c:3056-3102; AcquireOpenCLCommandQueue :656) but gives an operator one
// device; this is the part of section 8e it has no counterpart for.
#include "mh_internal.hpp"

#include <rccl/rccl.h>          // types and enums only: the symbols are resolved with dlopen
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mh {

// ---------------------------------------------------------------- streams
// Worker streams live until MhTerminus: the workspace pool remembers the stream a block was
// last used on, so a stream must outlive the blocks it tagged.
static std::mutex g_stream_lock;
static std::map<std::pair<int,int>,hipStream_t> g_streams;

// MhTerminus: after the pools (whose blocks the streams tagged) have been trimmed
void release_batch_streams()
{
  std::lock_guard<std::mutex> guard(g_stream_lock);
  for (auto &entry : g_streams)
    {
      DeviceGuard device;
      if (device.enter(entry.first.first) == hipSuccess)
        {
          (void) hipStreamSynchronize(entry.second);
          (void) hipStreamDestroy(entry.second);
        }
    }
  g_streams.clear();
}

static hipStream_t batch_stream(int physical,int index)
{
  std::lock_guard<std::mutex> guard(g_stream_lock);
  auto key=std::make_pair(physical,index);
  auto it=g_streams.find(key);
  if (it != g_streams.end())
    return it->second;
  DeviceGuard device;
  if (device.enter(physical) != hipSuccess)
    return nullptr;
  hipStream_t stream=nullptr;
  if (hipStreamCreateWithFlags(&stream,hipStreamNonBlocking) != hipSuccess)
    return nullptr;
  g_streams[key]=stream;
  return stream;
}

// ---------------------------------------------------------------- operators
static size_t image_bytes(const MhImage &image)
{
  return image.columns*image.rows*(size_t) image.number_channels*
    (image.quantum == MH_QUANTUM_U16 ? 2u : 4u);
}

// what an operator needs besides its arguments, built once per call
struct PreparedOperator
{
  MhOperator op;
  std::unique_ptr<MhKernelInfo,MhKernelInfo *(*)(MhKernelInfo *)> kernel{nullptr,MhDestroyKernelInfo};
  const MhKernelInfo *borrowed=nullptr;   // a caller's kernel list (host_banded_operator)
  const MhKernelInfo *kernels() const { return borrowed != nullptr ? borrowed : kernel.get(); }
  size_t reach=0;              // rows a stencil reads above / below an output row
  bool stencil=false,histogram=false;
};

static MhStatus prepare(const MhOperator *operators,size_t count,std::vector<PreparedOperator> &out,
  const MhKernelInfo *borrowed_kernel=nullptr)
{
  out.resize(count);
  for (size_t i=0; i < count; i++)
    {
      PreparedOperator &p=out[i];
      p.op=operators[i];
      p.borrowed=borrowed_kernel;
      switch (p.op.kind)
      {
        case MH_OP_BLUR: case MH_OP_UNSHARP_MASK: case MH_OP_GAUSSIAN_BLUR:
          {
            // blur:RxS is 1 x W and W x 1 (effect.c:773); GaussianBlurImage's kernel is
            // Gaussian:RxS, (2*radius+1)^2 or GetOptimalKernelWidth2D (effect.c:1725)
            const double radius=p.op.args[0],sigma=p.op.args[1];
            size_t width;
            if (p.op.kind == MH_OP_GAUSSIAN_BLUR)
              width=radius >= 1.0 ? (size_t) radius*2+1 : MhGetOptimalKernelWidth2D(radius,sigma);
            else
              width=radius >= 1.0 ? (size_t) radius*2+1 : MhGetOptimalKernelWidth1D(radius,sigma);
            p.reach=(width-1)/2;
            p.stencil=true;
            break;
          }
        case MH_OP_MORPHOLOGY:
          {
            if (p.borrowed == nullptr)
              {
                if (p.op.text == nullptr)
                  return fail(MH_BAD_ARGUMENT,"operator %zu: morphology needs a kernel string",i);
                p.kernel.reset(MhAcquireKernelInfo(p.op.text));
                if (!p.kernel)
                  return fail(MH_BAD_ARGUMENT,"operator %zu: cannot parse kernel '%s'",i,p.op.text);
              }
            size_t reach=0,kernels=0;
            for (const MhKernelInfo *k=p.kernels(); k != nullptr; k=k->next)
              {
                const size_t up=(size_t) k->y,down=k->height-1-(size_t) k->y;
                reach+=up > down ? up : down;
                kernels++;
              }
            // compound methods run up to four primitives per kernel (Smooth); iterations
            // multiply the reach.  Iterate-until-convergence (-1) has no bound.
            const ptrdiff_t iterations=(ptrdiff_t) p.op.args[1];
            size_t stages=1;
            switch ((MhMorphologyMethod) (int) p.op.args[0])
            {
              case MH_MORPHOLOGY_SMOOTH: stages=4; break;
              case MH_MORPHOLOGY_OPEN: case MH_MORPHOLOGY_CLOSE: case MH_MORPHOLOGY_OPEN_INTENSITY:
              case MH_MORPHOLOGY_CLOSE_INTENSITY: case MH_MORPHOLOGY_TOP_HAT:
              case MH_MORPHOLOGY_BOTTOM_HAT: case MH_MORPHOLOGY_EDGE: stages=2; break;
              default: break;
            }
            p.reach=iterations < 1 ? (size_t) -1 : reach*stages*(size_t) iterations;
            p.stencil=true;
            (void) kernels;
            break;
          }
        case MH_OP_RESIZE: case MH_OP_COLORSPACE:
          break;
        case MH_OP_CONTRAST_STRETCH: case MH_OP_EQUALIZE:
          p.histogram=true;
          break;
        default:
          return fail(MH_BAD_ARGUMENT,"operator %zu: unknown kind %u",i,p.op.kind);
      }
    }
  return MH_OK;
}

// A device-resident image a chain works on.  `owned`: its pixels came from the pool.
struct Working
{
  MhImage image;
  bool owned=false;
  int device=0;
  hipStream_t stream=nullptr;
  void release()
  {
    if (owned)
      pool_free(device,image.pixels,stream);
    owned=false;
  }
};

// One operator on `cur`.  New-image operators allocate their result from the pool and release
// their input; in-place operators mutate cur.  Everything is enqueued on cur.stream.
static MhStatus apply_operator(const PreparedOperator &p,Working &cur)
{
  const MhOperator &op=p.op;
  switch (op.kind)
  {
    case MH_OP_COLORSPACE:
      return MagickHipTransformImageColorspace(&cur.image,(MhColorspace) (int) op.args[0]);
    case MH_OP_CONTRAST_STRETCH:
      return MagickHipContrastStretchImage(&cur.image,op.args[0],op.args[1],nullptr);
    case MH_OP_EQUALIZE:
      return MagickHipEqualizeImage(&cur.image);
    default:
      break;
  }
  MhImage next=cur.image;
  if (op.kind == MH_OP_RESIZE)
    {
      next.columns=(size_t) op.args[0];
      next.rows=(size_t) op.args[1];
      if ((next.columns == 0) || (next.rows == 0))
        return fail(MH_BAD_ARGUMENT,"resize to %zux%zu",next.columns,next.rows);
    }
  void *memory=nullptr;
  MH_TRY(pool_alloc(cur.device,image_bytes(next),cur.stream,&memory));
  next.pixels=memory;
  MhStatus status=MH_BAD_ARGUMENT;
  switch (op.kind)
  {
    case MH_OP_BLUR:
      status=MagickHipBlurImage(&cur.image,&next,op.args[0],op.args[1]);
      break;
    case MH_OP_GAUSSIAN_BLUR:
      status=MagickHipGaussianBlurImage(&cur.image,&next,op.args[0],op.args[1]);
      break;
    case MH_OP_UNSHARP_MASK:
      status=MagickHipUnsharpMaskImage(&cur.image,&next,op.args[0],op.args[1],op.args[2],op.args[3]);
      break;
    case MH_OP_RESIZE:
      status=MagickHipResizeImage(&cur.image,&next,(MhFilterType) (int) op.args[2]);
      break;
    case MH_OP_MORPHOLOGY:
      status=MagickHipMorphologyImage(&cur.image,&next,(MhMorphologyMethod) (int) op.args[0],
        (ptrdiff_t) op.args[1],p.kernels(),op.args[2]);
      break;
    default:
      break;
  }
  if (status != MH_OK)
    {
      pool_free(cur.device,memory,cur.stream);
      return status;
    }
  cur.release();
  cur.image=next;
  cur.owned=true;
  return MH_OK;
}

// A device-resident image belongs to the CALLER's stream (image.stream; null = the default
// stream of image.device): work the caller has enqueued there — a kernel still filling the
// tensor, a previous reader of the result buffer — must be ordered before anything this
// library enqueues on its own worker streams.  The wait is a device-side event, not a host sync.
static MhStatus wait_for_caller(const MhImage &image,hipStream_t mine)
{
  if (image.memory != MH_MEMORY_DEVICE)
    return MH_OK;
  const int home=resolve_device(&image);
  hipStream_t theirs=resolve_stream(&image,home);
  if (theirs == mine)
    return MH_OK;
  DeviceGuard guard;
  MH_HIP(guard.enter(home));
  hipEvent_t event=nullptr;
  MH_HIP(hipEventCreateWithFlags(&event,hipEventDisableTiming));
  hipError_t err=hipEventRecord(event,theirs);
  if (err == hipSuccess)
    err=hipStreamWaitEvent(mine,event,0);
  (void) hipEventDestroy(event);                  // released once the wait has consumed it
  MH_HIP(err);
  return MH_OK;
}

// Bring `source` onto (device, stream) as a pool-owned working copy.
static MhStatus working_copy(const MhImage &source,int device,hipStream_t stream,Working &out)
{
  out.image=source;
  out.device=device;
  out.stream=stream;
  out.image.memory=MH_MEMORY_DEVICE;
  out.image.device=device;
  out.image.stream=stream;
  void *memory=nullptr;
  const size_t bytes=image_bytes(source);
  MH_TRY(pool_alloc(device,bytes,stream,&memory));
  out.image.pixels=memory;
  out.owned=true;
  if (source.memory == MH_MEMORY_HOST)
    return MhUpload(device,memory,source.pixels,bytes,stream);
  MH_TRY(wait_for_caller(source,stream));
  const int home=resolve_device(&source);
  if (home == device)
    MH_HIP(hipMemcpyAsync(memory,source.pixels,bytes,hipMemcpyDeviceToDevice,stream));
  else                                            // works with and without peer access
    MH_HIP(hipMemcpyPeerAsync(memory,device,source.pixels,home,bytes,stream));
  return MH_OK;
}

// Copy the chain's output into the caller's descriptor and wait for it.
// pending (optional): streams whose work the CALLER of deliver() waits for later — a result in
// device memory is then only enqueued here, and the worker does not drain its stream after every
// image (a drained stream idles the GPU while the host prepares the next chain: 512 images of
// config C4 took 109 ms for 69 ms of kernels)
static MhStatus deliver(const Working &cur,MhImage &result,std::vector<std::pair<int,hipStream_t>> *pending=nullptr)
{
  if ((result.columns != cur.image.columns) || (result.rows != cur.image.rows) ||
      (result.number_channels != cur.image.number_channels) || (result.quantum != cur.image.quantum))
    return fail(MH_BAD_ARGUMENT,"result descriptor is %zux%zu, the chain produced %zux%zu",
      result.columns,result.rows,cur.image.columns,cur.image.rows);
  const size_t bytes=image_bytes(cur.image);
  if (result.memory == MH_MEMORY_HOST)
    MH_TRY(MhDownload(cur.device,result.pixels,cur.image.pixels,bytes,cur.stream));
  else
    {
      if (result.pixels != cur.image.pixels)
        {
          MH_TRY(wait_for_caller(result,cur.stream));       // earlier readers / writers of the buffer
          const int home=resolve_device(&result);
          if (home == cur.device)
            MH_HIP(hipMemcpyAsync(result.pixels,cur.image.pixels,bytes,hipMemcpyDeviceToDevice,cur.stream));
          else
            MH_HIP(hipMemcpyPeerAsync(result.pixels,home,cur.image.pixels,cur.device,bytes,cur.stream));
        }
      // the caller's stream is not ours: the result is complete when this returns, or — pending —
      // when the caller has waited for the stream
      if (pending == nullptr)
        MH_HIP(hipStreamSynchronize(cur.stream));
      else
        {
          bool known=false;
          for (const auto &entry : *pending)
            known=known || (entry.second == cur.stream);
          if (!known)
            pending->emplace_back(cur.device,cur.stream);
        }
    }
  result.colorspace=cur.image.colorspace;
  return MH_OK;
}

static int logical_devices(int requested)
{
  const int physical=device_count();
  int n=requested <= 0 ? physical : requested;
  if (const char *e=option("MAGICKHIP_LOGICAL_DEVICES"))
    if ((requested <= 0) && (atoi(e) > 0))
      n=atoi(e);
  return n > 16 ? 16 : (n < 1 ? 1 : n);
}

// ---------------------------------------------------------------- all-reduce
// RCCL through dlopen: the library has no link-time dependency on librccl, and a node without
// it (or a single GPU) still runs everything.
struct Rccl
{
  void *handle=nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *,int,const int *)=nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t)=nullptr;
  ncclResult_t (*GroupStart)()=nullptr;
  ncclResult_t (*GroupEnd)()=nullptr;
  ncclResult_t (*AllReduce)(const void *,void *,size_t,ncclDataType_t,ncclRedOp_t,ncclComm_t,hipStream_t)=nullptr;
  bool ready=false;
};

static Rccl &rccl()
{
  static Rccl r;
  static std::once_flag once;
  std::call_once(once,[]()
  {
    const char *off=option("MAGICKHIP_RCCL");
    if ((off != nullptr) && (strcmp(off,"0") == 0))
      return;
    r.handle=dlopen("librccl.so.1",RTLD_NOW | RTLD_LOCAL);
    if (r.handle == nullptr)
      r.handle=dlopen("librccl.so",RTLD_NOW | RTLD_LOCAL);
    if (r.handle == nullptr)
      return;
    *(void **) &r.CommInitAll=dlsym(r.handle,"ncclCommInitAll");
    *(void **) &r.CommDestroy=dlsym(r.handle,"ncclCommDestroy");
    *(void **) &r.GroupStart=dlsym(r.handle,"ncclGroupStart");
    *(void **) &r.GroupEnd=dlsym(r.handle,"ncclGroupEnd");
    *(void **) &r.AllReduce=dlsym(r.handle,"ncclAllReduce");
    r.ready=(r.CommInitAll != nullptr) && (r.CommDestroy != nullptr) && (r.GroupStart != nullptr) &&
      (r.GroupEnd != nullptr) && (r.AllReduce != nullptr);
  });
  return r;
}

// communicators by device set (all_reduce_tables), destroyed by MhTerminus
struct CommSet { std::mutex lock; std::vector<ncclComm_t> comms; };
static std::mutex g_comm_lock;
static std::map<std::vector<int>,std::shared_ptr<CommSet>> g_comms;

void release_rccl_communicators()
{
  std::lock_guard<std::mutex> guard(g_comm_lock);
  Rccl &r=rccl();
  for (auto &entry : g_comms)
    {
      std::lock_guard<std::mutex> set_guard(entry.second->lock);
      for (ncclComm_t comm : entry.second->comms)
        if (r.CommDestroy != nullptr)
          (void) r.CommDestroy(comm);
      entry.second->comms.clear();
    }
  g_comms.clear();
}

// Band b's table (device memory on band b's device, stream b) becomes the sum over all bands.
struct TableView { unsigned long long *table; int device; hipStream_t stream; };

static MhStatus all_reduce_tables(std::vector<TableView> &bands,size_t count,bool *used_rccl)
{
  *used_rccl=false;
  if (bands.size() < 2)
    return MH_OK;
  bool distinct=true;
  for (size_t a=0; a < bands.size(); a++)
    for (size_t b=a+1; b < bands.size(); b++)
      distinct=distinct && (bands[a].device != bands[b].device);
  Rccl &r=rccl();
  if (distinct && r.ready)
    {
      // one communicator per set of devices, created on the first use and kept until MhTerminus
      // (ncclCommInitAll costs tens to hundreds of milliseconds; the all-reduce of a 2 MB table
      // is latency-bound)
      std::vector<int> devices;
      for (const TableView &v : bands)
        devices.push_back(v.device);
      // the global lock covers the lookup only; a set of devices has its own lock for the
      // collective itself, so operators on disjoint sets of GPUs do not wait for each other
      std::shared_ptr<CommSet> set;
      {
        std::lock_guard<std::mutex> guard(g_comm_lock);
        auto it=g_comms.find(devices);
        if (it == g_comms.end())
          {
            auto fresh=std::make_shared<CommSet>();
            fresh->comms.resize(bands.size());
            if (r.CommInitAll(fresh->comms.data(),(int) bands.size(),devices.data()) == ncclSuccess)
              it=g_comms.emplace(devices,fresh).first;
          }
        if (it != g_comms.end())
          set=it->second;
      }
      if (set)
        {
          bool ok;
          {
            std::lock_guard<std::mutex> guard(set->lock);        // one collective at a time per communicator
            ok=r.GroupStart() == ncclSuccess;
            for (size_t b=0; ok && (b < bands.size()); b++)
              ok=r.AllReduce(bands[b].table,bands[b].table,count,ncclUint64,ncclSum,set->comms[b],
                bands[b].stream) == ncclSuccess;
            ok=(r.GroupEnd() == ncclSuccess) && ok;
          }
          for (size_t b=0; b < bands.size(); b++)
            ok=(hipStreamSynchronize(bands[b].stream) == hipSuccess) && ok;
          if (ok)
            {
              *used_rccl=true;
              return MH_OK;
            }
          // a communicator that failed once is not retried: drop it (the next call builds a new one)
          {
            std::lock_guard<std::mutex> guard(g_comm_lock);
            auto it=g_comms.find(devices);
            if ((it != g_comms.end()) && (it->second == set))
              g_comms.erase(it);
          }
          {
            std::lock_guard<std::mutex> guard(set->lock);
            for (ncclComm_t comm : set->comms)
              if (r.CommDestroy != nullptr)
                (void) r.CommDestroy(comm);
            set->comms.clear();
          }
          // (the tables may be partly reduced: the caller cannot recover the inputs)
          return fail(MH_DEVICE_ERROR,"ncclAllReduce of the histogram table failed");
        }
    }
  // peer copies: gather on band 0, add, broadcast
  const size_t bytes=count*sizeof(unsigned long long);
  Temp scratch;
  MH_TRY(scratch.alloc(bands[0].device,bytes,bands[0].stream));
  for (size_t b=1; b < bands.size(); b++)
    {
      MH_HIP(hipStreamSynchronize(bands[b].stream));            // band b's table is complete
      MH_HIP(hipMemcpyPeerAsync(scratch.ptr,bands[0].device,bands[b].table,bands[b].device,bytes,
        bands[0].stream));
      MH_TRY(launch_table_add(bands[0].table,scratch.as<unsigned long long>(),count,bands[0].device,
        bands[0].stream));
    }
  MH_HIP(hipStreamSynchronize(bands[0].stream));
  for (size_t b=1; b < bands.size(); b++)
    MH_HIP(hipMemcpyPeerAsync(bands[b].table,bands[b].device,bands[0].table,bands[0].device,bytes,
      bands[b].stream));
  return MH_OK;
}

// bands a logical device has finished (MhBandedBands: tests, bench)
static std::atomic<unsigned long long> g_banded_bands[16];

// ---------------------------------------------------------------- one host image, pipelined
// A new-image stencil operator on HOST memory (the pixel cache): upload, kernels and download of
// the whole frame in sequence leave the copy engines idle two thirds of the time (8192^2 RGBA
// Q16 BlurImage: 22 ms up, 0.5 ms of kernels, 22 ms down).  The frame is cut into row bands
// with the operator's reach as halo rows on both sides (recomputed, not exchanged); a few host
// threads, one stream each, take bands from a queue: upload band k+2, kernels of band k+1 and
// download of band k overlap, PCIe runs in both directions at once.
// *handled = false: not a case for this path (small image, unbounded reach, device memory).
MhStatus host_banded_operator(const MhOperator &op,const MhKernelInfo *kernel,const MhImage *image,
  MhImage *result,bool *handled)
{
  *handled=false;
  if ((image->memory != MH_MEMORY_HOST) || (result->memory != MH_MEMORY_HOST) ||
      (option("MAGICKHIP_NO_BANDED") != nullptr))
    return MH_OK;
  // the bands are described by the SOURCE's descriptor: a result with other channel traits, alpha
  // trait or channel mask (the whole-frame path honours them, channel_roles(image, result)) keeps
  // that path
  if ((result->alpha_trait != image->alpha_trait) || (result->channel_mask != image->channel_mask) ||
      (result->number_channels != image->number_channels) ||
      (memcmp(result->channel_traits,image->channel_traits,sizeof(image->channel_traits)) != 0))
    return MH_OK;
  const size_t row_bytes=image->columns*(size_t) image->number_channels*
    (image->quantum == MH_QUANTUM_U16 ? 2u : 4u);
  size_t minimum=64u << 20;
  if (const char *e=option("MAGICKHIP_BANDED_MIN_BYTES"))
    minimum=(size_t) atoll(e);
  if (row_bytes*image->rows < minimum)
    return MH_OK;
  std::vector<PreparedOperator> chain;
  if (prepare(&op,1,chain,kernel) != MH_OK)
    return MH_OK;
  const size_t reach=chain[0].reach;
  if (!chain[0].stencil || (reach == (size_t) -1) || (8*reach > image->rows))
    return MH_OK;
  // bands of ~32 MiB, at least four times the halo they carry
  size_t band_rows=(32u << 20)/row_bytes;
  band_rows=band_rows < 8*reach ? 8*reach : band_rows;
  band_rows=band_rows < 16 ? 16 : band_rows;
  const size_t H=image->rows;
  const size_t nbands=(H+band_rows-1)/band_rows;
  if (nbands < 3)
    return MH_OK;
  int workers=4;
  if (const char *e=option("MAGICKHIP_BANDED_WORKERS"))
    workers=atoi(e) < 1 ? 1 : (atoi(e) > 16 ? 16 : atoi(e));
  // MhImage::device = MH_DEVICE_ALL: the bands go round the (logical) devices of the node, `workers`
  // threads per device — one big pixel cache moves over every GPU's host link instead of one
  // (the reference hands an operator one device, opencl.c:3056-3102: 2.1 GB up and down for a
  // 16384^2 frame against 2 ms of kernel)
  const bool spread=image->device == MH_DEVICE_ALL;
  const int devices=spread ? logical_devices(0) : 1;
  const int physical=device_count();
  const int first_device=spread ? 0 : resolve_device(image);
  // per device: at most 8 threads in spread mode (the stream index below gives a logical device 8 slots of its
  // physical device's table) and at most 64 threads in all, cut evenly across the devices
  if (spread)
    {
      workers=workers > 8 ? 8 : workers;
      workers=workers > 64/devices ? (64/devices > 0 ? 64/devices : 1) : workers;
    }
  workers*=devices;
  std::atomic<size_t> next{0};
  std::mutex error_lock;
  MhStatus first_status=MH_OK;
  std::string first_error;
  auto work=[&](int w)
  {
    const int logical=w % devices;
    const int device=spread ? logical % physical : first_device;
    DeviceGuard guard;
    hipStream_t stream=batch_stream(device,32+(spread ? w/devices+8*(logical/physical) : w));
    MhStatus setup=MH_OK;
    if ((guard.enter(device) != hipSuccess) || (stream == nullptr))
      setup=fail(MH_DEVICE_ERROR,"banded operator: cannot set up device %d",device);
    for (;;)
      {
        const size_t b=next.fetch_add(1);
        if (b >= nbands)
          break;
        MhStatus status=setup;
        const size_t y0=b*band_rows,y1=y0+band_rows < H ? y0+band_rows : H;
        const size_t top=y0 < reach ? y0 : reach,bottom=H-y1 < reach ? H-y1 : reach;
        Working cur;
        if (status == MH_OK)
          {
            MhImage slice=*image;
            slice.rows=y1-y0+top+bottom;
            slice.pixels=static_cast<char *>(image->pixels)+(y0-top)*row_bytes;
            status=working_copy(slice,device,stream,cur);
            if (status == MH_OK)
              status=apply_operator(chain[0],cur);
            if (status == MH_OK)
              status=MhDownload(device,static_cast<char *>(result->pixels)+y0*row_bytes,
                static_cast<const char *>(cur.image.pixels)+top*row_bytes,(y1-y0)*row_bytes,stream);
            else
              (void) hipStreamSynchronize(stream);
            cur.release();
            if (status == MH_OK)
              g_banded_bands[logical & 15].fetch_add(1,std::memory_order_relaxed);
          }
        if (status != MH_OK)
          {
            std::lock_guard<std::mutex> lock(error_lock);
            if (first_status == MH_OK)
              {
                first_status=status;
                first_error=MhGetLastError();
              }
          }
      }
  };
  std::vector<std::thread> pool;
  for (int w=1; w < workers; w++)
    pool.emplace_back(work,w);
  work(0);
  for (std::thread &t : pool)
    t.join();
  if (first_status != MH_OK)
    return fail(first_status,"%s",first_error.c_str());
  *handled=true;
  return MH_OK;
}

} // namespace mh

using namespace mh;

extern "C" {

MH_API unsigned long long MhBandedBands(int logical_device)
{
  if ((logical_device < 0) || (logical_device >= 16))
    return 0;
  return g_banded_bands[logical_device].load(std::memory_order_relaxed);
}

MH_API MhStatus MagickHipBatchImages(const MhOperator *operators,size_t number_operators,
  const MhImage *images,MhImage *results,size_t number_images,int number_devices,
  int streams_per_device,MhBatchReport *report)
{
  MH_TRY(runtime_ready());
  if ((operators == nullptr) || (number_operators == 0) || (images == nullptr))
    return fail(MH_BAD_ARGUMENT,"BatchImages: null operators or images");
  const auto t0=std::chrono::steady_clock::now();
  std::vector<PreparedOperator> chain;
  MH_TRY(prepare(operators,number_operators,chain));
  if (results == nullptr)
    for (const PreparedOperator &p : chain)
      if (p.op.kind == MH_OP_RESIZE)
        return fail(MH_BAD_ARGUMENT,"BatchImages: a chain that resizes needs result descriptors");
  for (size_t i=0; i < number_images; i++)
    {
      MH_TRY(validate_image(&images[i],"BatchImages"));
      if (results != nullptr)
        MH_TRY(validate_image(&results[i],"BatchImages"));
    }
  const int devices=logical_devices(number_devices);
  const int physical=device_count();
  int per_device=streams_per_device <= 0 ? 3 : (streams_per_device > 8 ? 8 : streams_per_device);
  size_t workers=(size_t) devices*(size_t) per_device;
  if (workers > number_images)
    workers=number_images > 0 ? number_images : 1;
  std::atomic<size_t> next{0};
  std::mutex error_lock;
  MhStatus first_status=MH_OK;
  std::string first_error;
  std::vector<std::atomic<uint64_t>> counts(16);
  for (auto &c : counts)
    c=0;
  auto work=[&](size_t w)
  {
    // worker w serves logical device w mod devices: the first `devices` workers cover every device
    const int logical=(int) (w % (size_t) devices);
    const int device=logical % physical;
    DeviceGuard guard;
    hipStream_t stream=batch_stream(device,(int) (w/(size_t) devices)+8*(logical/physical));
    MhStatus setup=MH_OK;
    if ((guard.enter(device) != hipSuccess) || (stream == nullptr))
      setup=fail(MH_DEVICE_ERROR,"BatchImages: cannot set up device %d",device);
    // results in device memory are waited for once, when the worker has enqueued all its images
    std::vector<std::pair<int,hipStream_t>> pending;
    for (;;)
      {
        const size_t i=next.fetch_add(1);
        if (i >= number_images)
          break;
        MhStatus status=setup;
        Working cur;
        if (status == MH_OK)
          {
            // a device-resident input that is not to be overwritten is worked on as a copy
            const bool in_place=(results == nullptr) && (images[i].memory == MH_MEMORY_DEVICE);
            DeviceGuard home_guard;
            if (in_place)
              {
                // The kernels dereference the caller's memory: they must run on the GPU that
                // owns it (no peer access is ever enabled; a kernel of another GPU would fault).
                // The worker lends its thread and a stream of THAT device.
                const int home=resolve_device(&images[i]);
                hipStream_t home_stream=home == device ? stream :
                  batch_stream(home,(int) (w/(size_t) devices)+8*(logical/physical));
                if ((home_stream == nullptr) || (home_guard.enter(home) != hipSuccess))
                  status=fail(MH_DEVICE_ERROR,"BatchImages: cannot set up device %d",home);
                else
                  {
                    cur.image=images[i];
                    cur.image.stream=home_stream;
                    cur.image.device=home;
                    cur.device=home;
                    cur.stream=home_stream;
                    cur.owned=false;
                    status=wait_for_caller(images[i],home_stream);
                  }
              }
            else
              status=working_copy(images[i],device,stream,cur);
            for (size_t k=0; (status == MH_OK) && (k < chain.size()); k++)
              {
                // a colourspace transform directly followed by a contrast stretch: one entry point
                // (they share the pass over the pixels where the library has a fused kernel)
                if ((chain[k].op.kind == MH_OP_COLORSPACE) && (k+1 < chain.size()) &&
                    (chain[k+1].op.kind == MH_OP_CONTRAST_STRETCH))
                  {
                    status=MagickHipTransformColorspaceContrastStretchImage(&cur.image,
                      (MhColorspace) (int) chain[k].op.args[0],chain[k+1].op.args[0],chain[k+1].op.args[1]);
                    k++;
                    continue;
                  }
                status=apply_operator(chain[k],cur);
              }
            if (status == MH_OK)
              {
                MhImage *target=results != nullptr ? &results[i] : const_cast<MhImage *>(&images[i]);
                status=deliver(cur,*target,&pending);
              }
            else if (cur.stream != nullptr)
              (void) hipStreamSynchronize(cur.stream);
            cur.release();
          }
        if (status != MH_OK)
          {
            std::lock_guard<std::mutex> lock(error_lock);
            if (first_status == MH_OK)
              {
                first_status=status;
                first_error=MhGetLastError();
              }
          }
        else
          counts[(size_t) logical]++;
      }
    for (const auto &entry : pending)
      {
        DeviceGuard home;
        MhStatus status=MH_OK;
        if ((home.enter(entry.first) != hipSuccess) || (hipStreamSynchronize(entry.second) != hipSuccess))
          status=fail(MH_DEVICE_ERROR,"BatchImages: a stream of device %d failed",entry.first);
        if (status != MH_OK)
          {
            std::lock_guard<std::mutex> lock(error_lock);
            if (first_status == MH_OK)
              {
                first_status=status;
                first_error=MhGetLastError();
              }
          }
      }
  };
  std::vector<std::thread> pool;
  for (size_t w=1; w < workers; w++)
    pool.emplace_back(work,w);
  work(0);
  for (std::thread &t : pool)
    t.join();
  if (report != nullptr)
    {
      memset(report,0,sizeof(*report));
      report->devices=(uint32_t) devices;
      report->workers=(uint32_t) workers;
      for (int d=0; d < 16; d++)
        report->images_per_device[d]=counts[(size_t) d];
      report->seconds=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
    }
  if (first_status != MH_OK)
    return fail(first_status,"%s",first_error.c_str());
  return MH_OK;
}

MH_API MhStatus MagickHipShardedImage(const MhOperator *operators,size_t number_operators,
  const MhImage *image,MhImage *result,int number_devices,MhBatchReport *report)
{
  MH_TRY(runtime_ready());
  if ((operators == nullptr) || (number_operators == 0) || (image == nullptr) || (result == nullptr))
    return fail(MH_BAD_ARGUMENT,"ShardedImage: null argument");
  MH_TRY(validate_image(image,"ShardedImage"));
  MH_TRY(validate_image(result,"ShardedImage"));
  if ((result->columns != image->columns) || (result->rows != image->rows) ||
      (result->number_channels != image->number_channels) || (result->quantum != image->quantum))
    return fail(MH_BAD_ARGUMENT,"ShardedImage: the result must have the image's geometry and layout");
  const auto t0=std::chrono::steady_clock::now();
  std::vector<PreparedOperator> chain;
  MH_TRY(prepare(operators,number_operators,chain));
  size_t reach=0;
  for (const PreparedOperator &p : chain)
    {
      if (p.op.kind == MH_OP_RESIZE)
        return fail(MH_UNSUPPORTED,"ShardedImage: resize changes the geometry");
      if (p.reach == (size_t) -1)
        return fail(MH_UNSUPPORTED,"ShardedImage: iterate-until-convergence has no halo bound");
      reach=p.reach > reach ? p.reach : reach;
    }
  const size_t H=image->rows;
  int devices=logical_devices(number_devices);
  // a band must be at least as tall as the halo it lends to its neighbours
  while ((devices > 1) && (H/(size_t) devices < (reach > 0 ? reach : 1)))
    devices--;
  const int physical=device_count();
  const size_t row_bytes=image->columns*(size_t) image->number_channels*
    (image->quantum == MH_QUANTUM_U16 ? 2u : 4u);

  struct Band
  {
    int device=0;
    hipStream_t stream=nullptr;
    size_t y0=0,y1=0,top=0,bottom=0;        // owned rows [y0,y1), halo rows above / below
    Working cur;
    hipEvent_t done=nullptr;                // the band's last operator has been enqueued up to here
  };
  std::vector<Band> bands((size_t) devices);
  MhStatus status=MH_OK;
  uint32_t exchanges=0;
  bool rccl_used=false;
  for (int b=0; (status == MH_OK) && (b < devices); b++)
    {
      Band &band=bands[(size_t) b];
      band.device=b % physical;
      band.stream=batch_stream(band.device,64+b);
      band.y0=H*(size_t) b/(size_t) devices;
      band.y1=H*(size_t) (b+1)/(size_t) devices;
      band.top=band.y0 < reach ? band.y0 : reach;
      band.bottom=H-band.y1 < reach ? H-band.y1 : reach;
      DeviceGuard guard;
      if ((band.stream == nullptr) || (guard.enter(band.device) != hipSuccess) ||
          (hipEventCreateWithFlags(&band.done,hipEventDisableTiming) != hipSuccess))
        {
          status=fail(MH_DEVICE_ERROR,"ShardedImage: cannot set up device %d",band.device);
          break;
        }
      // the band and its halo rows straight from the source: the first stencil needs no exchange
      MhImage slice=*image;
      slice.rows=band.y1-band.y0+band.top+band.bottom;
      slice.pixels=static_cast<char *>(image->pixels)+(band.y0-band.top)*row_bytes;
      status=working_copy(slice,band.device,band.stream,band.cur);
    }
  bool halos_valid=true;
  for (size_t k=0; (status == MH_OK) && (k < chain.size()); k++)
    {
      const PreparedOperator &p=chain[k];
      if (p.stencil && !halos_valid && (devices > 1))
        {
          // every band fetches its neighbours' current edge rows (their owned rows next to the
          // cut) into its halo rows, behind the neighbours' last operator
          for (int b=0; b < devices; b++)
            (void) hipEventRecord(bands[(size_t) b].done,bands[(size_t) b].stream);
          for (int b=0; (status == MH_OK) && (b < devices); b++)
            {
              Band &band=bands[(size_t) b];
              char *mine=static_cast<char *>(band.cur.image.pixels);
              if ((b > 0) && (band.top > 0))
                {
                  Band &up=bands[(size_t) b-1];
                  const char *from=static_cast<const char *>(up.cur.image.pixels)+
                    (up.top+(up.y1-up.y0)-band.top)*row_bytes;
                  if ((hipStreamWaitEvent(band.stream,up.done,0) != hipSuccess) ||
                      (hipMemcpyPeerAsync(mine,band.device,from,up.device,band.top*row_bytes,
                         band.stream) != hipSuccess))
                    status=fail(MH_DEVICE_ERROR,"ShardedImage: halo copy from band %d failed",b-1);
                  exchanges++;
                }
              if ((status == MH_OK) && (b+1 < devices) && (band.bottom > 0))
                {
                  Band &down=bands[(size_t) b+1];
                  const char *from=static_cast<const char *>(down.cur.image.pixels)+down.top*row_bytes;
                  char *to=mine+(band.top+(band.y1-band.y0))*row_bytes;
                  if ((hipStreamWaitEvent(band.stream,down.done,0) != hipSuccess) ||
                      (hipMemcpyPeerAsync(to,band.device,from,down.device,band.bottom*row_bytes,
                         band.stream) != hipSuccess))
                    status=fail(MH_DEVICE_ERROR,"ShardedImage: halo copy from band %d failed",b+1);
                  exchanges++;
                }
            }
          // a neighbour may only overwrite the rows just read (its next operator works in place
          // or recycles the buffer) once the copies are done
          for (int b=0; b < devices; b++)
            (void) hipEventRecord(bands[(size_t) b].done,bands[(size_t) b].stream);
          for (int b=0; b < devices; b++)
            {
              if (b > 0)
                (void) hipStreamWaitEvent(bands[(size_t) b].stream,bands[(size_t) b-1].done,0);
              if (b+1 < devices)
                (void) hipStreamWaitEvent(bands[(size_t) b].stream,bands[(size_t) b+1].done,0);
            }
          halos_valid=true;
        }
      if (status != MH_OK)
        break;
      if (p.histogram && (devices > 1))
        {
          // local tables over the OWNED rows, one all-reduce, identical LUT, local apply
          const bool equalize=p.op.kind == MH_OP_EQUALIZE;
          const size_t channels=image->number_channels;
          const size_t count=(size_t) MH_HISTOGRAM_BINS*channels;
          std::vector<Temp> tables((size_t) devices),flags((size_t) devices);
          std::vector<TableView> views;
          const MhImage &described=bands[0].cur.image;
          const int mode=equalize ? ((described.channel_mask & MH_SYNC_CHANNELS) != 0 ? 1 : 0) :
            (described.channel_mask == MH_ALL_CHANNELS ? 1 : 0);
          const uint32_t colour=described.number_channels-(described.alpha_offset >= 0 ? 1u : 0u);
          const bool scan=!equalize && (colour >= 3) && ((described.colorspace == MH_COLORSPACE_SRGB) ||
            (described.colorspace == MH_COLORSPACE_RGB));
          for (int b=0; (status == MH_OK) && (b < devices); b++)
            {
              Band &band=bands[(size_t) b];
              DeviceGuard guard;
              (void) guard.enter(band.device);
              status=tables[(size_t) b].alloc(band.device,count*sizeof(unsigned long long),band.stream);
              if (status != MH_OK)
                break;
              if (hipMemsetAsync(tables[(size_t) b].ptr,0,count*sizeof(unsigned long long),band.stream) != hipSuccess)
                status=fail(MH_DEVICE_ERROR,"ShardedImage: memset failed");
              View owned;
              owned.pixels=static_cast<char *>(band.cur.image.pixels)+band.top*row_bytes;
              owned.columns=image->columns;
              owned.rows=band.y1-band.y0;
              owned.channels=(int) channels;
              owned.quantum=(MhQuantumKind) image->quantum;
              owned.device=band.device;
              owned.stream=band.stream;
              if (status == MH_OK)
                status=launch_histogram(owned,mode,&band.cur.image,tables[(size_t) b].as<unsigned long long>());
              if ((status == MH_OK) && scan)
                {
                  // IdentifyImageType (enhance.c:1586): the image is gray only if every band is
                  status=flags[(size_t) b].alloc(band.device,sizeof(unsigned int),band.stream);
                  if ((status == MH_OK) &&
                      (hipMemsetAsync(flags[(size_t) b].ptr,0,sizeof(unsigned int),band.stream) != hipSuccess))
                    status=fail(MH_DEVICE_ERROR,"ShardedImage: memset failed");
                  if (status == MH_OK)
                    status=launch_gray_check(owned,&band.cur.image,flags[(size_t) b].as<unsigned int>());
                }
              views.push_back(TableView{tables[(size_t) b].as<unsigned long long>(),band.device,band.stream});
            }
          if ((status == MH_OK) && scan)
            {
              unsigned int any_colour=0;
              for (int b=0; (status == MH_OK) && (b < devices); b++)
                {
                  unsigned int host=0;
                  if ((hipMemcpyAsync(&host,flags[(size_t) b].ptr,sizeof(host),hipMemcpyDeviceToHost,
                         bands[(size_t) b].stream) != hipSuccess) ||
                      (hipStreamSynchronize(bands[(size_t) b].stream) != hipSuccess))
                    status=fail(MH_DEVICE_ERROR,"ShardedImage: gray scan failed");
                  any_colour|=host;
                }
              if ((status == MH_OK) && (any_colour == 0))
                status=fail(MH_UNSUPPORTED,"ContrastStretchImage: image is gray; convert it to the "
                  "GRAY colourspace first (IdentifyImageType, enhance.c:1586)");
            }
          bool used=false;
          if (status == MH_OK)
            status=all_reduce_tables(views,count,&used);
          rccl_used=rccl_used || used;
          for (int b=0; (status == MH_OK) && (b < devices); b++)
            {
              Band &band=bands[(size_t) b];
              DeviceGuard guard;
              (void) guard.enter(band.device);
              View whole;
              whole.pixels=band.cur.image.pixels;
              whole.columns=image->columns;
              whole.rows=band.cur.image.rows;
              whole.channels=(int) channels;
              whole.quantum=(MhQuantumKind) image->quantum;
              whole.device=band.device;
              whole.stream=band.stream;
              // the LUT builders take the WHOLE image's pixel count (enhance.c:1674)
              MhImage full=band.cur.image;
              full.rows=H;
              status=apply_histogram_lut(whole,&full,tables[(size_t) b].as<unsigned long long>(),mode,equalize,
                p.op.args[0],(double) image->columns*(double) H-p.op.args[1]);
              if (status == MH_OK)
                (void) hipStreamSynchronize(band.stream);       // the tables go back to the pool
            }
          continue;                               // pointwise: halo rows stay consistent
        }
      for (int b=0; (status == MH_OK) && (b < devices); b++)
        {
          DeviceGuard guard;
          (void) guard.enter(bands[(size_t) b].device);
          status=apply_operator(p,bands[(size_t) b].cur);
        }
      if (p.stencil)
        halos_valid=false;
    }
  // the owned rows of every band, into the caller's image
  for (int b=0; (status == MH_OK) && (b < devices); b++)
    {
      Band &band=bands[(size_t) b];
      const char *from=static_cast<const char *>(band.cur.image.pixels)+band.top*row_bytes;
      char *to=static_cast<char *>(result->pixels)+band.y0*row_bytes;
      const size_t bytes=(band.y1-band.y0)*row_bytes;
      if (result->memory == MH_MEMORY_HOST)
        status=MhDownload(band.device,to,from,bytes,band.stream);
      else if ((wait_for_caller(*result,band.stream) != MH_OK) ||
               (hipMemcpyPeerAsync(to,result->device < 0 ? default_device() : result->device,from,
                  band.device,bytes,band.stream) != hipSuccess) ||
               (hipStreamSynchronize(band.stream) != hipSuccess))
        status=fail(MH_DEVICE_ERROR,"ShardedImage: cannot deliver band %d",b);
      if (status == MH_OK)
        result->colorspace=band.cur.image.colorspace;
    }
  for (Band &band : bands)
    {
      if (band.stream != nullptr)
        (void) hipStreamSynchronize(band.stream);
      band.cur.release();
      if (band.done != nullptr)
        (void) hipEventDestroy(band.done);
    }
  if (report != nullptr)
    {
      memset(report,0,sizeof(*report));
      report->devices=(uint32_t) devices;
      report->workers=1;
      report->used_rccl=rccl_used ? 1u : 0u;
      report->halo_exchanges=exchanges;
      for (int b=0; b < devices; b++)
        report->images_per_device[b]=1;
      report->seconds=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
    }
  return status;
}

} // extern "C"
