// ResizeImage (enlargement: VerticalFilter then HorizontalFilter, MagickCore/resize.c:3846-3861)
// in ONE launch on the fp64 matrix pipe — the FAST (+-1 ULP / +-1 level) form of the four-channel
// frame.  The two-pass form writes and re-reads the Quantum-typed intermediate (8.6 GB of 26.8 GB
// for 8192^2 -> 32768^2 float RGBA) and its horizontal pass is bound by LDS reads and fp64 vector
// issue (7 taps x 32 B per output pixel out of LDS); here both filters are banded matrix products
// on v_mfma_f64_16x16x4_f64 (resize_mfma_plan.hpp), the vector pipe only finishes pixels
// (gamma, rounding to Quantum) and LDS holds nothing but the source patch and the weights:
//
//   workgroup = 4 waves = one strip of tps*16 output columns x 64 output rows per step, `steps`
//               steps down the image (the strip's horizontal weights are staged in LDS once)
//   wave      = one row group (16 output rows); walks the strip's source columns in blocks of 16:
//     vertical   I[x][y] = sum_k P[x][k] Wv[k][y]: A = the patch (lane: column x = lane&15, row
//                k = lane>>4 of the K-block, alpha-premultiplied while it is read out of LDS),
//                B = the row group's weight block (registers), 4 channel accumulators.  Register r
//                of lane (g = lane>>4, y = lane&15) ends up holding column g+4r.
//     finish     HorizontalFilter's input is the Quantum-ROUNDED intermediate (the reference
//                stores filter_image): gamma, ClampToQuantum, premultiply again — per lane, all
//                four channels of a pixel are in the same lane.
//     horizontal O[y][xo] = sum_k I'[y][k] Wh[k][xo]: A = those registers (K-block r = columns
//                4r..4r+3 of the block, lane group g = column 4r+g: exactly the layout above), B =
//                the tile's weight blocks from LDS.  An out tile's window spans at most two
//                blocks, so a ring of 8 K-blocks (previous + current block) is enough.
//                D: lane (n = lane&15) = output column, register r of group g = row g+4r — a
//                whole pixel per lane, 16 lanes = 256 contiguous bytes per row.
// Semantics are resize.c:3494-3530 / :3709-3745 with the derived gamma of the Fma64 policy
// (resize_acc.hpp): colour = sum(w*alpha*p) / sum(w*alpha), alpha = sum(w*alpha); the matrix
// pipe's fused sums differ from the reference's separately rounded ones by ~1e-16 relative.
// Zero-weight padding multiplies samples outside an output's window by 0, which is exact unless
// a sample is not finite: a float frame's patch is watched while it is staged and a workgroup
// step that holds a non-finite sample (or produces a non-finite intermediate) recomputes its
// outputs tap by tap in the reference's own window (careful_step).
#include "mh_internal.hpp"
#include "resize_filter.hpp"
#include "resize_mfma_plan.hpp"
#include "device_common.hpp"
#include "resize_acc.hpp"
#include <memory>
#include <mutex>

namespace mh {

typedef double d4 __attribute__((ext_vector_type(4)));

struct MfmaResizeArgs
{
  const void *src;
  void *dst;
  int src_columns,src_rows,dst_columns,dst_rows;
  int tps,nstrips,nrg,nvb_max,strips_per_xcd,steps;
  unsigned wlds_bytes;
  const int *strip_col_lo,*strip_nvb,*strip_wbase,*strip_wcount,*strip_ready;
  const int *tile_kb0,*tile_nkb,*tile_woff;
  const double *wh;
  const int *rg_row_lo,*rg_nvk,*rg_woff;
  const double *wv;
  // the contribution lists themselves (careful_step)
  const int *vstart,*vcount,*hstart,*hcount;
  const double *vweight,*hweight;            // [tap][out]
};

template<typename Q>
static __device__ __forceinline__ float4 load_patch_pixel(const Q *p)
{
  Q v[4];
  load_pixel<Q,4>(p,v);
  return make_float4((float) v[0],(float) v[1],(float) v[2],(float) v[3]);
}

static __device__ __forceinline__ bool not_finite_f32(float v)
{
  return (__builtin_bit_cast(unsigned,v) & 0x7f800000u) == 0x7f800000u;
}

// the four sums of one pixel -> the Quantum the reference stores
template<typename Q,bool BLEND>
static __device__ __forceinline__ void finish_pixel(double s0,double s1,double s2,double s3,Q (&q)[4])
{
  ResizeAcc<Q,4,BLEND,Fma64> f;
  f.s[0]=s0; f.s[1]=s1; f.s[2]=s2; f.s[3]=s3;
  f.g=0.0;
  Q copy[4]={(Q) 0,(Q) 0,(Q) 0,(Q) 0};
  f.finish(copy,0u,q);
}

// One workgroup step recomputed tap by tap, every sample multiplied only inside its output's
// window (resize.c:3494-3530 twice), from global memory: the rare path of a float frame.
template<typename Q,bool BLEND>
static __device__ __forceinline__ void careful_step(const MfmaResizeArgs &a,int x0,int x1,int y0,int y1)
{
  const Q *src=static_cast<const Q *>(a.src);
  Q *dst=static_cast<Q *>(a.dst);
  const int w=x1-x0,n=w*(y1-y0);
  for (int i=(int) threadIdx.x; i < n; i+=(int) blockDim.x)
    {
      const int y=y0+i/w,x=x0+i%w;
      const int hs=a.hstart[x],hc=a.hcount[x];
      const int vs=a.vstart[y],vc=a.vcount[y];
      ResizeAcc<Q,4,BLEND,Fma64> h;
      h.init();
      for (int j=0; j < hc; j++)
        {
          ResizeAcc<Q,4,BLEND,Fma64> v;
          v.init();
          for (int k=0; k < vc; k++)
            {
              Q p[4];
              load_pixel<Q,4>(src+((size_t) (vs+k)*(size_t) a.src_columns+(size_t) (hs+j))*4,p);
              v.tap(a.vweight[(size_t) k*(size_t) a.dst_rows+(size_t) y],0.0,p);
            }
          Q copy[4]={(Q) 0,(Q) 0,(Q) 0,(Q) 0},q[4];
          v.finish(copy,0u,q);
          h.tap(a.hweight[(size_t) j*(size_t) a.dst_columns+(size_t) x],0.0,q);
        }
      Q copy[4]={(Q) 0,(Q) 0,(Q) 0,(Q) 0},out[4];
      h.finish(copy,0u,out);
      store_pixel<Q,4>(dst+((size_t) y*(size_t) a.dst_columns+(size_t) x)*4,out);
    }
}

template<typename Q,bool BLEND>
__global__ __launch_bounds__(256,2)
void resize_mfma_kernel(MfmaResizeArgs a)
{
  constexpr bool kFloat=QuantumOps<Q>::is_float;
  constexpr int kMaxVK=MfmaResizePlan::kMaxVK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double *wlds=reinterpret_cast<double *>(smem_raw);
  float4 *patch=reinterpret_cast<float4 *>(smem_raw+a.wlds_bytes);
  __shared__ int flag[2];

  // workgroup -> (strip, chunk of steps).  Block b runs on XCD b%8: give an XCD a contiguous set
  // of strips so that neighbouring strips' shared source columns meet in one L2.
  const int xcd=(int) (blockIdx.x & 7u),slot=(int) (blockIdx.x >> 3);
  const int strip=xcd*a.strips_per_xcd+slot%a.strips_per_xcd;
  const int chunk=slot/a.strips_per_xcd;
  if (strip >= a.nstrips)
    return;
  const int tid=(int) threadIdx.x;
  const int lane=tid & 63;
  const int wave=__builtin_amdgcn_readfirstlane(tid >> 6);
  const int g=lane >> 4,n=lane & 15;
  const int col_lo=a.strip_col_lo[strip],nvb=a.strip_nvb[strip];
  const int pc=16*nvb;                       // patch pitch in pixels
  const Q *src=static_cast<const Q *>(a.src);
  Q *dst=static_cast<Q *>(a.dst);

  {
    const int count=a.strip_wcount[strip]*64;
    const double *from=a.wh+(size_t) a.strip_wbase[strip]*64;
    for (int i=tid; i < count; i+=256)
      wlds[i]=from[i];
    if (tid < 2)
      flag[tid]=0;
  }

  for (int step=0; step < a.steps; step++)
    {
      const int rg0=(chunk*a.steps+step)*MfmaResizePlan::kWaves;
      if (rg0 >= a.nrg)
        break;
      // source rows of this step's row groups
      const int prow_lo=a.rg_row_lo[rg0];
      int prow_hi=prow_lo;
      for (int i=0; (i < MfmaResizePlan::kWaves) && (rg0+i < a.nrg); i++)
        {
          const int hi=a.rg_row_lo[rg0+i]+4*a.rg_nvk[rg0+i];
          prow_hi=hi > prow_hi ? hi : prow_hi;
        }
      const int items=(prow_hi-prow_lo)*pc;
      __syncthreads();                       // the previous step's readers are done (weights staged)
      {
        constexpr int BATCH=8;
        bool bad=false;
        for (int i0=tid; i0 < items; i0+=256*BATCH)
          {
            float4 v[BATCH];
#pragma unroll
            for (int k=0; k < BATCH; k++)
              {
                int idx=i0+256*k;
                idx=idx < items ? idx : items-1;
                const int r=idx/pc,i=idx-r*pc;
                int row=prow_lo+r,col=col_lo+i;
                row=row < a.src_rows ? row : a.src_rows-1;
                col=col < a.src_columns ? col : a.src_columns-1;
                v[k]=load_patch_pixel<Q>(src+((size_t) row*(size_t) a.src_columns+(size_t) col)*4);
              }
#pragma unroll
            for (int k=0; k < BATCH; k++)
              {
                if constexpr (kFloat)
                  bad=bad || not_finite_f32(v[k].x) || not_finite_f32(v[k].y) || not_finite_f32(v[k].z) ||
                    not_finite_f32(v[k].w);
                if (i0+256*k < items)
                  patch[i0+256*k]=v[k];
              }
          }
        if constexpr (kFloat)
          if (bad)
            flag[step & 1]=1;
      }
      __syncthreads();
      bool careful=false;
      if constexpr (kFloat)
        {
          careful=flag[step & 1] != 0;
          if (tid == 0)
            flag[(step+1) & 1]=0;            // nobody reads the other word between these barriers
        }
      const int rg=rg0+wave;
      if (!careful && (rg < a.nrg))
        {
          const int rrow=a.rg_row_lo[rg]-prow_lo,nvk=a.rg_nvk[rg];
          double wvr[kMaxVK];
          {
            const double *wvp=a.wv+(size_t) a.rg_woff[rg]*64+lane;
#pragma unroll
            for (int kb=0; kb < kMaxVK; kb++)
              wvr[kb]=kb < nvk ? wvp[kb*64] : 0.0;
          }
          double ring[8][4];
#pragma unroll
          for (int s=0; s < 8; s++)
#pragma unroll
            for (int c=0; c < 4; c++)
              ring[s][c]=0.0;
          bool bad=false;
          int tdone=0;
          const int y_base=16*rg+g;
          for (int vb=0; vb < nvb; vb++)
            {
              d4 acc[4];
#pragma unroll
              for (int c=0; c < 4; c++)
                acc[c]=(d4) {0.0,0.0,0.0,0.0};
#pragma unroll
              for (int kb=0; kb < kMaxVK; kb++)
                if (kb < nvk)
                  {
                    const float4 px=patch[(rrow+4*kb+g)*pc+16*vb+n];
                    double av[4];
                    if constexpr (BLEND)
                      {
                        av[3]=(double) px.w;
                        av[0]=av[3]*(double) px.x;       // exact: two 24-bit significands
                        av[1]=av[3]*(double) px.y;
                        av[2]=av[3]*(double) px.z;
                      }
                    else
                      {
                        av[0]=(double) px.x; av[1]=(double) px.y; av[2]=(double) px.z; av[3]=(double) px.w;
                      }
#pragma unroll
                    for (int c=0; c < 4; c++)
                      acc[c]=__builtin_amdgcn_mfma_f64_16x16x4f64(av[c],wvr[kb],acc[c],0,0,0);
                  }
              // the previous block's K-blocks move down, this block's take slots 4..7
#pragma unroll
              for (int s=0; s < 4; s++)
#pragma unroll
                for (int c=0; c < 4; c++)
                  ring[s][c]=ring[s+4][c];
#pragma unroll
              for (int r=0; r < 4; r++)
                {
                  Q q[4];
                  finish_pixel<Q,BLEND>(acc[0][r],acc[1][r],acc[2][r],acc[3][r],q);
                  if constexpr (kFloat)
                    bad=bad || not_finite_f32((float) q[0]) || not_finite_f32((float) q[1]) ||
                      not_finite_f32((float) q[2]) || not_finite_f32((float) q[3]);
                  if constexpr (BLEND)
                    {
                      const double qa=(double) q[3];
                      ring[4+r][0]=qa*(double) q[0];
                      ring[4+r][1]=qa*(double) q[1];
                      ring[4+r][2]=qa*(double) q[2];
                      ring[4+r][3]=qa;
                    }
                  else
                    {
#pragma unroll
                      for (int c=0; c < 4; c++)
                        ring[4+r][c]=(double) q[c];
                    }
                }
              // out tiles whose window ends in this block
              const int tend=a.strip_ready[strip*a.nvb_max+vb];
              for (; tdone < tend; tdone++)
                {
                  const int t=strip*a.tps+tdone;
                  const int sl0=a.tile_kb0[t]-4*(vb-1),sl1=sl0+a.tile_nkb[t];
                  const double *wb=wlds+(size_t) a.tile_woff[t]*64+lane;
                  d4 o[4];
#pragma unroll
                  for (int c=0; c < 4; c++)
                    o[c]=(d4) {0.0,0.0,0.0,0.0};
#pragma unroll
                  for (int s=0; s < 8; s++)
                    if ((s >= sl0) && (s < sl1))
                      {
                        const double b=wb[(s-sl0)*64];
#pragma unroll
                        for (int c=0; c < 4; c++)
                          o[c]=__builtin_amdgcn_mfma_f64_16x16x4f64(ring[s][c],b,o[c],0,0,0);
                      }
                  const int x=16*t+n;
#pragma unroll
                  for (int r=0; r < 4; r++)
                    {
                      Q out[4];
                      finish_pixel<Q,BLEND>(o[0][r],o[1][r],o[2][r],o[3][r],out);
                      const int y=y_base+4*r;
                      if ((x < a.dst_columns) && (y < a.dst_rows))
                        store_pixel<Q,4>(dst+((size_t) y*(size_t) a.dst_columns+(size_t) x)*4,out);
                    }
                }
            }
          if constexpr (kFloat)
            if (__any(bad))
              flag[step & 1]=1;
        }
      if constexpr (kFloat)
        {
          __syncthreads();
          if (flag[step & 1] != 0)
            {
              const int x0=strip*a.tps*16;
              int x1=x0+a.tps*16,y1=16*rg0+16*MfmaResizePlan::kWaves;
              x1=x1 < a.dst_columns ? x1 : a.dst_columns;
              y1=y1 < a.dst_rows ? y1 : a.dst_rows;
              careful_step<Q,BLEND>(a,x0,x1,16*rg0,y1);
            }
        }
    }
}

// ------------------------------------------------------------------ host side
struct MfmaPlanDevice
{
  MfmaResizePlan plan;
  TableBundle tables;
  size_t i_col_lo=0,i_nvb=0,i_wbase=0,i_wcount=0,i_ready=0,i_kb0=0,i_nkb=0,i_woff=0,i_wh=0;
  size_t i_row_lo=0,i_nvk=0,i_rwoff=0,i_wv=0,i_vstart=0,i_vcount=0,i_hstart=0,i_hcount=0,i_vw=0,i_hw=0;
  hipEvent_t ready=nullptr;
  int device=-1;
  bool ok=false;
  ~MfmaPlanDevice()
  {
    if (device >= 0)
      {
        DeviceGuard guard;
        if (guard.enter(device) == hipSuccess)
          (void) hipDeviceSynchronize();     // shared across streams, as PassTables (resize.hip)
      }
    if (ready != nullptr)
      (void) hipEventDestroy(ready);
  }
};

struct MfmaPlanEntry
{
  unsigned long long vserial,hserial; int device,tps; std::shared_ptr<MfmaPlanDevice> plan;
};
static std::mutex &mfma_plans_lock() { static std::mutex &m=*new std::mutex; return m; }
static std::vector<MfmaPlanEntry> &mfma_plans() { static std::vector<MfmaPlanEntry> &v=*new std::vector<MfmaPlanEntry>; return v; }

void release_resize_mfma_plans()
{
  std::lock_guard<std::mutex> guard(mfma_plans_lock());
  mfma_plans().clear();
}

static MhStatus build_plan_device(MfmaPlanDevice &d,const TapTable &vt,const TapTable &ht,int tps,int device,
  hipStream_t stream)
{
  d.ok=build_mfma_resize_plan(d.plan,vt,ht,tps);
  if (!d.ok)
    return MH_OK;
  const MfmaResizePlan &p=d.plan;
#define MH_ADD(vec) d.tables.add((vec).data(),(vec).size()*sizeof((vec)[0]))
  d.i_col_lo=MH_ADD(p.strip_col_lo); d.i_nvb=MH_ADD(p.strip_nvb); d.i_wbase=MH_ADD(p.strip_wbase);
  d.i_wcount=MH_ADD(p.strip_wcount); d.i_ready=MH_ADD(p.strip_ready);
  d.i_kb0=MH_ADD(p.tile_kb0); d.i_nkb=MH_ADD(p.tile_nkb); d.i_woff=MH_ADD(p.tile_woff); d.i_wh=MH_ADD(p.wh);
  d.i_row_lo=MH_ADD(p.rg_row_lo); d.i_nvk=MH_ADD(p.rg_nvk); d.i_rwoff=MH_ADD(p.rg_woff); d.i_wv=MH_ADD(p.wv);
  d.i_vstart=MH_ADD(vt.start); d.i_vcount=MH_ADD(vt.count); d.i_hstart=MH_ADD(ht.start); d.i_hcount=MH_ADD(ht.count);
  d.i_vw=MH_ADD(vt.weight); d.i_hw=MH_ADD(ht.weight);
#undef MH_ADD
  MH_TRY(d.tables.upload(device,stream));
  d.device=device;
  MH_HIP(hipEventCreateWithFlags(&d.ready,hipEventDisableTiming));
  MH_HIP(hipEventRecord(d.ready,stream));
  // the weight blocks are only read by the kernel: free the host copies
  d.plan.wh.clear(); d.plan.wh.shrink_to_fit();
  d.plan.wv.clear(); d.plan.wv.shrink_to_fit();
  return MH_OK;
}

static MhStatus acquire_plan(std::shared_ptr<MfmaPlanDevice> *out,const TapTable &vt,const TapTable &ht,int tps,
  int device,hipStream_t stream)
{
  const bool shared=(vt.serial != 0) && (ht.serial != 0);
  constexpr size_t kEntries=6;
  if (shared)
    {
      std::lock_guard<std::mutex> guard(mfma_plans_lock());
      std::vector<MfmaPlanEntry> &entries=mfma_plans();
      for (size_t i=0; i < entries.size(); i++)
        if ((entries[i].vserial == vt.serial) && (entries[i].hserial == ht.serial) &&
            (entries[i].device == device) && (entries[i].tps == tps))
          {
            MfmaPlanEntry hit=entries[i];
            entries.erase(entries.begin()+(ptrdiff_t) i);
            entries.insert(entries.begin(),hit);
            *out=hit.plan;
            if (hit.plan->ok)
              MH_HIP(hipStreamWaitEvent(stream,hit.plan->ready,0));
            return MH_OK;
          }
    }
  auto built=std::make_shared<MfmaPlanDevice>();
  MH_TRY(build_plan_device(*built,vt,ht,tps,device,stream));
  *out=built;
  if (shared)
    {
      std::lock_guard<std::mutex> guard(mfma_plans_lock());
      std::vector<MfmaPlanEntry> &entries=mfma_plans();
      entries.insert(entries.begin(),MfmaPlanEntry{vt.serial,ht.serial,device,tps,built});
      if (entries.size() > kEntries)
        entries.pop_back();
    }
  return MH_OK;
}

template<typename Q,bool BLEND>
static MhStatus launch_mfma_typed(const View &src,const View &dst,const MfmaPlanDevice &d,int steps)
{
  const MfmaResizePlan &p=d.plan;
  const TableBundle &t=d.tables;
  MfmaResizeArgs a;
  a.src=src.pixels; a.dst=dst.pixels;
  a.src_columns=(int) src.columns; a.src_rows=(int) src.rows;
  a.dst_columns=(int) dst.columns; a.dst_rows=(int) dst.rows;
  a.tps=p.tps; a.nstrips=p.nstrips; a.nrg=p.nrg; a.nvb_max=p.nvb_max;
  a.strips_per_xcd=(p.nstrips+7)/8;
  a.steps=steps;
  a.wlds_bytes=(unsigned) (((size_t) p.wblocks_max*512u+15u) & ~(size_t) 15u);
  a.strip_col_lo=t.at<int>(d.i_col_lo); a.strip_nvb=t.at<int>(d.i_nvb); a.strip_wbase=t.at<int>(d.i_wbase);
  a.strip_wcount=t.at<int>(d.i_wcount); a.strip_ready=t.at<int>(d.i_ready);
  a.tile_kb0=t.at<int>(d.i_kb0); a.tile_nkb=t.at<int>(d.i_nkb); a.tile_woff=t.at<int>(d.i_woff);
  a.wh=t.at<double>(d.i_wh);
  a.rg_row_lo=t.at<int>(d.i_row_lo); a.rg_nvk=t.at<int>(d.i_nvk); a.rg_woff=t.at<int>(d.i_rwoff);
  a.wv=t.at<double>(d.i_wv);
  a.vstart=t.at<int>(d.i_vstart); a.vcount=t.at<int>(d.i_vcount);
  a.hstart=t.at<int>(d.i_hstart); a.hcount=t.at<int>(d.i_hcount);
  a.vweight=t.at<double>(d.i_vw); a.hweight=t.at<double>(d.i_hw);
  const size_t lds=(size_t) a.wlds_bytes+(size_t) p.patch_rows_max*(size_t) (16*p.nvb_max)*16u;
  const int quads=(p.nrg+MfmaResizePlan::kWaves-1)/MfmaResizePlan::kWaves;
  const int chunks=(quads+steps-1)/steps;
  dim3 grid((unsigned) (8*a.strips_per_xcd*chunks));
  if (lds > 64u*1024u)
    MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&resize_mfma_kernel<Q,BLEND>),
      hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  ProfileScope prof("resize_mfma",src.stream);
  hipLaunchKernelGGL((resize_mfma_kernel<Q,BLEND>),grid,dim3(256),lds,src.stream,a);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// *handled = false (nothing launched) when the frame or the geometry is not this kernel's.
MhStatus launch_resize_mfma(const View &src,const View &dst,const TapTable &vertical,const TapTable &horizontal,
  const Roles &roles,bool *handled)
{
  *handled=false;
  if ((src.channels != 4) || (dst.channels != 4) || (src.quantum != dst.quantum) || (roles.copy_mask != 0))
    return MH_OK;
  if (roles.blend && (roles.alpha != 3))
    return MH_OK;
  if (((int) dst.rows != vertical.out_size) || ((int) dst.columns != horizontal.out_size))
    return fail(MH_BAD_ARGUMENT,"resize: geometry mismatch");
  // enlargements only: a reduction's windows are wider than the two blocks the ring holds
  if ((dst.rows < src.rows) || (dst.columns < src.columns))
    return MH_OK;
  int tps=16,steps=8;
  if (const char *e=option("MAGICKHIP_RESIZE_MFMA_TPS"))
    tps=atoi(e) > 0 ? atoi(e) : tps;
  if (const char *e=option("MAGICKHIP_RESIZE_MFMA_STEPS"))
    steps=atoi(e) > 0 ? atoi(e) : steps;
  std::shared_ptr<MfmaPlanDevice> plan;
  MH_TRY(acquire_plan(&plan,vertical,horizontal,tps,src.device,src.stream));
  if (!plan->ok)
    return MH_OK;
  const size_t lds=(((size_t) plan->plan.wblocks_max*512u+15u) & ~(size_t) 15u)+
    (size_t) plan->plan.patch_rows_max*(size_t) (16*plan->plan.nvb_max)*16u;
  if (lds > 76u*1024u)                       // two workgroups a CU
    return MH_OK;
  *handled=true;
  if (src.quantum == MH_QUANTUM_U16)
    return roles.blend ? launch_mfma_typed<uint16_t,true>(src,dst,*plan,steps) :
                         launch_mfma_typed<uint16_t,false>(src,dst,*plan,steps);
  return roles.blend ? launch_mfma_typed<float,true>(src,dst,*plan,steps) :
                       launch_mfma_typed<float,false>(src,dst,*plan,steps);
}

} // namespace mh
