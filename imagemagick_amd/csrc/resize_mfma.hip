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
// step that holds a non-finite (or huge, see not_tame_f32) sample computes its outputs tap by tap
// in the reference's own windows and operation order instead (resize_acc.hpp: resize_redo_rect).
// So does, after the fact, a step in which an intermediate value lay on a rounding boundary or an
// alpha sum was small enough for the quotient to amplify the sums' last bits (finish_pixel): the
// intermediate is rounded, and one level of a small intermediate alpha is thousands of levels of
// the colours HorizontalFilter weights with it.
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
  int tps,ntiles,nstrips,nrg,strips_per_xcd,steps;
  int nvk_max;
  unsigned wlds_bytes,meta_bytes,wv_bytes;
  const int *strip_col_lo,*strip_nvb,*strip_wbase,*strip_wcount;
  const unsigned *tile_meta;
  const double *wh;
  const int *rg_row_lo,*rg_nvk,*rg_woff;
  const double *wv;
  // the contribution lists themselves (resize_redo_rect)
  const int *vstart,*vcount,*hstart,*hcount;
  const double *vweight,*hweight;            // [tap][out]
  unsigned patch_bytes;                      // of the source patch in LDS: resize_redo_rect's scratch
};

template<typename Q>
static __device__ __forceinline__ float4 load_patch_pixel(const Q *p)
{
  Q v[4];
  load_pixel<Q,4>(p,v);
  return make_float4((float) v[0],(float) v[1],(float) v[2],(float) v[3]);
}

// Inf, NaN or a magnitude above 2^20: such a sample sends its workgroup step down the careful
// path.  With every staged sample at most 2^20 the intermediate stays finite — gamma is at most
// QuantumRange/MagickEpsilon (PerceptibleReciprocal's clamp), so |colour| < 1.3*2^40*6.6e16 — and a
// finite intermediate times a zero weight is an exact zero.
static __device__ __forceinline__ unsigned not_tame_f32(float v)
{
  return (__builtin_bit_cast(unsigned,v) & 0x7fffffffu) > 0x49800000u ? 1u : 0u;
}

// the four sums of one pixel -> the Quantum the reference stores (ResizeAcc<Fma64>::finish), and
// whether the fused sums can vouch for it (resize_acc.hpp: TIES, the intermediate: a value too close
// to a rounding boundary; the outputs: an alpha sum so small that the quotient's error counts)
template<typename Q,bool BLEND,bool TIES>
static __device__ __forceinline__ bool finish_pixel(double s0,double s1,double s2,double s3,Q (&q)[4])
{
  double v[4]={s0,s1,s2,s3};
  bool doubt=false;
  TieWatch<Q> plain,colour;
  plain.plain();
  colour.plain();
  if constexpr (BLEND)
    {
      const double sa=s3;
      const double mag=sa < 0.0 ? -sa : sa;
      double r=__builtin_amdgcn_rcp(sa);
      double e=__builtin_fma(-sa,r,1.0);
      r=__builtin_fma(r,e,r);
      e=__builtin_fma(-sa,r,1.0);
      r=__builtin_fma(r,e,r);
      const double clamped=(sa < 0.0 ? -kInvEps : kInvEps)*kQS;
      const bool divides=(mag*kQS) >= kEps;
      const double inv=divides ? r : clamped;
      v[0]=s0*inv; v[1]=s1*inv; v[2]=s2*inv;
      if constexpr (TIES)
        {
          if (divides)
            colour.quotient(r);
        }
      else
        doubt=divides && (mag < OutputAlphaLimit<Q>::value);
      doubt=doubt || (!divides && clamped_sums_count(v));
    }
#pragma unroll
  for (int c=0; c < 4; c++)
    q[c]=QuantumOps<Q>::clamp(v[c]);
  if constexpr (TIES)
    doubt=doubt || colour.near(v[0]) || colour.near(v[1]) || colour.near(v[2]) || plain.near(v[3]);
  return doubt;
}

// the matrix chain of one out tile whose first K-block sits in ring slot S0: straight-line code
template<int S0,int NK>
static __device__ __forceinline__ void tile_chain(const double (&ring)[8][4],const double *wb,d4 (&o)[4])
{
  if constexpr (S0+NK <= 8)
    {
#pragma unroll
      for (int j=0; j < NK; j++)
        {
          const double b=wb[j*64];
#pragma unroll
          for (int c=0; c < 4; c++)
            o[c]=__builtin_amdgcn_mfma_f64_16x16x4f64(ring[S0+j][c],b,o[c],0,0,0);
        }
    }
}

template<typename Q,bool BLEND,int WAVES,int NK>
__global__ __launch_bounds__(64*WAVES)
void resize_mfma_kernel(MfmaResizeArgs a)
{
  constexpr bool kFloat=QuantumOps<Q>::is_float;
  constexpr int THREADS=64*WAVES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double *wlds=reinterpret_cast<double *>(smem_raw);                          // [tps*NK][64]
  unsigned *tmeta=reinterpret_cast<unsigned *>(smem_raw+a.wlds_bytes);       // [tps+1]
  double *wvlds=reinterpret_cast<double *>(smem_raw+a.wlds_bytes+a.meta_bytes);   // [WAVES][nvk_max][64]
  float4 *patch=reinterpret_cast<float4 *>(smem_raw+a.wlds_bytes+a.meta_bytes+a.wv_bytes);
  __shared__ int flag[2];
  __shared__ int redo[2];                    // a step of this parity met a value the fused sums cannot vouch for

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
  int strip_tiles=a.ntiles-strip*a.tps;
  strip_tiles=strip_tiles < a.tps ? strip_tiles : a.tps;

  // everything the walk looks up per tile goes to LDS once: no global load (and so no vmcnt wait
  // behind the pixel stores) inside the walk
  {
    const int count=a.strip_wcount[strip]*64;
    const double *from=a.wh+(size_t) a.strip_wbase[strip]*64;
    for (int i=tid; i < count; i+=THREADS)
      wlds[i]=from[i];
    for (int i=tid; i <= a.tps; i+=THREADS)
      tmeta[i]=i < strip_tiles ? a.tile_meta[strip*a.tps+i] : 0xff00u;        // sentinel: never ready
    if (tid < 2)
      flag[tid]=redo[tid]=0;
  }
  RedoTables tables;
  tables.vstart=a.vstart; tables.vcount=a.vcount; tables.hstart=a.hstart; tables.hcount=a.hcount;
  tables.vweight=a.vweight; tables.hweight=a.hweight;
  tables.src_columns=a.src_columns; tables.dst_columns=a.dst_columns; tables.dst_rows=a.dst_rows;
  // the outputs of the step that began at row group rg0, again, in the reference's own operation
  // order and windows (the patch is free between two steps: the rectangle's intermediate goes there)
  auto redo_step=[&](int rg0)
  {
    const int x0=strip*a.tps*16;
    int x1=x0+a.tps*16,y1=16*rg0+16*WAVES;
    x1=x1 < a.dst_columns ? x1 : a.dst_columns;
    y1=y1 < a.dst_rows ? y1 : a.dst_rows;
    resize_redo_rect<Q,BLEND,0>(tables,src,dst,x0,x1,16*rg0,y1,reinterpret_cast<unsigned char *>(patch),(int) a.patch_bytes);
  };
  int previous_rg0=-1,last_parity=0;
  // a staging thread keeps its patch column and walks rows
  const int stage_r0=tid/pc,stage_i=tid-stage_r0*pc;
  const int stage_rows=THREADS/pc;           // rows one sweep of the workgroup covers (pc <= THREADS)
  int stage_col=col_lo+stage_i;
  stage_col=stage_col < a.src_columns ? stage_col : a.src_columns-1;
  double *wvmine=wvlds+(size_t) wave*(size_t) a.nvk_max*64+lane;

  for (int step=0; step < a.steps; step++)
    {
      const int rg0=(chunk*a.steps+step)*WAVES;
      if (rg0 >= a.nrg)
        break;
      // source rows of this step's row groups
      const int prow_lo=a.rg_row_lo[rg0];
      int prow_hi=prow_lo;
#pragma unroll
      for (int i=0; i < WAVES; i++)
        if (rg0+i < a.nrg)
          {
            const int hi=a.rg_row_lo[rg0+i]+4*a.rg_nvk[rg0+i];
            prow_hi=hi > prow_hi ? hi : prow_hi;
          }
      const int prows=prow_hi-prow_lo;
      // this wave's row group
      const int rg=rg0+wave;
      const bool active=rg < a.nrg;
      const int rgc=active ? rg : a.nrg-1;
      const int rrow=a.rg_row_lo[rgc]-prow_lo,nvk=a.rg_nvk[rgc];
      const double *wvp=a.wv+(size_t) a.rg_woff[rgc]*64+lane;
      __syncthreads();                       // the previous step's readers are done (weights staged)
      if ((previous_rg0 >= 0) && (redo[(step+1) & 1] != 0))
        {
          redo_step(previous_rg0);
          if (tid == 0)
            redo[(step+1) & 1]=0;              // (written again two steps on)
        }
      previous_rg0=rg0;
      for (int kb=0; kb < nvk; kb++)         // the wave's own weight blocks, beside the patch
        wvmine[kb*64]=wvp[kb*64];
      {
        constexpr int BATCH=4;
        unsigned bad=0u;
        if (stage_r0 < stage_rows)
          for (int r0=stage_r0; r0 < prows; r0+=stage_rows*BATCH)
            {
              float4 v[BATCH];
#pragma unroll
              for (int k=0; k < BATCH; k++)
                {
                  int row=prow_lo+r0+stage_rows*k;
                  row=row < a.src_rows ? row : a.src_rows-1;
                  v[k]=load_patch_pixel<Q>(src+((size_t) row*(size_t) a.src_columns+(size_t) stage_col)*4);
                }
#pragma unroll
              for (int k=0; k < BATCH; k++)
                if (r0+stage_rows*k < prows)
                  {
                    if constexpr (kFloat)
                      bad=bad | not_tame_f32(v[k].x) | not_tame_f32(v[k].y) | not_tame_f32(v[k].z) | not_tame_f32(v[k].w);
                    patch[(r0+stage_rows*k)*pc+stage_i]=v[k];
                  }
            }
        if constexpr (kFloat)
          if (bad != 0u)
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
      if (careful)
        redo_step(rg0);                        // (the patch holds samples that are not tame: nobody multiplies them by a zero)
      else if (active)
        {
          unsigned long long doubt=0ull;
          double ring[8][4];
#pragma unroll
          for (int s=0; s < 8; s++)
#pragma unroll
            for (int c=0; c < 4; c++)
              ring[s][c]=0.0;
          int tdone=0;
          unsigned meta=__builtin_amdgcn_readfirstlane(tmeta[0]);
          const int y_base=16*rg+g;
          const float4 *prow=patch+(rrow+g)*pc+n;
          for (int vb=0; vb < nvb; vb++)
            {
              d4 acc[4];
#pragma unroll
              for (int c=0; c < 4; c++)
                acc[c]=(d4) {0.0,0.0,0.0,0.0};
              for (int kb=0; kb < nvk; kb++)
                {
                  const float4 px=prow[4*kb*pc+16*vb];
                  const double b=wvmine[kb*64];
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
                    acc[c]=__builtin_amdgcn_mfma_f64_16x16x4f64(av[c],b,acc[c],0,0,0);
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
                  doubt|=__builtin_amdgcn_ballot_w64(finish_pixel<Q,BLEND,true>(acc[0][r],acc[1][r],acc[2][r],acc[3][r],q));
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
              // out tiles whose window ends in this block (the word of the next tile is read a tile ahead)
              while ((int) (meta >> 8) <= vb)
                {
                  const int sl0=(int) (meta & 255u);
                  const double *wb=wlds+(size_t) tdone*(NK*64)+lane;
                  const int x=16*(strip*a.tps+tdone)+n;
                  tdone++;
                  meta=__builtin_amdgcn_readfirstlane(tmeta[tdone]);
                  d4 o[4];
#pragma unroll
                  for (int c=0; c < 4; c++)
                    o[c]=(d4) {0.0,0.0,0.0,0.0};
                  switch (sl0)
                  {
                    case 0: tile_chain<0,NK>(ring,wb,o); break;
                    case 1: tile_chain<1,NK>(ring,wb,o); break;
                    case 2: tile_chain<2,NK>(ring,wb,o); break;
                    case 3: tile_chain<3,NK>(ring,wb,o); break;
                    case 4: tile_chain<4,NK>(ring,wb,o); break;
                    case 5: tile_chain<5,NK>(ring,wb,o); break;
                    case 6: tile_chain<6,NK>(ring,wb,o); break;
                    default: tile_chain<7,NK>(ring,wb,o); break;
                  }
#pragma unroll
                  for (int r=0; r < 4; r++)
                    {
                      Q out[4];
                      doubt|=__builtin_amdgcn_ballot_w64(finish_pixel<Q,BLEND,false>(o[0][r],o[1][r],o[2][r],o[3][r],out));
                      const int y=y_base+4*r;
                      if ((x < a.dst_columns) && (y < a.dst_rows))
                        store_pixel<Q,4>(dst+((size_t) y*(size_t) a.dst_columns+(size_t) x)*4,out);
                    }
                }
            }
          if (doubt != 0ull)
            redo[step & 1]=1;                  // settled behind the next barrier
        }
      last_parity=step & 1;
    }
  __syncthreads();
  if ((previous_rg0 >= 0) && (redo[last_parity] != 0))
    redo_step(previous_rg0);
}

// ------------------------------------------------------------------ host side
struct MfmaPlanDevice
{
  MfmaResizePlan plan;
  TableBundle tables;
  size_t i_col_lo=0,i_nvb=0,i_wbase=0,i_wcount=0,i_meta=0,i_wh=0;
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
  unsigned long long vserial,hserial; int device,tps,waves; std::shared_ptr<MfmaPlanDevice> plan;
};
static std::mutex &mfma_plans_lock() { static std::mutex &m=*new std::mutex; return m; }
static std::vector<MfmaPlanEntry> &mfma_plans() { static std::vector<MfmaPlanEntry> &v=*new std::vector<MfmaPlanEntry>; return v; }

void release_resize_mfma_plans()
{
  std::lock_guard<std::mutex> guard(mfma_plans_lock());
  mfma_plans().clear();
}

static MhStatus build_plan_device(MfmaPlanDevice &d,const TapTable &vt,const TapTable &ht,int tps,int waves,
  int device,hipStream_t stream)
{
  d.ok=build_mfma_resize_plan(d.plan,vt,ht,tps,waves);
  if (!d.ok)
    return MH_OK;
  const MfmaResizePlan &p=d.plan;
#define MH_ADD(vec) d.tables.add((vec).data(),(vec).size()*sizeof((vec)[0]))
  d.i_col_lo=MH_ADD(p.strip_col_lo); d.i_nvb=MH_ADD(p.strip_nvb); d.i_wbase=MH_ADD(p.strip_wbase);
  d.i_wcount=MH_ADD(p.strip_wcount); d.i_meta=MH_ADD(p.tile_meta); d.i_wh=MH_ADD(p.wh);
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
  int waves,int device,hipStream_t stream)
{
  const bool shared=(vt.serial != 0) && (ht.serial != 0);
  constexpr size_t kEntries=16;
  if (shared)
    {
      std::lock_guard<std::mutex> guard(mfma_plans_lock());
      std::vector<MfmaPlanEntry> &entries=mfma_plans();
      for (size_t i=0; i < entries.size(); i++)
        if ((entries[i].vserial == vt.serial) && (entries[i].hserial == ht.serial) &&
            (entries[i].device == device) && (entries[i].tps == tps) && (entries[i].waves == waves))
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
  MH_TRY(build_plan_device(*built,vt,ht,tps,waves,device,stream));
  *out=built;
  // (the evicted plan is released after the lock: its destructor drains the device)
  std::shared_ptr<MfmaPlanDevice> evicted;
  if (shared)
    {
      std::lock_guard<std::mutex> guard(mfma_plans_lock());
      std::vector<MfmaPlanEntry> &entries=mfma_plans();
      entries.insert(entries.begin(),MfmaPlanEntry{vt.serial,ht.serial,device,tps,waves,built});
      if (entries.size() > kEntries)
        {
          evicted=std::move(entries.back().plan);
          entries.pop_back();
        }
    }
  return MH_OK;
}

static size_t mfma_weight_bytes(const MfmaResizePlan &p) { return ((size_t) p.wblocks_max*512u+15u) & ~(size_t) 15u; }
static size_t mfma_meta_bytes(const MfmaResizePlan &p) { return ((size_t) (p.tps+1)*4u+15u) & ~(size_t) 15u; }
static size_t mfma_wv_bytes(const MfmaResizePlan &p) { return (size_t) p.waves*(size_t) p.nvk_max*512u; }
static size_t mfma_lds_bytes(const MfmaResizePlan &p)
{
  return mfma_weight_bytes(p)+mfma_meta_bytes(p)+mfma_wv_bytes(p)+(size_t) p.patch_rows_max*(size_t) (16*p.nvb_max)*16u;
}

template<typename Q,bool BLEND,int WAVES,int NK>
static MhStatus launch_mfma_typed(const View &src,const View &dst,const MfmaPlanDevice &d,int steps)
{
  const MfmaResizePlan &p=d.plan;
  const TableBundle &t=d.tables;
  MfmaResizeArgs a;
  a.src=src.pixels; a.dst=dst.pixels;
  a.src_columns=(int) src.columns; a.src_rows=(int) src.rows;
  a.dst_columns=(int) dst.columns; a.dst_rows=(int) dst.rows;
  a.tps=p.tps; a.ntiles=p.ntiles; a.nstrips=p.nstrips; a.nrg=p.nrg;
  a.strips_per_xcd=(p.nstrips+7)/8;
  a.steps=steps;
  a.wlds_bytes=(unsigned) mfma_weight_bytes(p);
  a.meta_bytes=(unsigned) mfma_meta_bytes(p);
  a.wv_bytes=(unsigned) mfma_wv_bytes(p);
  a.nvk_max=p.nvk_max;
  a.patch_bytes=(unsigned) ((size_t) p.patch_rows_max*(size_t) (16*p.nvb_max)*16u);
  a.strip_col_lo=t.at<int>(d.i_col_lo); a.strip_nvb=t.at<int>(d.i_nvb); a.strip_wbase=t.at<int>(d.i_wbase);
  a.strip_wcount=t.at<int>(d.i_wcount); a.tile_meta=t.at<unsigned>(d.i_meta);
  a.wh=t.at<double>(d.i_wh);
  a.rg_row_lo=t.at<int>(d.i_row_lo); a.rg_nvk=t.at<int>(d.i_nvk); a.rg_woff=t.at<int>(d.i_rwoff);
  a.wv=t.at<double>(d.i_wv);
  a.vstart=t.at<int>(d.i_vstart); a.vcount=t.at<int>(d.i_vcount);
  a.hstart=t.at<int>(d.i_hstart); a.hcount=t.at<int>(d.i_hcount);
  a.vweight=t.at<double>(d.i_vw); a.hweight=t.at<double>(d.i_hw);
  const size_t lds=mfma_lds_bytes(p);
  const int groups=(p.nrg+WAVES-1)/WAVES;
  const int chunks=(groups+steps-1)/steps;
  dim3 grid((unsigned) (8*a.strips_per_xcd*chunks));
  if (lds > 64u*1024u)
    MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&resize_mfma_kernel<Q,BLEND,WAVES,NK>),
      hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  ProfileScope prof("resize_mfma",src.stream);
  hipLaunchKernelGGL((resize_mfma_kernel<Q,BLEND,WAVES,NK>),grid,dim3(64*WAVES),lds,src.stream,a);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

template<int WAVES,int NK>
static MhStatus launch_mfma_waves(const View &src,const View &dst,const MfmaPlanDevice &d,int steps,bool blend)
{
  if (src.quantum == MH_QUANTUM_U16)
    return blend ? launch_mfma_typed<uint16_t,true,WAVES,NK>(src,dst,d,steps) :
                   launch_mfma_typed<uint16_t,false,WAVES,NK>(src,dst,d,steps);
  return blend ? launch_mfma_typed<float,true,WAVES,NK>(src,dst,d,steps) :
                 launch_mfma_typed<float,false,WAVES,NK>(src,dst,d,steps);
}

template<int WAVES>
static MhStatus launch_mfma_blocks(const View &src,const View &dst,const MfmaPlanDevice &d,int steps,bool blend)
{
  switch (d.plan.nk)
  {
    case 1: return launch_mfma_waves<WAVES,1>(src,dst,d,steps,blend);
    case 2: return launch_mfma_waves<WAVES,2>(src,dst,d,steps,blend);
    case 3: return launch_mfma_waves<WAVES,3>(src,dst,d,steps,blend);
    case 4: return launch_mfma_waves<WAVES,4>(src,dst,d,steps,blend);
    default: return launch_mfma_waves<WAVES,5>(src,dst,d,steps,blend);
  }
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
  // 16 tiles (256 columns) a strip, four waves a workgroup: three workgroups = three waves a SIMD
  // fit a CU's LDS (weights 24 KB + patch 29 KB each for a 4x Lanczos); measured 4.83 ms against
  // 5.75 with six waves a workgroup (8192^2 -> 32768^2, profiles/r5_notes)
  int tps=16,steps=8,waves=4;
  if (const char *e=option("MAGICKHIP_RESIZE_MFMA_TPS"))
    tps=atoi(e) > 0 ? atoi(e) : tps;
  if (const char *e=option("MAGICKHIP_RESIZE_MFMA_STEPS"))
    steps=atoi(e) > 0 ? atoi(e) : steps;
  if (const char *e=option("MAGICKHIP_RESIZE_MFMA_WAVES"))
    waves=atoi(e);
  if ((waves != 4) && (waves != 6))
    waves=4;
  // (sized for the 160 KiB of a gfx950 CU: two workgroups of up to 78 KiB; a part with less declines)
  if (lds_bytes_per_workgroup(src.device) < 160*1024)
    return MH_OK;
  std::shared_ptr<MfmaPlanDevice> plan;
  // (the staging threads each keep one patch column: the patch is at most a workgroup wide; a
  // barely-enlarging geometry's wide patch gets narrower strips)
  for ( ; ; tps/=2)
    {
      MH_TRY(acquire_plan(&plan,vertical,horizontal,tps,waves,src.device,src.stream));
      if (!plan->ok || (plan->plan.nk > 5))
        return MH_OK;
      if ((mfma_lds_bytes(plan->plan) <= 78u*1024u) && (16*plan->plan.nvb_max <= 64*waves))
        break;
      if (tps <= 4)
        return MH_OK;
    }
  *handled=true;
  if (waves == 4)
    return launch_mfma_blocks<4>(src,dst,*plan,steps,roles.blend);
  return launch_mfma_blocks<6>(src,dst,*plan,steps,roles.blend);
}

} // namespace mh
