// Separable resampling passes of ResizeImage.
//
// Reference semantics restated from HorizontalFilter / VerticalFilter
// (MagickCore/resize.c:3333-3547, :3549-3759).  The contribution lists (start,
// count, normalised weights) are built on the host in double precision
// (resize_filter.cpp) — they depend only on the output index — and each pass
// evaluates, per output sample o and Update channel c,
//     plain : out = ClampToQuantum( sum_j w[j] * src[start+j] )
//     blend : a_j = (w[j]*QuantumScale) * alpha_src[start+j]
//             out = ClampToQuantum( PerceptibleReciprocal(sum a_j) * sum a_j*src[start+j] )
//     copy  : out = src[nearest]
// with the sums in ascending j, as the CPU does.
//
// MI355X mapping (a 4x Lanczos enlargement writes 16x the pixels it reads, so
// both passes are bound by the coalesced output stream):
//   vertical   lane = output column (loads and stores are whole coalesced row
//              segments); a lane owns RY consecutive output rows and walks the
//              union of their source rows once, so each source row is fetched
//              once per RY outputs instead of once per tap; the weights are
//              wave-uniform and live in SGPRs.
//   horizontal lane = output column; a workgroup stages the source span of its
//              256 output columns x TH rows in LDS with coalesced loads, each
//              lane keeps its own (per-column) weights in registers across the
//              TH rows and reads its taps from LDS (neighbouring lanes share
//              taps, so the reads broadcast).
#include "mh_internal.hpp"
#include "resize_filter.hpp"
#include <memory>
#include <mutex>
#include "device_common.hpp"
#include "resize_acc.hpp"
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace mh {

struct ResizeArgs
{
  const void *src;
  void *dst;
  int src_columns,src_rows;
  int dst_columns,dst_rows;
  int out_size;               // size of the resampled axis in dst
  int max_taps;
  const int *start;
  const int *count;
  const int *nearest;
  const void *weight;         // T[max_taps][out_size]
  const void *weight_qs;      // T[max_taps][out_size] : weight*QuantumScale
  const int *tile_lo;         // horizontal pass: first source column of each 256-column tile
  const int *tile_span;       //                  and the number of source columns it needs
  uint32_t copy_mask;
};


// ------------------------------------------------------------- vertical pass
// Dense form: for a tile of RY consecutive output rows the host flattens the
// contribution lists into a small matrix W[k][r] (k = source row lo+k of the
// tile's union window, r = output row of the tile; 0 where row k is outside
// r's window), so the kernel is a loop over the union rows that loads each
// source row once and feeds all RY accumulators with wave-uniform weights —
// no per-(row, output) window tests.  Zero weights are exact no-ops
// (s + 0*p == s) for finite pixels; a FLOAT frame still skips them (both
// policies) through a per-row bit mask — a wave-uniform test, the weights are
// scalars — so that a non-finite pixel outside an output's window cannot leak
// into it (0*inf = NaN; resize.c:3494-3530 only ever multiplies the samples of
// the window).
struct VerticalDenseArgs
{
  const void *src;
  void *dst;
  int columns;                // == source columns
  int out_rows;
  int kmax;                   // rows of the widest union window
  const int *tile_lo;         // [tiles] first source row of the union window
  const int *tile_rows;       // [tiles] its height (<= kmax)
  const unsigned *row_mask;   // [tiles][kmax] bit r set: source row k contributes to output r
  const double *w;            // [tiles][kmax][RY]
  const double *wq;           // [tiles][kmax][RY]  weight*QuantumScale
  const int *nearest;         // [out_rows] source row of Copy-trait channels
  const int *count;           // [out_rows]
  uint32_t copy_mask;
};

template<typename Q,int C,bool BLEND,class A,int RY>
__global__ __launch_bounds__(256)
void resize_vertical_kernel(VerticalDenseArgs args)
{
  typedef typename A::T T;
  constexpr bool kSkipZeros=QuantumOps<Q>::is_float;
  const int W=args.columns;
  const int lane_x=(int) (blockIdx.x*blockDim.x+threadIdx.x);
  const int tile=(int) blockIdx.y;
  const int y0=tile*RY;
  const int x=lane_x < W ? lane_x : W-1;
  const Q *src=static_cast<const Q *>(args.src)+(size_t) x*C;
  Q *dst=static_cast<Q *>(args.dst);
  const size_t pitch=(size_t) W*C;
  const int lo=args.tile_lo[tile],nrows=args.tile_rows[tile];
  const size_t tbase=(size_t) tile*(size_t) args.kmax;
  const double *wt=args.w+tbase*RY;
  const double *wqt=args.wq+tbase*RY;
  const unsigned *mask=args.row_mask+tbase;

  ResizeAcc<Q,C,BLEND,A> acc[RY];
#pragma unroll
  for (int r=0; r < RY; r++)
    acc[r].init();
  constexpr int KB=4;                       // source rows fetched per batch
  for (int k0=0; k0 < nrows; k0+=KB)
    {
      Q q[KB][C];
#pragma unroll
      for (int kk=0; kk < KB; kk++)
        {
          int k=k0+kk;
          k=k < nrows ? k : nrows-1;
          load_pixel<Q,C>(src+(size_t) (lo+k)*pitch,q[kk]);
        }
#pragma unroll
      for (int kk=0; kk < KB; kk++)
        {
          const int k=k0+kk;
          if (k < nrows)
            {
              const unsigned m=kSkipZeros ? mask[k] : 0xffffffffu;
#pragma unroll
              for (int r=0; r < RY; r++)
                if (!kSkipZeros || ((m >> r) & 1u))
                  acc[r].tap((T) wt[(size_t) k*RY+r],BLEND ? (T) wqt[(size_t) k*RY+r] : (T) 0,q[kk]);
            }
        }
    }
  if (lane_x >= W)
    return;
#pragma unroll
  for (int r=0; r < RY; r++)
    {
      int y=y0+r;
      if ((y < args.out_rows) && (args.count[y] > 0))
        {
          Q copy[C],out[C];
#pragma unroll
          for (int c=0; c < C; c++)
            copy[c]=(Q) 0;
          if (args.copy_mask != 0)
            load_pixel<Q,C>(src+(size_t) args.nearest[y]*pitch,copy);
          acc[r].finish(copy,args.copy_mask,out);
          store_pixel<Q,C>(dst+(size_t) y*pitch+(size_t) x*C,out);
        }
    }
}

// ----------------------------------------------------------- horizontal pass
template<typename Q,int C,bool BLEND,class A,int MAXT>
__global__ __launch_bounds__(256)
void resize_horizontal_kernel(ResizeArgs args,int tile_rows)
{
  typedef typename A::T T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Q *tile=reinterpret_cast<Q *>(smem_raw);
  const int OUT=args.out_size;              // dst_columns
  const int x0=(int) blockIdx.x*256;
  const int x=x0+(int) threadIdx.x;
  const int y0=(int) blockIdx.y*tile_rows;
  const Q *src=static_cast<const Q *>(args.src);
  Q *dst=static_cast<Q *>(args.dst);
  const T *weight=static_cast<const T *>(args.weight);
  const T *weight_qs=static_cast<const T *>(args.weight_qs);
  const size_t src_pitch=(size_t) args.src_columns*C;
  const size_t dst_pitch=(size_t) args.dst_columns*C;

  // source span of this tile (computed on the host, wave-uniform)
  const int lo=args.tile_lo[blockIdx.x],span=args.tile_span[blockIdx.x];
  int rows=args.dst_rows-y0;
  rows=rows < tile_rows ? rows : tile_rows;
  {
    // eight pixels of a thread in flight together: a reduction stages 4x the columns it writes (7 rows x 1048 pixels a
    // workgroup for 4x Lanczos), and one pixel at a time that was 29 memory round trips in a row — most of the launch
    constexpr int BATCH=8;
    const int items=span*rows;
    for (int i0=(int) threadIdx.x; i0 < items; i0+=256*BATCH)
      {
        Q v[BATCH][C];
#pragma unroll
        for (int k=0; k < BATCH; k++)
          {
            int idx=i0+256*k;
            idx=idx < items ? idx : items-1;
            const int r=idx/span,i=idx-r*span;
            load_pixel<Q,C>(src+(size_t) (y0+r)*src_pitch+(size_t) (lo+i)*C,v[k]);
          }
#pragma unroll
        for (int k=0; k < BATCH; k++)
          if (i0+256*k < items)
            store_pixel<Q,C>(tile+(size_t) (i0+256*k)*C,v[k]);
      }
  }
  __syncthreads();
  if (x >= OUT)
    return;
  const int start=args.start[x]-lo;
  const int count=args.count[x];
  if (count <= 0)
    return;
  const int nearest=args.nearest[x]-lo;
  // the lane's weights, in registers across the tile's rows (weight*QuantumScale is formed where it is used: the
  // same double product the host's table holds, and half the registers — with both arrays the compiler gave up
  // at 32 contributions and re-read them from memory for every row)
  T w[MAXT > 0 ? MAXT : 1];
  if constexpr (MAXT > 0)
    {
#pragma unroll
      for (int j=0; j < MAXT; j++)
        {
          w[j]=(T) 0;
          if (j < count)
            w[j]=weight[(size_t) j*OUT+x];
        }
    }
  for (int r=0; r < rows; r++)
    {
      const Q *line=tile+(size_t) r*span*C;
      ResizeAcc<Q,C,BLEND,A> acc;
      acc.init();
      if constexpr (MAXT > 0)
        {
#pragma unroll
          for (int j=0; j < MAXT; j++)
            if (j < count)
              {
                Q q[C];
                load_pixel<Q,C>(line+(size_t) (start+j)*C,q);
                acc.tap(w[j],BLEND ? w[j]*(T) kQS : (T) 0,q);
              }
        }
      else
        {
          for (int j=0; j < count; j++)
            {
              Q q[C];
              load_pixel<Q,C>(line+(size_t) (start+j)*C,q);
              T wj=weight[(size_t) j*OUT+x];
              T wqj=BLEND ? weight_qs[(size_t) j*OUT+x] : (T) 0;
              acc.tap(wj,wqj,q);
            }
        }
      Q copy[C],out[C];
      load_pixel<Q,C>(line+(size_t) nearest*C,copy);
      acc.finish(copy,args.copy_mask,out);
      store_pixel<Q,C>(dst+(size_t) (y0+r)*dst_pitch+(size_t) x*C,out);
    }
}


// --------------------------------------------- horizontal pass, converted tile
// Same tiling as resize_horizontal_kernel, but the staged source span is
// converted to the accumulation type once, when it is written to LDS (each
// source sample is used by ~4*taps output pixels of a 4x enlargement, and a
// Quantum->double conversion costs as much issue time as a multiply-add), and
// every lane runs the same number of taps (the tile's maximum; the extra taps
// carry zero weights) so the tap loop has no per-lane predicates.
template<typename Q,int C,bool BLEND,class A,int MAXT,bool PREMUL=false>
__global__ __launch_bounds__(256)
void resize_horizontal_cvt_kernel(ResizeArgs args,int tile_rows,int lds_span)
{
  typedef typename A::T T;
  // PREMUL (Fma64, alpha-weighted, every channel updated): the tile holds alpha*colour, alpha
  static_assert(!PREMUL || ResizeAcc<Q,C,BLEND,A>::kDerive,"premultiplied staging needs the derived gamma");
  // Zero-weight padding taps (j >= count) multiply a sample OUTSIDE the output's window by 0: an
  // exact no-op unless that sample is not finite (0*inf = NaN where the reference never looks,
  // resize.c:3494-3530).  EXACT on a float frame tests every tap; FAST on a float frame looks at
  // the samples while it stages them and only a tile that holds a non-finite one takes the tested
  // loop (kWatch) — two integer instructions per staged sample instead of a test per tap.
  constexpr bool kSkipZeros=std::is_same<A,Exact64>::value && QuantumOps<Q>::is_float;
  constexpr bool kWatch=!kSkipZeros && QuantumOps<Q>::is_float;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *tile=reinterpret_cast<T *>(smem_raw);
  __shared__ int wave_not_finite[4];            // one word per wave of the workgroup: written once, no init
  bool not_finite=false;
  const int OUT=args.out_size;
  const int x0=(int) blockIdx.x*256;
  const int x=x0+(int) threadIdx.x;
  const int y0=(int) blockIdx.y*tile_rows;
  const Q *src=static_cast<const Q *>(args.src);
  Q *dst=static_cast<Q *>(args.dst);
  const T *weight=static_cast<const T *>(args.weight);
  const T *weight_qs=static_cast<const T *>(args.weight_qs);
  const size_t src_pitch=(size_t) args.src_columns*C;
  const size_t dst_pitch=(size_t) args.dst_columns*C;
  const int lo=args.tile_lo[blockIdx.x];
  int rows=args.dst_rows-y0;
  rows=rows < tile_rows ? rows : tile_rows;

  // stage lds_span columns (the tile's span plus the zero-weight overhang,
  // clamped to the image) of `rows` rows, converted to T
  {
    constexpr int BATCH=4;
    const int items=lds_span*rows;
    for (int i0=(int) threadIdx.x; i0 < items; i0+=256*BATCH)
      {
        Q v[BATCH][C];
        int slot[BATCH];
#pragma unroll
        for (int k=0; k < BATCH; k++)
          {
            int idx=i0+256*k;
            idx=idx < items ? idx : items-1;
            const int r=idx/lds_span,i=idx-r*lds_span;
            int col=lo+i;
            col=col < args.src_columns ? col : args.src_columns-1;
            slot[k]=idx;
            load_pixel<Q,C>(src+(size_t) (y0+r)*src_pitch+(size_t) col*C,v[k]);
          }
        if constexpr (kWatch)
          {
#pragma unroll
            for (int k=0; k < BATCH; k++)
#pragma unroll
              for (int c=0; c < C; c++)
                not_finite=not_finite || ((__builtin_bit_cast(unsigned,(float) v[k][c]) & 0x7f800000u) == 0x7f800000u);
          }
#pragma unroll
        for (int k=0; k < BATCH; k++)
          if (i0+256*k < items)
            {
              const T scale=PREMUL ? (T) v[k][C-1] : (T) 1;
#pragma unroll
              for (int c=0; c < C; c++)
                tile[(size_t) slot[k]*C+c]=(PREMUL && (c != C-1)) ? scale*(T) v[k][c] : (T) v[k][c];
            }
      }
  }
  if constexpr (kWatch)
    {
      const int raised=__any(not_finite) ? 1 : 0;
      if ((threadIdx.x & 63) == 0)
        wave_not_finite[threadIdx.x >> 6]=raised;
    }
  __syncthreads();
  const bool careful=kSkipZeros ||
    (kWatch && ((wave_not_finite[0] | wave_not_finite[1] | wave_not_finite[2] | wave_not_finite[3]) != 0));
  if (x >= OUT)
    return;
  const int start=args.start[x]-lo;
  const int count=args.count[x];
  if (count <= 0)
    return;
  const int nearest=args.nearest[x]-lo;
  T w[MAXT],wq[MAXT];
#pragma unroll
  for (int j=0; j < MAXT; j++)
    {
      w[j]=(T) 0;
      wq[j]=(T) 0;
      if (j < count)
        {
          w[j]=weight[(size_t) j*OUT+x];
          if constexpr (BLEND)
            wq[j]=weight_qs[(size_t) j*OUT+x];
        }
    }
  for (int r=0; r < rows; r++)
    {
      const T *line=tile+(size_t) r*lds_span*C+(size_t) start*C;
      ResizeAcc<Q,C,BLEND,A> acc;
      acc.init();
      // all MAXT samples are read first (the staged span has the overhang for it),
      // so the LDS reads are in flight together instead of one latency per tap
      T p[MAXT][C];
#pragma unroll
      for (int j=0; j < MAXT; j++)
#pragma unroll
        for (int c=0; c < C; c++)
          p[j][c]=line[(size_t) j*C+c];
#pragma unroll
      for (int j=0; j < MAXT; j++)
        if (!careful || (j < count))              // zero-weight taps are exact no-ops on finite samples
          {
            if constexpr (PREMUL)
              acc.tap_premultiplied(w[j],p[j]);
            else
              acc.tap_converted(w[j],wq[j],p[j]);
          }
      Q copy[C],out[C];
#pragma unroll
      for (int c=0; c < C; c++)
        copy[c]=(Q) 0;
      if (args.copy_mask != 0)
        {
#pragma unroll
          for (int c=0; c < C; c++)
            copy[c]=(Q) tile[((size_t) r*lds_span+(size_t) nearest)*C+c];   // exact: it was a Quantum
        }
      acc.finish(copy,args.copy_mask,out);
      store_pixel<Q,C>(dst+(size_t) (y0+r)*dst_pitch+(size_t) x*C,out);
    }
}

// VerticalFilter followed by HorizontalFilter in one launch.  *handled is false
// (and nothing was launched) when the tiles do not fit LDS or the tap count is
// too large; the caller then runs the two passes separately.
MhStatus launch_resize_fused(const View &src,const View &dst,const TapTable &vertical,
  const TapTable &horizontal,const Roles &roles,MhPrecision prec,bool *handled)
{
  *handled=false;
  if ((src.channels != dst.channels) || (src.quantum != dst.quantum))
    return fail(MH_BAD_ARGUMENT,"resize: layout mismatch");
  if (roles.blend && (roles.alpha != src.channels-1))
    return MH_OK;
  if (((int) dst.rows != vertical.out_size) || ((int) dst.columns != horizontal.out_size))
    return fail(MH_BAD_ARGUMENT,"resize: geometry mismatch");
  // The one-launch forms are FAST forms (fused multiply-adds, derived gamma: within one ULP / one
  // level).  (Rounds 1 and 2 had two EXACT-capable forms that kept the intermediate tile in LDS and
  // cost as much as the two passes together — 8.6 and 6.0 ms on config C3; removed in round 5,
  // DESIGN.md section 4.3.)
  if (prec != MH_PRECISION_FAST)
    return MH_OK;
  // The one-launch walks need a frame that fills the chip: a wave of the streaming form walks a strip of 58 source
  // columns down 32..128 result rows, a workgroup of the matrix form a 16-column strip.  Below that the two passes
  // are the faster FAST form (tools/probe_resize_rows.py, Lanczos RGBA: 512^2 x2 0.05-0.25 ms against 0.03;
  // 2048^2 x3 0.28-0.44 against 0.26; 4096^2 x3 0.92 against 1.06, x1.5 on the matrix pipe 0.57 against 0.48;
  // 8192^2 x2.5 3.35 against 3.89).  MAGICKHIP_RESIZE_ONE_LAUNCH_MIN_PIXELS (MhSetOption) moves both thresholds —
  // the suites set it to 0 so that small frames keep exercising these kernels.
  const long long pixels=(long long) src.columns*(long long) src.rows;
  const long long forced=option_long("MAGICKHIP_RESIZE_ONE_LAUNCH_MIN_PIXELS",-1);
  const long long stream_from=forced >= 0 ? forced : 6000000ll,matrix_from=forced >= 0 ? forced : 40000000ll;
  // four channels, enlargement by a whole-number horizontal factor: plain fp64 multiply-adds out of
  // registers with scalar-register weights (resize_stream.hip); MAGICKHIP_NO_RESIZE_STREAM=1 skips it
  if ((option("MAGICKHIP_NO_RESIZE_STREAM") == nullptr) && (pixels >= stream_from))
    {
      MH_TRY(launch_resize_stream(src,dst,vertical,horizontal,roles,handled));
      if (*handled)
        return MH_OK;
    }
  // any other enlargement of a four-channel frame: both filters on the fp64 matrix pipe, the
  // intermediate in registers (resize_mfma.hip); MAGICKHIP_NO_RESIZE_MFMA=1 keeps the two passes
  if ((option("MAGICKHIP_NO_RESIZE_MFMA") == nullptr) && (pixels >= matrix_from))
    return launch_resize_mfma(src,dst,vertical,horizontal,roles,handled);
  return MH_OK;
}

// ---------------------------------------------------------------- launcher
// Everything a resize pass reads besides the pixels, resident on the device: the contribution
// lists, the per-tile dense weight matrices of the vertical kernel or the per-tile spans of the
// horizontal one, and the launch geometry that follows from them.  Built once per (contribution
// table, axis, weight type) and kept (a pass of config C3 uploads 4 MB: 70-140 us of idle GPU in
// front of a 1.3 ms kernel, and ~0.5 ms of host loops per call).
struct PassTables
{
  TableBundle tables;
  size_t i_start=0,i_count=0,i_near=0,i_w=0,i_wq=0;
  // vertical: dense W[tile][k][RY]
  size_t i_lo=0,i_rows=0,i_mask=0,i_dw=0,i_dwq=0;
  int tiles=0,kmax=1,ry=0;
  // horizontal: per 256-column tile
  size_t i_tile_lo=0,i_tile_span=0;
  int max_span=1,overhang=0,maxt=8;
  hipEvent_t ready=nullptr;          // the upload, for callers on another stream
  int device=-1;
  ~PassTables()
  {
    // The tables are shared across streams; their block goes back to a pool that tags it with
    // the BUILDER's stream only.  Evictions are rare (more than eight geometries in flight):
    // wait for the device, so that no kernel of another stream can still be reading them.
    if (device >= 0)
      {
        DeviceGuard guard;
        if (guard.enter(device) == hipSuccess)
          (void) hipDeviceSynchronize();
      }
    if (ready != nullptr)
      (void) hipEventDestroy(ready);
  }
};

#ifndef MH_VERTICAL_ROWS
#define MH_VERTICAL_ROWS 4
#endif
constexpr int kVerticalRows=MH_VERTICAL_ROWS;      // output rows per tile of resize_vertical_kernel
// (A reduction's windows are long and overlap — 4x Lanczos: 25 source rows per output, consecutive outputs four
// rows apart — so a tile of 4 outputs reads 40 rows for 16 new ones, the frame 2.5 times: the pass is bound by
// that traffic, 0.27 ms for 8192^2 -> 8192x2048 Q16.  Sixteen outputs a tile read it 1.4 times but, in the dense
// form, run 88 x 16 multiply-adds for 16 x 25 useful ones: 0.82 ms; profiles/r6_notes/resize_reduction.txt.)
static inline int vertical_rows_of(const TapTable &)
{
  return kVerticalRows;
}

template<typename T>
static MhStatus build_pass_tables(PassTables &p,const TapTable &table,bool vertical,int device,
  hipStream_t stream)
{
  const size_t n=(size_t) table.max_taps*(size_t) table.out_size;
  std::vector<T> w(n),wq(n);
  for (size_t i=0; i < n; i++)
    {
      w[i]=(T) table.weight[i];
      wq[i]=(T) (table.weight[i]*kQuantumScale);     // contribution.weight*QuantumScale
    }
  const size_t ib=(size_t) table.out_size*sizeof(int);
  p.i_start=p.tables.add(table.start.data(),ib);
  p.i_count=p.tables.add(table.count.data(),ib);
  p.i_near=p.tables.add(table.nearest.data(),ib);
  p.i_w=p.tables.add(w.data(),n*sizeof(T));
  p.i_wq=p.tables.add(wq.data(),n*sizeof(T));
  std::vector<int> lo,nrows,tile_lo,tile_span;
  std::vector<double> dw,dwq;
  std::vector<unsigned> mask;
  if (vertical)
    {
      const int RY=vertical_rows_of(table);
      p.ry=RY;
      const int tiles=(table.out_size+RY-1)/RY;
      lo.assign((size_t) tiles,0);
      nrows.assign((size_t) tiles,0);
      int kmax=1;
      for (int t=0; t < tiles; t++)
        {
          int l=0x7fffffff,h=0;
          for (int r=0; r < RY; r++)
            {
              int y=t*RY+r;
              if ((y >= table.out_size) || (table.count[(size_t) y] <= 0))
                continue;
              int st=table.start[(size_t) y],en=st+table.count[(size_t) y];
              l=st < l ? st : l;
              h=en > h ? en : h;
            }
          if (h <= l)
            { l=0; h=0; }
          lo[(size_t) t]=l;
          nrows[(size_t) t]=h-l;
          kmax=(h-l) > kmax ? (h-l) : kmax;
        }
      dw.assign((size_t) tiles*kmax*RY,0.0);
      dwq.assign((size_t) tiles*kmax*RY,0.0);
      mask.assign((size_t) tiles*kmax,0u);
      for (int t=0; t < tiles; t++)
        for (int r=0; r < RY; r++)
          {
            int y=t*RY+r;
            if ((y >= table.out_size) || (table.count[(size_t) y] <= 0))
              continue;
            for (int j=0; j < table.count[(size_t) y]; j++)
              {
                int k=table.start[(size_t) y]+j-lo[(size_t) t];
                double wv=table.weight[(size_t) j*table.out_size+(size_t) y];
                size_t at=((size_t) t*kmax+(size_t) k)*RY+(size_t) r;
                dw[at]=wv;
                dwq[at]=wv*kQuantumScale;       // contribution.weight*QuantumScale
                mask[(size_t) t*kmax+(size_t) k]|=1u << r;
              }
          }
      p.tiles=tiles;
      p.kmax=kmax;
      p.i_lo=p.tables.add(lo.data(),lo.size()*sizeof(int));
      p.i_rows=p.tables.add(nrows.data(),nrows.size()*sizeof(int));
      p.i_mask=p.tables.add(mask.data(),mask.size()*sizeof(unsigned));
      p.i_dw=p.tables.add(dw.data(),dw.size()*sizeof(double));
      p.i_dwq=p.tables.add(dwq.data(),dwq.size()*sizeof(double));
    }
  else
    {
      // widest source span of any 256-column tile decides how many rows fit in LDS
      int max_span=1;
      for (int x0=0; x0 < table.out_size; x0+=256)
        {
          int xh=(x0+255) < table.out_size ? (x0+255) : table.out_size-1;
          int l=table.start[(size_t) x0],hi=0;
          for (int i=x0; i <= xh; i++)
            {
              int st=table.start[(size_t) i],e=st+table.count[(size_t) i];
              l=st < l ? st : l;
              hi=e > hi ? e : hi;
            }
          if (hi < l)
            hi=l;
          tile_lo.push_back(l);
          tile_span.push_back(hi-l);
          if ((hi-l) > max_span)
            max_span=hi-l;
        }
      // per tile: the largest tap count (run by every lane of the converted-tile kernel)
      const size_t ntiles=tile_lo.size();
      // (a 4x Lanczos enlargement has 6 or 7 contributions per output: 7 taps, not 8)
      const int maxt=table.max_taps <= 4 ? 4 : (table.max_taps <= 6 ? 6 : (table.max_taps <= 7 ? 7 : 8));
      int overhang=0;
      for (size_t t=0; t < ntiles; t++)
        {
          int x0t=(int) t*256,xh=(x0t+255) < table.out_size ? (x0t+255) : table.out_size-1;
          int cmaxt=0,endmax=0;
          for (int i=x0t; i <= xh; i++)
            cmaxt=table.count[(size_t) i] > cmaxt ? table.count[(size_t) i] : cmaxt;
          for (int i=x0t; i <= xh; i++)
            {
              int e=table.start[(size_t) i]+maxt;       // the kernel reads MAXT samples per output
              endmax=e > endmax ? e : endmax;
            }
          int over=endmax-(tile_lo[t]+tile_span[t]);
          overhang=over > overhang ? over : overhang;
          tile_span.push_back(cmaxt);                 // second half of the array: taps per tile
        }
      p.max_span=max_span;
      p.overhang=overhang;
      p.maxt=maxt;
      p.i_tile_lo=p.tables.add(tile_lo.data(),tile_lo.size()*sizeof(int));
      p.i_tile_span=p.tables.add(tile_span.data(),tile_span.size()*sizeof(int));
    }
  MH_TRY(p.tables.upload(device,stream));
  p.device=device;
  MH_HIP(hipEventCreateWithFlags(&p.ready,hipEventDisableTiming));
  MH_HIP(hipEventRecord(p.ready,stream));
  return MH_OK;
}

// cached by the serial number of a shared contribution table (resize_filter.cpp); tables built
// for one call (serial 0: the MagickCore shim's callback filters) are not kept
struct PassTablesEntry { unsigned long long serial; int device; bool vertical; size_t weight_bytes; std::shared_ptr<PassTables> tables; };
// (never destroyed: at process exit the runtime the device blocks belong to may be gone)
static std::mutex &pass_tables_lock() { static std::mutex &m=*new std::mutex; return m; }
static std::vector<PassTablesEntry> &pass_tables() { static std::vector<PassTablesEntry> &v=*new std::vector<PassTablesEntry>; return v; }

void release_resize_tables()
{
  std::lock_guard<std::mutex> guard(pass_tables_lock());
  pass_tables().clear();
}

template<typename T>
static MhStatus acquire_pass_tables(std::shared_ptr<PassTables> *out,const TapTable &table,bool vertical,
  int device,hipStream_t stream)
{
  typedef PassTablesEntry Entry;
  std::mutex &lock=pass_tables_lock();
  std::vector<Entry> &entries=pass_tables();
  constexpr size_t kEntries=8;
  if (table.serial != 0)
    {
      std::lock_guard<std::mutex> guard(lock);
      for (size_t i=0; i < entries.size(); i++)
        if ((entries[i].serial == table.serial) && (entries[i].device == device) &&
            (entries[i].vertical == vertical) && (entries[i].weight_bytes == sizeof(T)))
          {
            Entry hit=entries[i];
            entries.erase(entries.begin()+(ptrdiff_t) i);
            entries.insert(entries.begin(),hit);
            *out=hit.tables;
            MH_HIP(hipStreamWaitEvent(stream,hit.tables->ready,0));
            return MH_OK;
          }
    }
  auto built=std::make_shared<PassTables>();
  MH_TRY(build_pass_tables<T>(*built,table,vertical,device,stream));
  *out=built;
  if (table.serial != 0)
    {
      std::lock_guard<std::mutex> guard(lock);
      entries.insert(entries.begin(),Entry{table.serial,device,vertical,sizeof(T),built});
      if (entries.size() > kEntries)
        entries.pop_back();
    }
  return MH_OK;
}

template<typename Q,int C,bool BLEND,class A>
static MhStatus launch_typed(const View &src,const View &dst,bool vertical,
  const TapTable &table,const Roles &roles)
{
  typedef typename A::T T;
  std::shared_ptr<PassTables> pass;
  MH_TRY(acquire_pass_tables<T>(&pass,table,vertical,src.device,src.stream));
  const TableBundle &tables=pass->tables;

  ResizeArgs args;
  args.src=src.pixels;
  args.dst=dst.pixels;
  args.src_columns=(int) src.columns;
  args.src_rows=(int) src.rows;
  args.dst_columns=(int) dst.columns;
  args.dst_rows=(int) dst.rows;
  args.out_size=table.out_size;
  args.max_taps=table.max_taps;
  args.copy_mask=roles.copy_mask;
  args.start=tables.at<int>(pass->i_start);
  args.count=tables.at<int>(pass->i_count);
  args.nearest=tables.at<int>(pass->i_near);
  args.weight=tables.at<void>(pass->i_w);
  args.weight_qs=tables.at<void>(pass->i_wq);
  args.tile_lo=nullptr;
  args.tile_span=nullptr;

  if (vertical)
    {
      VerticalDenseArgs va;
      va.src=src.pixels;
      va.dst=dst.pixels;
      va.columns=(int) dst.columns;
      va.out_rows=table.out_size;
      va.kmax=pass->kmax;
      va.tile_lo=tables.at<int>(pass->i_lo);
      va.tile_rows=tables.at<int>(pass->i_rows);
      va.row_mask=tables.at<unsigned>(pass->i_mask);
      va.w=tables.at<double>(pass->i_dw);
      va.wq=tables.at<double>(pass->i_dwq);
      va.nearest=tables.at<int>(pass->i_near);
      va.count=tables.at<int>(pass->i_count);
      va.copy_mask=roles.copy_mask;
      dim3 grid((unsigned) ((dst.columns+255)/256),(unsigned) pass->tiles);
      ProfileScope prof("resize_vertical",src.stream);
      hipLaunchKernelGGL((resize_vertical_kernel<Q,C,BLEND,A,kVerticalRows>),grid,dim3(256),0,src.stream,va);
    }
  else
    {
      const int max_span=pass->max_span,overhang=pass->overhang,maxt=pass->maxt;
      args.tile_lo=tables.at<int>(pass->i_tile_lo);
      args.tile_span=tables.at<int>(pass->i_tile_span);
      const size_t px=(size_t) C*sizeof(Q);
      const size_t budget=60u*1024u;
      if (table.max_taps <= 8)
        {
          const int lds_span=max_span+overhang;
          const size_t cpx=(size_t) C*sizeof(T);
          if ((size_t) lds_span*cpx <= 150u*1024u)
            {
              int tile_rows=(int) (budget/((size_t) lds_span*cpx));
              int cap=16;
              if (const char *e=option("MAGICKHIP_HTILE"))
                cap=atoi(e);
              tile_rows=tile_rows < 1 ? 1 : (tile_rows > cap ? cap : tile_rows);
              size_t lds=(size_t) lds_span*cpx*(size_t) tile_rows;
              dim3 grid((unsigned) ((dst.columns+255)/256),(unsigned) ((dst.rows+tile_rows-1)/tile_rows));
              ProfileScope prof("resize_horizontal",src.stream);
              constexpr bool kCanPremultiply=ResizeAcc<Q,C,BLEND,A>::kDerive;
              const bool premultiply=kCanPremultiply && (args.copy_mask == 0) &&
                (option("MAGICKHIP_NO_RESIZE_PREMULTIPLY") == nullptr);
#define MH_LAUNCH_H(N)                                                                        \
              {                                                                                \
                if (premultiply)                                                               \
                  {                                                                            \
                    if (lds > 64u*1024u)                                                       \
                      MH_HIP(hipFuncSetAttribute(                                              \
                        reinterpret_cast<const void *>(&resize_horizontal_cvt_kernel<Q,C,BLEND,A,N,kCanPremultiply>),\
                        hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));                \
                    hipLaunchKernelGGL((resize_horizontal_cvt_kernel<Q,C,BLEND,A,N,kCanPremultiply>),grid,dim3(256),lds, \
                      src.stream,args,tile_rows,lds_span);                                     \
                  }                                                                            \
                else                                                                           \
                  {                                                                            \
                    if (lds > 64u*1024u)                                                       \
                      MH_HIP(hipFuncSetAttribute(                                              \
                        reinterpret_cast<const void *>(&resize_horizontal_cvt_kernel<Q,C,BLEND,A,N>),\
                        hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));                \
                    hipLaunchKernelGGL((resize_horizontal_cvt_kernel<Q,C,BLEND,A,N>),grid,dim3(256),lds, \
                      src.stream,args,tile_rows,lds_span);                                     \
                  }                                                                            \
              }
              if (maxt == 4) MH_LAUNCH_H(4)
              else if (maxt == 6) MH_LAUNCH_H(6)
              else if (maxt == 7) MH_LAUNCH_H(7)
              else MH_LAUNCH_H(8)
#undef MH_LAUNCH_H
              MH_HIP(hipGetLastError());
              return MH_OK;
            }
        }
      if ((size_t) max_span*px > 150u*1024u)
        return fail(MH_UNSUPPORTED,"resize: source span of %d pixels does not fit LDS",max_span);
      int tile_rows=(int) (budget/((size_t) max_span*px));
      tile_rows=tile_rows < 1 ? 1 : (tile_rows > 16 ? 16 : tile_rows);
      size_t lds=(size_t) max_span*px*(size_t) tile_rows;
      dim3 grid((unsigned) ((dst.columns+255)/256),(unsigned) ((dst.rows+tile_rows-1)/tile_rows));
      ProfileScope prof("resize_horizontal",src.stream);
      if (table.max_taps <= 8)
        {
          if (lds > 64u*1024u)
            MH_HIP(hipFuncSetAttribute(
              reinterpret_cast<const void *>(&resize_horizontal_kernel<Q,C,BLEND,A,8>),
              hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
          hipLaunchKernelGGL((resize_horizontal_kernel<Q,C,BLEND,A,8>),grid,dim3(256),lds,
            src.stream,args,tile_rows);
        }
      else if (table.max_taps <= 16)
        {
          // reductions (2x .. 2.6x Lanczos): the lane's weights in registers across the tile's rows
          if (lds > 64u*1024u)
            MH_HIP(hipFuncSetAttribute(
              reinterpret_cast<const void *>(&resize_horizontal_kernel<Q,C,BLEND,A,16>),
              hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
          hipLaunchKernelGGL((resize_horizontal_kernel<Q,C,BLEND,A,16>),grid,dim3(256),lds,
            src.stream,args,tile_rows);
        }
      else if (table.max_taps <= 32)
        {
          // ... up to 5x (a 4x Lanczos reduction: 25 contributions per output)
          if (lds > 64u*1024u)
            MH_HIP(hipFuncSetAttribute(
              reinterpret_cast<const void *>(&resize_horizontal_kernel<Q,C,BLEND,A,32>),
              hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
          hipLaunchKernelGGL((resize_horizontal_kernel<Q,C,BLEND,A,32>),grid,dim3(256),lds,
            src.stream,args,tile_rows);
        }
      else
        {
          if (lds > 64u*1024u)
            MH_HIP(hipFuncSetAttribute(
              reinterpret_cast<const void *>(&resize_horizontal_kernel<Q,C,BLEND,A,0>),
              hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
          hipLaunchKernelGGL((resize_horizontal_kernel<Q,C,BLEND,A,0>),grid,dim3(256),lds,
            src.stream,args,tile_rows);
        }
    }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

template<typename Q,class A>
static MhStatus dispatch(const View &src,const View &dst,bool vertical,const TapTable &table,
  const Roles &roles)
{
  const bool blend=roles.blend && (roles.alpha == src.channels-1);
  switch (src.channels)
  {
    case 1: return launch_typed<Q,1,false,A>(src,dst,vertical,table,roles);
    case 2:
      if (blend) return launch_typed<Q,2,true,A>(src,dst,vertical,table,roles);
      return launch_typed<Q,2,false,A>(src,dst,vertical,table,roles);
    case 3: return launch_typed<Q,3,false,A>(src,dst,vertical,table,roles);
    case 4:
      if (blend) return launch_typed<Q,4,true,A>(src,dst,vertical,table,roles);
      return launch_typed<Q,4,false,A>(src,dst,vertical,table,roles);
    default: break;
  }
  return fail(MH_UNSUPPORTED,"%d channels",src.channels);
}

MhStatus launch_resize_pass(const View &src,const View &dst,bool vertical,
  const TapTable &table,const Roles &roles,MhPrecision prec)
{
  if ((src.channels != dst.channels) || (src.quantum != dst.quantum))
    return fail(MH_BAD_ARGUMENT,"resize: layout mismatch");
  if (vertical ? ((src.columns != dst.columns) || ((int) dst.rows != table.out_size)) :
                 ((src.rows != dst.rows) || ((int) dst.columns != table.out_size)))
    return fail(MH_BAD_ARGUMENT,"resize: geometry mismatch");
  if (roles.blend && (roles.alpha != src.channels-1))
    return fail(MH_UNSUPPORTED,"alpha channel must be the last channel");
  // Both precision modes resample in fp64.  A two-pass resize hands a
  // Quantum-rounded intermediate to the second pass, and with alpha-weighted
  // channels a +-1 difference in a small intermediate alpha moves the final colour
  // by many levels (measured: 15 levels on uniform-random alpha), so an f32 first
  // pass cannot keep the +-1 contract.  FAST selects the fused-multiply-add fp64
  // policy (Fma64), whose intermediate differs from the CPU's only when a value
  // lies within ~1e-11 of a rounding boundary.
  // (operators.cpp hands FAST to the SECOND filter only: the first one's result is rounded and
  // feeds it.)  Alpha-weighted channels keep the reference's order in the second filter too: where
  // the alpha sum cancels to nearly nothing the quotient — or PerceptibleReciprocal's clamp, a
  // factor of 1.5e7 — turns the last bits of the sums into whole levels, and these kernels have no
  // way back to the taps of a single output (the one-launch forms recompute such rows).
  if ((prec == MH_PRECISION_FAST) && !(roles.blend && (roles.alpha == src.channels-1)))
    {
      if (src.quantum == MH_QUANTUM_U16)
        return dispatch<uint16_t,Fma64>(src,dst,vertical,table,roles);
      return dispatch<float,Fma64>(src,dst,vertical,table,roles);
    }
  if (src.quantum == MH_QUANTUM_U16)
    return dispatch<uint16_t,Exact64>(src,dst,vertical,table,roles);
  return dispatch<float,Exact64>(src,dst,vertical,table,roles);
}

} // namespace mh
