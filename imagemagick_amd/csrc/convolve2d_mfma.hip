// Non-separable 2-D ConvolveMorphology on the f16 matrix cores (FAST, Q16; RGBA, four plain
// channels or RGB).
//
// Reference: the reflected-kernel loops of MorphologyPrimitive, MagickCore/morphology.c:2925-2979
// (a w x h weighted sum per channel, alpha-weighted for Blend channels, NaN cells skipped) with
// the epilogue :3195-3198.  SURVEY section 8d asks for C5's `ConvolveMorphology Disk:15` as the
// MAC-bound variant: 16384^2 x 4 channels x 709 active cells.
//
// Formulation: one kernel ROW is a 1-D horizontal convolution, i.e. the banded (Toeplitz) product
// of convolve_mfma.hip / convolve_fused.hip; the h rows accumulate into the same f32 tile:
//     D[e][n] += sum_k  data_v[e][k] * T_v[k][n],    T_v[k][n] = 256 * tap[v][k-n]
// for v = 0..h-1, where entry e = 4*row + channel pairs output row r with source row r+v.
// Operands are hi/lo-split f16 with three products per term (mfma_common.hpp); the scale
// factors and the epilogue are those of the 1-D passes.
//
// MI355X mapping.  A workgroup (8 waves) walks DOWN a strip of 64 output columns, 32 output rows
// a step:
//   * the (32+h-1) x (64+w-1) source window lives in LDS as alpha-premultiplied hi/lo f16 planes
//     [channel][row][column] — 119 KB for 31 x 31; the stride comes from the same bank-conflict
//     search as the fused blur's planes.  The rows are a RING: a step replaces the 32 oldest rows
//     with the 32 new ones, which were fetched into registers (two pixel quads per thread) while
//     the previous step's products ran, so neither the h-1 shared rows nor the tap tables are
//     staged again and no load latency stands in front of the products (round 2 staged a whole
//     window per 64 x 32 tile behind a barrier: 9.8 ms for Disk:15 on 16384^2, 52 % of the
//     matrix pipe);
//   * wave = one quad of output rows (entries e = 4*row+channel: D hands a lane the four
//     channels of one pixel, so the division by the alpha sum is lane-local and the result
//     leaves as one 8-byte store) x the four 16-column tiles of the strip.  The Toeplitz operand
//     of a (kernel row, 32-sample chunk) is the same for every tile: it is read once per wave
//     and used for 4 tiles; the data blocks of tile t chunk c and tile t+2 chunk c-1 are the
//     same 16 x 32 samples: six loads feed eight products;
//   * the Toeplitz operands cannot live in registers (31 rows x 16 VGPRs), and a lane's eight
//     taps tap[v][32c+8kq+i-n] start at an address that depends on n: the table is kept in LDS
//     in four copies shifted by 0..3 taps, which makes every lane's window 8-byte aligned
//     (two ds_read_b64); 40 KB for 31 x 31.
// 24 MFMAs per kernel row and wave against 12 + 8 LDS reads: the matrix pipe is the bound
// (16384^2, 31 x 31: 5.4 ms of v_mfma_f32_16x16x32_f16 at 100 %; measured 9.7 ms against 127 ms
// for the generic 2-D kernel, tools/time_convolve2d.py).
#include "mh_internal.hpp"
#include "device_common.hpp"
#include "mfma_common.hpp"
#include <vector>
#include <cmath>
#include <cstdlib>

namespace mh {

struct Conv2DArgs
{
  const uint16_t *src;
  uint16_t *dst;
  int columns,rows;
  int kw,kh;                  // kernel size
  int shiftx,shifty;          // output (x,y) reads source (x-shiftx+u, y-shifty+v)
  const float *taps;          // float[4][kh][TL]: the four shifted Toeplitz tables, 256*tap, reflected walk
  int stage_rows;             // 32+kh-1: rows of the ring
  int stride,plane;           // LDS row stride and channel-plane size of the staged window (halves)
  int strips,groups;          // 64-column strips, 32-row steps of a whole strip
  int segments,steps_per_segment,items_per_xcd;   // vertical cuts of a strip: work items = strips*segments
};

constexpr int kC2Rows=32;     // output rows per workgroup
constexpr int kC2Cols=64;     // output columns per workgroup

template<int NC>
struct Conv2DGeometry
{
  static constexpr int TL=32*NC+16;              // entries of one (shift, kernel row) of the tap table
  static constexpr int XS=16*3+32*(NC-1)+32;     // staged columns: the last tile's last chunk ends here
  static constexpr int BLOCKS=4+2*(NC-1);        // distinct 16 x 32 data blocks per kernel row
};

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

template<int NC,int MODE>
__global__ __launch_bounds__(512)
void conv2d_mfma_kernel(Conv2DArgs args)
{
  // MFMA_PLAIN3 (RGB, 6-byte pixels) runs as four plain channels whose fourth is zero: only the
  // pixel loads and stores differ
  constexpr int PX=MODE == MFMA_PLAIN3 ? 3 : 4;          // u16 per pixel in memory
  constexpr int SAMPLES=MODE == MFMA_PLAIN3 ? MFMA_PLAIN4 : MODE;
  typedef unsigned __attribute__((aligned(2))) LooseDword;
  typedef Conv2DGeometry<NC> G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int SR=args.stride,CH=args.plane;
  _Float16 *stage_hi=reinterpret_cast<_Float16 *>(smem_raw);
  _Float16 *stage_lo=stage_hi+4*CH;
  _Float16 *taps_hi=stage_lo+4*CH;               // [4 shifts][kh][TL]
  _Float16 *taps_lo=taps_hi+4*args.kh*G::TL;
  const int tid=(int) threadIdx.x,lane=tid & 63;
  const int wave=__builtin_amdgcn_readfirstlane(tid >> 6);
  const int W=args.columns,H=args.rows;
  const int items=args.strips*args.segments;
  const int item=((int) blockIdx.x & 7)*args.items_per_xcd+((int) blockIdx.x >> 3);
  if (item >= items)
    return;
  const int segment=item/args.strips,strip=item-segment*args.strips;
  const int step_begin=segment*args.steps_per_segment;
  const int step_end=step_begin+args.steps_per_segment < args.groups ? step_begin+args.steps_per_segment : args.groups;
  const int x0=kC2Cols*strip;
  const int xin0=x0-args.shiftx;
  const int R=args.stage_rows;

  // ---- tap tables: copy b holds T[v][m] = 256*tap[v][m-16-b], zero outside the kernel row
  // (laid out by the host: args.taps is float[4][kh][TL])
  {
    const int total=4*args.kh*G::TL;
    for (int idx=tid; idx < total; idx+=512)
      {
        _Float16 h,l;
        split_f16(args.taps[idx],h,l);
        taps_hi[idx]=h;
        taps_lo[idx]=l;
      }
  }
  // ---- source rows, edge-clamped (cache.c:2663-2679), as quads of four pixels: item idx = (row,
  // quad) of a block of rows that starts at image row `first`
  constexpr int QUADS=G::XS/4;
  auto load_quad=[&](int first,int idx,uint2 (&raw)[4])
  {
    const int row=idx/QUADS,quad=idx-row*QUADS;
    int y=first+row;
    y=y < 0 ? 0 : (y > H-1 ? H-1 : y);
#pragma unroll
    for (int i=0; i < 4; i++)
      {
        int x=xin0+4*quad+i;
        x=x < 0 ? 0 : (x > W-1 ? W-1 : x);
        const uint16_t *at=args.src+pixel_index(y,W,x)*PX;
        if constexpr (MODE == MFMA_PLAIN3)
          raw[i]=make_uint2(*reinterpret_cast<const LooseDword *>(at),(unsigned) at[2]);
        else
          raw[i]=*reinterpret_cast<const uint2 *>(at);
      }
  };
  // ... converted and written to ring row (ring_first + row) mod R
  auto store_quad=[&](int ring_first,int idx,const uint2 (&raw)[4])
  {
    const int row=idx/QUADS,quad=idx-row*QUADS;
    int ring_row=ring_first+row;
    ring_row=ring_row >= R ? ring_row-R : ring_row;
    f32x2 v[4][2];
    quantum_to_samples<SAMPLES>(raw,v);
#pragma unroll
    for (int c=0; c < 4; c++)
      {
        uint2 hi,lo;
        split_f16_pair(v[c][0],hi.x,lo.x);
        split_f16_pair(v[c][1],hi.y,lo.y);
        const int at=c*CH+ring_row*SR+4*quad;
        *reinterpret_cast<uint2 *>(stage_hi+at)=hi;
        *reinterpret_cast<uint2 *>(stage_lo+at)=lo;
      }
  };
  // the whole window of the first step: every load of a thread's batch is in flight before the
  // first conversion
  {
    constexpr int ITEMS=4;
    const int total=R*QUADS;
    const int first=kC2Rows*step_begin-args.shifty;
    for (int i0=tid; i0 < total; i0+=512*ITEMS)
      {
        uint2 raw[ITEMS][4];
#pragma unroll
        for (int k=0; k < ITEMS; k++)
          {
            const int idx=i0+512*k;
            load_quad(first,idx < total ? idx : total-1,raw[k]);
          }
#pragma unroll
        for (int k=0; k < ITEMS; k++)
          if (i0+512*k < total)
            store_quad(0,i0+512*k,raw[k]);
      }
  }
  __syncthreads();

  // ---- wave = row quad `wave`, tiles 0..3; lane (e, kq): entry e = 4*row+channel
  const int e=lane & 15,kq=lane >> 4;
  const int a_column=(e & 3)*CH+8*kq;           // + ring row * SR
  const int a_row=4*wave+(e >> 2);               // window row of kernel row 0
  // Toeplitz window of lane (n = e, kq): taps 32c+8kq+i-n = T_b[32c + 8kq - 4a + 16 + i], n = 4a+b
  const int t_base=(e & 3)*args.kh*G::TL+8*kq-4*(e >> 2)+16;
  // the 32 new rows of the next step: 32 x QUADS items, two per thread (QUADS <= 32)
  constexpr int NEW_ITEMS=(kC2Rows*QUADS+511)/512;
  static_assert(NEW_ITEMS <= 2,"two pixel quads per thread");
  int origin=0;                                  // ring row of the window's first row
  for (int step=step_begin; step < step_end; step++)
    {
  const int y0=kC2Rows*step;
  uint2 ahead[NEW_ITEMS][4];
  if (step+1 < step_end)
    {
#pragma unroll
      for (int k=0; k < NEW_ITEMS; k++)
        {
          const int idx=tid+512*k;
          load_quad(y0+kC2Rows-args.shifty+R-kC2Rows,idx < kC2Rows*QUADS ? idx : kC2Rows*QUADS-1,ahead[k]);
        }
    }
  floatx4 acc[4];
#pragma unroll
  for (int t=0; t < 4; t++)
    acc[t]=floatx4{0.0f,0.0f,0.0f,0.0f};
  // The operands of kernel row v+1 are read while the 24 products of row v run (two register
  // sets, the loop unrolled by two), so that the LDS latency of a row's 20 reads does not sit in
  // front of its first product.  (Tried and dropped, round 3: sliding the data operand — a row
  // quad's window moves by one row per kernel row, so three quarters of it can come from the lane
  // four up with v_mov_b32_dpp row_shl:4 and only the new row from LDS: 7 KB instead of 16 KB of
  // LDS reads per wave and kernel row, same time, 8.8 against 8.4 ms.)
  struct Operands
  {
    half8 b_hi[NC],b_lo[NC];
    half8 a_hi[G::BLOCKS],a_lo[G::BLOCKS];
  };
  auto fetch_taps=[&](Operands &o,int v)
  {
#pragma unroll
    for (int c=0; c < NC; c++)
      {
        const int at=t_base+v*G::TL+32*c;
        const half4 h0=*reinterpret_cast<const half4 *>(taps_hi+at);
        const half4 h1=*reinterpret_cast<const half4 *>(taps_hi+at+4);
        const half4 l0=*reinterpret_cast<const half4 *>(taps_lo+at);
        const half4 l1=*reinterpret_cast<const half4 *>(taps_lo+at+4);
        o.b_hi[c]=half8{h0[0],h0[1],h0[2],h0[3],h1[0],h1[1],h1[2],h1[3]};
        o.b_lo[c]=half8{l0[0],l0[1],l0[2],l0[3],l1[0],l1[1],l1[2],l1[3]};
      }
  };
  auto read_rows=[&](Operands &o,int v)
  {
    int ring_row=origin+a_row+v;                 // < 2R
    ring_row=ring_row >= R ? ring_row-R : ring_row;
    const int row_at=a_column+ring_row*SR;
#pragma unroll
    for (int q=0; q < G::BLOCKS; q++)
      {
        o.a_hi[q]=*reinterpret_cast<const half8 *>(stage_hi+row_at+16*q);
        o.a_lo[q]=*reinterpret_cast<const half8 *>(stage_lo+row_at+16*q);
      }
  };
  auto fetch=[&](Operands &o,int v)
  {
    fetch_taps(o,v);
    read_rows(o,v);
  };
  auto multiply=[&](const Operands &o)
  {
    // (tile t, chunk c: columns 16t+32c = data block t+2c.)  Three other products stand between
    // two that accumulate into the same tile: a dependent v_mfma waits for its predecessor's passes
#pragma unroll
    for (int c=0; c < NC; c++)
      {
#pragma unroll
        for (int t=0; t < 4; t++)
          acc[t]=__builtin_amdgcn_mfma_f32_16x16x32_f16(o.a_hi[t+2*c],o.b_hi[c],acc[t],0,0,0);
#pragma unroll
        for (int t=0; t < 4; t++)
          acc[t]=__builtin_amdgcn_mfma_f32_16x16x32_f16(o.a_lo[t+2*c],o.b_hi[c],acc[t],0,0,0);
#pragma unroll
        for (int t=0; t < 4; t++)
          acc[t]=__builtin_amdgcn_mfma_f32_16x16x32_f16(o.a_hi[t+2*c],o.b_lo[c],acc[t],0,0,0);
      }
  };
  Operands even,odd;
  fetch(even,0);
  for (int v=0; v < args.kh; v+=2)
    {
      if (v+1 < args.kh)
        fetch(odd,v+1);
      __builtin_amdgcn_sched_barrier(0);
      multiply(even);
      __builtin_amdgcn_sched_barrier(0);
      if (v+1 >= args.kh)
        break;
      if (v+2 < args.kh)
        fetch(even,v+2);
      __builtin_amdgcn_sched_barrier(0);
      multiply(odd);
      __builtin_amdgcn_sched_barrier(0);
    }
  // ---- D: lane (n, kq) holds the four channels of pixel (row 4*wave+kq, column 16t+n)
  const int y=y0+4*wave+kq;
  if (y < H)
    {
#pragma unroll
      for (int t=0; t < 4; t++)
        {
          const int x=x0+16*t+e;
          if (x < W)
            {
              const uint2 result=sums_to_quantum<SAMPLES>(acc[t][0],acc[t][1],acc[t][2],acc[t][3]);
              uint16_t *at=args.dst+pixel_index(y,W,x)*PX;
              if constexpr (MODE == MFMA_PLAIN3)
                {
                  *reinterpret_cast<LooseDword *>(at)=result.x;
                  at[2]=(uint16_t) result.y;
                }
              else
                *reinterpret_cast<uint2 *>(at)=result;
            }
        }
    }
  if (step+1 < step_end)
    {
      __syncthreads();                           // every read of the 32 oldest rows is done
#pragma unroll
      for (int k=0; k < NEW_ITEMS; k++)
        if (tid+512*k < kC2Rows*QUADS)
          store_quad(origin,tid+512*k,ahead[k]);
      origin=origin+kC2Rows >= R ? origin+kC2Rows-R : origin+kC2Rows;
      __syncthreads();
    }
    }
}

template<int NC,int MODE>
static MhStatus launch_conv2d_typed(const View &src,Conv2DArgs &args,size_t lds)
{
  MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2d_mfma_kernel<NC,MODE>),
    hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  ProfileScope prof("conv2d_mfma",src.stream);
  hipLaunchKernelGGL((conv2d_mfma_kernel<NC,MODE>),dim3((unsigned) (8*args.items_per_xcd)),dim3(512),lds,
    src.stream,args);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// w x h Convolve of an RGBA (alpha-weighted colour, alpha last), four-plain-channel or RGB Q16 frame.
// *handled stays false (nothing launched) when the kernel or the frame does not qualify.
MhStatus launch_conv2d_mfma(const View &src,const View &dst,const MhKernelInfo *kernel,bool blend,
  bool *handled)
{
  *handled=false;
  if ((src.quantum != MH_QUANTUM_U16) || (dst.quantum != MH_QUANTUM_U16) ||
      ((src.channels != 4) && ((src.channels != 3) || blend)) ||
      (dst.channels != src.channels) || (src.columns != dst.columns) || (src.rows != dst.rows))
    return MH_OK;
  if ((src.columns >= (1u << 24)) || (src.rows >= (1u << 24)) ||
      ((unsigned long long) src.columns*src.rows >= (1ull << 32)))
    return MH_OK;                                // pixel_index()
  const int kw=(int) kernel->width,kh=(int) kernel->height;
  // below 5 x 5 the generic kernel's 25 taps are cheaper than a staged window
  if ((kw < 2) || (kh < 2) || (kw*kh < 25) || (kw > 49) || (kh > 64) || (kernel->x < 0) || (kernel->y < 0) ||
      (kernel->x >= kw) || (kernel->y >= kh))
    return MH_OK;
  const int nc=(16+kw-1+31)/32;                  // 16 outputs + kw-1 halo, in 32-sample chunks
  if (nc > 2)
    return MH_OK;
  // the reflected walk of morphology.c:2925: cell (v,u) carries values[(kh-1-v)*kw+(kw-1-u)]
  std::vector<float> taps((size_t) kw*kh);
  {
    // three f16 products per term are good to 2^-21 of sum|k|*65535: within a level for kernels
    // that average, not for one with a gain of tens (and those of both signs cancel)
    double magnitude=0.0;
    for (int i=0; i < kw*kh; i++)
      if (!std::isnan(kernel->values[i]))
        magnitude+=std::fabs(kernel->values[i]);
    if (!(magnitude <= 8.0))
      return MH_OK;
  }
  if (blend)
    {
      // An alpha-weighted result is a QUOTIENT of sums: where a small cell is the only one that meets an opaque
      // sample — a sprite on a transparent ground — it is the whole result (sum(k*alpha*p)/sum(k*alpha),
      // morphology.c:2968-2977), and the f16 terms of a cell far below the largest carry too few bits for it (a 9 x 9
      // Gaussian of sigma 1 on such a frame: ten levels off, found by the randomised run once small Gaussians were
      // sent here).  The flat shapes this kernel was built for have one cell value; anything with a cell below
      // 2^-13 of the largest keeps the fp64 / separated routes (as f16_taps_resolved does for the 1-D passes).
      double largest=0.0,smallest=INFINITY;
      for (int i=0; i < kw*kh; i++)
        {
          const double t=std::fabs(kernel->values[i]);
          if (std::isnan(t) || (t == 0.0))
            continue;
          largest=t > largest ? t : largest;
          smallest=t < smallest ? t : smallest;
        }
      if (!(largest > 0.0) || (smallest < std::ldexp(largest,-13)))
        return MH_OK;
    }
  for (int v=0; v < kh; v++)
    for (int u=0; u < kw; u++)
      {
        const double value=kernel->values[(size_t) (kh-1-v)*kw+(size_t) (kw-1-u)];
        if (std::isnan(value))
          taps[(size_t) v*kw+u]=0.0f;            // `if (!IsNaN(*k))`: the cell is skipped
        else
          {
            // alpha-weighted sums with taps of both signs stay on the fp64 kernels (cancellation)
            if (blend && (value < 0.0))
              return MH_OK;
            taps[(size_t) v*kw+u]=(float) value;
          }
      }
  Conv2DArgs args;
  args.src=static_cast<const uint16_t *>(src.pixels);
  args.dst=static_cast<uint16_t *>(dst.pixels);
  args.columns=(int) src.columns;
  args.rows=(int) src.rows;
  args.kw=kw;
  args.kh=kh;
  args.shiftx=kw-1-(int) kernel->x;
  args.shifty=kh-1-(int) kernel->y;
  args.stage_rows=kC2Rows+kh-1;
  const int xs=nc == 1 ? Conv2DGeometry<1>::XS : Conv2DGeometry<2>::XS;
  const int tl=nc == 1 ? Conv2DGeometry<1>::TL : Conv2DGeometry<2>::TL;
  const int layout=fused16_layout(xs,args.stage_rows,false,false);
  args.stride=layout/256;
  args.plane=args.stage_rows*args.stride+layout % 256;
  const size_t lds=((size_t) 2*4*args.plane+(size_t) 2*4*kh*tl)*sizeof(_Float16);
  if (lds > 160u*1024u)
    return MH_OK;
  // copy b of the table: T[v][m] = 256*tap[v][m-16-b] (conv2d_mfma_kernel)
  std::vector<float> shifted((size_t) 4*kh*tl,0.0f);
  for (int b=0; b < 4; b++)
    for (int v=0; v < kh; v++)
      for (int m=0; m < tl; m++)
        {
          const int t=m-16-b;
          if ((t >= 0) && (t < kw))
            shifted[((size_t) b*kh+(size_t) v)*tl+(size_t) m]=256.0f*taps[(size_t) v*kw+t];
        }
  Temp table;
  MH_TRY(upload_table(table,src.device,src.stream,shifted.data(),shifted.size()*sizeof(float)));
  args.taps=table.as<float>();
  args.strips=(args.columns+kC2Cols-1)/kC2Cols;
  args.groups=(args.rows+kC2Rows-1)/kC2Rows;
  {
    // cuts of a strip: the schedule (one workgroup per CU) that finishes first; a cut costs the
    // staging of its first window, about a step's worth
    const int cus=compute_units(src.device);
    int best=1;
    double best_cost=1.0e300;
    for (int cuts=1; cuts <= args.groups; cuts++)
      {
        const int steps=(args.groups+cuts-1)/cuts;
        const int rounds=(args.strips*((args.groups+steps-1)/steps)+cus-1)/cus;
        const double cost=(double) rounds*((double) steps+1.0);
        if (cost < best_cost-1.0e-9)
          {
            best_cost=cost;
            best=cuts;
          }
      }
    if (const char *e=option("MAGICKHIP_CONV2D_CUTS"))          // tests: long walks on small frames
      best=(atoi(e) >= 1) && (atoi(e) <= args.groups) ? atoi(e) : best;
    args.steps_per_segment=(args.groups+best-1)/best;
    args.segments=(args.groups+args.steps_per_segment-1)/args.steps_per_segment;
  }
  args.items_per_xcd=(args.strips*args.segments+7)/8;
  *handled=true;
  if (src.channels == 3)
    return nc == 1 ? launch_conv2d_typed<1,MFMA_PLAIN3>(src,args,lds) : launch_conv2d_typed<2,MFMA_PLAIN3>(src,args,lds);
  if (nc == 1)
    return blend ? launch_conv2d_typed<1,MFMA_BLEND4>(src,args,lds) : launch_conv2d_typed<1,MFMA_PLAIN4>(src,args,lds);
  return blend ? launch_conv2d_typed<2,MFMA_BLEND4>(src,args,lds) : launch_conv2d_typed<2,MFMA_PLAIN4>(src,args,lds);
}

} // namespace mh
