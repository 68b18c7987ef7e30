// Separable-convolution pass on the matrix cores (FAST precision, Q16 RGBA with
// alpha-weighted colour channels — BlurImage's case, MagickCore/morphology.c:2654-2979).
//
// A K-tap 1-D convolution of a tile is a banded (Toeplitz) matrix product
//
//     out[e][n] = sum_k  data[e][k] * T[k][n],      T[k][n] = tap[k-n]  (0 <= k-n < K)
//
// with e = (pixel column or row, channel), n = output position along the filter axis and
// k = input position.  At 79 taps the vector-ALU formulation needs 316 f32 multiply-adds
// per pixel and pass and is issue-bound at ~35 % of the vector peak while
// HBM idles at 17 %; the f16 MFMA pipe has 16x the f32 FMA rate, so the same sums are
// formed there and the pass becomes a stream again.
//
// Precision.  The reference evaluates gamma*sum(k*alpha*p) with gamma=1/sum(k*alpha) in
// fp64 and rounds to Q16; FAST promises +-1 level.  f16 carries 11 significant bits, so
// every operand is split into two f16 terms (22 bits, relative error 2^-22):
//     v  = alpha*p*2^-17 (colour) or alpha/2 (alpha channel)   -> v_hi + v_lo
//     t  = 256*tap                                             -> t_hi + t_lo
// and the product is accumulated in f32 as v_hi*t_hi + v_lo*t_hi + v_hi*t_lo (the dropped
// v_lo*t_lo term is 2^-22 of the result).  f16 x f16 products are exact in f32; the f32
// accumulation is the same kind of error the vector FAST path has.  The scale factors are
// powers of two and cancel exactly in the epilogue.
//
// Shape.  v_mfma_f32_32x32x16_f16: A = data (32 entries e x 16 k), B = Toeplitz taps
// (16 k x 32 n), D (32 e x 32 n) with D row = (reg&3)+8*(reg>>2)+4*(lane>>5), col = lane&31:
// with e = 4*pixel+channel the four channels of a pixel sit in four consecutive registers
// of ONE lane, so the gamma division is lane-local.  The tap operands depend only on
// (lane, k-chunk) and stay in registers for the lifetime of a persistent workgroup.
#include "mh_internal.hpp"
#include "device_common.hpp"
#include <cstdlib>

namespace mh {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

struct ConvMfmaArgs
{
  const uint16_t *src;
  uint16_t *dst;
  int columns,rows;
  int ntaps;
  int shift;                 // K-1-origin: offset of the first input sample
  const float *taps;         // float[K], taps[v] multiplies input o-shift+v
  int tiles_minor,tiles_major,total_tiles;
  int skip;                  // experiment mask (MAGICKHIP_MFMA_SKIP): 1 multiply, 2 staging, 4 loads, 8 stores
};

// LDS image of a tile: two planes (hi, lo) of f16, entry-major with the filter axis
// contiguous: plane[channel][unit 0..31][k 0..KR), KR = 32*(NG-1)+16*NQ input positions.
//   stride S = KR+4 halves: (S/2) mod 64 = 2 mod 8  -> the 32 units of a channel fall into
//   32 different banks for the 8-byte staging writes;
//   channel skew of 8 halves -> the four channels of a pixel read different banks.
template<int NQ,int NG>
struct MfmaGeometry
{
  static constexpr int KR=32*(NG-1)+16*NQ;
  static constexpr int S=KR+8;                 // multiple of 8 halves: 16-byte aligned ds_read_b128
  static constexpr int CH=32*S+8;              // halves per channel
  static constexpr int PLANE=4*CH;             // halves per plane
  static constexpr size_t lds_bytes=(size_t) 2*PLANE*sizeof(_Float16);
};

// v = hi + lo with hi the top 11 significant bits of v (mantissa truncated in the integer
// domain, so the f32 -> f16 conversion of hi is exact whatever its rounding rule) and lo the
// remainder.  Converting v itself and subtracting the result back is NOT safe: on gfx950 the
// packed and the scalar f32 -> f16 conversions the compiler mixes disagree on ties (measured:
// v = 9060.0 between 9056 and 9064 stored one neighbour and subtracted the other, an error
// of a whole f16 ulp in one sample, +-3 Quantum levels after the pass).
static __device__ __forceinline__ void split_f16(float v,_Float16 &hi,_Float16 &lo)
{
  const float top=__builtin_bit_cast(float,__builtin_bit_cast(unsigned,v) & 0xffffe000u);
  hi=(_Float16) top;
  lo=(_Float16) (v-top);
}

// VERTICAL: units are pixel columns, the filter axis runs down the rows (column pass).
// Workgroup tile: 32 units x (32*NG) outputs along the axis; wave w owns pixel group w
// (8 units = 32 entries) and loops over the NG output groups.
template<bool VERTICAL,int NQ,int NG>
__global__ __launch_bounds__(256)
void conv_mfma_kernel(ConvMfmaArgs args)
{
  typedef MfmaGeometry<NQ,NG> G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *plane_hi=reinterpret_cast<_Float16 *>(smem_raw);
  _Float16 *plane_lo=plane_hi+G::PLANE;
  const int tid=(int) threadIdx.x,lane=tid & 63,wave=tid >> 6;
  const int n=lane & 31,half=lane >> 5;
  const int K=args.ntaps;
  const int W=args.columns,H=args.rows;

  // ---- Toeplitz operands: T[q][i] = 256*tap[16q+8*half+i-n]
  half8 t_hi[NQ],t_lo[NQ];
#pragma unroll
  for (int q=0; q < NQ; q++)
#pragma unroll
    for (int i=0; i < 8; i++)
      {
        const int j=16*q+8*half+i-n;
        const float t=((j >= 0) && (j < K)) ? 256.0f*args.taps[j] : 0.0f;
        _Float16 h,l;
        split_f16(t,h,l);
        t_hi[q][i]=h;
        t_lo[q][i]=l;
      }

  // ---- persistent loop over tiles, a contiguous range per XCD (8 XCDs, round-robin ids)
  const int nblocks=(int) gridDim.x,xcd=(int) blockIdx.x & 7,slot=(int) blockIdx.x >> 3;
  const int per_xcd=(args.total_tiles+7)/8,blocks_per_xcd=nblocks >> 3;
  const int range_begin=xcd*per_xcd;
  const int range_end=range_begin+per_xcd < args.total_tiles ? range_begin+per_xcd : args.total_tiles;
  // Raw pixels of a tile, fetched one tile ahead: the global loads of tile i+1 are in flight
  // while tile i is multiplied, otherwise every tile pays the full load latency five times
  // (measured: 0.67 ms per pass without the prefetch — latency-, not bandwidth-bound).
  constexpr int NIT=((G::KR/4)*32+255)/256;
  uint2 raw[NIT][4];
  auto fetch=[&](int tile)
  {
    const int t_major=tile/args.tiles_minor,t_minor=tile-t_major*args.tiles_minor;
    const int unit0=32*t_major,in0=32*NG*t_minor-args.shift;
#pragma unroll
    for (int it=0; it < NIT; it++)
      {
        int u=tid+256*it;
        u=u < (G::KR/4)*32 ? u : (G::KR/4)*32-1;
        int unit,group;
        if (VERTICAL)
          {
            group=u >> 5;                        // lanes of a wave: 32 adjacent columns
            unit=u & 31;
          }
        else
          {
            unit=u/(G::KR/4);                    // lanes of a wave: adjacent x groups of a row
            group=u-unit*(G::KR/4);
          }
#pragma unroll
        for (int i=0; i < 4; i++)
          {
            const int pos=in0+4*group+i;
            int x=VERTICAL ? unit0+unit : pos;
            int y=VERTICAL ? pos : unit0+unit;
            x=x < 0 ? 0 : (x > W-1 ? W-1 : x);   // edge clamp, cache.c:2663-2679
            y=y < 0 ? 0 : (y > H-1 ? H-1 : y);
            if ((args.skip & 4) == 0)
              raw[it][i]=*reinterpret_cast<const uint2 *>(args.src+((size_t) y*W+x)*4);
            else
              raw[it][i]=make_uint2((unsigned) x,(unsigned) y);
          }
      }
  };
  const int first_tile=range_begin+slot;
  if (first_tile < range_end)
    fetch(first_tile);
  for (int tile=first_tile; tile < range_end; tile+=blocks_per_xcd)
    {
      // tiles_minor runs along the filter axis so that consecutive tiles share their halo
      const int t_major=tile/args.tiles_minor,t_minor=tile-t_major*args.tiles_minor;
      const int unit0=32*t_major;                  // first pixel column (V) / row (H)
      const int out0=32*NG*t_minor;                // first output position along the axis
      __syncthreads();                             // previous tile's readers are done
      // ---- stage: 4 consecutive axis positions of one unit per thread and step
#pragma unroll
      for (int it=0; it < NIT; it++)
        {
          const int u=tid+256*it;
          if ((u >= (G::KR/4)*32) || ((args.skip & 2) != 0))
            break;
          int unit,group;
          if (VERTICAL)
            {
              group=u >> 5;
              unit=u & 31;
            }
          else
            {
              unit=u/(G::KR/4);
              group=u-unit*(G::KR/4);
            }
          float v[4][4];
#pragma unroll
          for (int i=0; i < 4; i++)
            {
              const uint2 r=raw[it][i];
              const float alpha=(float) (r.y >> 16)*0.5f;
              const float weight=alpha*(1.0f/65536.0f);
              v[0][i]=(float) (r.x & 0xffffu)*weight;
              v[1][i]=(float) (r.x >> 16)*weight;
              v[2][i]=(float) (r.y & 0xffffu)*weight;
              v[3][i]=alpha;
            }
#pragma unroll
          for (int c=0; c < 4; c++)
            {
              half4 hi,lo;
#pragma unroll
              for (int i=0; i < 4; i++)
                {
                  _Float16 h,l;
                  split_f16(v[c][i],h,l);
                  hi[i]=h;
                  lo[i]=l;
                }
              const int at=c*G::CH+unit*G::S+4*group;
              *reinterpret_cast<half4 *>(plane_hi+at)=hi;
              *reinterpret_cast<half4 *>(plane_lo+at)=lo;
            }
        }
      if (tile+blocks_per_xcd < range_end)
        fetch(tile+blocks_per_xcd);
      __syncthreads();
      // ---- multiply: wave = pixel group (8 units), loop over the output groups
      const int unit_local=8*wave+(n >> 2),channel=n & 3;
      const int entry=channel*G::CH+unit_local*G::S+8*half;
      floatx16 acc[NG];
#pragma unroll
      for (int g=0; g < NG; g++)
        {
#pragma unroll
          for (int r=0; r < 16; r++)
            acc[g][r]=0.0f;
#pragma unroll
          for (int q=0; q < ((args.skip & 1) != 0 ? 0 : NQ); q++)
            {
              const half8 a_hi=*reinterpret_cast<const half8 *>(plane_hi+entry+32*g+16*q);
              const half8 a_lo=*reinterpret_cast<const half8 *>(plane_lo+entry+32*g+16*q);
              acc[g]=__builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi,t_hi[q],acc[g],0,0,0);
              acc[g]=__builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo,t_hi[q],acc[g],0,0,0);
              acc[g]=__builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi,t_lo[q],acc[g],0,0,0);
            }
        }
      if (VERTICAL)
        __syncthreads();                           // the planes become the output tile
      // ---- epilogue: lane holds 4 pixels (reg>>2) x 4 channels (reg&3) of output n
      uint16_t *tile_out=reinterpret_cast<uint16_t *>(smem_raw);      // [32*NG][32][4] (VERTICAL)
#pragma unroll
      for (int g=0; g < NG; g++)
#pragma unroll
        for (int pg=0; pg < 4; pg++)
          {
            // S_c = 2^-9 * sum k*alpha*p, S_a = 128 * sum k*alpha:
            //   gamma*pixel = sum(k*alpha*p)/sum(k*alpha) = 65536 * S_c / S_a
            // v_rcp_f32(0) = inf and 0*inf = NaN convert to 0: PerceptibleReciprocal's clamp
            // for an all-transparent window (as the vector FAST epilogue)
            const float sa=acc[g][4*pg+3];
            const float inv=__builtin_amdgcn_rcpf(sa)*65536.0f;
            uint16_t out[4];
#pragma unroll
            for (int c=0; c < 4; c++)
              {
                const float pixel=c == 3 ? sa*(1.0f/128.0f) : acc[g][4*pg+c]*inv;
                unsigned q=(unsigned) (pixel+0.5f);          // NaN and negatives -> 0
                out[c]=(uint16_t) (q > 65535u ? 65535u : q);
              }
            const int unit_out=8*wave+2*pg+half;             // D row = (reg&3)+8*(reg>>2)+4*half
            const int pos_out=32*g+n;
            if ((args.skip & 8) != 0)
              continue;
            if (VERTICAL)
              store_pixel<uint16_t,4>(tile_out+((size_t) pos_out*32+unit_out)*4,out);
            else
              {
                const int x=out0+pos_out,y=unit0+unit_out;
                if ((x < W) && (y < H))
                  store_pixel<uint16_t,4>(args.dst+((size_t) y*W+x)*4,out);
              }
          }
      if (VERTICAL)
        {
          __syncthreads();
          // coalesced copy-out: 16 bytes (2 pixels) per thread and step, rows of 256 bytes
          for (int u=tid; u < 32*NG*16; u+=256)
            {
              const int row=u >> 4,pair=u & 15;
              const int x=unit0+2*pair,y=out0+row;
              if (y >= H)
                continue;
              const uint16_t *from=tile_out+((size_t) row*32+2*pair)*4;
              uint16_t *to=args.dst+((size_t) y*W+x)*4;
              if (x+1 < W)
                *reinterpret_cast<uint4 *>(to)=*reinterpret_cast<const uint4 *>(from);
              else if (x < W)
                *reinterpret_cast<uint2 *>(to)=*reinterpret_cast<const uint2 *>(from);
            }
        }
    }
}

template<bool VERTICAL,int NQ>
static MhStatus launch_mfma_typed(const View &src,ConvMfmaArgs &args)
{
  constexpr int NG=2;
  typedef MfmaGeometry<NQ,NG> G;
  const int units=VERTICAL ? args.columns : args.rows;
  const int axis=VERTICAL ? args.rows : args.columns;
  args.tiles_major=(units+31)/32;
  args.tiles_minor=(axis+32*NG-1)/(32*NG);
  args.total_tiles=args.tiles_major*args.tiles_minor;
  const size_t lds=G::lds_bytes;
  const int per_cu=lds <= 80u*1024u ? 2 : 1;
  int nblocks=compute_units(src.device)*per_cu;
  nblocks=(nblocks/8)*8;
  if (nblocks < 8)
    nblocks=8;
  MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<VERTICAL,NQ,NG>),
    hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  ProfileScope prof(VERTICAL ? "conv_column" : "conv_row",src.stream);
  hipLaunchKernelGGL((conv_mfma_kernel<VERTICAL,NQ,NG>),dim3((unsigned) nblocks),dim3(256),lds,
    src.stream,args);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// *handled = false: shape outside this kernel's reach, nothing launched
MhStatus launch_conv1d_mfma(const View &src,const View &dst,bool vertical,const float *taps_device,
  int ntaps,int shift,bool *handled)
{
  *handled=false;
  if ((src.quantum != MH_QUANTUM_U16) || (src.channels != 4) || (ntaps < 2))
    return MH_OK;
  const int nq=(ntaps+31+15)/16;                 // 32 outputs + K-1 halo, in 16-sample chunks
  if (nq > 9)
    return MH_OK;
  ConvMfmaArgs args;
  args.src=static_cast<const uint16_t *>(src.pixels);
  args.dst=static_cast<uint16_t *>(dst.pixels);
  args.columns=(int) src.columns;
  args.rows=(int) src.rows;
  args.ntaps=ntaps;
  args.shift=shift;
  args.taps=taps_device;
  args.skip=getenv("MAGICKHIP_MFMA_SKIP") != nullptr ? atoi(getenv("MAGICKHIP_MFMA_SKIP")) : 0;
  *handled=true;
#define MH_NQ(NQV) \
  case NQV: return vertical ? launch_mfma_typed<true,NQV>(src,args) : launch_mfma_typed<false,NQV>(src,args);
  switch (nq)
  {
    MH_NQ(3) MH_NQ(4) MH_NQ(5) MH_NQ(6) MH_NQ(7) MH_NQ(8) MH_NQ(9)
    default: break;
  }
#undef MH_NQ
  return vertical ? launch_mfma_typed<true,3>(src,args) : launch_mfma_typed<false,3>(src,args);
}

} // namespace mh
