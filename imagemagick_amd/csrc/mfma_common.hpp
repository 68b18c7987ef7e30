// Device helpers shared by the matrix-core convolution kernels (convolve_mfma.hip: one pass
// per launch; convolve_fused.hip: BlurImage's two passes in one launch).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cmath>

namespace mh {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short pknorm2 __attribute__((ext_vector_type(2)));

// What the four entries of a pixel are
//   MFMA_BLEND4  R,G,B weighted by alpha + alpha itself; the epilogue divides by the alpha sum
//   MFMA_PLAIN4  four independent channels (RGBA without alpha weighting)
//   MFMA_PLAIN3  three independent channels of a 6-byte pixel (RGB), the fourth entry is zero
// In the plain modes a sample is a 16-bit integer, so hi (top 11 bits) + lo (the other 5) is exact.
enum MfmaMode { MFMA_BLEND4=0,MFMA_PLAIN4=1,MFMA_PLAIN3=2 };

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

// 4 x 4 byte transpose: x[t] = the 32-bit sample of position t -> p[i] = byte i of positions 0..3
static __device__ __forceinline__ void byte_planes(const unsigned (&x)[4],unsigned (&p)[4])
{
  const unsigned l01=__builtin_amdgcn_perm(x[1],x[0],0x05010400u),h01=__builtin_amdgcn_perm(x[1],x[0],0x07030602u);
  const unsigned l23=__builtin_amdgcn_perm(x[3],x[2],0x05010400u),h23=__builtin_amdgcn_perm(x[3],x[2],0x07030602u);
  p[0]=__builtin_amdgcn_perm(l23,l01,0x05040100u);
  p[1]=__builtin_amdgcn_perm(l23,l01,0x07060302u);
  p[2]=__builtin_amdgcn_perm(h23,h01,0x05040100u);
  p[3]=__builtin_amdgcn_perm(h23,h01,0x07060302u);
}

// y*W+x for rows and columns below 2^24 and fewer than 2^32 pixels (the launchers check):
// one full-rate v_mad_u32_u24 instead of a 64-bit multiply
static __device__ __forceinline__ size_t pixel_index(int y,int W,int x)
{
  return (size_t) (__umul24((unsigned) y,(unsigned) W)+(unsigned) x);
}

// The same split for two non-negative values at once, packed for the LDS planes.
// v_cvt_pkrtz_f16_f32 truncates (for v >= 0 that is the mantissa mask above) and packs both hi
// halves; v_fma_mixlo_f16 / v_fma_mixhi_f16 read an f16 half as an operand and write the
// rounded f16 result into the low / high half of the destination: lo = v - hi, packed, in two
// instructions (the f32 difference is exact, the rounding to f16 is the only one).
static __device__ __forceinline__ void split_f16_pair(f32x2 v,unsigned &hi,unsigned &lo)
{
  hi=__builtin_bit_cast(unsigned,__builtin_amdgcn_cvt_pkrtz(v[0],v[1]));
  unsigned packed;
  asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(packed) : "v"(hi),"v"(v[0]));
  asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(packed) : "v"(hi),"v"(v[1]));
  lo=packed;
}

// Toeplitz tap operand of chunk q for output n (lane & 31) and k-half (lane >> 5):
// T[i] = 256*tap[16q+8*half+i-n], split into hi and lo f16 terms.  tap_lds: float[K] in LDS.
static __device__ __forceinline__ void toeplitz_operand(const float *tap_lds,int K,int q,int half,int n,
  half8 &t_hi,half8 &t_lo)
{
#pragma unroll
  for (int i=0; i < 8; i++)
    {
      const int j=16*q+8*half+i-n;
      const float t=((j >= 0) && (j < K)) ? 256.0f*tap_lds[j] : 0.0f;
      _Float16 h,l;
      split_f16(t,h,l);
      t_hi[i]=h;
      t_lo[i]=l;
    }
}

// Four samples (positions p..p+3) of four channels, raw Quantum pixels r[0..3] -> the
// matrix-core sample values v[channel][pair], pairs of neighbouring positions:
//   MFMA_BLEND4: alpha*p*2^-17 for the colour channels, alpha/2 for alpha
//   plain:       p/2 (65535 stays inside the f16 range)
template<int MODE>
static __device__ __forceinline__ void quantum_to_samples(const uint2 (&r)[4],f32x2 (&v)[4][2])
{
#pragma unroll
  for (int j=0; j < 2; j++)
    {
      const unsigned r0x=r[2*j].x,r0y=r[2*j].y,r1x=r[2*j+1].x,r1y=r[2*j+1].y;
      const f32x2 c0={(float) (r0x & 0xffffu),(float) (r1x & 0xffffu)};
      const f32x2 c1={(float) (r0x >> 16),(float) (r1x >> 16)};
      const f32x2 c2={(float) (r0y & 0xffffu),(float) (r1y & 0xffffu)};
      const f32x2 c3={(float) (r0y >> 16),(float) (r1y >> 16)};
      if (MODE == MFMA_BLEND4)
        {
          const f32x2 weight=c3*(0.5f/65536.0f);
          v[0][j]=c0*weight;
          v[1][j]=c1*weight;
          v[2][j]=c2*weight;
          v[3][j]=c3*0.5f;
        }
      else
        {
          v[0][j]=c0*0.5f;
          v[1][j]=c1*0.5f;
          v[2][j]=c2*0.5f;
          v[3][j]=c3*0.5f;
        }
    }
}

// The epilogue of one pixel: the four f32 sums of a 1-D pass -> four Quantum levels, packed.
// With the taps' factor 256 (`unit` = 2/(scale*65535) for any other):
//   S_c = 2^-9 * sum k*alpha*p, S_a = 128 * sum k*alpha:
//     gamma*pixel = sum(k*alpha*p)/sum(k*alpha) = 65536 * S_c / S_a
//   v_rcp_f32(0) = inf and 0*inf = NaN convert to 0: PerceptibleReciprocal's clamp for an
//   all-transparent window.  Plain modes: S_c = 128 * sum k*p.
// v_cvt_pknorm_u16_f32 rounds 65535*x to the nearest level, clamps to [0,65535], maps NaN to 0
// and packs two results: the whole quantisation in one instruction.
template<int MODE>
static __device__ __forceinline__ uint2 sums_to_quantum(float s0,float s1,float s2,float sa,
  float unit=1.0f/(128.0f*65535.0f))
{
  const float inv=MODE == MFMA_BLEND4 ? __builtin_amdgcn_rcpf(sa)*(65536.0f/65535.0f) : unit;
  const f32x2 scale01={inv,inv};
  const f32x2 scale23={inv,MODE == MFMA_BLEND4 ? unit : inv};
  const f32x2 p01=f32x2{s0,s1}*scale01;
  const f32x2 p23=f32x2{s2,sa}*scale23;
  const pknorm2 lo2=__builtin_amdgcn_cvt_pknorm_u16(p01[0],p01[1]);
  const pknorm2 hi2=__builtin_amdgcn_cvt_pknorm_u16(p23[0],p23[1]);
  return make_uint2(__builtin_bit_cast(unsigned,lo2),__builtin_bit_cast(unsigned,hi2));
}

// BlurImage's row pass hands a Quantum-rounded alpha to the column pass, where it is a WEIGHT:
// one level more or less in an alpha of a few levels changes that sample's weight by tens of
// per cent, and the second pass turns that into many levels of colour.  The f32 alpha sum is
// within alpha_sum_error levels of the exact sum below kSmallAlpha levels (22-bit operands,
// f32 accumulation of <= 4*3*NQ partial sums); when it lies closer than that to a rounding tie
// the level is recomputed as the reference does (fp64, its operation order,
// morphology.c:2941-2951 with the alpha channel's own Update trait) — the intermediate alpha is
// then bit-identical to the reference's wherever it can matter.
constexpr float kSmallAlpha=8192.0f;
constexpr float kAlphaSumError=0.0625f;

// (levels = the alpha sum in levels: S_a * 2/scale)
static __device__ __forceinline__ bool alpha_sum_is_ambiguous(float levels)
{
  const float fraction=levels-__builtin_floorf(levels);
  return (levels < kSmallAlpha) && (__builtin_fabsf(fraction-0.5f) < kAlphaSumError);
}

// sum over v of taps64[v]*alpha(position o-shift+v), edge-clamped, in the reference's order
// (kernel walked backwards = taps64 forwards), one rounding per multiply and per add
// (-ffp-contract=off), then ClampToQuantum.  `stride` = distance between consecutive
// positions in pixels (1 along a row, W down a column).
static __device__ __forceinline__ unsigned exact_alpha_level(const uint16_t *src,size_t line0,int stride,
  int extent,int first,const double *taps64,int K)
{
  double sum=0.0;
  for (int v=0; v < K; v++)
    {
      int at=first+v;
      at=at < 0 ? 0 : (at > extent-1 ? extent-1 : at);
      const double alpha=(double) src[(line0+(size_t) at*(size_t) stride)*4+3];
      sum=sum+taps64[v]*alpha;
    }
  if (!(sum > 0.0))
    return 0u;
  if (sum >= 65535.0)
    return 65535u;
  return (unsigned) (sum+0.5);
}

// UnsharpMaskImage's epilogue for one Quantum-rounded blurred sample b of the sample p
// (effect.c:4364-4369): p if |2(p-b)| < QuantumRange*threshold, else p+gain*(p-b), clamped and
// rounded.  2(p-b) is an integer, so comparing it with the ceiling of the threshold is exact.
static __device__ __forceinline__ unsigned unsharp_sample(unsigned p,unsigned b,float gain,int threshold)
{
  const int d=(int) p-(int) b;
  const int twice=d < 0 ? -2*d : 2*d;
  // ClampToQuantum: round half up (gain*d has exact halves for gains like 2.5, so the
  // round-to-even of v_cvt_pknorm would differ from the reference on every such tie)
  const float sharpened=(float) p+gain*(float) d+0.5f;
  const unsigned level=(unsigned) (sharpened < 0.0f ? 0.0f : sharpened);
  return twice < threshold ? p : (level > 65535u ? 65535u : level);
}

static __device__ __forceinline__ uint2 unsharp_pixel(uint2 p,uint2 b,float gain,int threshold)
{
  return make_uint2(
    unsharp_sample(p.x & 0xffffu,b.x & 0xffffu,gain,threshold) |
      (unsharp_sample(p.x >> 16,b.x >> 16,gain,threshold) << 16),
    unsharp_sample(p.y & 0xffffu,b.y & 0xffffu,gain,threshold) |
      (unsharp_sample(p.y >> 16,b.y >> 16,gain,threshold) << 16));
}

// Worst number of operand lines of one ds_read_b128 lane group (convolve_fused.hip, fused_reads_conflict_free)
// that share a 16-byte slot of the 256-byte bank row, for the 16x16x32 operand map: entry =
// lane&15, k quarter = lane>>4.  ring: the k quarters 2,3 sit in the next ring group.
static constexpr int fused16_read_degree(int S,int PAD,int units,bool channel_major,bool ring)
{
  const int CH=units*S+PAD;
  int worst=1;
  for (int g=0; g < 4; g++)
    {
      int count[16]={0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
      for (int i=0; i < 16; i++)
        {
          const int base=(g & 1) == 0 ? (i < 4 ? i : (i < 8 ? i+8 : i+12)) : (i < 8 ? i+4 : (i < 12 ? i+8 : i+16));
          const int lane=base+32*(g >> 1);
          const int e=lane & 15,kq=lane >> 4;
          const int channel=channel_major ? e >> 2 : e & 3;
          const int unit=channel_major ? e & 3 : e >> 2;
          const int offset=ring ? 16*(kq >> 1)+8*(kq & 1) : 8*kq;
          const int bytes=(channel*CH+unit*S+offset)*2;
          const int slot=(bytes % 256)/16;
          count[slot]++;
          worst=count[slot] > worst ? count[slot] : worst;
        }
    }
  return worst;
}

// line stride (>= extent, multiple of 8 halves) and channel padding with the fewest read
// conflicts; encoded S*256+PAD
static constexpr int fused16_layout(int extent,int units,bool channel_major,bool ring)
{
  int best=extent*256+8,best_degree=99;
  for (int S=extent; S <= extent+16; S+=8)
    for (int PAD=8; PAD <= 64; PAD+=8)
      {
        const int degree=fused16_read_degree(S,PAD,units,channel_major,ring);
        if (degree < best_degree)
          {
            best_degree=degree;
            best=S*256+PAD;
          }
      }
  return best;
}


// LDS geometry of the fused BlurImage walk on 16x16 tiles (convolve_fused.hip,
// convolve_fused_exact.hip): a strip of 64 columns, groups of 16 rows, a ring of NR groups.
template<int NC>
struct Fused16Geometry
{
  static constexpr int COLS=64;                // strip width
  static constexpr int GROUP=16;               // rows per iteration: one ring group, one output block
  static constexpr int NG=2*NC;                // ring groups a 16-output tile reads: 32*NC rows
  static constexpr int NR=NG+1;                // ring groups held
  static constexpr int RC=GROUP*NR;            // ring rows
  static constexpr int XS=32*NC+48;            // staged columns: 64 outputs + band
  static constexpr bool ROW_CHANNEL_MAJOR=false;  // row pass entries e = 4*row + channel
  static constexpr int SR=fused16_layout(XS,GROUP,ROW_CHANNEL_MAJOR,false)/256,PADR=fused16_layout(XS,GROUP,ROW_CHANNEL_MAJOR,false) % 256;
  // Ring plane of a channel: [8-row octet][column][8 rows].  A column-pass lane's 8 operand rows
  // are one 16-byte unit; the 16 lanes of a ds_read_b128 group (4 columns x 4 channels) land in 16
  // different slots of a bank row when the channel stride is 4 units mod 16 (PADC = 32 halves);
  // the row pass's 8-byte stores (16 consecutive columns per lane group) are 2-way conflicted.
  // (Round 2b kept a column's rows in one line of SC halves: its stores were 4-way conflicted —
  // SQ_LDS_BANK_CONFLICT 63 % of the LDS cycles, 385 of them per iteration, right in front of
  // barrier Y.)
  static constexpr int OB=COLS*8;              // halves per octet block
  static constexpr int SC=8,PADC=32;
  static constexpr int CHR=GROUP*SR+PADR;
  static constexpr int CHC=(RC/8)*OB+PADC;
  static constexpr int STAGE_PLANE=4*CHR,RING_PLANE=4*CHC;
  static constexpr size_t planes_bytes=(size_t) 2*(STAGE_PLANE+RING_PLANE)*sizeof(_Float16);
  // the column pass's 16 x 64 result pixels on their way to row-contiguous stores; 65 pixels per
  // row: the 16 rows a quarter-wave writes fall into 16 different bank pairs
  static constexpr int OUT_STRIDE=COLS+1;
  static constexpr size_t lds_bytes=planes_bytes+(size_t) GROUP*OUT_STRIDE*sizeof(uint2);
  static_assert(lds_bytes <= 163840,"more than the 160 KiB of a CU");
  static constexpr int GROUPS_PER_ROW=XS/4;
  static constexpr int FETCH_GROUPS=GROUP*GROUPS_PER_ROW;
  static_assert(FETCH_GROUPS <= 1024,"one staging round");
};


} // namespace mh
