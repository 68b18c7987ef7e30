// Shared by the two kernels that form BlurImage's sums as EXACT INTEGERS on the i8 matrix cores:
// convolve_fused_exact.hip (both passes exact, or exact row pass + f16 column pass) and
// convolve_fused_hybrid.hip (FAST: f16 colour sums + exact alpha sums).  See the header of
// convolve_fused_exact.hip for the arithmetic and its certificate.
#pragma once

#include "mh_internal.hpp"
#include "device_common.hpp"
#include "mfma_common.hpp"
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

namespace mh {

typedef int intx4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct BlurExactArgs
{
  const uint16_t *src;
  uint16_t *dst;
  int columns,rows;
  int ntaps;
  int shift;                 // K-1-origin: offset of the first input sample (both axes)
  const double *taps64;      // double[K], taps64[v] multiplies input o-shift+v (the recomputation)
  const float *taps;         // the same as floats (the hybrid kernel's f16 products)
  const signed char *digits; // [kExactDigits][kExactDigitPitch]: balanced digits of rint(k*2^F)
  double offset;             // 128 * sum over the kept products of sum_v d_j[v] * 2^(8(i+j-3))
  double alpha_scale;        // level of a plain / alpha sum = M * alpha_scale  (2^(8-F))
  double colour_window;      // colour level ambiguous within colour_window / M_alpha of a tie
  double alpha_half_window;  // plain / alpha level ambiguous within alpha_window of a tie: 0.5 - alpha_window
  double alpha_floor;        // M_alpha below this: the bound says nothing, recompute
  int strips;                // ceil(columns/64)
  int segments;              // vertical cuts of a strip
  int blocks;                // ceil(rows/16) output blocks per strip
  int blocks_per_segment;
  int items_per_xcd;         // ceil(strips*segments/8)
  float gain;                // UnsharpMaskImage's epilogue
  int threshold;             // ceil(QuantumRange*threshold), see unsharp_sample
  // the f16 products' taps are tap_scale*tap, a power of two that puts the largest tap just under 2^15
  // (f16_tap_scale); two_over_scale = 2/tap_scale undoes it in the epilogues, quantum_unit =
  // two_over_scale/65535 in sums_to_quantum
  float tap_scale,two_over_scale,quantum_unit;
  unsigned long long *recomputed;   // optional device counter of recomputed samples (diagnostics)
  // optional device word.  A frame whose alpha is a few levels everywhere widens the certificate's
  // window (it grows with 1/alpha) until several samples of every group go down the
  // reference-order recomputation, 79 dependent taps on one lane each: 22 ms per 8192^2 frame
  // instead of 0.9.  A wave that has recomputed more than a sample per group on average raises
  // the word, every workgroup leaves at its next group, and the fp64 passes queued behind the
  // kernel (Conv1DParams::only_if) compute the frame: bit-identical too, 2.2 ms.
  unsigned *give_up;
  unsigned give_up_token;    // "raised" = the word holds this value (the caller's own: nothing clears the word first)
  unsigned long long *trace;        // diagnostic build (-DMH_EXACT_TRACE) only
};

// Diagnostic build only (-DMH_EXACT_TRACE, tools/trace_exact_blur.py): waves 0, 4, 8 and 12 of the
// first four workgroups stamp the shader clock at the phase boundaries of 48 steady-state
// iterations: trace[block][wave>>2][iteration][mark].
#ifdef MH_EXACT_TRACE
#define MH_XTRACE_MARK(id) \
  do { \
    if (traced && (g >= 64) && (g < 112)) \
      { \
        const unsigned long long now=__builtin_readcyclecounter(); \
        if (lane == 0) \
          args.trace[((((int) blockIdx.x*4+(wave >> 2))*48)+(g-64))*12+(id)]=now; \
      } \
  } while (0)
#else
#define MH_XTRACE_MARK(id) do { } while (0)
#endif

// The kept digit products of one chunk of the band: a[i] = byte plane i of the samples, t[j] =
// digit j of the Toeplitz taps, class i+j-3.  First chunk: 64 slots (v_mfma_i32_16x16x64_i8,
// 16 bytes per lane); second chunk (kernels of more than 49 taps): 32 slots
// (v_mfma_i32_16x16x32_i8, 8 bytes per lane: 64+32 slots hold the 96-slot band of 81 taps, at
// the same instruction time and half the operand registers).  The order keeps three other
// instructions between two that write the same tile (a dependent v_mfma waits for its
// predecessor's passes), also across the chunk boundary.
static __device__ __forceinline__ intx4 digit_product(intx4 a,intx4 t,intx4 acc)
{
  return __builtin_amdgcn_mfma_i32_16x16x64_i8(a,t,acc,0,0,0);
}
static __device__ __forceinline__ intx4 digit_product(long a,long t,intx4 acc)
{
  return __builtin_amdgcn_mfma_i32_16x16x32_i8(a,t,acc,0,0,0);
}
// Second chunk of the band: 32 slots of the legacy-K instruction (the bare instruction issues at
// the rate of the 64-slot one, tools/ubench/mfma_i8_shapes.hip, and the operands take half the
// registers: no spills at four waves per SIMD), or — -DMH_EXACT_K64 — 64 slots, half of them
// zero taps.
#ifndef MH_EXACT_K64
typedef long SecondOperand;
constexpr int kSecondBytes=8;
#else
typedef intx4 SecondOperand;
constexpr int kSecondBytes=16;
#endif

template<bool PLAIN,bool SECOND,typename Operand>
static __device__ __forceinline__ void exact_products(const Operand (&a)[4],const Operand (&t)[5],intx4 (&acc)[5])
{
  if constexpr (PLAIN)
    {
      constexpr int order[9][2]={{3,0},{3,1},{3,2},{3,3},{3,4},{2,1},{2,2},{2,3},{2,4}};
#pragma unroll
      for (int k=0; k < 9; k++)
        acc[order[k][0]+order[k][1]-3]=digit_product(a[order[k][0]],t[order[k][1]],acc[order[k][0]+order[k][1]-3]);
    }
  else if constexpr (!SECOND)
    {
      constexpr int order[14][2]={{3,0},{3,1},{3,2},{3,3},{2,1},{2,2},{2,3},{3,4},{1,2},{1,3},{1,4},{2,4},{0,3},{0,4}};
#pragma unroll
      for (int k=0; k < 14; k++)
        acc[order[k][0]+order[k][1]-3]=digit_product(a[order[k][0]],t[order[k][1]],acc[order[k][0]+order[k][1]-3]);
    }
  else
    {
      constexpr int order[14][2]={{3,2},{3,3},{3,0},{3,1},{2,3},{2,1},{2,2},{1,4},{1,2},{1,3},{2,4},{0,3},{0,4},{3,4}};
#pragma unroll
      for (int k=0; k < 14; k++)
        acc[order[k][0]+order[k][1]-3]=digit_product(a[order[k][0]],t[order[k][1]],acc[order[k][0]+order[k][1]-3]);
    }
}

// Wait states between a chain's last v_mfma and the first VALU read of a tile.  hipcc (ROCm 7.2)
// pads them per basic block; a branch target that begins with such a read got `s_nop 0` (seen with
// store_row's branch between the row chain and its sums: during the first lap of the ring, where
// the branch is taken, channel 0 of the sums came from a tile still in the pipe).  The asm "uses"
// every tile, so nothing that reads one can be scheduled above it.
static __device__ __forceinline__ void settle_tiles(intx4 (&acc)[5])
{
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]),"+v"(acc[1]),"+v"(acc[2]),"+v"(acc[3]),"+v"(acc[4]));
}

// The five class tiles of one lane's pixel (register = channel) -> the exact sums
//   M = sum_c acc_c * 2^(8c) + offset  (integers below 2^53; offset = the constant
//   128 * sum(digit) of the signed-byte samples), in units of 2^(24-F).
// Partial sums in i32: a product's sum over K <= 81 taps is below 81*128*128 = 1.33e6, so
// class 4 + 256 * class 5 (4 and 3 products) stays below 1.03e9 < 2^31.
static __device__ __forceinline__ void exact_sums(intx4 (&acc)[5],double offset,double (&M)[4])
{
  settle_tiles(acc);
#pragma unroll
  for (int ch=0; ch < 4; ch++)
    {
      const int mid=acc[1][ch]+(acc[2][ch] << 8);
      const int top=acc[3][ch]+(acc[4][ch] << 8);
      M[ch]=__builtin_fma((double) top,16777216.0,__builtin_fma((double) mid,256.0,(double) acc[0][ch]+offset));
    }
}

// ... -> the four Quantum levels, and whether any of them is one the error bound cannot decide.
// BLEND: level_c = round(65536*M_c/M_a), level_a = round(M_a*alpha_scale); plain: the latter
// for every channel.  Branch-free: M_a = 0 (every alpha of the window is zero; the host checked
// that the smallest tap times one alpha level is far above the error bound) gives 0*inf = NaN,
// which converts to level 0 and compares as "not doubtful" — PerceptibleReciprocal's clamp times
// pixel = 0.  ClampToQuantum (quantum.h:86-97) = v_cvt_u32_f64 of value+0.5: it truncates, maps
// negatives and NaN to 0, and the values cannot exceed 65535.5 by more than the error bound.
template<bool BLEND>
static __device__ __forceinline__ bool exact_levels(const double (&M)[4],const BlurExactArgs &args,
  unsigned (&q)[4])
{
  bool doubtful=false;
  // doubtful: the fraction of value+0.5 lies within the window of 0 or 1
  auto level_of=[&](double shifted,double half_window,int ch)
  {
    q[ch]=(unsigned) shifted;
    doubtful=doubtful || (__builtin_fabs(__builtin_amdgcn_fract(shifted)-0.5) > half_window);
  };
  if constexpr (BLEND)
    {
      const double Ma=M[3];
      double r=__builtin_amdgcn_rcp(Ma);
      double e=__builtin_fma(-Ma,r,1.0);
      r=__builtin_fma(r,e,r);
      e=__builtin_fma(-Ma,r,1.0);
      r=__builtin_fma(r,e,r);
      // 0.5 - window, window = colour_window/M_a + 4e-9
      const double half_window=__builtin_fma(-args.colour_window,r,0.5-4.0e-9);
      const double scale=65536.0*r;
#pragma unroll
      for (int ch=0; ch < 3; ch++)
        level_of(__builtin_fma(M[ch],scale,0.5),half_window,ch);
      level_of(__builtin_fma(Ma,args.alpha_scale,0.5),args.alpha_half_window,3);
      // an alpha sum the bound says nothing about (tiny or, by the dropped classes, negative)
      doubtful=doubtful || !((Ma >= args.alpha_floor) || (Ma == 0.0));
    }
  else
    {
#pragma unroll
      for (int ch=0; ch < 4; ch++)
        level_of(__builtin_fma(M[ch],args.alpha_scale,0.5),args.alpha_half_window,ch);
    }
  return doubtful;
}

// ---------------------------------------------------------------------------------------------
// The pixels the integer sums cannot decide (a few per million).  Fifteen other waves wait at the
// next barrier for the wave that handles one, so the WHOLE wave handles it: lane v fetches the
// window's sample v (and v+64), and
//   step 2  the pixel's sums once more, in fp64 with fused multiply-adds over the EXACT taps,
//           reduced over the wave.  That value is within 2e-9 level of the real one, the
//           reference's own result (79 separately rounded operations) within another 2e-9: unless
//           it lies within 1e-8 of a rounding tie, its level is the reference's.
//   step 3  (some tens of pixels per 8192^2 frame) the reference's loop itself: every lane forms
//           its tap's terms with the reference's operations (alpha = QuantumScale*a, alpha*k,
//           (alpha*k)*p: each rounded once, independent of the order), and the sums run over the
//           lanes in the reference's order (morphology.c:2743-2776, :2941-2977).
// No dependent memory accesses, no calls: about two microseconds per pixel.
constexpr double kSecondWindow=1.0e-8;

static __device__ __forceinline__ double wave_total(double value)
{
#pragma unroll
  for (int step=1; step < 64; step<<=1)
    value+=__shfl_xor(value,step,64);
  return value;
}

static __device__ __forceinline__ double lane_value(double value,int source)
{
  const unsigned long long bits=__builtin_bit_cast(unsigned long long,value);
  const unsigned lo=(unsigned) __builtin_amdgcn_readlane((int) (unsigned) bits,source);
  const unsigned hi=(unsigned) __builtin_amdgcn_readlane((int) (unsigned) (bits >> 32),source);
  return __builtin_bit_cast(double,((unsigned long long) hi << 32) | (unsigned long long) lo);
}

// mine: this lane's pixel needs it.  fetch(source_lane, v, levels): the four Quantum levels of
// sample v of the window of source_lane's pixel (CHANNELS of them meaningful).  q: this lane's
// levels, replaced when `mine`.  Returns the number of pixels handled (wave-uniform).
template<bool BLEND,int CHANNELS,class Fetch>
static __device__ __forceinline__ unsigned settle_doubtful_pixels(bool mine,int lane,const double *taps,int K,
  const Fetch &fetch,unsigned (&q)[4])
{
  unsigned long long todo=__ballot(mine);
  unsigned handled=0u;
  while (todo != 0ull)
    {
      const int source=(int) __builtin_ctzll(todo);
      todo&=todo-1ull;
      handled++;
      // this lane's one or two samples of the window
      unsigned level[2][4];
      double tap[2];
#pragma unroll
      for (int half=0; half < 2; half++)
        {
          const int v=lane+64*half;
          const bool inside=v < K;
          tap[half]=inside ? taps[v] : 0.0;
          fetch(source,inside ? v : 0,level[half]);
        }
      unsigned result[4];
      bool certain=true;
      {
        double sum[4];
#pragma unroll
        for (int c=0; c < 4; c++)
          {
            double part=0.0;
#pragma unroll
            for (int half=0; half < 2; half++)
              {
                // alpha*p is an exact integer below 2^32
                const unsigned sample=(BLEND && (c != 3)) ? level[half][c]*level[half][3] : level[half][c];
                part=__builtin_fma(tap[half],(double) sample,part);
              }
            sum[c]=c < CHANNELS ? wave_total(part) : 0.0;
          }
        double scale=1.0;
        if constexpr (BLEND)
          scale=sum[3] > 0.0 ? 1.0/sum[3] : 0.0;   // all-transparent window: colour 0
#pragma unroll
        for (int c=0; c < 4; c++)
          {
            const double value=(BLEND && (c != 3)) ? sum[c]*scale : sum[c];
            const double shifted=value+0.5;
            const double whole=__builtin_floor(shifted);
            const double fraction=shifted-whole;
            certain=certain && (fraction > kSecondWindow) && (fraction < 1.0-kSecondWindow);
            result[c]=whole >= 65535.0 ? 65535u : (whole > 0.0 ? (unsigned) whole : 0u);
          }
        // PerceptibleReciprocal's branch (gamma = QuantumScale*S_a below 1e-12) is the reference's
        if constexpr (BLEND)
          certain=certain && ((sum[3] == 0.0) || (sum[3] > 1.0e-6));
      }
      if (!certain)                              // wave-uniform: every lane holds the same sums
        {
          double term[2][4],weight[2];
#pragma unroll
          for (int half=0; half < 2; half++)
            {
              if constexpr (BLEND)
                {
                  const double alpha=kQS*(double) level[half][3];
                  weight[half]=alpha*tap[half];
#pragma unroll
                  for (int c=0; c < 3; c++)
                    term[half][c]=weight[half]*(double) level[half][c];
                  term[half][3]=tap[half]*(double) level[half][3];
                }
              else
                {
                  weight[half]=0.0;
#pragma unroll
                  for (int c=0; c < 4; c++)
                    term[half][c]=tap[half]*(double) level[half][c];
                }
            }
          double sum[4]={0.0,0.0,0.0,0.0},gamma=0.0;
          for (int v=0; v < K; v++)
            {
              const int from=v & 63;
#pragma unroll
              for (int c=0; c < CHANNELS; c++)
                sum[c]+=lane_value(v < 64 ? term[0][c] : term[1][c],from);
              if constexpr (BLEND)
                gamma+=lane_value(v < 64 ? weight[0] : weight[1],from);
            }
          if constexpr (BLEND)
            {
              const double g=perceptible_reciprocal(gamma);
#pragma unroll
              for (int c=0; c < 3; c++)
                sum[c]=g*sum[c];
            }
#pragma unroll
          for (int c=0; c < 4; c++)
            result[c]=(unsigned) QuantumOps<uint16_t>::clamp(sum[c]);
        }
      if (lane == source)
        {
#pragma unroll
          for (int c=0; c < 4; c++)
            q[c]=result[c];
        }
    }
  return handled;
}

// ---------------------------------------------------------------------------------------------
// Host side: the taps as fixed-point digits and the error bound that goes with them.

struct ExactTapPlan
{
  bool ok=false;
  int fraction_bits=0;
  std::vector<signed char> digits;     // [kExactDigits][kExactDigitPitch]
  double offset_blend=0.0,offset_plain=0.0;
  double alpha_scale=0.0;
  double colour_window=0.0,alpha_window_blend=0.0,alpha_window_plain=0.0,alpha_floor=0.0;
};

// taps[v] > 0, K <= kExactDigitPitch.  See the header of this file for the derivation.
static inline ExactTapPlan plan_exact_taps(const double *taps,int K)
{
  ExactTapPlan plan;
  if ((K < 2) || (K > kExactDigitPitch))
    return plan;
  double largest=0.0,smallest=INFINITY;
  for (int v=0; v < K; v++)
    {
      if (!(taps[v] > 0.0) || !std::isfinite(taps[v]))
        return plan;
      largest=taps[v] > largest ? taps[v] : largest;
      smallest=taps[v] < smallest ? taps[v] : smallest;
    }
  // five balanced digits hold |q| <= 127*(256^5-1)/255 = 5.476e11
  int exponent=0;
  (void) std::frexp(5.4e11/largest,&exponent);   // 5.4e11/largest = m * 2^exponent, 0.5 <= m < 1
  const int F=exponent-1;
  if ((F < 24) || (F > 62))
    return plan;
  plan.fraction_bits=F;
  plan.digits.assign((size_t) kExactDigits*kExactDigitPitch,0);
  double quantisation=0.0;                     // sum |k - q 2^-F|
  double magnitude[kExactDigits]={0,0,0,0,0};  // sum_v |d_j[v]|
  double signed_sum[kExactDigits]={0,0,0,0,0}; // sum_v d_j[v]
  for (int v=0; v < K; v++)
    {
      const double scaled=std::ldexp(taps[v],F);             // exact
      const double nearest=std::nearbyint(scaled);
      quantisation+=std::fabs(scaled-nearest);               // exact difference, in units of 2^-F
      long long rest=(long long) nearest;
      for (int j=0; j < kExactDigits; j++)
        {
          long long d=((rest+128) & 255)-128;
          plan.digits[(size_t) j*kExactDigitPitch+(size_t) v]=(signed char) d;
          magnitude[j]+=(double) (d < 0 ? -d : d);
          signed_sum[j]+=(double) d;
          rest=(rest-d) >> 8;
        }
      if (rest != 0)
        return plan;
    }
  const double unit=std::ldexp(1.0,-F);                      // 2^-F
  // kept products: blend i = 0..3, plain i = 2..3; j = 0..4; i+j >= 3; weight 2^(8(i+j-3))
  auto offset_of=[&](int i0)
  {
    double total=0.0;
    for (int i=i0; i < 4; i++)
      for (int j=0; j < kExactDigits; j++)
        if (i+j >= 3)
          total+=128.0*signed_sum[j]*std::ldexp(1.0,8*(i+j-3));
    return total;
  };
  plan.offset_blend=offset_of(0);
  plan.offset_plain=offset_of(2);
  // dropped products (i+j <= 2): |sum_v b_i d_j| <= 255 * sum_v |d_j|, weight 2^(8(i+j)) * 2^-F
  auto dropped_of=[&](int i0)
  {
    double total=0.0;
    for (int i=i0; i < 4; i++)
      for (int j=0; j < kExactDigits; j++)
        if (i+j <= 2)
          total+=255.0*magnitude[j]*std::ldexp(1.0,8*(i+j));
    return total*unit;
  };
  // error bounds of the sums, in sample units (alpha*p, or level*2^16)
  const double e_colour=quantisation*unit*65535.0*65535.0+dropped_of(0);
  const double e_shifted=quantisation*unit*65535.0*65536.0+dropped_of(2);
  // M is in units of 2^(24-F) sample units
  const double m_unit=std::ldexp(1.0,24-F);
  plan.alpha_scale=std::ldexp(1.0,8-F);                      // M -> levels of a shifted sample
  plan.alpha_window_plain=e_shifted/65536.0+4.0e-9;
  plan.alpha_window_blend=plan.alpha_window_plain;
  // colour: |value - value~| <= 65536*(E_N+E_D)/(D~-E_D); for D~ >= 1024 E_D the factor 1.002 covers
  // the denominator
  plan.colour_window=1.002*65536.0*(e_colour+e_shifted)/m_unit;
  plan.alpha_floor=1024.0*e_shifted/m_unit;
  // "M_alpha == 0 <=> every alpha of the window is 0" needs one level under the smallest tap to be
  // far above the alpha sum's error bound
  if (!(smallest*65536.0 > 8.0*e_shifted) || !(smallest > 1.0e-7))
    return plan;
  plan.ok=true;
  return plan;
}

// One device block per tap set: K doubles, K floats, the digits (shared_table keeps it).
struct ExactDeviceTaps
{
  const double *taps64=nullptr;
  const float *taps=nullptr;
  const signed char *digits=nullptr;
  std::shared_ptr<void> keep;           // the device block, until the launch is enqueued
};

static inline MhStatus upload_exact_taps(const View &src,const double *taps,int K,const ExactTapPlan &plan,
  ExactDeviceTaps *out)
{
  const size_t doubles=(size_t) K,floats=((size_t) K+1)/2;
  const size_t digit_words=((size_t) kExactDigits*kExactDigitPitch+7)/8;
  std::vector<double> host(doubles+floats+digit_words,0.0);
  float *host_floats=reinterpret_cast<float *>(host.data()+doubles);
  signed char *host_digits=reinterpret_cast<signed char *>(host.data()+doubles+floats);
  for (int v=0; v < K; v++)
    {
      host[(size_t) v]=taps[v];
      host_floats[v]=(float) taps[v];
    }
  std::memcpy(host_digits,plan.digits.data(),plan.digits.size());
  const void *device=nullptr;
  MH_TRY(shared_table(src.device,src.stream,host.data(),host.size()*sizeof(double),&device,&out->keep));
  out->taps64=static_cast<const double *>(device);
  out->taps=reinterpret_cast<const float *>(out->taps64+doubles);
  out->digits=reinterpret_cast<const signed char *>(out->taps64+doubles+floats);
  return MH_OK;
}


// MhExactBlurRecomputed's device counter of `device` (nullptr unless counting is on)
unsigned long long *exact_recomputed_counter(int device);

} // namespace mh
