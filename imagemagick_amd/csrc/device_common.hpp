// Device-side helpers shared by the HIP kernels: Quantum conversion and
// clamping, pixel vector I/O, arithmetic policies.
//
// IMPORTANT: every .hip file is compiled with -ffp-contract=off.  The
// reference CPU build has no FMA contraction (x86-64 baseline, configure's
// "-O2 -mtune=core2"), so the EXACT policy must keep every multiply and add a
// separately rounded IEEE operation, in the CPU's order.  Where a fused
// multiply-add is wanted (FAST policy) it is written explicitly.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace mh {

constexpr double kEps = 1.0e-12;          // MagickEpsilon, magick-type.h:114
constexpr double kQR = 65535.0;           // QuantumRange
constexpr double kQS = 1.0/65535.0;       // QuantumScale, magick-type.h:119
constexpr double kInvEps = 1.0/kEps;       // 1/MagickEpsilon, as PerceptibleReciprocal's division yields

// ----------------------------------------------------------------- Quantum
template<typename Q> struct QuantumOps;

template<> struct QuantumOps<uint16_t>
{
  static constexpr bool is_float=false;
  // ClampToQuantum, non-HDRI branch: MagickCore/quantum.h:86-97
  static __device__ __forceinline__ uint16_t clamp(double v)
  {
    if (!(v > 0.0))            // NaN or <= 0
      return 0;
    if (v >= kQR)
      return 65535;
    return (uint16_t) (v+0.5);
  }
  // ScaleQuantumToMap, quantum-private.h:504-514 (Q16: identity)
  static __device__ __forceinline__ unsigned map_index(uint16_t q) { return q; }
};

template<> struct QuantumOps<float>
{
  static constexpr bool is_float=true;
  // ClampToQuantum, HDRI branch: a plain cast (quantum.h:88-89)
  static __device__ __forceinline__ float clamp(double v) { return (float) v; }
  // ScaleQuantumToMap, HDRI branch
  static __device__ __forceinline__ unsigned map_index(float q)
  {
    if (q >= 65535.0f)
      return 65535u;
    if (!(q > 0.0f))
      return 0u;
    return (unsigned) (q+0.5f);
  }
};

// fp64 sums between the passes of a separable 2-D kernel (convolve_separable.hip): no Quantum
template<> struct QuantumOps<double>
{
  static constexpr bool is_float=true;
  static __device__ __forceinline__ double clamp(double v) { return v; }
};

// PerceptibleReciprocal, MagickCore/pixel-accessor.h:242-254
static __device__ __forceinline__ double perceptible_reciprocal(double x)
{
  // sign/MagickEpsilon == sign*(1.0/MagickEpsilon) exactly (sign is +-1 and the quotient is
  // folded by the compiler with IEEE rounding); written as a multiply so that the select does
  // not evaluate a second fp64 division sequence per pixel
  double sign=x < 0.0 ? -1.0 : 1.0;
  if ((sign*x) >= kEps)
    return 1.0/x;
  return sign*kInvEps;
}

// 1/x to ~1 ulp without the IEEE division sequence (v_rcp_f64 + two Newton steps);
// same clamp as PerceptibleReciprocal.
static __device__ __forceinline__ double perceptible_reciprocal_fast(double x)
{
  double sign=x < 0.0 ? -1.0 : 1.0;
  if ((sign*x) < kEps)
    return sign*kInvEps;
  double r=__builtin_amdgcn_rcp(x);
  double e=__builtin_fma(-x,r,1.0);
  r=__builtin_fma(r,e,r);
  e=__builtin_fma(-x,r,1.0);
  r=__builtin_fma(r,e,r);
  return r;
}

// ------------------------------------------------------------- pixel I/O
template<typename Q,int C>
static __device__ __forceinline__ void load_pixel(const Q *p,Q (&v)[C])
{
  constexpr int bytes=(int) sizeof(Q)*C;
  if constexpr (bytes == 16)
    {
      uint4 t=*reinterpret_cast<const uint4 *>(p);
      __builtin_memcpy(v,&t,16);
    }
  else if constexpr (bytes == 8)
    {
      uint2 t=*reinterpret_cast<const uint2 *>(p);
      __builtin_memcpy(v,&t,8);
    }
  else if constexpr (bytes == 4)
    {
      uint32_t t=*reinterpret_cast<const uint32_t *>(p);
      __builtin_memcpy(v,&t,4);
    }
  else
    {
#pragma unroll
      for (int c=0; c < C; c++)
        v[c]=p[c];
    }
}

template<typename Q,int C>
static __device__ __forceinline__ void store_pixel(Q *p,const Q (&v)[C])
{
  constexpr int bytes=(int) sizeof(Q)*C;
  if constexpr (bytes == 16)
    {
      uint4 t;
      __builtin_memcpy(&t,v,16);
      *reinterpret_cast<uint4 *>(p)=t;
    }
  else if constexpr (bytes == 8)
    {
      uint2 t;
      __builtin_memcpy(&t,v,8);
      *reinterpret_cast<uint2 *>(p)=t;
    }
  else if constexpr (bytes == 4)
    {
      uint32_t t;
      __builtin_memcpy(&t,v,4);
      *reinterpret_cast<uint32_t *>(p)=t;
    }
  else
    {
#pragma unroll
      for (int c=0; c < C; c++)
        p[c]=v[c];
    }
}

// ------------------------------------------------------ arithmetic policies
// EXACT: double, separately rounded multiply and add, CPU operation order.
struct Exact64
{
  typedef double T;
  static constexpr bool premultiply=false;
  static constexpr bool taps_in_lds=false;
  static constexpr bool tie_check=false;
  static __device__ __forceinline__ T mul(T a,T b) { return a*b; }
  static __device__ __forceinline__ T add(T a,T b) { return a+b; }
  static __device__ __forceinline__ T mac(T acc,T a,T b) { return acc+a*b; }   // two roundings
};

// FAST for the fp64 kernels (resampling): double with fused multiply-adds and a
// Newton reciprocal.  Results differ from the CPU's separately rounded doubles by
// ~1e-16 relative, i.e. after the final rounding to Quantum they are identical
// except when the exact value lies within ~1e-11 of a rounding boundary: Q16
// output is then off by one level, float output by one float ULP (the +-1 contract).
struct Fma64
{
  typedef double T;
  static constexpr bool premultiply=false;
  static constexpr bool taps_in_lds=false;
  static constexpr bool tie_check=false;
  static __device__ __forceinline__ T mul(T a,T b) { return a*b; }
  static __device__ __forceinline__ T add(T a,T b) { return a+b; }
  static __device__ __forceinline__ T mac(T acc,T a,T b) { return __builtin_fma(a,b,acc); }
};

// EXACT results at less than half the fp64 work: the sums are formed with fused multiply-adds
// over alpha-premultiplied samples (alpha*p is an exact integer below 2^32, so a tap costs one FMA
// per channel instead of the reference's multiply, multiply, add, add), which agrees with the
// reference's separately rounded sums to ~1e-14 relative — 1e-9 Quantum levels.  Rounding to a
// Quantum level can then only differ when the value lies within 1e-9 of a rounding tie; every
// result within kTieMargin (1e-6) of a tie, and every pixel whose alpha sum is small enough for
// PerceptibleReciprocal's clamp to act, is recomputed in the reference's own operation order
// (about two samples per million).  The output is bit-identical to Exact64's; Q16 only.
struct Tie64
{
  typedef double T;
  static constexpr bool premultiply=true;
  static constexpr bool taps_in_lds=false;
  static constexpr bool tie_check=true;
  static __device__ __forceinline__ T mul(T a,T b) { return a*b; }
  static __device__ __forceinline__ T add(T a,T b) { return a+b; }
  static __device__ __forceinline__ T mac(T acc,T a,T b) { return __builtin_fma(a,b,acc); }
};
constexpr double kTieMargin=1.0e-6;

// FAST: float with explicit FMA; alpha is folded into the colour channels once
// per input pixel.  Only offered for Q16, where the accumulated error stays
// well inside one Quantum level (DESIGN.md "Precision").
struct Fast32
{
  typedef float T;
  static constexpr bool premultiply=true;
  static constexpr bool taps_in_lds=true;
  static constexpr bool tie_check=false;
  static __device__ __forceinline__ T mul(T a,T b) { return a*b; }
  static __device__ __forceinline__ T add(T a,T b) { return a+b; }
  static __device__ __forceinline__ T mac(T acc,T a,T b) { return __builtin_fmaf(a,b,acc); }
};

// One output sample of a 1-D Convolve pass exactly as the reference forms it: the kernel walked
// backwards over the edge-clamped window, every multiply and add separately rounded
// (morphology.c:2743-2776 column path, :2941-2977 row path), PerceptibleReciprocal, ClampToQuantum.
// taps: reversed doubles (taps[v] multiplies the input at o-shift+v).  Used by the Tie64 policy
// for the few results its fused sums cannot decide.
template<typename Q,int C,bool BLEND>
static __device__ __noinline__ Q conv1d_reference_sample(const Q *src,int W,int H,bool vertical,int x,int y,
  int c,const double *taps,int K,int shift,double bias)
{
  double pixel=bias,gamma=0.0;
  const bool weighted=BLEND && (c != C-1);
  for (int v=0; v < K; v++)
    {
      int xx=x,yy=y;
      if (vertical)
        {
          yy=y-shift+v;
          yy=yy < 0 ? 0 : (yy > H-1 ? H-1 : yy);
        }
      else
        {
          xx=x-shift+v;
          xx=xx < 0 ? 0 : (xx > W-1 ? W-1 : xx);
        }
      const Q *p=src+((size_t) yy*(size_t) W+(size_t) xx)*C;
      if (weighted)
        {
          const double alpha=kQS*(double) p[C-1];
          pixel+=alpha*taps[v]*(double) p[c];
          gamma+=alpha*taps[v];
        }
      else
        pixel+=taps[v]*(double) p[c];
    }
  if (weighted)
    pixel=perceptible_reciprocal(gamma)*pixel;
  return QuantumOps<Q>::clamp(pixel);
}

// wave64 sum of a small non-negative count; lane 0 gets the total
static __device__ __forceinline__ unsigned wave_sum(unsigned v)
{
#pragma unroll
  for (int off=32; off > 0; off>>=1)
    v+=__shfl_down(v,off,64);
  return v;
}

} // namespace mh
