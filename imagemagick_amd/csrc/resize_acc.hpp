// The per-output accumulator of HorizontalFilter / VerticalFilter (MagickCore/resize.c:3494-3530,
// :3709-3745) under an arithmetic policy; shared by resize.hip and resize_mfma.hip.
#pragma once

#include "device_common.hpp"
#include <type_traits>

namespace mh {

// accumulate one tap into (s[],g)
template<typename Q,int C,bool BLEND,class A>
struct ResizeAcc
{
  typedef typename A::T T;
  // Fma64 (the +-1 mode): the colour sums are weighted by weight*alpha without QuantumScale
  // and gamma is derived from the alpha sum afterwards (gamma*pixel = S_c/S_alpha, the scale
  // cancels) — 5 fp64 operations per tap instead of 6, and no weight*QuantumScale table
  static constexpr bool kDerive=BLEND && std::is_same<A,Fma64>::value;
  T s[C];
  T g;
  __device__ __forceinline__ void init()
  {
#pragma unroll
    for (int c=0; c < C; c++)
      s[c]=(T) 0;
    g=(T) 0;
  }
  __device__ __forceinline__ void tap(T w,T wq,const Q (&q)[C])
  {
    if constexpr (BLEND)
      {
        // alpha=weight*QuantumScale*GetPixelAlpha(); pixel+=alpha*p; gamma+=alpha  (resize.c:3515-3520)
        T a=A::mul(kDerive ? w : wq,(T) q[C-1]);
#pragma unroll
        for (int c=0; c < C-1; c++)
          s[c]=A::mac(s[c],a,(T) q[c]);
        if constexpr (kDerive)
          s[C-1]=A::add(s[C-1],a);
        else
          {
            g=A::add(g,a);
            s[C-1]=A::mac(s[C-1],w,(T) q[C-1]);
          }
      }
    else
      {
#pragma unroll
        for (int c=0; c < C; c++)
          s[c]=A::mac(s[c],w,(T) q[c]);           // resize.c:3503-3505
      }
  }
  // same as tap() for a pixel already converted to T
  __device__ __forceinline__ void tap_converted(T w,T wq,const T (&p)[C])
  {
    if constexpr (BLEND)
      {
        T a=A::mul(kDerive ? w : wq,p[C-1]);
#pragma unroll
        for (int c=0; c < C-1; c++)
          s[c]=A::mac(s[c],a,p[c]);
        if constexpr (kDerive)
          s[C-1]=A::add(s[C-1],a);
        else
          {
            g=A::add(g,a);
            s[C-1]=A::mac(s[C-1],w,p[C-1]);
          }
      }
    else
      {
#pragma unroll
        for (int c=0; c < C; c++)
          s[c]=A::mac(s[c],w,p[c]);
      }
  }
  // kDerive with a sample staged as (alpha*p .., alpha): sum w*(alpha*p) instead of
  // sum (w*alpha)*p — one fused multiply-add per channel and tap, no product per tap (the staging
  // pays the three products once per SOURCE sample, which ~4*taps outputs share)
  __device__ __forceinline__ void tap_premultiplied(T w,const T (&p)[C])
  {
    static_assert(kDerive,"the premultiplied form is the derived-gamma mode's");
#pragma unroll
    for (int c=0; c < C; c++)
      s[c]=A::mac(s[c],w,p[c]);
  }
  __device__ __forceinline__ void finish(const Q (&copy)[C],uint32_t copy_mask,Q (&out)[C]) const
  {
    if constexpr (kDerive)
      {
        // gamma = PerceptibleReciprocal(QuantumScale*S_alpha); gamma*(QuantumScale*S_c) = S_c*inv
        // with inv = 1/S_alpha, or (+-1/MagickEpsilon)*QuantumScale under the clamp
        const double sa=(double) s[C-1];
        const double mag=sa < 0.0 ? -sa : sa;
        // (branch-free: the Newton reciprocal of a clamped sum is computed and discarded)
        double r=__builtin_amdgcn_rcp(sa);
        double e=__builtin_fma(-sa,r,1.0);
        r=__builtin_fma(r,e,r);
        e=__builtin_fma(-sa,r,1.0);
        r=__builtin_fma(r,e,r);
        const double clamped=(sa < 0.0 ? -kInvEps : kInvEps)*kQS;
        const double inv=(mag*kQS) >= kEps ? r : clamped;
        if (copy_mask == 0)
          {
#pragma unroll
            for (int c=0; c < C-1; c++)
              out[c]=QuantumOps<Q>::clamp((double) s[c]*inv);
            out[C-1]=QuantumOps<Q>::clamp(sa);
            return;
          }
#pragma unroll
        for (int c=0; c < C; c++)
          {
            double pixel=c != C-1 ? (double) s[c]*inv : sa;
            out[c]=((copy_mask >> c) & 1u) ? copy[c] : QuantumOps<Q>::clamp(pixel);
          }
        return;
      }
#pragma unroll
    for (int c=0; c < C; c++)
      {
        if ((copy_mask >> c) & 1u)
          {
            out[c]=copy[c];
            continue;
          }
        double pixel=(double) s[c];
        if (BLEND && (c != C-1))
          {
            if constexpr (std::is_same<A,Fma64>::value)
              pixel=perceptible_reciprocal_fast((double) g)*pixel;
            else
              pixel=perceptible_reciprocal((double) g)*pixel;
          }
        out[c]=QuantumOps<Q>::clamp(pixel);
      }
  }
};

} // namespace mh
