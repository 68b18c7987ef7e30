// Shared by the kernels that evaluate a 2-D ConvolveMorphology some other way than the reference's
// w x h walk and still return its bits (convolve_separable.hip: two fp64 passes; convolve2d_exact.hip:
// integer sums on the i8 matrix cores): given the sums to a known absolute error, which Quantum
// does the reference produce — and the reference's own walk for the samples that cannot be told.
#pragma once
#include "device_common.hpp"

namespace mh {

// One output sample exactly as morphology.c:2919-2979 forms it, by a whole wave (every lane calls
// it with the same arguments and gets the same result).  `values`: the kernel's cells in its own
// order ([kh][kw]; a NaN cell is skipped like `if (!IsNaN(*k))` does — its term is +0.0, which
// leaves both running sums as they are); the window's top-left is (x-shiftx, y-shifty).  The
// reference's terms alpha*k*p and alpha*k are rounded on their own, so 64 of them are formed at
// a time, one per lane; its two running sums take them in the reference's order through
// v_readlane — a scalar walk of the 6241 cells of GaussianBlur 0x10 paid a full memory latency
// per cell (1.5 ms per sample; 86 000 undecided samples of a float 8192^2 frame: 46 ms).
template<typename Q,int C,bool BLEND>
static __device__ __forceinline__ Q conv2d_reference_sample(const Q *src,int W,int H,int x,int y,int c,
  const double *values,int kw,int kh,int shiftx,int shifty,int lane)
{
  const int cells=kw*kh;
  const bool weighted=BLEND && (c != C-1);
  double pixel=0.0,gamma=weighted ? 0.0 : 1.0;
  // this lane's term (and weight) of the 64 cells from `base` on
  auto terms_of=[&](int base,double &term,double &weight)
  {
    term=0.0;
    weight=0.0;
    const int at=base+lane;
    if (at < cells)
      {
        const int v=at/kw,u=at-v*kw;
        int yy=y-shifty+v,xx=x-shiftx+u;
        yy=yy < 0 ? 0 : (yy > H-1 ? H-1 : yy);
        xx=xx < 0 ? 0 : (xx > W-1 ? W-1 : xx);
        const Q *sample=src+((size_t) yy*W+(size_t) xx)*C;
        const double cell=values[cells-1-at];      // k starts at the last cell and walks backwards
        if (cell == cell)
          {
            if (weighted)
              {
                const double alpha=kQS*(double) sample[C-1];
                weight=alpha*cell;
                term=weight*(double) sample[c];      // alpha*(*k)*pixels[i]
              }
            else
              term=cell*(double) sample[c];
          }
      }
  };
  // (the next 64 cells are fetched while the two running sums take the current ones: the walk of
  // one sample was a chain of memory latencies — 70 us for a 31 x 31 kernel, 0.2 ms for 79 x 79 —
  // and it sits behind the column pass of convolve_separable.hip's folded form)
  double term,weight;
  terms_of(0,term,weight);
  for (int base=0; base < cells; base+=64)
    {
      double next_term=0.0,next_weight=0.0;
      if (base+64 < cells)
        terms_of(base+64,next_term,next_weight);
      const int count=cells-base < 64 ? cells-base : 64;
      for (int j=0; j < count; j++)
        {
          const int lo=__builtin_amdgcn_readlane((int) (unsigned) __double_as_longlong(term),j);
          const int hi=__builtin_amdgcn_readlane((int) (unsigned) (__double_as_longlong(term) >> 32),j);
          pixel+=__longlong_as_double(((long long) hi << 32) | (long long) (unsigned) lo);
          if (weighted)
            {
              const int wlo=__builtin_amdgcn_readlane((int) (unsigned) __double_as_longlong(weight),j);
              const int whi=__builtin_amdgcn_readlane((int) (unsigned) (__double_as_longlong(weight) >> 32),j);
              gamma+=__longlong_as_double(((long long) whi << 32) | (long long) (unsigned) wlo);
            }
        }
      term=next_term;
      weight=next_weight;
    }
  gamma=perceptible_reciprocal(gamma);
  return QuantumOps<Q>::clamp(gamma*pixel);
}

// s[c] = sum k*P_c in real arithmetic to within error[c] (P = alpha*p for the alpha-weighted colour
// channels of a BLEND layout, the plain sample otherwise; the alpha sums are in Quantum units, not
// scaled by QuantumScale).  out[c] = the Quantum the reference's epilogue (gamma = Perceptible-
// Reciprocal(sum alpha*k), ClampToQuantum(gamma*pixel), morphology.c:3192-3194) makes of it; the
// returned mask has bit c set where the error bound cannot tell which side of a rounding boundary
// the reference lands on.  mixed_signs: the kernel has cells of both signs, so an alpha sum of
// exactly zero does not mean "every alpha of the window is zero".
template<typename Q,int C,bool BLEND>
static __device__ __forceinline__ uint32_t settle_sums(const double (&s)[4],const double (&error)[4],
  int mixed_signs,Q (&out)[C])
{
  double inverse=1.0,alpha_error=0.0;
  bool unsure=false;
  if constexpr (BLEND)
    {
      const double sa=s[C-1];
      alpha_error=error[C-1];
      // PerceptibleReciprocal's clamp acts on QuantumScale*S_alpha below MagickEpsilon; and an
      // alpha sum of mixed-sign cells that has cancelled down to its own error says nothing
      unsure=((sa != 0.0) || (mixed_signs != 0)) &&
        (!(__builtin_fabs(kQS*sa) >= kEps*1.000001) || !(__builtin_fabs(sa) > 8.0*alpha_error));
      inverse=sa == 0.0 ? 0.0 : perceptible_reciprocal_fast(sa);
    }
  uint32_t doubtful=0;
#pragma unroll
  for (int c=0; c < C; c++)
    {
      const bool weighted=BLEND && (c != C-1);
      double value=s[c];
      double bound=error[c];
      if (weighted)
        {
          value=value*inverse;
          bound=__builtin_fma(__builtin_fabs(value),alpha_error,bound)*__builtin_fabs(inverse)+
            __builtin_fabs(value)*1.0e-15;
        }
      if constexpr (QuantumOps<Q>::is_float)
        {
          const float nearest=(float) value;
          const uint32_t bits=__float_as_uint(nearest);
          const int exponent=(int) ((bits >> 23) & 0xffu);
          const bool power_of_two=(bits & 0x7fffffu) == 0u;
          const bool ordinary=(exponent != 0xff) && ((exponent != 0) || ((bits & 0x7fffffffu) == 0u));
          const int half_exponent=(exponent > 0 ? exponent : 1)-151-(power_of_two ? 1 : 0);
          const double half_ulp=__longlong_as_double((long long) (half_exponent+1023) << 52);
          const double distance=__builtin_fabs(value-(double) nearest);
          const bool decided=ordinary && (half_ulp-distance > bound);
          out[c]=nearest;
          if (!decided || (unsure && weighted))
            doubtful|=1u << c;
        }
      else
        {
          // ClampToQuantum (quantum.h:86-97): the only boundaries are the n+1/2 inside the
          // range; beyond it the level is 0 or 65535 whatever the last bits say
          const double shifted=value+0.5;
          const double fraction=shifted-__builtin_floor(shifted);
          const double distance=fraction < 0.5 ? fraction : 1.0-fraction;
          const bool inside=(value > -1.0) && (value < 65536.0);
          out[c]=QuantumOps<Q>::clamp(value);
          if ((inside && !(distance > bound+1.0e-9)) || (unsure && weighted) || !(value == value))
            doubtful|=1u << c;
        }
    }
  return doubtful;
}

} // namespace mh
