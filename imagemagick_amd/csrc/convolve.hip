// Separable (1-D) ConvolveMorphology passes — the BlurImage hot loop.
//
// Reference semantics restated (not ported) from MorphologyPrimitive:
//   horizontal 1 x K kernel  : general row path   MagickCore/morphology.c:2811-2979, 3192-3200
//   vertical   K x 1 kernel  : column fast path   MagickCore/morphology.c:2654-2807
// Both evaluate, per Update channel c of output pixel o along the filter axis,
//     s = bias + sum_{v=0..K-1} [alpha_v] * values[K-1-v] * in(o - origin' + v)
//     out = ClampToQuantum(gamma * s),  gamma = 1/sum(alpha_v*values[..]) for
//     Blend channels, 1 otherwise;  origin' = K-1-origin (reflected kernel)
// with edge-clamped reads (cache.c:2663-2679) and the sum taken in ascending v.
//
// MI355X mapping: the 79-tap sigma=10 blur is VALU-bound, so the
// kernels are register-blocked along the filter axis: each lane owns R
// consecutive outputs, streams the R+K-1 inputs they depend on exactly once,
// and keeps the taps in SGPRs (the tap index depends only on wave-uniform
// loop counters).  The column pass reads rows straight from global memory
// (lane = pixel column => every load is one coalesced 512 B row segment); the
// row pass stages a (64R+K-1)-pixel strip per wavefront in LDS with a
// one-slot-per-R padding that makes the stride-R reads bank-conflict free.
#include "mh_internal.hpp"
#include "device_common.hpp"
#include "tie_check.hpp"
#include "separable_args.hpp"
#include <cstdlib>
#include <vector>
#include <type_traits>

namespace mh {

// (conv1d_reference_sample: device_common.hpp)

// finish()'s fallback for policies without a tie check
struct NoReference
{
  __device__ __forceinline__ unsigned operator()(int) const { return 0u; }
};

// ... and the real one: output (x,y) of a pass over `src`
template<typename Q,int C,bool BLEND>
struct ReferenceSample
{
  const Q *src;
  const double *taps;
  int W,H,K,shift,x,y;
  bool vertical;
  double bias;
  __device__ __forceinline__ Q operator()(int c) const
  {
    return conv1d_reference_sample<Q,C,BLEND>(src,W,H,vertical,x,y,c,taps,K,shift,bias);
  }
};

// ---------------------------------------------------------------- Accum
template<typename Q,int C,bool BLEND,class A,int R>
struct Accum
{
  typedef typename A::T T;
  // channel pairs live in 2-element vectors so the f32 policy can issue one
  // v_pk_fma_f32 per pair: a wave can issue only one VALU instruction per ~4.6
  // cycles (tools/ubench), so the packed form halves the issue slots a wave needs
  // and a single ready wave keeps the SIMD's FMA pipe busy.
  typedef T V2 __attribute__((ext_vector_type(2)));
  static constexpr int NP=(C+1)/2;
  V2 sv[R][NP];
  T g[R];
  // float Quantum under the tie check: the largest |sample| each channel has streamed (the error bound of a
  // sum is relative to it) and (2K+6)*2^-53, see finish().  Kept as a float, two samples an instruction
  // (v_max3_f32 with |x| source modifiers) where a running fp64 maximum of the premultiplied samples cost one
  // v_max_f64 a sample and channel.  (The OR of the samples' bit patterns — one v_or3_b32 for two — is a bound too,
  // but not a tight one: exponents 0x7f and 0x80 OR to 0xff, so a frame holding a 1.5 beside a 2.5 has an infinite
  // "bound" and every lane takes the reference's order: the row pass of a uniform-random frame ran 3.5x slower,
  // profiles/r6_notes/float_passes_and_dilate_order.txt.)  A NaN is dropped by the maximum as it was by
  // v_max_f64 (the sums are NaN then, and decide nothing); an Inf makes the bound infinite.
  static constexpr bool kTracksMagnitude=A::tie_check && (sizeof(Q) == 4);
  float seen[kTracksMagnitude ? C : 1];
  T error_unit;
#define MH_S(r,c) sv[r][(c) >> 1][(c) & 1]

  struct In { V2 pv[NP]; T a; };
#define MH_P(in,c) (in).pv[(c) >> 1][(c) & 1]

  __device__ __forceinline__ void init(T bias,int ntaps=0)
  {
#pragma unroll
    for (int r=0; r < R; r++)
      {
#pragma unroll
        for (int c=0; c < C; c++)
          MH_S(r,c)=A::premultiply ? (T) 0 : bias;
        g[r]=(T) 0;
      }
    error_unit=(T) 0;
    if constexpr (kTracksMagnitude)
      {
#pragma unroll
        for (int c=0; c < C; c++)
          seen[c]=0.0f;
        error_unit=(T) (2*ntaps+6)*(T) 1.1102230246251565e-16;
      }
  }

  __device__ __forceinline__ void observe(const Q (&q)[C])
  {
    if constexpr (kTracksMagnitude)
      {
#pragma unroll
        for (int c=0; c < C; c++)
          asm("v_max_f32 %0, %1, |%2|" : "=v"(seen[c]) : "v"(seen[c]),"v"(q[c]));
      }
  }

  __device__ __forceinline__ void observe(const Q (&q0)[C],const Q (&q1)[C])
  {
    if constexpr (kTracksMagnitude)
      {
#pragma unroll
        for (int c=0; c < C; c++)
          asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(seen[c]) : "v"(seen[c]),"v"(q0[c]),"v"(q1[c]));
      }
  }

  // ... as a magnitude of what channel c's sums are made of: alpha * p for a weighted colour channel
  // (max |alpha * p| <= max |alpha| * max |p|; the product of two floats is exact in fp64)
  __device__ __forceinline__ T magnitude(int c) const
  {
    if constexpr (kTracksMagnitude)
      {
        T m=(T) seen[c];
        if (BLEND && A::premultiply && (c != C-1))
          m=m*(T) seen[C-1];
        return m;
      }
    else
      return (T) 0;
  }

  static __device__ __forceinline__ In prepare(const Q (&q)[C])
  {
    In in;
#pragma unroll
    for (int q2=0; q2 < NP; q2++)
      in.pv[q2]=V2{(T) 0,(T) 0};
#pragma unroll
    for (int c=0; c < C; c++)
      MH_P(in,c)=(T) q[c];
    in.a=(T) 0;
    if constexpr (BLEND)
      {
        if constexpr (A::premultiply)
          {
            // gamma*pixel = (sum k*QS*alpha*p)/(sum k*QS*alpha): QuantumScale cancels, so
            // the FAST policy weights by the raw alpha quantum
            // whole channel pairs with one packed multiply, the odd colour channel alone
            const T alpha=MH_P(in,C-1);
#pragma unroll
            for (int q2=0; q2 < (C-1)/2; q2++)
              in.pv[q2]=in.pv[q2]*V2{alpha,alpha};
            if constexpr (((C-1) & 1) != 0)
              MH_P(in,C-2)=A::mul(alpha,MH_P(in,C-2));
          }
        else
          in.a=A::mul((T) kQS,MH_P(in,C-1));    // alpha=QuantumScale*GetPixelAlpha(): morphology.c:2766, :2965
      }
    return in;
  }

  __device__ __forceinline__ void packed_mac(int r,T kv,const In &in)
  {
    V2 k2={kv,kv};
#pragma unroll
    for (int q=0; q < C/2; q++)
      sv[r][q]=__builtin_elementwise_fma(k2,in.pv[q],sv[r][q]);
    if constexpr ((C & 1) != 0)
      MH_S(r,C-1)=A::mac(MH_S(r,C-1),kv,MH_P(in,C-1));
  }

  __device__ __forceinline__ void tap(int r,T kv,const In &in)
  {
    if constexpr (BLEND)
      {
        if constexpr (A::premultiply)
          {
            packed_mac(r,kv,in);
            // gamma = sum kv*QuantumScale*alpha = QuantumScale*(s[alpha]-bias): no
            // separate accumulator in this mode (finish() derives it)
          }
        else
          {
            // pixel+=alpha*(*k)*pixels[i]; gamma+=alpha*(*k);  morphology.c:2767-2768
            T w=A::mul(in.a,kv);
#pragma unroll
            for (int c=0; c < C-1; c++)
              MH_S(r,c)=A::add(MH_S(r,c),A::mul(w,MH_P(in,c)));
            g[r]=A::add(g[r],w);
            MH_S(r,C-1)=A::mac(MH_S(r,C-1),kv,MH_P(in,C-1));      // alpha channel: no weighting
          }
      }
    else
      {
        if constexpr (A::premultiply)
          packed_mac(r,kv,in);
        else
          {
#pragma unroll
            for (int c=0; c < C; c++)
              MH_S(r,c)=A::mac(MH_S(r,c),kv,MH_P(in,c));          // morphology.c:2750
          }
      }
  }

  // returns the number of channels that count as "changed" (morphology.c:2772, :3199)
  // reference(c): the sample recomputed in the reference's order (Tie64's doubtful results)
  template<class Reference=NoReference>
  __device__ __forceinline__ unsigned finish(int r,const Q (&center)[C],uint32_t copy_mask,
    Q (&out)[C],T bias,bool count_changed=true,const Reference &reference=Reference()) const
  {
    unsigned changed=0;
    if constexpr (A::tie_check)
      {
        // fused fp64 sums of alpha-premultiplied samples (bias 0, no change count: the launcher
        // sends everything else to Exact64).  S_c = sum k*alpha*p, S_a = sum k*alpha.
        static_assert(((sizeof(Q) == 2) || (sizeof(Q) == 4)) && (sizeof(T) == 8),
          "the tie check is for Q16 or float samples summed in fp64");
        if constexpr (sizeof(Q) == 4)
          {
            // float Quantum (HDRI): the result is the fp64 value rounded to float, so a "tie" is a
            // value closer to the midpoint of two neighbouring floats than the two summation
            // orders can differ.  Per channel that is at most (2K+6)*2^-53 * max|sample| * sum|k|
            // (K roundings of each order's running sum plus the reference's three per term; the
            // launcher admits normalised positive taps only, sum|k| <= 1); the alpha-weighted
            // colour channels add the quotient's share.  NaN fails every comparison below and
            // lands in the reference path, as do infinities, denormals and powers of two (whose
            // lower neighbour is half as far away).
            T inverse=(T) 1,alpha_error=(T) 0;
            bool doubtful=false;
            if constexpr (BLEND)
              {
                const T sa=MH_S(r,C-1);
                doubtful=(sa != (T) 0) && !(__builtin_fabs((T) kQS*sa) >= (T) kEps*(T) 1.000001);
                inverse=sa == (T) 0 ? (T) 0 : perceptible_reciprocal_fast(sa);
                alpha_error=error_unit*magnitude(C-1);
              }
#pragma unroll
            for (int c=0; c < C; c++)
              {
                if ((copy_mask >> c) & 1u)
                  {
                    out[c]=center[c];
                    continue;
                  }
                T value=MH_S(r,c);
                T error=error_unit*magnitude(c);
                if (BLEND && (c != C-1))
                  {
                    value=value*inverse;
                    error=__builtin_fma(__builtin_fabs(value),alpha_error,error)*__builtin_fabs(inverse)+
                      __builtin_fabs(value)*(T) 1.0e-15;
                  }
                const float nearest=(float) value;
                const uint32_t bits=__float_as_uint(nearest);
                const int exponent=(int) ((bits >> 23) & 0xffu);
                const bool power_of_two=(bits & 0x7fffffu) == 0u;
                const bool ordinary=(exponent != 0xff) && ((exponent != 0) || ((bits & 0x7fffffffu) == 0u));
                // half an ulp of `nearest`: 2^(e-127-24), a quarter below a power of two
                const int half_exponent=(exponent > 0 ? exponent : 1)-151-(power_of_two ? 1 : 0);
                const T half_ulp=__longlong_as_double((long long) (half_exponent+1023) << 52);
                const T distance=__builtin_fabs(value-(T) nearest);
                const bool decided=ordinary && (half_ulp-distance > error);
                Q level=nearest;
                if (!decided || (doubtful && BLEND && (c != C-1)))
                  level=reference(c);
                out[c]=level;
              }
            return 0;
          }
        T inverse=(T) 1;
        bool doubtful=false;
        if constexpr (BLEND)
          {
            const T sa=MH_S(r,C-1);
            // PerceptibleReciprocal acts below MagickEpsilon (the reference's gamma is QuantumScale*S_a)
            doubtful=(sa != (T) 0) && !((T) kQS*sa >= (T) kEps);
            inverse=sa == (T) 0 ? (T) 0 : perceptible_reciprocal_fast(sa);
          }
#pragma unroll
        for (int c=0; c < C; c++)
          {
            if ((copy_mask >> c) & 1u)
              {
                out[c]=center[c];
                continue;
              }
            T value=MH_S(r,c);
            if (BLEND && (c != C-1))
              value=value*inverse;
            const T shifted=value+(T) 0.5;
            const T fraction=shifted-__builtin_floor(shifted);
            const bool tie=(fraction < (T) kTieMargin) || (fraction > (T) 1-(T) kTieMargin);
            Q level=QuantumOps<Q>::clamp(value);
            if (tie || (doubtful && BLEND && (c != C-1)))
              level=(Q) reference(c);
            out[c]=level;
          }
        return 0;
      }
    else if constexpr (A::premultiply)
      {
        // FAST epilogue, all in f32 and branch-free.  s[] holds the un-biased sums
        // S_c = sum k*alpha*p (colour), S_a = sum k*alpha; the reference's
        // gamma*pixel is (bias*QuantumRange + S_c)/S_a, the alpha channel bias + S_a.
        // v_rcp_f32(0) = inf and 0*inf = NaN convert to 0, which is what
        // PerceptibleReciprocal's clamp yields for an all-transparent window.
        static_assert(sizeof(Q) == 2,"the FAST policy is Q16 only");
        T inv=(T) 1,cbias=bias;
        if constexpr (BLEND)
          {
            inv=__builtin_amdgcn_rcpf(MH_S(r,C-1));
            cbias=bias*(T) kQR;
          }
#pragma unroll
        for (int c=0; c < C; c++)
          {
            T pixel=MH_S(r,c);
            if (BLEND && (c != C-1))
              pixel=(pixel+cbias)*inv;
            else
              pixel=pixel+bias;
            if (count_changed)
              {
                // compares the un-normalised sum (morphology.c:2772, :3199)
                T raw=BLEND && (c != C-1) ? (T) kQS*MH_S(r,c)+bias : MH_S(r,c)+bias;
                T d=raw-(T) center[c];
                if (!((copy_mask >> c) & 1u) && ((d < (T) 0 ? -d : d) >= (T) kEps))
                  changed++;
              }
            // ClampToQuantum (quantum.h:86-97): v_cvt_u32_f32 maps NaN and negatives to 0
            unsigned q=(unsigned) (pixel+(T) 0.5);
            q=q > 65535u ? 65535u : q;
            out[c]=((copy_mask >> c) & 1u) ? center[c] : (Q) q;
          }
        return changed;
      }
    T gsum=g[r];
#pragma unroll
    for (int c=0; c < C; c++)
      {
        if ((copy_mask >> c) & 1u)
          {
            out[c]=center[c];
            continue;
          }
        double pixel=(double) MH_S(r,c);
        if (fabs(pixel-(double) center[c]) >= kEps)
          changed++;
        if (BLEND && (c != C-1))
          pixel=perceptible_reciprocal((double) gsum)*pixel;
        out[c]=QuantumOps<Q>::clamp(pixel);
      }
    return changed;
  }
};

struct Conv1DArgs
{
  const void *src;
  void *dst;
  int columns,rows;
  int ntaps;
  int shift;                 // K-1-origin: offset of the first input sample
  double bias;
  uint32_t copy_mask;
  const void *taps;          // T[K], reversed so that taps[v] multiplies input o-shift+v
  unsigned long long *changed;
  int nblocks;               // blocked kernels: number of U-sample blocks of the padded tap table
  int wave_bytes;            // separable_row_sums_kernel: bytes of LDS a wave owns (strip / stash)
  // triangular column kernels: UnsharpMaskImage's epilogue on the way out (effect.c:4364-4369);
  // unsharp_src = the unblurred frame, or NULL
  const void *unsharp_src;
  double unsharp_gain,unsharp_threshold;
  const unsigned *only_if;   // nullptr, or: every workgroup leaves at once unless the word is set:
  unsigned only_if_token;    //   0: set = not zero;  otherwise: set = equal to the token (no clearing launch needed)
};

static __device__ __forceinline__ bool pass_not_wanted(const Conv1DArgs &args)
{
  if (args.only_if == nullptr)
    return false;
  const unsigned word=*args.only_if;
  return args.only_if_token == 0u ? word == 0u : word != args.only_if_token;
}

// blurred sample -> unsharp-masked sample, as unsharp_kernel (pointwise.hip) does in its own pass
template<typename Q,int C>
static __device__ __forceinline__ void unsharp_on_the_way_out(const Conv1DArgs &args,size_t pixel,Q (&out)[C])
{
  Q p[C];
  load_pixel<Q,C>(static_cast<const Q *>(args.unsharp_src)+pixel*C,p);
#pragma unroll
  for (int c=0; c < C; c++)
    {
      if ((args.copy_mask >> c) & 1u)
        {
          out[c]=p[c];
          continue;
        }
      double value=(double) p[c]-(double) out[c];
      if (fabs(2.0*value) < args.unsharp_threshold)
        value=(double) p[c];
      else
        value=(double) p[c]+args.unsharp_gain*value;
      out[c]=QuantumOps<Q>::clamp(value);
    }
}

template<typename Q,int C,bool BLEND,class A>
static __device__ __forceinline__ auto make_reference(const Conv1DArgs &args,bool vertical,int x,int y)
{
  if constexpr (A::tie_check)
    return ReferenceSample<Q,C,BLEND>{static_cast<const Q *>(args.src),static_cast<const double *>(args.taps),
      args.columns,args.rows,args.ntaps,args.shift,x,y,vertical,args.bias};
  else
    return NoReference();
}

// --------------------------------------------------------------- column pass
template<typename Q,int C,bool BLEND,class A,int R,int WAVES>
__global__ __launch_bounds__(64*WAVES)
void conv_column_kernel(Conv1DArgs args)
{
  if (pass_not_wanted(args))
    return;
  typedef typename A::T T;
  typedef Accum<Q,C,BLEND,A,R> Acc;
  const int lane=(int) (threadIdx.x & 63);
  const int wave=__builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int W=args.columns,H=args.rows,K=args.ntaps;
  // XCD-aware tile order: consecutive workgroup ids land on different XCDs
  // (id % 8), so give each XCD a contiguous range of tiles, y fastest, and
  // vertically adjacent tiles (which share K-1 input rows) meet in one L2.
  const unsigned ntx=(unsigned) ((W+63)/64);
  const unsigned nty=(unsigned) ((H+R*WAVES-1)/(R*WAVES));
  const unsigned total=ntx*nty;
  unsigned id=blockIdx.x;
  const unsigned per=(total+7u)/8u;
  unsigned tile=(id & 7u)*per+(id >> 3);
  if (tile >= total)
    return;
  const int tx=(int) (tile/nty),ty=(int) (tile%nty);
  const int x=tx*64+lane;
  const int xc=x < W ? x : W-1;
  const int y0=(ty*WAVES+wave)*R;
  if (y0 >= H)
    return;
  const Q *src=static_cast<const Q *>(args.src);
  Q *dst=static_cast<Q *>(args.dst);
  const T *taps=static_cast<const T *>(args.taps);
  const size_t pitch=(size_t) W*C;

  Acc acc;
  acc.init((T) args.bias);
  const int NJ=R+K-1;
  Q cur[C],nxt[C];
  {
    int yy=y0-args.shift;
    yy=yy < 0 ? 0 : (yy > H-1 ? H-1 : yy);
    load_pixel<Q,C>(src+(size_t) yy*pitch+(size_t) xc*C,cur);
  }
  // three loops instead of one with a branch: ramp-up (j < R-1), steady state
  // (every one of the R outputs takes a tap) and ramp-down (j >= K)
  auto fetch_next=[&](int j)
  {
    int yy=y0-args.shift+j+1;
    yy=yy < 0 ? 0 : (yy > H-1 ? H-1 : yy);
    load_pixel<Q,C>(src+(size_t) yy*pitch+(size_t) xc*C,nxt);
  };
  auto rotate=[&]()
  {
#pragma unroll
    for (int c=0; c < C; c++)
      cur[c]=nxt[c];
  };
  int j=0;
  const int ramp=(R-1) < NJ ? (R-1) : NJ;
  for ( ; j < ramp; j++)
    {
      fetch_next(j);
      typename Acc::In in=Acc::prepare(cur);
#pragma unroll
      for (int r=0; r < R; r++)
        {
          int t=j-r;
          if ((t >= 0) && (t < K))
            acc.tap(r,taps[t],in);
        }
      rotate();
    }
  for ( ; j < K; j++)
    {
      fetch_next(j);
      typename Acc::In in=Acc::prepare(cur);
#pragma unroll
      for (int r=0; r < R; r++)
        acc.tap(r,taps[j-r],in);
      rotate();
    }
  for ( ; j < NJ; j++)
    {
      fetch_next(j);
      typename Acc::In in=Acc::prepare(cur);
#pragma unroll
      for (int r=0; r < R; r++)
        {
          int t=j-r;
          if ((t >= 0) && (t < K))
            acc.tap(r,taps[t],in);
        }
      rotate();
    }
  unsigned changed=0;
#pragma unroll
  for (int r=0; r < R; r++)
    {
      int y=y0+r;
      if (y < H)
        {
          Q center[C],out[C];
          load_pixel<Q,C>(src+(size_t) y*pitch+(size_t) xc*C,center);
          unsigned ch=acc.finish(r,center,args.copy_mask,out,(T) args.bias);
          if (x < W)
            {
              if (args.unsharp_src != nullptr)
                unsharp_on_the_way_out<Q,C>(args,(size_t) y*W+(size_t) x,out);
              store_pixel<Q,C>(dst+(size_t) y*pitch+(size_t) x*C,out);
              changed+=ch;
            }
        }
    }
  if (args.changed != nullptr)
    {
      changed=wave_sum(changed);
      if ((lane == 0) && (changed != 0))
        atomicAdd(args.changed,(unsigned long long) changed);
    }
}

// ------------------------------------------------------------------ row pass
// LDS slot of strip sample i: one padding slot every R samples, so the lanes of
// a wave (stride R samples) hit distinct banks.
template<int R> static __device__ __forceinline__ int lds_slot(int i) { return i+i/R; }

template<typename Q,int C,bool BLEND,class A,int R,int WAVES>
__global__ __launch_bounds__(64*WAVES)
void conv_row_kernel(Conv1DArgs args)
{
  if (pass_not_wanted(args))
    return;
  typedef typename A::T T;
  typedef Accum<Q,C,BLEND,A,R> Acc;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane=(int) (threadIdx.x & 63);
  const int wave=__builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int W=args.columns,H=args.rows,K=args.ntaps;
  const int SEG=64*R;                         // outputs per wave
  const int NS=SEG+K-1;                       // input samples per wave
  const int slots=NS+NS/R+1;
  Q *strip=reinterpret_cast<Q *>(smem_raw)+(size_t) wave*slots*C;

  const unsigned ntx=(unsigned) ((W+SEG-1)/SEG);
  const unsigned nty=(unsigned) ((H+WAVES-1)/WAVES);
  const unsigned total=ntx*nty;
  unsigned id=blockIdx.x;
  const unsigned per=(total+7u)/8u;
  unsigned tile=(id & 7u)*per+(id >> 3);
  if (tile >= total)
    return;
  const int tx=(int) (tile%ntx),ty=(int) (tile/ntx);
  const int y=ty*WAVES+wave;
  const int x0=tx*SEG;
  const Q *src=static_cast<const Q *>(args.src);
  Q *dst=static_cast<Q *>(args.dst);
  const T *taps=static_cast<const T *>(args.taps);
  const size_t pitch=(size_t) W*C;
  const bool row_ok=y < H;
  const Q *row=src+(size_t) (row_ok ? y : H-1)*pitch;

  // stage the strip: sample i is input column clamp(x0-shift+i)
  for (int i=lane; i < NS; i+=64)
    {
      int xx=x0-args.shift+i;
      xx=xx < 0 ? 0 : (xx > W-1 ? W-1 : xx);
      Q v[C];
      load_pixel<Q,C>(row+(size_t) xx*C,v);
      store_pixel<Q,C>(strip+(size_t) lds_slot<R>(i)*C,v);
    }
  __syncthreads();
  if (!row_ok)
    return;

  Acc acc;
  acc.init((T) args.bias);
  const int NJ=R+K-1;
  const int base=lane*(R+1);                  // lds_slot(lane*R)
  Q cur[C],nxt[C];
  load_pixel<Q,C>(strip+(size_t) base*C,cur);
  auto fetch_next=[&](int j)
  {
    int jn=j+1 < NJ ? j+1 : j;
    load_pixel<Q,C>(strip+(size_t) (base+jn+jn/R)*C,nxt);
  };
  auto rotate=[&]()
  {
#pragma unroll
    for (int c=0; c < C; c++)
      cur[c]=nxt[c];
  };
  int j=0;
  const int ramp=(R-1) < NJ ? (R-1) : NJ;
  for ( ; j < ramp; j++)
    {
      fetch_next(j);
      typename Acc::In in=Acc::prepare(cur);
#pragma unroll
      for (int r=0; r < R; r++)
        {
          int t=j-r;
          if ((t >= 0) && (t < K))
            acc.tap(r,taps[t],in);
        }
      rotate();
    }
  for ( ; j < K; j++)
    {
      fetch_next(j);
      typename Acc::In in=Acc::prepare(cur);
#pragma unroll
      for (int r=0; r < R; r++)
        acc.tap(r,taps[j-r],in);
      rotate();
    }
  for ( ; j < NJ; j++)
    {
      fetch_next(j);
      typename Acc::In in=Acc::prepare(cur);
#pragma unroll
      for (int r=0; r < R; r++)
        {
          int t=j-r;
          if ((t >= 0) && (t < K))
            acc.tap(r,taps[t],in);
        }
      rotate();
    }
  unsigned changed=0;
  const int xo=x0+lane*R;
#pragma unroll
  for (int r=0; r < R; r++)
    {
      int x=xo+r;
      if (x < W)
        {
          Q center[C],out[C];
          int ci=lane*R+r+args.shift;         // strip index of input column x
          load_pixel<Q,C>(strip+(size_t) lds_slot<R>(ci)*C,center);
          changed+=acc.finish(r,center,args.copy_mask,out,(T) args.bias);
          store_pixel<Q,C>(dst+(size_t) y*pitch+(size_t) x*C,out);
        }
    }
  if (args.changed != nullptr)
    {
      changed=wave_sum(changed);
      if ((lane == 0) && (changed != 0))
        atomicAdd(args.changed,(unsigned long long) changed);
    }
}


// =================================================================== blocked
// Q16 kernels.  Same register blocking (each lane owns R consecutive outputs
// along the filter axis and streams the R+K-1 inputs they depend on once),
// but the input stream is cut into blocks of U samples and the tap table is
// zero-padded by R-1 entries on both sides: inside a block the tap that
// output r takes from sample jj is  table[jb + (jj-r+R-1)], a compile-time
// offset from the wave-uniform block base, so one block needs R+U-1 taps that
// are fetched with a few scalar loads into SGPRs and every FMA reads its tap
// from a fixed SGPR.  No per-sample tap reloads, no ramp-up/ramp-down branches
// (the zero taps make the edge samples contribute nothing; Q16 samples are
// always finite, so  s + 0*p == s  exactly and the EXACT order is unchanged).
// The samples of block b+1 are fetched into registers while block b is
// accumulated.
//
// Where the taps live matters on gfx950 (tools/ubench/valu_rate.hip, 4 waves/SIMD):
// v_fmac_f32 with an SGPR multiplier sustains 64.6 TFLOP/s, the all-VGPR form
// 122 TFLOP/s.  The f32 policy therefore stages the tap table in LDS once per
// workgroup and reads each block's R+U-1 taps with broadcast ds_reads into
// VGPRs; the f64 policy (no such penalty measured) keeps scalar loads.
template<typename T,bool IN_LDS>
static __device__ __forceinline__ const T *stage_taps(const Conv1DArgs &args,unsigned char *smem,
  int count,bool sync_done)
{
  const T *global=static_cast<const T *>(args.taps);
  if constexpr (!IN_LDS)
    return global;
  else
    {
      T *lds=reinterpret_cast<T *>(smem);
      for (int i=(int) threadIdx.x; i < count; i+=(int) blockDim.x)
        lds[i]=global[i];
      if (!sync_done)
        __syncthreads();
      return lds;
    }
}

template<typename Q,int C,bool BLEND,class A,int R,int U,int WAVES>
__global__ __launch_bounds__(64*WAVES)
void conv_column_blocked(Conv1DArgs args)
{
  if (pass_not_wanted(args))
    return;
  typedef typename A::T T;
  typedef Accum<Q,C,BLEND,A,R> Acc;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane=(int) (threadIdx.x & 63);
  const int wave=__builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int W=args.columns,H=args.rows;
  const unsigned ntx=(unsigned) ((W+63)/64);
  const unsigned nty=(unsigned) ((H+R*WAVES-1)/(R*WAVES));
  const unsigned total=ntx*nty;
  const unsigned id=blockIdx.x;
  const unsigned per=(total+7u)/8u;
  const unsigned tile=(id & 7u)*per+(id >> 3);       // XCD-aware: see conv_column_kernel
  if (tile >= total)
    return;
  const int tx=(int) (tile/nty),ty=(int) (tile%nty);
  const int x=tx*64+lane;
  const int xc=x < W ? x : W-1;
  const int y0=(ty*WAVES+wave)*R;
  const Q *src=static_cast<const Q *>(args.src)+(size_t) xc*C;
  Q *dst=static_cast<Q *>(args.dst);
  const size_t pitch=(size_t) W*C;
  const int ybase=y0-args.shift;
  const int nblocks=args.nblocks;
  const T *table=stage_taps<T,A::taps_in_lds>(args,smem_raw,nblocks*U+R-1,false);
  if (y0 >= H)
    return;

  Acc acc;
  acc.init((T) args.bias);
  Q nxt[U][C];
  // interior tiles (every row this wave touches exists) address row jb+jj as
  // uniform base + jj*pitch + lane offset: two scalar adds per row and an
  // SGPR-base load, instead of a clamp and a 64-bit multiply per row
  const bool interior=(ybase >= 0) && (ybase+nblocks*U <= H);
  const unsigned lane_off=(unsigned) xc*(unsigned) (C*sizeof(Q));
  const char *base0=reinterpret_cast<const char *>(args.src);
  const size_t pitch_bytes=pitch*sizeof(Q);
  auto fetch=[&](int jb)
  {
    if (interior)
      {
        const char *rowp=base0+(size_t) (ybase+jb)*pitch_bytes;
#pragma unroll
        for (int jj=0; jj < U; jj++)
          {
            load_pixel<Q,C>(reinterpret_cast<const Q *>(rowp+lane_off),nxt[jj]);
            rowp+=pitch_bytes;
          }
      }
    else
      {
#pragma unroll
        for (int jj=0; jj < U; jj++)
          {
            int yy=ybase+jb+jj;
            yy=yy < 0 ? 0 : (yy > H-1 ? H-1 : yy);
            load_pixel<Q,C>(src+(size_t) yy*pitch,nxt[jj]);
          }
      }
  };
  fetch(0);
  for (int b=0; b < nblocks; b++)
    {
      const int jb=b*U;
      T tw[R+U-1];
#pragma unroll
      for (int i=0; i < R+U-1; i++)
        tw[i]=table[jb+i];
      Q cur[U][C];
#pragma unroll
      for (int jj=0; jj < U; jj++)
#pragma unroll
        for (int c=0; c < C; c++)
          cur[jj][c]=nxt[jj][c];
      if (b+1 < nblocks)
        fetch(jb+U);
#pragma unroll
      for (int jj=0; jj < U; jj++)
        {
          typename Acc::In in=Acc::prepare(cur[jj]);
#pragma unroll
          for (int r=0; r < R; r++)
            acc.tap(r,tw[jj-r+R-1],in);
        }
    }
  unsigned changed=0;
  // the centre pixel is only needed for the `changed` count and for Copy channels
  const bool need_center=(args.changed != nullptr) || (args.copy_mask != 0);
#pragma unroll
  for (int r=0; r < R; r++)
    {
      int y=y0+r;
      if (y < H)
        {
          Q center[C],out[C];
#pragma unroll
          for (int c=0; c < C; c++)
            center[c]=(Q) 0;
          if (need_center)
            load_pixel<Q,C>(src+(size_t) y*pitch,center);
          unsigned ch=acc.finish(r,center,args.copy_mask,out,(T) args.bias,args.changed != nullptr);
          if (x < W)
            {
              if (args.unsharp_src != nullptr)
                unsharp_on_the_way_out<Q,C>(args,(size_t) y*W+(size_t) x,out);
              store_pixel<Q,C>(dst+(size_t) y*pitch+(size_t) x*C,out);
              changed+=ch;
            }
        }
    }
  if (args.changed != nullptr)
    {
      changed=wave_sum(changed);
      if ((lane == 0) && (changed != 0))
        atomicAdd(args.changed,(unsigned long long) changed);
    }
}

// Row pass: a wave stages the 63R+NJ input pixels its 64R outputs depend on in
// LDS (coalesced global reads), with one padding slot every R samples so the
// stride-R reads of the 64 lanes fall on distinct banks; blocks are R samples
// long (U == R) so the slot of sample jb+jj is  base + jb + jb/R + jj.
template<typename Q,int C,bool BLEND,class A,int R,int U,int WAVES>
__global__ __launch_bounds__(64*WAVES)
void conv_row_blocked(Conv1DArgs args)
{
  if (pass_not_wanted(args))
    return;
  typedef typename A::T T;
  typedef Accum<Q,C,BLEND,A,R> Acc;
  static_assert((R%U) == 0,"a block of U samples must not straddle an LDS padding slot");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane=(int) (threadIdx.x & 63);
  const int wave=__builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int W=args.columns,H=args.rows;
  const int nblocks=args.nblocks;
  const int SEG=64*R;                         // outputs per wave
  const int NS=63*R+nblocks*U;                // input samples per wave
  const int slots=NS+NS/R+1;
  const int table_bytes=A::taps_in_lds ? (((nblocks*U+R-1)*(int) sizeof(T)+15) & ~15) : 0;
  Q *strip=reinterpret_cast<Q *>(smem_raw+table_bytes)+(size_t) wave*slots*C;

  const unsigned ntx=(unsigned) ((W+SEG-1)/SEG);
  const unsigned nty=(unsigned) ((H+WAVES-1)/WAVES);
  const unsigned total=ntx*nty;
  const unsigned id=blockIdx.x;
  const unsigned per=(total+7u)/8u;
  const unsigned tile=(id & 7u)*per+(id >> 3);
  if (tile >= total)
    return;
  const int tx=(int) (tile%ntx),ty=(int) (tile/ntx);
  const int y=ty*WAVES+wave;
  const int x0=tx*SEG;
  const Q *src=static_cast<const Q *>(args.src);
  Q *dst=static_cast<Q *>(args.dst);
  const T *table=stage_taps<T,A::taps_in_lds>(args,smem_raw,nblocks*U+R-1,true);
  const size_t pitch=(size_t) W*C;
  const bool row_ok=y < H;
  const Q *row=src+(size_t) (row_ok ? y : H-1)*pitch;

  // stage the strip in batches: all loads of a batch are in flight before the
  // first LDS store waits for one (a load-store loop would pay one memory
  // latency per 64 samples)
  {
    constexpr int BATCH=10;
    for (int i0=lane; i0 < NS; i0+=64*BATCH)
      {
        Q v[BATCH][C];
#pragma unroll
        for (int k=0; k < BATCH; k++)
          {
            int i=i0+64*k;
            int xx=x0-args.shift+(i < NS ? i : NS-1);
            xx=xx < 0 ? 0 : (xx > W-1 ? W-1 : xx);
            load_pixel<Q,C>(row+(size_t) xx*C,v[k]);
          }
#pragma unroll
        for (int k=0; k < BATCH; k++)
          {
            int i=i0+64*k;
            if (i < NS)
              store_pixel<Q,C>(strip+(size_t) (i+i/R)*C,v[k]);
          }
      }
  }
  __syncthreads();
  if (!row_ok)
    return;

  Acc acc;
  acc.init((T) args.bias);
  const Q *mine=strip+(size_t) lane*(R+1)*C;     // slot of sample lane*R
  // the samples of block b+1 are read from LDS into registers while block b is
  // accumulated (sample jb+jj sits in slot jb + jb/R + jj: U divides R)
  Q nxt[U][C];
  auto fetch=[&](int jb)
  {
    const Q *blk=mine+(size_t) (jb+jb/R)*C;
#pragma unroll
    for (int jj=0; jj < U; jj++)
      load_pixel<Q,C>(blk+(size_t) jj*C,nxt[jj]);
  };
  fetch(0);
  for (int b=0; b < nblocks; b++)
    {
      const int jb=b*U;
      T tw[R+U-1];
#pragma unroll
      for (int i=0; i < R+U-1; i++)
        tw[i]=table[jb+i];
      Q cur[U][C];
#pragma unroll
      for (int jj=0; jj < U; jj++)
#pragma unroll
        for (int c=0; c < C; c++)
          cur[jj][c]=nxt[jj][c];
      if (b+1 < nblocks)
        fetch(jb+U);
#pragma unroll
      for (int jj=0; jj < U; jj++)
        {
          typename Acc::In in=Acc::prepare(cur[jj]);
#pragma unroll
          for (int r=0; r < R; r++)
            acc.tap(r,tw[jj-r+R-1],in);
        }
    }
  unsigned changed=0;
  const int xo=x0+lane*R;
#pragma unroll
  for (int r=0; r < R; r++)
    {
      int x=xo+r;
      if (x < W)
        {
          Q center[C],out[C];
          int ci=lane*R+r+args.shift;           // strip index of input column x
          load_pixel<Q,C>(strip+(size_t) (ci+ci/R)*C,center);
          changed+=acc.finish(r,center,args.copy_mask,out,(T) args.bias,args.changed != nullptr);
          store_pixel<Q,C>(dst+(size_t) y*pitch+(size_t) x*C,out);
        }
    }
  if (args.changed != nullptr)
    {
      changed=wave_sum(changed);
      if ((lane == 0) && (changed != 0))
        atomicAdd(args.changed,(unsigned long long) changed);
    }
}

template<typename Q,int C,bool BLEND,class A,int R,int U>
static MhStatus launch_blocked(const View &src,const View &dst,bool vertical,
  const Conv1DParams &p,const Roles &roles,unsigned long long *changed)
{
  typedef typename A::T T;
  constexpr int WAVES=4;
  const int K=p.ntaps;
  const int UU=U;
  const int NJ=R+K-1;
  const int nblocks=(NJ+UU-1)/UU;
  // padded, reversed tap table: entry e multiplies, for output r, the sample
  // j = e-(R-1)+r; taps are reversed as in launch_one (morphology.c:2746, :2919)
  std::vector<T> host((size_t) (nblocks*UU+R-1+UU),(T) 0);
  for (int v=0; v < K; v++)
    host[(size_t) (v+R-1)]=(T) p.taps[K-1-v];
  Temp taps;
  MH_TRY(upload_table(taps,src.device,src.stream,host.data(),host.size()*sizeof(T)));

  Conv1DArgs args;
  args.unsharp_src=nullptr;
  args.unsharp_gain=0.0;
  args.unsharp_threshold=0.0;
  args.src=src.pixels;
  args.dst=dst.pixels;
  args.columns=(int) src.columns;
  args.rows=(int) src.rows;
  args.ntaps=K;
  args.shift=K-1-p.origin;
  args.bias=p.bias;
  args.copy_mask=roles.copy_mask;
  args.taps=taps.ptr;
  args.changed=changed;
  args.nblocks=nblocks;
  args.wave_bytes=0;
  args.only_if=p.only_if;
  args.only_if_token=p.only_if_token;
  const int W=args.columns,H=args.rows;
  if (vertical)
    {
      unsigned ntx=(unsigned) ((W+63)/64),nty=(unsigned) ((H+R*WAVES-1)/(R*WAVES));
      unsigned grid=((ntx*nty+7u)/8u)*8u;
      size_t lds=A::taps_in_lds ? (size_t) (nblocks*UU+R-1)*sizeof(T) : 0;
      ProfileScope prof("conv_column",src.stream);
      hipLaunchKernelGGL((conv_column_blocked<Q,C,BLEND,A,R,U,WAVES>),dim3(grid),dim3(64*WAVES),lds,
        src.stream,args);
    }
  else
    {
      const int SEG=64*R,NS=63*R+nblocks*UU,slots=NS+NS/R+1;
      size_t table_bytes=A::taps_in_lds ? ((((size_t) (nblocks*UU+R-1))*sizeof(T)+15u) & ~(size_t) 15u) : 0;
      size_t lds=table_bytes+(size_t) WAVES*slots*C*sizeof(Q);
      if (lds > 160u*1024u)
        return fail(MH_UNSUPPORTED,"row kernel of %d taps needs %zu bytes of LDS",K,lds);
      unsigned ntx=(unsigned) ((W+SEG-1)/SEG),nty=(unsigned) ((H+WAVES-1)/WAVES);
      unsigned grid=((ntx*nty+7u)/8u)*8u;
      if (lds > 64u*1024u)
        MH_HIP(hipFuncSetAttribute(
          reinterpret_cast<const void *>(&conv_row_blocked<Q,C,BLEND,A,R,U,WAVES>),
          hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
      ProfileScope prof("conv_row",src.stream);
      hipLaunchKernelGGL((conv_row_blocked<Q,C,BLEND,A,R,U,WAVES>),dim3(grid),dim3(64*WAVES),lds,
        src.stream,args);
    }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

template<class A,int R,int U>
static MhStatus dispatch_blocked(const View &src,const View &dst,bool vertical,
  const Conv1DParams &p,const Roles &roles,unsigned long long *changed)
{
  typedef uint16_t Q;
  const bool blend=roles.blend && (roles.alpha == src.channels-1);
  switch (src.channels)
  {
    case 1: return launch_blocked<Q,1,false,A,R,U>(src,dst,vertical,p,roles,changed);
    case 2:
      if (blend) return launch_blocked<Q,2,true,A,R,U>(src,dst,vertical,p,roles,changed);
      return launch_blocked<Q,2,false,A,R,U>(src,dst,vertical,p,roles,changed);
    case 3: return launch_blocked<Q,3,false,A,R,U>(src,dst,vertical,p,roles,changed);
    case 4:
      if (blend) return launch_blocked<Q,4,true,A,R,U>(src,dst,vertical,p,roles,changed);
      return launch_blocked<Q,4,false,A,R,U>(src,dst,vertical,p,roles,changed);
    default: break;
  }
  return fail(MH_UNSUPPORTED,"%d channels",src.channels);
}


// ================================================================ triangular
// The blocked kernels above spend (R-1)/(R+K-1) of their multiply-adds on the
// zero-padded ramp taps (15 % at K=79, R=16).  These variants split the sample
// stream of a lane's R outputs into
//   A  samples 0 .. R-1           output r takes sample j only when r <= j
//   B  samples R .. K-2           every output takes every sample (blocks of U, then single samples)
//   C  samples K-1 .. K+R-2       output r takes sample K-1+s only when r >= s
// A and C are fully unrolled with compile-time (sample, output) pairs, so exactly
// K*R*C multiply-adds are issued and larger R (fewer, longer sample streams per
// output) becomes profitable.  Tap table: the reversed taps t[0..K-1], t[v]
// multiplying sample j for output r when v = j-r.  Requires K >= R+1.
template<typename Q,int C,bool BLEND,class A,int R,int U,class FETCHU,class FETCH1>
static __device__ __forceinline__ void tri_accumulate(Accum<Q,C,BLEND,A,R> &acc,
  const typename A::T *table,int K,FETCHU fetch_u,FETCH1 fetch_1,Q (&nxt)[U][C])
{
  typedef typename A::T T;
  typedef Accum<Q,C,BLEND,A,R> Acc;
  static_assert((R%U) == 0,"R must be a multiple of U");
  static_assert((U%2) == 0,"samples are observed two at a time");
  const int M=K-1-R;                       // samples of phase B
  const int nfull=M/U,nrem=M-nfull*U;
  const int b2_begin=R+nfull*U,b2_end=K-1;
  auto prefetch=[&](int pos)
  {
    if ((pos >= b2_begin) && (pos < b2_end))
      fetch_1(pos);
    else
      fetch_u(pos);
  };
  Q cur[U][C];
  auto grab=[&]()
  {
#pragma unroll
    for (int jj=0; jj < U; jj++)
#pragma unroll
      for (int c=0; c < C; c++)
        cur[jj][c]=nxt[jj][c];
  };
  int pos=0;
  fetch_u(0);
  // ---- A
  {
    T ta[R];
#pragma unroll
    for (int i=0; i < R; i++)
      ta[i]=table[i];
#pragma unroll
    for (int c0=0; c0 < R/U; c0++)
      {
        grab();
        prefetch(pos+U);
#pragma unroll
        for (int jj=0; jj < U; jj++)
          {
            typename Acc::In in=Acc::prepare(cur[jj]);
            if ((jj & 1) == 0)
              acc.observe(cur[jj],cur[jj+1]);
#pragma unroll
            for (int r=0; r < R; r++)
              if (r <= c0*U+jj)
                acc.tap(r,ta[c0*U+jj-r],in);
          }
        pos+=U;
      }
  }
  // ---- B, blocks of U samples
  for (int b=0; b < nfull; b++)
    {
      grab();
      prefetch(pos+U);
      T tw[R+U-1];
#pragma unroll
      for (int i=0; i < R+U-1; i++)
        tw[i]=table[pos-(R-1)+i];
#pragma unroll
      for (int jj=0; jj < U; jj++)
        {
          typename Acc::In in=Acc::prepare(cur[jj]);
          if ((jj & 1) == 0)
            acc.observe(cur[jj],cur[jj+1]);
#pragma unroll
          for (int r=0; r < R; r++)
            acc.tap(r,tw[jj-r+R-1],in);
        }
      pos+=U;
    }
  // ---- B, the M mod U remaining samples one at a time
  for (int b=0; b < nrem; b++)
    {
      grab();
      prefetch(pos+1);
      T tw[R];
#pragma unroll
      for (int i=0; i < R; i++)
        tw[i]=table[pos-(R-1)+i];
      typename Acc::In in=Acc::prepare(cur[0]);
      acc.observe(cur[0]);
#pragma unroll
      for (int r=0; r < R; r++)
        acc.tap(r,tw[R-1-r],in);
      pos+=1;
    }
  // ---- C
  {
    T tc[R];
#pragma unroll
    for (int i=0; i < R; i++)
      tc[i]=table[K-R+i];
#pragma unroll
    for (int c0=0; c0 < R/U; c0++)
      {
        grab();
        if (c0+1 < R/U)
          fetch_u(pos+U);
#pragma unroll
        for (int jj=0; jj < U; jj++)
          {
            typename Acc::In in=Acc::prepare(cur[jj]);
            if ((jj & 1) == 0)
              acc.observe(cur[jj],cur[jj+1]);
#pragma unroll
            for (int r=0; r < R; r++)
              if (r >= c0*U+jj)
                acc.tap(r,tc[R-1+c0*U+jj-r],in);
          }
        pos+=U;
      }
  }
}

template<typename Q,int C,bool BLEND,class A,int R,int U,int WAVES>
__global__ __launch_bounds__(64*WAVES)
void conv_column_tri(Conv1DArgs args)
{
  if (pass_not_wanted(args))
    return;
  typedef typename A::T T;
  typedef Accum<Q,C,BLEND,A,R> Acc;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane=(int) (threadIdx.x & 63);
  const int wave=__builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int W=args.columns,H=args.rows,K=args.ntaps;
  const unsigned ntx=(unsigned) ((W+63)/64);
  const unsigned nty=(unsigned) ((H+R*WAVES-1)/(R*WAVES));
  const unsigned total=ntx*nty;
  const unsigned id=blockIdx.x;
  const unsigned per=(total+7u)/8u;
  const unsigned tile=(id & 7u)*per+(id >> 3);       // XCD-aware: see conv_column_kernel
  if (tile >= total)
    return;
  const int tx=(int) (tile/nty),ty=(int) (tile%nty);
  const int x=tx*64+lane;
  const int xc=x < W ? x : W-1;
  const int y0=(ty*WAVES+wave)*R;
  const Q *src=static_cast<const Q *>(args.src)+(size_t) xc*C;
  Q *dst=static_cast<Q *>(args.dst);
  const size_t pitch=(size_t) W*C;
  const int ybase=y0-args.shift;
  const T *table=stage_taps<T,A::taps_in_lds>(args,smem_raw,K,false);
  if (y0 >= H)
    return;

  Acc acc;
  acc.init((T) args.bias,K);
  Q nxt[U][C];
  const bool interior=(ybase >= 0) && (ybase+R+K+U <= H);
  const char *base0=reinterpret_cast<const char *>(args.src)+(size_t) xc*(size_t) (C*sizeof(Q));
  const size_t pitch_bytes=pitch*sizeof(Q);
  auto fetch_u=[&](int pos)
  {
    if (interior)
      {
        const char *rowp=base0+(size_t) (ybase+pos)*pitch_bytes;
#pragma unroll
        for (int jj=0; jj < U; jj++)
          {
            load_pixel<Q,C>(reinterpret_cast<const Q *>(rowp),nxt[jj]);
            rowp+=pitch_bytes;
          }
      }
    else
      {
#pragma unroll
        for (int jj=0; jj < U; jj++)
          {
            int yy=ybase+pos+jj;
            yy=yy < 0 ? 0 : (yy > H-1 ? H-1 : yy);
            load_pixel<Q,C>(src+(size_t) yy*pitch,nxt[jj]);
          }
      }
  };
  auto fetch_1=[&](int pos)
  {
    int yy=ybase+pos;
    yy=yy < 0 ? 0 : (yy > H-1 ? H-1 : yy);
    load_pixel<Q,C>(src+(size_t) yy*pitch,nxt[0]);
  };
  tri_accumulate<Q,C,BLEND,A,R,U>(acc,table,K,fetch_u,fetch_1,nxt);

  unsigned changed=0;
  const bool need_center=(args.changed != nullptr) || (args.copy_mask != 0);
#pragma unroll
  for (int r=0; r < R; r++)
    {
      int y=y0+r;
      if (y < H)
        {
          Q center[C],out[C];
#pragma unroll
          for (int c=0; c < C; c++)
            center[c]=(Q) 0;
          if (need_center)
            load_pixel<Q,C>(src+(size_t) y*pitch,center);
          unsigned ch=acc.finish(r,center,args.copy_mask,out,(T) args.bias,args.changed != nullptr,
            make_reference<Q,C,BLEND,A>(args,true,xc,y));
          if (x < W)
            {
              if (args.unsharp_src != nullptr)
                unsharp_on_the_way_out<Q,C>(args,(size_t) y*W+(size_t) x,out);
              store_pixel<Q,C>(dst+(size_t) y*pitch+(size_t) x*C,out);
              changed+=ch;
            }
        }
    }
  if (args.changed != nullptr)
    {
      changed=wave_sum(changed);
      if ((lane == 0) && (changed != 0))
        atomicAdd(args.changed,(unsigned long long) changed);
    }
}


// Column pass with the rows staged through LDS.  In conv_column_tri every wave
// fetches its own R+K-1 rows, so an image row is requested (R+K-1)/R = 5.9 times
// (R=16, K=79) and, with a 4 MB L2 per XCD, most of those re-reads go back to the
// fabric (PMC: 3.47 GB per launch against 1.07 GB algorithmic).  Here a workgroup
// of WAVES waves stages the WAVES*R+K-1 rows of its 64-column strip once, with
// batched coalesced loads, and the waves take their samples from LDS (lane = column
// => each row read is one conflict-free 512-byte ds_read_b64 sweep).
template<typename Q,int C,bool BLEND,class A,int R,int U,int WAVES>
__global__ __launch_bounds__(64*WAVES)
void conv_column_lds(Conv1DArgs args)
{
  if (pass_not_wanted(args))
    return;
  typedef typename A::T T;
  typedef Accum<Q,C,BLEND,A,R> Acc;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane=(int) (threadIdx.x & 63);
  const int wave=__builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int W=args.columns,H=args.rows,K=args.ntaps;
  const int table_bytes=A::taps_in_lds ? ((K*(int) sizeof(T)+15) & ~15) : 0;
  Q *tile=reinterpret_cast<Q *>(smem_raw+table_bytes);
  const unsigned ntx=(unsigned) ((W+63)/64);
  const unsigned nty=(unsigned) ((H+R*WAVES-1)/(R*WAVES));
  const unsigned total=ntx*nty;
  const unsigned id=blockIdx.x;
  const unsigned per=(total+7u)/8u;
  const unsigned tile_id=(id & 7u)*per+(id >> 3);    // XCD-aware: see conv_column_kernel
  if (tile_id >= total)
    return;
  const int tx=(int) (tile_id/nty),ty=(int) (tile_id%nty);
  const int x=tx*64+lane;
  const int yb=ty*WAVES*R;                             // first output row of the workgroup
  const int y0=yb+wave*R;
  const Q *src=static_cast<const Q *>(args.src);
  Q *dst=static_cast<Q *>(args.dst);
  const size_t pitch=(size_t) W*C;
  const T *table=stage_taps<T,A::taps_in_lds>(args,smem_raw,K,true);
  // stage rows yb-shift .. yb-shift+nrows-1 (edge clamp, cache.c:2663-2679), 64 pixels each
  const int nrows=WAVES*R+K-1;
  {
    constexpr int BATCH=18;
    const int items=nrows*64;
    const int x0=tx*64;
    for (int i0=(int) threadIdx.x; i0 < items; i0+=64*WAVES*BATCH)
      {
        Q v[BATCH][C];
#pragma unroll
        for (int k=0; k < BATCH; k++)
          {
            int idx=i0+64*WAVES*k;
            idx=idx < items ? idx : items-1;
            int yy=yb-args.shift+(idx >> 6),xx=x0+(idx & 63);
            yy=yy < 0 ? 0 : (yy > H-1 ? H-1 : yy);
            xx=xx > W-1 ? W-1 : xx;
            load_pixel<Q,C>(src+(size_t) yy*pitch+(size_t) xx*C,v[k]);
          }
#pragma unroll
        for (int k=0; k < BATCH; k++)
          if (i0+64*WAVES*k < items)
            store_pixel<Q,C>(tile+(size_t) (i0+64*WAVES*k)*C,v[k]);
      }
  }
  __syncthreads();
  if (y0 >= H)
    return;

  Acc acc;
  acc.init((T) args.bias,K);
  const Q *mine=tile+((size_t) wave*R*64+(size_t) lane)*C;      // sample j lives at mine + j*64*C
  Q nxt[U][C];
  auto fetch_u=[&](int pos)
  {
#pragma unroll
    for (int jj=0; jj < U; jj++)
      {
        int j=pos+jj;
        j=j < R+K-1 ? j : R+K-2;                                 // the last prefetch may run over
        load_pixel<Q,C>(mine+(size_t) j*64*C,nxt[jj]);
      }
  };
  auto fetch_1=[&](int pos)
  {
    load_pixel<Q,C>(mine+(size_t) pos*64*C,nxt[0]);
  };
  tri_accumulate<Q,C,BLEND,A,R,U>(acc,table,K,fetch_u,fetch_1,nxt);

  unsigned changed=0;
#pragma unroll
  for (int r=0; r < R; r++)
    {
      int y=y0+r;
      if (y < H)
        {
          Q center[C],out[C];
          load_pixel<Q,C>(mine+(size_t) (r+args.shift)*64*C,center);   // input row y
          unsigned ch=acc.finish(r,center,args.copy_mask,out,(T) args.bias,args.changed != nullptr,
            make_reference<Q,C,BLEND,A>(args,true,x < W ? x : W-1,y));
          if (x < W)
            {
              if (args.unsharp_src != nullptr)
                unsharp_on_the_way_out<Q,C>(args,(size_t) y*W+(size_t) x,out);
              store_pixel<Q,C>(dst+(size_t) y*pitch+(size_t) x*C,out);
              changed+=ch;
            }
        }
    }
  if (args.changed != nullptr)
    {
      changed=wave_sum(changed);
      if ((lane == 0) && (changed != 0))
        atomicAdd(args.changed,(unsigned long long) changed);
    }
}

template<typename Q,int C,bool BLEND,class A,int R,int U,int WAVES>
__global__ __launch_bounds__(64*WAVES)
void conv_row_tri(Conv1DArgs args)
{
  if (pass_not_wanted(args))
    return;
  typedef typename A::T T;
  typedef Accum<Q,C,BLEND,A,R> Acc;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane=(int) (threadIdx.x & 63);
  const int wave=__builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int W=args.columns,H=args.rows,K=args.ntaps;
  const int SEG=64*R;                         // outputs per wave
  const int NS=63*R+R+K-1+U;                  // input samples per wave (+U: the C prefetch may run over)
  const int slots=NS+NS/R+1;
  const int table_bytes=A::taps_in_lds ? ((K*(int) sizeof(T)+15) & ~15) : 0;
  Q *strip=reinterpret_cast<Q *>(smem_raw+table_bytes)+(size_t) wave*slots*C;

  const unsigned ntx=(unsigned) ((W+SEG-1)/SEG);
  const unsigned nty=(unsigned) ((H+WAVES-1)/WAVES);
  const unsigned total=ntx*nty;
  const unsigned id=blockIdx.x;
  const unsigned per=(total+7u)/8u;
  const unsigned tile=(id & 7u)*per+(id >> 3);
  if (tile >= total)
    return;
  const int tx=(int) (tile%ntx),ty=(int) (tile/ntx);
  const int y=ty*WAVES+wave;
  const int x0=tx*SEG;
  const Q *src=static_cast<const Q *>(args.src);
  Q *dst=static_cast<Q *>(args.dst);
  const T *table=stage_taps<T,A::taps_in_lds>(args,smem_raw,K,true);
  const size_t pitch=(size_t) W*C;
  const bool row_ok=y < H;
  const Q *row=src+(size_t) (row_ok ? y : H-1)*pitch;
  {
    constexpr int BATCH=10;
    for (int i0=lane; i0 < NS; i0+=64*BATCH)
      {
        Q v[BATCH][C];
#pragma unroll
        for (int k=0; k < BATCH; k++)
          {
            int i=i0+64*k;
            int xx=x0-args.shift+(i < NS ? i : NS-1);
            xx=xx < 0 ? 0 : (xx > W-1 ? W-1 : xx);
            load_pixel<Q,C>(row+(size_t) xx*C,v[k]);
          }
#pragma unroll
        for (int k=0; k < BATCH; k++)
          {
            int i=i0+64*k;
            if (i < NS)
              store_pixel<Q,C>(strip+(size_t) (i+i/R)*C,v[k]);
          }
      }
  }
  __syncthreads();
  if (!row_ok)
    return;

  Acc acc;
  acc.init((T) args.bias,K);
  const Q *mine=strip+(size_t) lane*(R+1)*C;     // slot of sample lane*R
  Q nxt[U][C];
  auto fetch_u=[&](int pos)
  {
    // U consecutive samples; a block may straddle a padding slot only when pos is not a
    // multiple of U inside R, which the single-sample phase makes possible: index each one
#pragma unroll
    for (int jj=0; jj < U; jj++)
      {
        const int j=pos+jj;
        load_pixel<Q,C>(mine+(size_t) (j+j/R)*C,nxt[jj]);
      }
  };
  auto fetch_1=[&](int pos)
  {
    load_pixel<Q,C>(mine+(size_t) (pos+pos/R)*C,nxt[0]);
  };
  tri_accumulate<Q,C,BLEND,A,R,U>(acc,table,K,fetch_u,fetch_1,nxt);

  unsigned changed=0;
  const int xo=x0+lane*R;
  // 16-byte pixels (float RGBA: the HDRI blur): a lane's R consecutive results are 64 scattered
  // 16-byte pieces per store instruction when the lane stores them itself.  They leave through the
  // wave's strip instead — dead once no centre sample is needed — as contiguous kilobytes (see
  // separable_row_sums_kernel).  Whole segments only; rows' last segments keep the direct stores.
  if constexpr (C*sizeof(Q) == 16)
    {
      constexpr int LS=R*16+16;                 // a lane's deposit + a 16-byte gap (banks)
      const bool whole=x0+SEG <= W;
      if (whole && (args.changed == nullptr) && (args.copy_mask == 0) && (64*LS <= slots*16))
        {
          unsigned char *stash=reinterpret_cast<unsigned char *>(strip);
#pragma unroll
          for (int r=0; r < R; r++)
            {
              Q center[C],out[C];
#pragma unroll
              for (int c=0; c < C; c++)
                center[c]=(Q) 0;
              (void) acc.finish(r,center,0u,out,(T) args.bias,false,make_reference<Q,C,BLEND,A>(args,false,xo+r,y));
              store_pixel<Q,C>(reinterpret_cast<Q *>(stash+lane*LS+r*16),out);
            }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE,"wavefront");
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE,"wavefront");
          unsigned char *to=reinterpret_cast<unsigned char *>(dst+(size_t) y*pitch+(size_t) x0*C);
#pragma unroll
          for (int k=0; k < R; k++)
            {
              const int item=lane+64*k;           // 64 R pixels of 16 bytes
              const int from=item/R,within=item-from*R;
              const uint4 v=*reinterpret_cast<const uint4 *>(stash+from*LS+within*16);
              *reinterpret_cast<uint4 *>(to+(size_t) item*16)=v;
            }
          return;
        }
    }
  // a lane's R outputs are contiguous in the row: store them two pixels (16 bytes for
  // RGBA Q16) at a time — half the store instructions of this lane-strided pattern
  constexpr bool kPair=((R & 1) == 0) && (C*sizeof(Q) == 8);
#pragma unroll
  for (int r=0; r < R; r+=(kPair ? 2 : 1))
    {
      int x=xo+r;
      if (x < W)
        {
          Q center[C],out[C];
          int ci=lane*R+r+args.shift;           // strip index of input column x
          load_pixel<Q,C>(strip+(size_t) (ci+ci/R)*C,center);
          changed+=acc.finish(r,center,args.copy_mask,out,(T) args.bias,args.changed != nullptr,
            make_reference<Q,C,BLEND,A>(args,false,x,y));
          if constexpr (kPair)
            {
              if (x+1 < W)
                {
                  Q center1[C],out1[C];
                  int c1=ci+1;
                  load_pixel<Q,C>(strip+(size_t) (c1+c1/R)*C,center1);
                  changed+=acc.finish(r+1,center1,args.copy_mask,out1,(T) args.bias,args.changed != nullptr,
                    make_reference<Q,C,BLEND,A>(args,false,x+1,y));
                  Q both[2*C];
#pragma unroll
                  for (int c=0; c < C; c++)
                    {
                      both[c]=out[c];
                      both[C+c]=out1[c];
                    }
                  store_pixel<Q,2*C>(dst+(size_t) y*pitch+(size_t) x*C,both);
                  continue;
                }
            }
          store_pixel<Q,C>(dst+(size_t) y*pitch+(size_t) x*C,out);
        }
    }
  if (args.changed != nullptr)
    {
      changed=wave_sum(changed);
      if ((lane == 0) && (changed != 0))
        atomicAdd(args.changed,(unsigned long long) changed);
    }
}

template<typename Q,int C,bool BLEND,class A,int R,int U,int WAVES>
static MhStatus launch_tri_waves(const View &src,const View &dst,bool vertical,
  const Conv1DParams &p,const Roles &roles,unsigned long long *changed)
{
  typedef typename A::T T;
  const int K=p.ntaps;
  std::vector<T> host((size_t) K);
  for (int v=0; v < K; v++)
    host[(size_t) v]=(T) p.taps[K-1-v];        // reversed: morphology.c:2746, :2919
  Temp taps;
  MH_TRY(upload_table(taps,src.device,src.stream,host.data(),host.size()*sizeof(T)));
  Conv1DArgs args;
  args.unsharp_src=nullptr;
  args.unsharp_gain=0.0;
  args.unsharp_threshold=0.0;
  args.src=src.pixels;
  args.dst=dst.pixels;
  args.columns=(int) src.columns;
  args.rows=(int) src.rows;
  args.ntaps=K;
  args.shift=K-1-p.origin;
  args.bias=p.bias;
  args.copy_mask=roles.copy_mask;
  args.taps=taps.ptr;
  args.changed=changed;
  args.nblocks=0;
  args.wave_bytes=0;
  args.only_if=p.only_if;
  args.only_if_token=p.only_if_token;
  if (vertical)
    {
      args.unsharp_src=p.unsharp_source;
      args.unsharp_gain=p.unsharp_gain;
      args.unsharp_threshold=p.unsharp_threshold;
    }
  const int W=args.columns,H=args.rows;
  if (vertical)
    {
      unsigned ntx=(unsigned) ((W+63)/64),nty=(unsigned) ((H+R*WAVES-1)/(R*WAVES));
      unsigned grid=((ntx*nty+7u)/8u)*8u;
      size_t lds=A::taps_in_lds ? (size_t) K*sizeof(T) : 0;
      // rows staged once per workgroup through LDS when the strip fits twice per CU
      size_t table_bytes=A::taps_in_lds ? (((size_t) K*sizeof(T)+15u) & ~(size_t) 15u) : 0;
      size_t lds_tile=table_bytes+(size_t) (WAVES*R+K-1)*64*C*sizeof(Q);
      ProfileScope prof("conv_column",src.stream);
      if ((lds_tile <= 80u*1024u) && (option("MAGICKHIP_NO_COLUMN_LDS") == nullptr))
        {
          if (lds_tile > 64u*1024u)
            MH_HIP(hipFuncSetAttribute(
              reinterpret_cast<const void *>(&conv_column_lds<Q,C,BLEND,A,R,U,WAVES>),
              hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds_tile));
          hipLaunchKernelGGL((conv_column_lds<Q,C,BLEND,A,R,U,WAVES>),dim3(grid),dim3(64*WAVES),lds_tile,
            src.stream,args);
        }
      else
      hipLaunchKernelGGL((conv_column_tri<Q,C,BLEND,A,R,U,WAVES>),dim3(grid),dim3(64*WAVES),lds,
        src.stream,args);
    }
  else
    {
      const int SEG=64*R,NS=63*R+R+K-1+U,slots=NS+NS/R+1;
      size_t table_bytes=A::taps_in_lds ? (((size_t) K*sizeof(T)+15u) & ~(size_t) 15u) : 0;
      size_t lds=table_bytes+(size_t) WAVES*slots*C*sizeof(Q);
      if (lds > 160u*1024u)
        return fail(MH_UNSUPPORTED,"row kernel of %d taps needs %zu bytes of LDS",K,lds);
      unsigned ntx=(unsigned) ((W+SEG-1)/SEG),nty=(unsigned) ((H+WAVES-1)/WAVES);
      unsigned grid=((ntx*nty+7u)/8u)*8u;
      if (lds > 64u*1024u)
        MH_HIP(hipFuncSetAttribute(
          reinterpret_cast<const void *>(&conv_row_tri<Q,C,BLEND,A,R,U,WAVES>),
          hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
      ProfileScope prof("conv_row",src.stream);
      hipLaunchKernelGGL((conv_row_tri<Q,C,BLEND,A,R,U,WAVES>),dim3(grid),dim3(64*WAVES),lds,
        src.stream,args);
    }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

template<typename Q,int C,bool BLEND,class A,int R,int U>
static MhStatus launch_tri(const View &src,const View &dst,bool vertical,
  const Conv1DParams &p,const Roles &roles,unsigned long long *changed)
{
  typedef typename A::T T;
  // fp64 column pass: eight waves share a staged strip when it still fits twice per CU (142
  // rows for 79 taps: 73 KB): the K-1 halo rows are staged per 64 output rows instead of per 32
  // and every SIMD holds four waves instead of two — conv_column 1.43 -> 1.21 ms on 8192^2
  // (Tie64).  The f32 policy measured slower with 8 and 16 waves and stays at four.
  if ((sizeof(T) == 8) && vertical)
    {
      const size_t table_bytes=A::taps_in_lds ? (((size_t) p.ntaps*sizeof(T)+15u) & ~(size_t) 15u) : 0;
      if (table_bytes+(size_t) (8*R+p.ntaps-1)*64*C*sizeof(Q) <= 80u*1024u)
        return launch_tri_waves<Q,C,BLEND,A,R,U,8>(src,dst,vertical,p,roles,changed);
    }
  return launch_tri_waves<Q,C,BLEND,A,R,U,4>(src,dst,vertical,p,roles,changed);
}

template<class A,int R,int U,typename Q=uint16_t>
static MhStatus dispatch_tri(const View &src,const View &dst,bool vertical,
  const Conv1DParams &p,const Roles &roles,unsigned long long *changed)
{
  const bool blend=roles.blend && (roles.alpha == src.channels-1);
  switch (src.channels)
  {
    case 1: return launch_tri<Q,1,false,A,R,U>(src,dst,vertical,p,roles,changed);
    case 2:
      if (blend) return launch_tri<Q,2,true,A,R,U>(src,dst,vertical,p,roles,changed);
      return launch_tri<Q,2,false,A,R,U>(src,dst,vertical,p,roles,changed);
    case 3: return launch_tri<Q,3,false,A,R,U>(src,dst,vertical,p,roles,changed);
    case 4:
      if (blend) return launch_tri<Q,4,true,A,R,U>(src,dst,vertical,p,roles,changed);
      return launch_tri<Q,4,false,A,R,U>(src,dst,vertical,p,roles,changed);
    default: break;
  }
  return fail(MH_UNSUPPORTED,"%d channels",src.channels);
}

// ---------------------------------------------------------------- folded separable passes
// convolve_separable.hip's four steps (premultiply -> row sums -> column sums -> finish: 216 bytes
// per RGBA Q16 pixel through HBM) as TWO launches moving 80: the row pass reads the frame's
// Quantum samples, forms P = (alpha*p .., alpha) on the way into its sums (Accum::prepare of a
// premultiplying fp64 policy: the product of two Quantum values is exact) and stores 4 doubles
// per pixel; the column pass reads those, and what leaves its accumulators goes straight through
// settle_sums (tie_check.hpp) to the destination frame — the undecided samples recomputed by
// the whole wave in the reference's order, as separable_finish_kernel does.
struct Premultiplied64
{
  typedef double T;
  static constexpr bool premultiply=true;
  static constexpr bool taps_in_lds=false;
  static constexpr bool tie_check=false;
  static __device__ __forceinline__ T mul(T a,T b) { return a*b; }
  static __device__ __forceinline__ T add(T a,T b) { return a+b; }
  static __device__ __forceinline__ T mac(T acc,T a,T b) { return __builtin_fma(a,b,acc); }
};

template<typename Q,int C,bool BLEND,int R,int U,int WAVES>
__global__ __launch_bounds__(64*WAVES)
void separable_row_sums_kernel(Conv1DArgs args,double *bound)
{
  typedef Premultiplied64 A;
  typedef double T;
  typedef Accum<Q,C,BLEND,A,R> Acc;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane=(int) (threadIdx.x & 63);
  const int wave=__builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int W=args.columns,H=args.rows,K=args.ntaps;
  const int SEG=64*R;
  const int NS=63*R+R+K-1+U;
  Q *strip=reinterpret_cast<Q *>(smem_raw+(size_t) wave*args.wave_bytes);

  const unsigned ntx=(unsigned) ((W+SEG-1)/SEG);
  const unsigned nty=(unsigned) ((H+WAVES-1)/WAVES);
  const unsigned total=ntx*nty;
  const unsigned id=blockIdx.x;
  const unsigned per=(total+7u)/8u;
  const unsigned tile=(id & 7u)*per+(id >> 3);
  if (tile >= total)
    return;
  const int tx=(int) (tile%ntx),ty=(int) (tile/ntx);
  const int y=ty*WAVES+wave;
  const int x0=tx*SEG;
  const Q *src=static_cast<const Q *>(args.src);
  const T *table=static_cast<const T *>(args.taps);
  const size_t pitch=(size_t) W*C;
  const bool row_ok=y < H;
  const Q *row=src+(size_t) (row_ok ? y : H-1)*pitch;
  // float Quantum: the largest |P_c| of the frame, which the column pass's error bound is
  // relative to — every pixel is staged by at least one wave
  double most[C];
#pragma unroll
  for (int c=0; c < C; c++)
    most[c]=0.0;
  {
    constexpr int BATCH=10;
    for (int i0=lane; i0 < NS; i0+=64*BATCH)
      {
        Q v[BATCH][C];
#pragma unroll
        for (int k=0; k < BATCH; k++)
          {
            int i=i0+64*k;
            int xx=x0-args.shift+(i < NS ? i : NS-1);
            xx=xx < 0 ? 0 : (xx > W-1 ? W-1 : xx);
            load_pixel<Q,C>(row+(size_t) xx*C,v[k]);
          }
#pragma unroll
        for (int k=0; k < BATCH; k++)
          {
            int i=i0+64*k;
            if (i < NS)
              store_pixel<Q,C>(strip+(size_t) (i+i/R)*C,v[k]);
            if constexpr (QuantumOps<Q>::is_float)
              {
                const double alpha=BLEND ? (double) v[k][C-1] : 1.0;
#pragma unroll
                for (int c=0; c < C; c++)
                  most[c]=__builtin_fmax(most[c],__builtin_fabs((BLEND && (c != C-1)) ? alpha*(double) v[k][c] :
                    (double) v[k][c]));                                    // (NaN: skipped)
              }
          }
      }
  }
  __syncthreads();
  if constexpr (QuantumOps<Q>::is_float)
    {
#pragma unroll
      for (int c=0; c < C; c++)
        {
          double m=most[c];
          for (int off=32; off > 0; off>>=1)
            m=__builtin_fmax(m,__shfl_xor(m,off,64));
          // non-negative doubles order as their bit patterns; an atomic only where it raises the bound
          const unsigned long long bits=(unsigned long long) __double_as_longlong(m);
          unsigned long long *word=reinterpret_cast<unsigned long long *>(bound+c);
          if ((lane == 0) && (bits > __hip_atomic_load(word,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT)))
            atomicMax(word,bits);
        }
    }
  if (!row_ok)
    return;

  Acc acc;
  acc.init((T) 0,K);
  const Q *mine=strip+(size_t) lane*(R+1)*C;
  Q nxt[U][C];
  auto fetch_u=[&](int pos)
  {
#pragma unroll
    for (int jj=0; jj < U; jj++)
      {
        const int j=pos+jj;
        load_pixel<Q,C>(mine+(size_t) (j+j/R)*C,nxt[jj]);
      }
  };
  auto fetch_1=[&](int pos)
  {
    load_pixel<Q,C>(mine+(size_t) (pos+pos/R)*C,nxt[0]);
  };
  tri_accumulate<Q,C,BLEND,A,R,U>(acc,table,K,fetch_u,fetch_1,nxt);

#ifdef MH_ROW_SUMS_DIRECT
  double *sums=static_cast<double *>(args.dst)+((size_t) y*W+(size_t) (x0+lane*R))*4;
#pragma unroll
  for (int r=0; r < R; r++)
    if (x0+lane*R+r < W)
      {
        double s[4]={0.0,0.0,0.0,0.0};
#pragma unroll
        for (int c=0; c < C; c++)
          s[c]=acc.MH_S(r,c);
        reinterpret_cast<double2 *>(sums+(size_t) r*4)[0]=make_double2(s[0],s[1]);
        reinterpret_cast<double2 *>(sums+(size_t) r*4)[1]=make_double2(s[2],s[3]);
      }
#else
  // A lane owns R consecutive pixels, 32 R bytes of sums: stored from the lane they are 64 scattered
  // 16-byte pieces an instruction (the pass ran at 3.5 TB/s).  They leave through the wave's own
  // strip instead (dead by now; LDS operations of one wave complete in order, no barrier): sixteen
  // lanes at a time deposit their 16 R pixels, then all 64 lanes store them as contiguous kilobytes.
  double2 *stash=reinterpret_cast<double2 *>(smem_raw+(size_t) wave*args.wave_bytes);
  double *out_row=static_cast<double *>(args.dst)+((size_t) y*W+(size_t) x0)*4;
#pragma unroll
  for (int pass=0; pass < 4; pass++)
    {
      if ((lane >> 4) == pass)
        {
#pragma unroll
          for (int r=0; r < R; r++)
            {
              double s[4]={0.0,0.0,0.0,0.0};
#pragma unroll
              for (int c=0; c < C; c++)
                s[c]=acc.MH_S(r,c);
              double2 *to=stash+((lane & 15)*(R+1)+r)*2;        // (a 32-byte gap a lane: the 16 lanes' banks)
              to[0]=make_double2(s[0],s[1]);
              to[1]=make_double2(s[2],s[3]);
            }
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE,"wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE,"wavefront");
#pragma unroll
      for (int k=0; k < (R+1)/2; k++)
        {
          const int item=lane+64*k;                             // 16 R pixels x two halves
          const int pixel=item >> 1,half=item & 1;
          if (pixel < 16*R)
            {
              const double2 v=stash[((pixel/R)*(R+1)+(pixel % R))*2+half];
              const int x=x0+pass*16*R+pixel;
              if (x < W)
                reinterpret_cast<double2 *>(out_row+(size_t) (pass*16*R+pixel)*4)[half]=v;
            }
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE,"wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE,"wavefront");
    }
#endif
}

template<typename Q,int C,bool BLEND,int R,int U,int WAVES>
__global__ __launch_bounds__(64*WAVES)
void separable_column_finish_kernel(Conv1DArgs args,SeparableArgs sep)
{
  typedef Fma64 A;
  typedef double T;
  typedef Accum<double,4,false,A,R> Acc;
  static_assert(R <= 8,"four doubtful bits per output row in one word");
  const int lane=(int) (threadIdx.x & 63);
  const int wave=__builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int W=args.columns,H=args.rows,K=args.ntaps;
  const unsigned ntx=(unsigned) ((W+63)/64);
  const unsigned nty=(unsigned) ((H+R*WAVES-1)/(R*WAVES));
  const unsigned total=ntx*nty;
  const unsigned id=blockIdx.x;
  const unsigned per=(total+7u)/8u;
  const unsigned tile=(id & 7u)*per+(id >> 3);       // XCD-aware: see conv_column_kernel
  if (tile >= total)
    return;
  const int tx=(int) (tile/nty),ty=(int) (tile%nty);
  const int x=tx*64+lane;
  const int xc=x < W ? x : W-1;
  const int y0=(ty*WAVES+wave)*R;
  if (y0 >= H)
    return;
  const double *in=static_cast<const double *>(args.src)+(size_t) xc*4;
  const size_t pitch=(size_t) W*4;
  const int ybase=y0-args.shift;
  const T *table=static_cast<const T *>(args.taps);

  Acc acc;
  acc.init((T) 0,K);
  double nxt[U][4];
  const bool interior=(ybase >= 0) && (ybase+R+K+U <= H);
  auto fetch_u=[&](int pos)
  {
    if (interior)
      {
        const double *rowp=in+(size_t) (ybase+pos)*pitch;
#pragma unroll
        for (int jj=0; jj < U; jj++)
          {
            load_pixel<double,4>(rowp,nxt[jj]);
            rowp+=pitch;
          }
      }
    else
      {
#pragma unroll
        for (int jj=0; jj < U; jj++)
          {
            int yy=ybase+pos+jj;
            yy=yy < 0 ? 0 : (yy > H-1 ? H-1 : yy);
            load_pixel<double,4>(in+(size_t) yy*pitch,nxt[jj]);
          }
      }
  };
  auto fetch_1=[&](int pos)
  {
    int yy=ybase+pos;
    yy=yy < 0 ? 0 : (yy > H-1 ? H-1 : yy);
    load_pixel<double,4>(in+(size_t) yy*pitch,nxt[0]);
  };
  tri_accumulate<double,4,false,A,R,U>(acc,table,K,fetch_u,fetch_1,nxt);

  const Q *src=static_cast<const Q *>(sep.src);
  Q *dst=static_cast<Q *>(sep.dst);
  double error[4]={0.0,0.0,0.0,0.0};
#pragma unroll
  for (int c=0; c < C; c++)
    error[c]=sep.error_unit*(QuantumOps<Q>::is_float ? sep.bound[c] : sep.fixed_bound[c]);
  uint32_t doubtful=0;                               // four bits per output row of this lane
#pragma unroll
  for (int r=0; r < R; r++)
    {
      const int y=y0+r;
      if (y < H)
        {
          double s[4];
#pragma unroll
          for (int c=0; c < 4; c++)
            s[c]=acc.MH_S(r,c);
          if (sep.delta != 0.0)
            {
              // + delta * (alpha*p .., alpha) of the one sample the extra cell sees
              int xx=xc+sep.delta_dx,yy=y+sep.delta_dy;
              xx=xx < 0 ? 0 : (xx > W-1 ? W-1 : xx);
              yy=yy < 0 ? 0 : (yy > H-1 ? H-1 : yy);
              Q q[C];
              load_pixel<Q,C>(src+((size_t) yy*W+(size_t) xx)*C,q);
              const double alpha=BLEND ? (double) q[C-1] : 1.0;
#pragma unroll
              for (int c=0; c < C; c++)
                s[c]=__builtin_fma(sep.delta,(BLEND && (c != C-1)) ? alpha*(double) q[c] : (double) q[c],s[c]);
            }
          Q out[C];
          const uint32_t which=settle_sums<Q,C,BLEND>(s,error,sep.mixed_signs,out);
          if (x < W)
            {
              store_pixel<Q,C>(dst+((size_t) y*W+(size_t) x)*C,out);
              doubtful|=which << (4*r);
            }
        }
    }
  // The undecided samples go to a queue that separable_settle_kernel works through with the whole
  // chip (a wave of this kernel holds 200 registers and its workgroup's tile while it walks the
  // reference's kw x kh cells: 15 000 such samples of a float 8192^2 frame under a 79 x 79 Gaussian
  // cost 4.4 ms here, 0.2 ms there).  What does not fit the queue is settled in place, by all lanes
  // of the wave (the loops are wave-uniform), and patched into the pixel its lane has just stored.
  unsigned recomputed=0;
  for (int r=0; r < R; r++)
    {
      const int y=y0+r;
      if (y >= H)
        break;
      const uint32_t mine=(doubtful >> (4*r)) & 15u;
      unsigned long long pending=__ballot(mine != 0u);
      if (pending == 0ull)
        continue;
      unsigned base=0;
      if (lane == 0)
        base=atomicAdd(sep.queue_count,(unsigned) __builtin_popcountll(pending));
      base=(unsigned) __builtin_amdgcn_readfirstlane((int) base);
      const unsigned rank=(unsigned) __builtin_popcountll(pending & ((1ull << lane)-1ull));
      const bool queued=(base < sep.queue_capacity) && (rank < sep.queue_capacity-base);
      if ((mine != 0u) && queued)
        sep.queue[base+rank]=((unsigned long long) ((size_t) y*W+(size_t) x) << 4) | (unsigned long long) mine;
      pending=__ballot((mine != 0u) && !queued);
      while (pending != 0ull)
        {
          const int who=__builtin_ctzll(pending);
          pending&=pending-1ull;
          const uint32_t which=(uint32_t) __builtin_amdgcn_readlane((int) mine,who);
          const int xw=tx*64+who;
          for (int c=0; c < C; c++)
            if ((which >> c) & 1u)
              {
                const Q settled=conv2d_reference_sample<Q,C,BLEND>(src,W,H,xw,y,c,sep.values,sep.kw,sep.kh,
                  sep.shiftx,sep.shifty,lane);
                if (lane == who)
                  {
                    dst[((size_t) y*W+(size_t) xw)*C+c]=settled;
                    recomputed++;
                  }
              }
        }
    }
  if (sep.recomputed != nullptr)
    {
      recomputed=wave_sum(recomputed);
      if ((lane == 0) && (recomputed != 0))
        atomicAdd(sep.recomputed,(unsigned long long) recomputed);
    }
}

// The queued samples of the column pass, one wave each, in the reference's own w x h order.
template<typename Q,int C,bool BLEND>
__global__ __launch_bounds__(256)
void separable_settle_kernel(SeparableArgs sep)
{
  const unsigned filled=*sep.queue_count;
  const unsigned count=filled < sep.queue_capacity ? filled : sep.queue_capacity;
  const int lane=(int) (threadIdx.x & 63);
  const unsigned wave=blockIdx.x*4u+(threadIdx.x >> 6),waves=gridDim.x*4u;
  const Q *src=static_cast<const Q *>(sep.src);
  Q *dst=static_cast<Q *>(sep.dst);
  unsigned recomputed=0;
  for (unsigned i=wave; i < count; i+=waves)
    {
      const unsigned long long entry=sep.queue[i];
      const uint32_t which=(uint32_t) (entry & 15ull);
      const size_t at=(size_t) (entry >> 4);
      const int y=(int) (at/(size_t) sep.columns),x=(int) (at-(size_t) y*sep.columns);
      for (int c=0; c < C; c++)
        if ((which >> c) & 1u)
          {
            const Q settled=conv2d_reference_sample<Q,C,BLEND>(src,sep.columns,sep.rows,x,y,c,sep.values,
              sep.kw,sep.kh,sep.shiftx,sep.shifty,lane);
            if (lane == 0)
              dst[at*C+c]=settled;
            recomputed++;
          }
    }
  if ((sep.recomputed != nullptr) && (lane == 0) && (recomputed != 0))
    atomicAdd(sep.recomputed,(unsigned long long) recomputed);
}

template<typename Q,int C,bool BLEND,int R,int U>
static MhStatus launch_folded_row(const View &src,const SeparableArgs &sep,const Conv1DParams &p)
{
  constexpr int WAVES=4;
  const int K=p.ntaps;
  std::vector<double> host((size_t) K);
  for (int v=0; v < K; v++)
    host[(size_t) v]=p.taps[K-1-v];             // reversed: morphology.c:2919
  Temp taps;
  MH_TRY(upload_table(taps,src.device,src.stream,host.data(),host.size()*sizeof(double)));
  Conv1DArgs args={};
  args.src=sep.src;
  args.dst=sep.sums;
  args.columns=sep.columns;
  args.rows=sep.rows;
  args.ntaps=K;
  args.shift=K-1-p.origin;
  args.taps=taps.ptr;
  const int W=args.columns,H=args.rows;
  const int SEG=64*R,NS=63*R+R+K-1+U,slots=NS+NS/R+1;
  // a wave's LDS: its strip of samples, and room for the 16 (R+1) pixels of sums it stores at a time
  size_t wave_bytes=(size_t) slots*C*sizeof(Q);
  if (wave_bytes < (size_t) 16*(R+1)*32)
    wave_bytes=(size_t) 16*(R+1)*32;
  wave_bytes=(wave_bytes+15u) & ~(size_t) 15u;
  args.wave_bytes=(int) wave_bytes;
  const size_t lds=(size_t) WAVES*wave_bytes;
  if (lds > 160u*1024u)
    return fail(MH_UNSUPPORTED,"row kernel of %d taps needs %zu bytes of LDS",K,lds);
  const unsigned ntx=(unsigned) ((W+SEG-1)/SEG),nty=(unsigned) ((H+WAVES-1)/WAVES);
  const unsigned grid=((ntx*nty+7u)/8u)*8u;
  if (lds > 64u*1024u)
    MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&separable_row_sums_kernel<Q,C,BLEND,R,U,WAVES>),
      hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  ProfileScope prof("separable_row_sums",src.stream);
  hipLaunchKernelGGL((separable_row_sums_kernel<Q,C,BLEND,R,U,WAVES>),dim3(grid),dim3(64*WAVES),lds,
    src.stream,args,sep.bound);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

template<typename Q,int C,bool BLEND,int R,int U>
static MhStatus launch_folded_column(const View &src,const SeparableArgs &sep,const Conv1DParams &p)
{
  constexpr int WAVES=4;
  const int K=p.ntaps;
  std::vector<double> host((size_t) K);
  for (int v=0; v < K; v++)
    host[(size_t) v]=p.taps[K-1-v];
  Temp taps;
  MH_TRY(upload_table(taps,src.device,src.stream,host.data(),host.size()*sizeof(double)));
  Conv1DArgs args={};
  args.src=sep.sums;
  args.dst=sep.dst;
  args.columns=sep.columns;
  args.rows=sep.rows;
  args.ntaps=K;
  args.shift=K-1-p.origin;
  args.taps=taps.ptr;
  const int W=args.columns,H=args.rows;
  const unsigned ntx=(unsigned) ((W+63)/64),nty=(unsigned) ((H+R*WAVES-1)/(R*WAVES));
  const unsigned grid=((ntx*nty+7u)/8u)*8u;
  ProfileScope prof("separable_column_finish",src.stream);
  hipLaunchKernelGGL((separable_column_finish_kernel<Q,C,BLEND,R,U,WAVES>),dim3(grid),dim3(64*WAVES),0,
    src.stream,args,sep);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

template<typename Q,int C,bool BLEND>
static MhStatus launch_folded(const View &src,const SeparableArgs &sep,const Conv1DParams &horizontal,
  const Conv1DParams &vertical)
{
  // a lane owns R consecutive outputs of a K-tap pass; the triangular walk needs K >= R+1
  // (measured on 4096^2, tools/time_fold_variants.py: the column pass of a short kernel is faster with four
  // outputs a lane — 106 registers, four waves a SIMD — than with eight: 0.19 / 0.22 / 0.25 ms against
  // 0.24 / 0.28 / 0.28 for 7 / 13 / 19 taps; from 25 taps on the fewer re-read rows of eight win)
  const int r8_rows=(int) option_long("MAGICKHIP_FOLD_R8_ROW_MIN",9),r8_columns=(int) option_long("MAGICKHIP_FOLD_R8_COLUMN_MIN",25);
  // (samples in flight: four instead of eight where that buys a wave more per SIMD — the float
  // frame's row pass, 194 -> ~140 registers, and every column pass, whose samples are doubles: the
  // fp64 pipe sustains more with more waves, tools/ubench/fma_f64_rate.hip.  0x10 on 8192^2: column
  // 1.44 -> 1.28 ms, float row 1.24 -> 1.14.)
  if ((horizontal.ntaps >= (r8_rows < 9 ? 9 : r8_rows)) && QuantumOps<Q>::is_float)
    MH_TRY((launch_folded_row<Q,C,BLEND,8,4>(src,sep,horizontal)));
  else if (horizontal.ntaps >= (r8_rows < 9 ? 9 : r8_rows))
    MH_TRY((launch_folded_row<Q,C,BLEND,8,8>(src,sep,horizontal)));
  else if (horizontal.ntaps >= 5)
    MH_TRY((launch_folded_row<Q,C,BLEND,4,4>(src,sep,horizontal)));
  else
    MH_TRY((launch_folded_row<Q,C,BLEND,2,2>(src,sep,horizontal)));
  if (vertical.ntaps >= (r8_columns < 9 ? 9 : r8_columns))
    MH_TRY((launch_folded_column<Q,C,BLEND,8,4>(src,sep,vertical)));
  else if (vertical.ntaps >= 5)
    MH_TRY((launch_folded_column<Q,C,BLEND,4,4>(src,sep,vertical)));
  else
    MH_TRY((launch_folded_column<Q,C,BLEND,2,2>(src,sep,vertical)));
  {
    // (the queue's fill is on the device: a grid that covers a full queue at four samples per wave,
    // whose workgroups leave at once when there is nothing for them)
    unsigned blocks=(sep.queue_capacity+15u)/16u;
    blocks=blocks > 1024u ? 1024u : (blocks < 1u ? 1u : blocks);
    ProfileScope prof("separable_settle",src.stream);
    hipLaunchKernelGGL((separable_settle_kernel<Q,C,BLEND>),dim3(blocks),dim3(256),0,src.stream,sep);
  }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// src: the frame (its layout picks the instantiation); sep: src, dst, sums ([rows][columns][4]
// doubles of work space), bound and the finish step's arguments.  Both kernels need 3 taps or more.
MhStatus launch_separable_folded(const View &src,const SeparableArgs &sep,const Conv1DParams &horizontal,
  const Conv1DParams &vertical,bool blend)
{
  if ((horizontal.ntaps < 3) || (vertical.ntaps < 3))
    return fail(MH_BAD_ARGUMENT,"folded separable passes: %d x %d taps",horizontal.ntaps,vertical.ntaps);
#define MH_LAYOUT(QT) \
  switch (src.channels) \
  { \
    case 1: return launch_folded<QT,1,false>(src,sep,horizontal,vertical); \
    case 2: return blend ? launch_folded<QT,2,true>(src,sep,horizontal,vertical) : \
      launch_folded<QT,2,false>(src,sep,horizontal,vertical); \
    case 3: return launch_folded<QT,3,false>(src,sep,horizontal,vertical); \
    case 4: return blend ? launch_folded<QT,4,true>(src,sep,horizontal,vertical) : \
      launch_folded<QT,4,false>(src,sep,horizontal,vertical); \
    default: break; \
  }
  if (src.quantum != MH_QUANTUM_U16)
    { MH_LAYOUT(float) }
  else
    { MH_LAYOUT(uint16_t) }
#undef MH_LAYOUT
  return fail(MH_UNSUPPORTED,"%d channels",src.channels);
}

// ---------------------------------------------------------------- launcher
template<typename Q,int C,bool BLEND,class A,int R>
static MhStatus launch_one(const View &src,const View &dst,bool vertical,
  const Conv1DParams &p,const Roles &roles,unsigned long long *changed)
{
  typedef typename A::T T;
  constexpr int WAVES=4;
  const int K=p.ntaps;
  // taps reversed: taps[v] = values[K-1-v]  (k starts at the last value and
  // walks backwards, morphology.c:2746 / :2919)
  std::vector<T> host((size_t) K);
  for (int v=0; v < K; v++)
    host[(size_t) v]=(T) p.taps[K-1-v];
  Temp taps;
  MH_TRY(upload_table(taps,src.device,src.stream,host.data(),host.size()*sizeof(T)));

  Conv1DArgs args;
  args.unsharp_src=nullptr;
  args.unsharp_gain=0.0;
  args.unsharp_threshold=0.0;
  args.src=src.pixels;
  args.dst=dst.pixels;
  args.columns=(int) src.columns;
  args.rows=(int) src.rows;
  args.ntaps=K;
  args.shift=K-1-p.origin;       // offset.x / offset.y, morphology.c:2623-2624
  args.bias=p.bias;
  args.copy_mask=roles.copy_mask;
  args.taps=taps.ptr;
  args.changed=changed;
  args.nblocks=0;
  args.wave_bytes=0;
  args.only_if=p.only_if;
  args.only_if_token=p.only_if_token;

  const int W=args.columns,H=args.rows;
  if (vertical)
    {
      unsigned ntx=(unsigned) ((W+63)/64),nty=(unsigned) ((H+R*WAVES-1)/(R*WAVES));
      unsigned total=ntx*nty;
      unsigned grid=((total+7u)/8u)*8u;
      ProfileScope prof("conv_column",src.stream);
      hipLaunchKernelGGL((conv_column_kernel<Q,C,BLEND,A,R,WAVES>),dim3(grid),dim3(64*WAVES),0,
        src.stream,args);
    }
  else
    {
      const int SEG=64*R,NS=SEG+K-1,slots=NS+NS/R+1;
      size_t lds=(size_t) WAVES*slots*C*sizeof(Q);
      if (lds > 160u*1024u)
        return fail(MH_UNSUPPORTED,"row kernel of %d taps needs %zu bytes of LDS",K,lds);
      unsigned ntx=(unsigned) ((W+SEG-1)/SEG),nty=(unsigned) ((H+WAVES-1)/WAVES);
      unsigned total=ntx*nty;
      unsigned grid=((total+7u)/8u)*8u;
      if (lds > 64u*1024u)
        MH_HIP(hipFuncSetAttribute(
          reinterpret_cast<const void *>(&conv_row_kernel<Q,C,BLEND,A,R,WAVES>),
          hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
      ProfileScope prof("conv_row",src.stream);
      hipLaunchKernelGGL((conv_row_kernel<Q,C,BLEND,A,R,WAVES>),dim3(grid),dim3(64*WAVES),lds,
        src.stream,args);
    }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

template<typename Q,class A,int R>
static MhStatus dispatch_channels(const View &src,const View &dst,bool vertical,
  const Conv1DParams &p,const Roles &roles,unsigned long long *changed)
{
  const bool blend=roles.blend && (roles.alpha == src.channels-1);
  switch (src.channels)
  {
    case 1: return launch_one<Q,1,false,A,R>(src,dst,vertical,p,roles,changed);
    case 2:
      if (blend) return launch_one<Q,2,true,A,R>(src,dst,vertical,p,roles,changed);
      return launch_one<Q,2,false,A,R>(src,dst,vertical,p,roles,changed);
    case 3: return launch_one<Q,3,false,A,R>(src,dst,vertical,p,roles,changed);
    case 4:
      if (blend) return launch_one<Q,4,true,A,R>(src,dst,vertical,p,roles,changed);
      return launch_one<Q,4,false,A,R>(src,dst,vertical,p,roles,changed);
    default: break;
  }
  return fail(MH_UNSUPPORTED,"%d channels",src.channels);
}

// FAST row pass of an alpha-weighted layout on the f32 vector kernels: the Quantum-rounded alpha
// it writes becomes a WEIGHT in a following column pass (BlurImage), where one level more or
// less in a small alpha moves the colour by many levels.  This audit recomputes every alpha
// result below kSmallAlpha (+1) levels as the reference does — fp64, its operation order
// (morphology.c:2941-2951), ClampToQuantum — so the intermediate alpha is bit-identical to the
// reference's wherever a level matters (the matrix-core kernels do the same inside their
// epilogue, mfma_common.hpp).  Frames without small alpha only pay one read of the alpha
// samples.
template<int C>
__global__ __launch_bounds__(256)
void conv_row_alpha_audit_kernel(const uint16_t *src,uint16_t *dst,int columns,int rows,
  const double *taps64,int K,int shift)
{
  const size_t total=(size_t) columns*(size_t) rows;
  for (size_t at=(size_t) blockIdx.x*256u+threadIdx.x; at < total; at+=(size_t) gridDim.x*256u)
    {
      if (dst[at*C+(C-1)] > 8192u)
        continue;
      const size_t y=at/(size_t) columns;
      const int x=(int) (at-y*(size_t) columns);
      const uint16_t *line=src+y*(size_t) columns*C;
      double sum=0.0;
      for (int v=0; v < K; v++)
        {
          int xx=x-shift+v;
          xx=xx < 0 ? 0 : (xx > columns-1 ? columns-1 : xx);
          sum=sum+taps64[v]*(double) line[(size_t) xx*C+(C-1)];
        }
      dst[at*C+(C-1)]=QuantumOps<uint16_t>::clamp(sum);
    }
}

static MhStatus launch_row_alpha_audit(const View &src,const View &dst,const Conv1DParams &params)
{
  const int K=params.ntaps;
  std::vector<double> host((size_t) K);
  for (int v=0; v < K; v++)
    host[(size_t) v]=params.taps[K-1-v];             // reversed walk, morphology.c:2746
  Temp taps;
  MH_TRY(upload_table(taps,src.device,src.stream,host.data(),host.size()*sizeof(double)));
  const size_t total=src.columns*src.rows;
  const unsigned blocks=(unsigned) ((total+255)/256 < 8192 ? (total+255)/256 : 8192);
  ProfileScope prof("conv_row_alpha_audit",src.stream);
  if (src.channels == 4)
    hipLaunchKernelGGL((conv_row_alpha_audit_kernel<4>),dim3(blocks),dim3(256),0,src.stream,
      static_cast<const uint16_t *>(src.pixels),static_cast<uint16_t *>(dst.pixels),(int) src.columns,
      (int) src.rows,taps.as<double>(),K,K-1-params.origin);
  else
    hipLaunchKernelGGL((conv_row_alpha_audit_kernel<2>),dim3(blocks),dim3(256),0,src.stream,
      static_cast<const uint16_t *>(src.pixels),static_cast<uint16_t *>(dst.pixels),(int) src.columns,
      (int) src.rows,taps.as<double>(),K,K-1-params.origin);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

MhStatus launch_conv1d_column_unsharp(const View &rows,const View &dst,const View &original,
  const Conv1DParams &params,const Roles &roles,MhPrecision prec,double gain,double threshold,
  bool *handled)
{
  *handled=false;
  const bool is_float=rows.quantum != MH_QUANTUM_U16;
  if ((!is_float && (prec != MH_PRECISION_EXACT)) || (params.ntaps < 16) || (params.bias != 0.0) ||
      (original.columns != rows.columns) || (original.rows != rows.rows) ||
      (original.channels != rows.channels) || (original.quantum != rows.quantum) ||
      (option("MAGICKHIP_NO_TRI") != nullptr) || (option("MAGICKHIP_NO_TIE64") != nullptr) ||
      (option("MAGICKHIP_NO_FUSED_UNSHARP") != nullptr))
    return MH_OK;
  double total=0.0;
  for (int v=0; v < params.ntaps; v++)
    {
      if (!(params.taps[v] >= 0.0))
        return MH_OK;
      total+=params.taps[v];
    }
  if (is_float && !(total <= 1.0+1.0e-9))
    return MH_OK;
  Conv1DParams with=params;
  with.unsharp_source=original.pixels;
  with.unsharp_gain=gain;
  with.unsharp_threshold=65535.0*threshold;    // QuantumRange*threshold, effect.c:4300
  *handled=true;
  if (is_float)
    return dispatch_tri<Tie64,8,8,float>(rows,dst,true,with,roles,nullptr);
  return dispatch_tri<Tie64,8,8>(rows,dst,true,with,roles,nullptr);
}

// One 1-D pass over [rows][columns][4] DOUBLES (un-normalised sums in, sums out): fused
// multiply-adds, nothing rounded to a Quantum — the passes of convolve_separable.hip.  The Views'
// `pixels` point at doubles; their quantum field is not looked at.
MhStatus launch_conv1d_sums64(const View &src,const View &dst,bool vertical,const Conv1DParams &params)
{
  Roles plain;
  plain.update_mask=0xfu;
  if ((params.ntaps >= 16) && (option("MAGICKHIP_NO_TRI") == nullptr))
    return launch_tri<double,4,false,Fma64,8,8>(src,dst,vertical,params,plain,nullptr);
  return launch_one<double,4,false,Fma64,8>(src,dst,vertical,params,plain,nullptr);
}

MhStatus launch_conv1d_sums(const View &src,const View &dst,bool vertical,const Conv1DParams &params,
  bool blend,bool *handled)
{
  *handled=false;
  if ((params.ntaps < 2) || (params.origin < 0) || (params.origin >= params.ntaps) ||
      (option("MAGICKHIP_NO_MFMA") != nullptr))
    return MH_OK;
  const int K=params.ntaps;
  std::vector<float> host((size_t) K);
  for (int v=0; v < K; v++)
    host[(size_t) v]=(float) params.taps[K-1-v];     // reversed walk, morphology.c:2746
  Temp taps;
  MH_TRY(upload_table(taps,src.device,src.stream,host.data(),host.size()*sizeof(float)));
  return launch_conv1d_mfma(src,dst,vertical,taps.as<float>(),K,K-1-params.origin,blend,
    vertical ? MFMA_FROM_SUMS : MFMA_TO_SUMS,handled);
}

MhStatus launch_conv1d_unsharp(const View &rows,const View &dst,const View &original,
  const Conv1DParams &params,bool blend,double gain,double threshold,bool *handled)
{
  *handled=false;
  if ((params.ntaps < 2) || (params.origin < 0) || (params.origin >= params.ntaps) ||
      (params.bias != 0.0) || (option("MAGICKHIP_NO_MFMA") != nullptr) ||
      (option("MAGICKHIP_NO_FUSED_UNSHARP") != nullptr))
    return MH_OK;
  const int K=params.ntaps;
  std::vector<float> host((size_t) K);
  for (int v=0; v < K; v++)
    host[(size_t) v]=(float) params.taps[K-1-v];     // reversed walk, morphology.c:2746
  if (blend && !f16_taps_resolved(params.taps,K))
    return MH_OK;                                // (see launch_conv1d)
  Temp taps;
  MH_TRY(upload_table(taps,rows.device,rows.stream,host.data(),host.size()*sizeof(float)));
  return launch_conv1d_mfma(rows,dst,true,taps.as<float>(),K,K-1-params.origin,blend,MFMA_UNSHARP,
    handled,&original,gain,threshold,nullptr,f16_tap_scale(params.taps,K));
}

MhStatus launch_conv1d(const View &src,const View &dst,bool vertical,
  const Conv1DParams &params,const Roles &roles,MhPrecision prec,
  unsigned long long *changed)
{
  if ((src.columns != dst.columns) || (src.rows != dst.rows) ||
      (src.channels != dst.channels) || (src.quantum != dst.quantum))
    return fail(MH_BAD_ARGUMENT,"conv1d: source/destination geometry mismatch");
  if (roles.blend && (roles.alpha != src.channels-1))
    return fail(MH_UNSUPPORTED,"alpha channel must be the last channel");
  if ((params.ntaps < 1) || (params.origin < 0) || (params.origin >= params.ntaps))
    return fail(MH_BAD_ARGUMENT,"conv1d: bad kernel geometry");
  // Alpha-weighted channels with taps of both signs: sum(k*alpha) can come out near zero, where
  // the reference's PerceptibleReciprocal clamp decides the pixel and no reduced-precision sum
  // stays within a level of it.  FAST keeps to the fp64 kernels there.
  if ((prec == MH_PRECISION_FAST) && roles.blend)
    for (int v=0; v < params.ntaps; v++)
      if (params.taps[v] < 0.0)
        {
          prec=MH_PRECISION_EXACT;
          break;
        }
  // ... and taps of both signs with a large gain on any layout: the float sums are good to 2^-21 of
  // sum|k|*65535, which cancellation does not shrink
  if (prec == MH_PRECISION_FAST)
    {
      double magnitude=0.0;
      bool negative=false;
      for (int v=0; v < params.ntaps; v++)
        {
          magnitude+=std::fabs(params.taps[v]);
          negative=negative || (params.taps[v] < 0.0);
        }
      if (negative && !(magnitude <= 8.0))
        prec=MH_PRECISION_EXACT;
    }
  // ... and alpha-weighted channels under a kernel whose outer taps are tiny beside the rest (BlurImage
  // with a radius far beyond its sigma): where such a tap is the only one that meets an opaque sample — a
  // sprite on a transparent ground — it IS the result (sum(k*alpha*p)/sum(k*alpha), morphology.c:2968-2977),
  // and neither the f16 terms nor the f32 sums resolve it.  The fp64 kernels take those.
  if ((prec == MH_PRECISION_FAST) && roles.blend && !f16_taps_resolved(params.taps,params.ntaps))
    prec=MH_PRECISION_EXACT;
  if (src.quantum == MH_QUANTUM_U16)
    {
      if (prec == MH_PRECISION_FAST)
        {
          // RGBA with alpha-weighted colour channels (BlurImage's case), or 3/4 independent
          // channels (RGB, CMYK): the banded-matrix formulation on the f16 matrix cores,
          // convolve_mfma.hip
          if ((roles.blend ? (src.channels == 4) && (roles.alpha == 3) :
               (src.channels == 3) || (src.channels == 4)) && (roles.copy_mask == 0) &&
              (params.bias == 0.0) && (changed == nullptr) && (option("MAGICKHIP_NO_MFMA") == nullptr))
            {
              const int K=params.ntaps;
              // one table: K doubles (the exact recomputation of ambiguous small alpha levels),
              // then K floats; both in the reversed walk of morphology.c:2746
              std::vector<double> host((size_t) K+((size_t) K+1)/2);
              float *host_floats=reinterpret_cast<float *>(host.data()+K);
              for (int v=0; v < K; v++)
                {
                  host[(size_t) v]=params.taps[K-1-v];
                  host_floats[v]=(float) params.taps[K-1-v];
                }
              const void *taps=nullptr;
              std::shared_ptr<void> keep;       // until the launch below is enqueued
              MH_TRY(shared_table(src.device,src.stream,host.data(),host.size()*sizeof(double),&taps,&keep));
              const double *taps64=static_cast<const double *>(taps);
              bool handled=false;
              MH_TRY(launch_conv1d_mfma(src,dst,vertical,reinterpret_cast<const float *>(taps64+K),K,
                K-1-params.origin,roles.blend,MFMA_Q16,&handled,nullptr,0.0,0.0,taps64,
                f16_tap_scale(params.taps,K)));
              if (handled)
                return MH_OK;
            }
          // long kernels: the triangular kernels with 32 outputs per lane; short ones: blocked
          // K >= R+1: the ramp-free triangular kernels (measured +7.5 % at K=79 on MI355X;
          // R=24/32 variants were slower: 240 VGPRs leave two waves per SIMD)
          if ((params.ntaps >= 24) && (option("MAGICKHIP_NO_TRI") == nullptr))
            MH_TRY((dispatch_tri<Fast32,16,4>(src,dst,vertical,params,roles,changed)));
          else
            MH_TRY((dispatch_blocked<Fast32,16,4>(src,dst,vertical,params,roles,changed)));
          // alpha-weighted layouts (gray + alpha, RGBA): small alpha results exact, see above
          if (!vertical && roles.blend && (params.bias == 0.0) && ((roles.copy_mask >> roles.alpha) & 1u) == 0 &&
              ((src.channels == 2) || (src.channels == 4)))
            MH_TRY(launch_row_alpha_audit(src,dst,params));
          return MH_OK;
        }
      if ((params.ntaps >= 16) && (option("MAGICKHIP_NO_TRI") == nullptr))
        {
          // fused sums + reference-order recomputation of the results they cannot decide: the
          // same bits at less than half the fp64 work (device_common.hpp, Tie64).  Positive taps
          // only (a weighted sum of both signs can cancel: the error bound is relative to the
          // sum of magnitudes, not to the result), no bias, no change count.
          bool positive=true;
          for (int v=0; v < params.ntaps; v++)
            positive=positive && (params.taps[v] >= 0.0);
          if (positive && (params.bias == 0.0) && (changed == nullptr) && (option("MAGICKHIP_NO_TIE64") == nullptr))
            return dispatch_tri<Tie64,8,8>(src,dst,vertical,params,roles,changed);
          return dispatch_tri<Exact64,8,8>(src,dst,vertical,params,roles,changed);
        }
      return dispatch_blocked<Exact64,8,8>(src,dst,vertical,params,roles,changed);
    }
  // float Quantum always accumulates in double: an FP32 sum cannot stay
  // within 1 ULP of a float result.  Long normalised positive kernels (BlurImage's) take the
  // fused sums with the float-rounding tie check of Accum::finish(): the same bits as the
  // reference's order at 4 fused multiply-adds per tap instead of 11 separately rounded operations.
  if ((params.ntaps >= 16) && (params.bias == 0.0) && (changed == nullptr) &&
      (option("MAGICKHIP_NO_TRI") == nullptr) && (option("MAGICKHIP_NO_TIE64") == nullptr))
    {
      bool positive=true;
      double total=0.0;
      for (int v=0; v < params.ntaps; v++)
        {
          positive=positive && (params.taps[v] >= 0.0);
          total+=params.taps[v];
        }
      if (positive && (total <= 1.0+1.0e-9))
        {
          // (the row pass with four samples in flight instead of eight: 138 registers instead of 194,
          // three waves a SIMD — the fp64 pipe sustains more with more waves, tools/ubench/
          // fma_f64_rate.hip — 1.25 -> 1.15 ms per 8192^2 frame; the column pass keeps its 182
          // registers either way and is slower with four)
          if (!vertical)
            return dispatch_tri<Tie64,8,4,float>(src,dst,vertical,params,roles,changed);
          return dispatch_tri<Tie64,8,8,float>(src,dst,vertical,params,roles,changed);
        }
    }
  return dispatch_channels<float,Exact64,8>(src,dst,vertical,params,roles,changed);
}

} // namespace mh
