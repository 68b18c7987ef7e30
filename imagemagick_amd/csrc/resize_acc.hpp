// The per-output accumulator of HorizontalFilter / VerticalFilter (MagickCore/resize.c:3494-3530,
// :3709-3745) under an arithmetic policy; shared by resize.hip and resize_mfma.hip.
#pragma once

#include "device_common.hpp"
#include "tie_watch.hpp"
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

// ---- what the one-launch FAST kernels (resize_stream.hip, resize_mfma.hip) need to know about a
// pixel of the INTERMEDIATE.  Their fused sums differ from the reference's separately rounded ones
// by ~1e-10 level — but the intermediate is ROUNDED, and where the exact value sits on a rounding
// boundary the last bits decide the level.  That is not a curiosity: polynomial filters at
// rational positions (Triangle, Box, Catrom ...) over small integers put whole families of sums
// exactly on x.5, and an intermediate ALPHA one level off moves the second filter's alpha-weighted
// colours by thousands of levels where alpha is a few levels (tests/stress_parity.py found it).
// A value closer to a boundary than the two summation orders can differ is reported, and the few
// rows that hold it are recomputed in the reference's own operation order (resize_redo_rect).
//
// How far the orders can differ: a sum of K <= 8 terms, fused against the reference's 2K+3
// separately rounded operations: 6e-15 * sum|terms|.  A plain sum of Q16 levels: 8e-10 level.  An
// alpha-weighted colour is a quotient S_c/S_a: 65535 * 6e-15 * A with A = sum|w*alpha| / |S_a|,
// and sum|w*alpha| <= 1.6 * 65535 for every filter of resize.c: 4.1e-5 / |S_a| level — the window
// widens with the reciprocal of the alpha sum, which the finish has at hand.  (A fixed window with
// a branch for small alpha sums cost more: on a frame of random alpha every other wave took the branch.)
template<typename Q> struct TieWatch : TieWatchBits<QuantumOps<Q>::is_float> {};

// an alpha sum of the SECOND filter below this many levels: the quotient's error (4.1e-5 / |S_a|
// level) is no longer a small fraction of a level.  Where PerceptibleReciprocal's clamp acts (either
// filter) the sums are multiplied by QuantumScale/MagickEpsilon = 1.5e7 instead: what cancelled under
// the window to (nearly) nothing is noise of thousands of levels, in the reference's order or in any
// other — unless every sum is an exact zero (a window of transparent pixels), the pixel is reported.
// (Q16: 4.1e-5 / 1e-3 = 0.04 level.  A float result's contract is relative: 6e-15 * 1.6 * 65535 / |S_a| has to
// stay a fraction of 2^-24 — an alpha sum of at least 0.05: the ringing of the first filter beside an
// opaque rectangle leaves intermediate alphas of 1e-3 on a transparent ground.  What no window can
// vouch for in a float frame is a PLAIN sum that is itself the residue of a cancellation — an alpha of
// -4e-20 out of terms of 1e-8: DESIGN.md section 2 has the qualifier.)
template<typename Q> struct OutputAlphaLimit { static constexpr double value=QuantumOps<Q>::is_float ? 0.05 : 1.0e-3; };
static __device__ __forceinline__ bool clamped_sums_count(const double (&s)[4])
{
  return (s[0] != 0.0) || (s[1] != 0.0) || (s[2] != 0.0) || (s[3] != 0.0);    // (a NaN counts)
}
// (Sums of rounding zeros — a sinc at a whole number, 1e-13, times any alpha: a 3x enlargement has one
// output in three whose window is (0 .. 0, 1, 0 .. 0) — over a transparent centre pixel between opaque ones
// are exactly this case: the two neighbours' weights are w and -w, the alpha sum is 0 or 1e-29, its SIGN
// picks +-1/MagickEpsilon, and the reference's order is the only arbiter.  Suppressing the report for
// such windows was tried and failed the binary-alpha runs.  Round 6 withdraws it for the LAST filter of a Q16 frame
// only where the window's own terms are all tiny AND the alpha sum is far above its rounding — resize_stream.hip,
// finish_fast: the w / -w window above fails the second condition and stays reported.)


// The reference's two filters over one rectangle of the output (VerticalFilter, then
// HorizontalFilter: the one-launch kernels' order), tap by tap in the reference's own operation
// order (Exact64; resize.c:3494-3530, :3709-3745), every sample multiplied only inside its
// output's window.  The whole workgroup calls it; `scratch` is workgroup-shared memory of
// `scratch_bytes` for the rectangle's intermediate (blocks of rows x columns that fit).  Ends
// behind a barrier.
struct RedoTables
{
  const int *vstart,*vcount,*hstart,*hcount;
  const double *vweight,*hweight;              // [tap][out]
  int src_columns,dst_columns,dst_rows;
};

// MAXT > 0: no contribution list is longer (the caller's plan has checked) — a thread then has the
// loads of all taps of GROUP samples in flight together: the rectangles are small, a redo is a
// chain of memory round trips, and the fewer of them the sooner the workgroup is gone.
template<typename Q,bool BLEND,int MAXT>
static __device__ __noinline__ void resize_redo_rect(const RedoTables &t,const Q *src,Q *dst,int x0,int x1,int y0,int y1,
  unsigned char *scratch,int scratch_bytes)
{
  constexpr int PXB=4*(int) sizeof(Q);
  constexpr int XC=256;                         // output columns per block
  constexpr int TB=MAXT > 0 ? MAXT : 4;         // taps whose loads are in flight together
  constexpr int GROUP=MAXT > 0 ? 2 : 1;         // samples of a thread in flight together
  const int tid=(int) threadIdx.x,nthreads=(int) blockDim.x;
  Q *inter=reinterpret_cast<Q *>(scratch);
  // one filter over `count` samples: sample i reads taps start[i] .. of `from(i)` with the weights
  // weight[tap*pitch+index(i)] and hands its pixel to put(i, pixel)
  auto filter=[&](int count,auto geometry,auto put)
  {
    for (int i0=tid; i0 < count; i0+=GROUP*nthreads)
      {
        const Q *from[GROUP];
        const double *weights[GROUP];
        size_t step[GROUP],pitch[GROUP];
        int taps[GROUP];
#pragma unroll
        for (int g=0; g < GROUP; g++)
          {
            const int i=i0+g*nthreads < count ? i0+g*nthreads : i0;
            geometry(i,from[g],step[g],weights[g],pitch[g],taps[g]);
          }
        int longest=0;
#pragma unroll
        for (int g=0; g < GROUP; g++)
          longest=taps[g] > longest ? taps[g] : longest;
        ResizeAcc<Q,4,BLEND,Exact64> acc[GROUP];
#pragma unroll
        for (int g=0; g < GROUP; g++)
          acc[g].init();
        for (int k0=0; k0 < longest; k0+=TB)
          {
            Q p[GROUP][TB][4];
            double w[GROUP][TB];
#pragma unroll
            for (int g=0; g < GROUP; g++)
#pragma unroll
              for (int k=0; k < TB; k++)
                {
                  const int kk=k0+k < taps[g] ? k0+k : taps[g]-1;
                  load_pixel<Q,4>(from[g]+(size_t) kk*step[g],p[g][k]);
                  w[g][k]=weights[g][(size_t) kk*pitch[g]];
                }
#pragma unroll
            for (int g=0; g < GROUP; g++)
#pragma unroll
              for (int k=0; k < TB; k++)
                if (k0+k < taps[g])
                  acc[g].tap(w[g][k],w[g][k]*kQS,p[g][k]);
          }
#pragma unroll
        for (int g=0; g < GROUP; g++)
          if (i0+g*nthreads < count)
            {
              Q copy[4]={(Q) 0,(Q) 0,(Q) 0,(Q) 0},q[4];
              acc[g].finish(copy,0u,q);
              put(i0+g*nthreads,q);
            }
      }
  };
  for (int xa=x0; xa < x1; )
    {
      int xb=xa+XC < x1 ? xa+XC : x1;
      const int cs=t.hstart[xa];
      // (contribution lists move right with the output: the last output's window ends last)
      while ((xb > xa+1) && ((t.hstart[xb-1]+t.hcount[xb-1]-cs)*PXB > scratch_bytes))
        xb--;
      const int span=t.hstart[xb-1]+t.hcount[xb-1]-cs;
      int rb=scratch_bytes/(span*PXB);
      rb=rb < 1 ? 1 : rb;
      for (int ya=y0; ya < y1; ya+=rb)
        {
          const int yb=ya+rb < y1 ? ya+rb : y1;
          __syncthreads();                      // the previous block's readers
          // VerticalFilter: the block's rows of the intermediate, columns cs .. cs+span-1
          filter((yb-ya)*span,
            [&](int i,const Q *&from,size_t &step,const double *&weights,size_t &pitch,int &taps)
            {
              const int r=i/span,y=ya+r;
              from=src+((size_t) t.vstart[y]*(size_t) t.src_columns+(size_t) (cs+(i-r*span)))*4;
              step=(size_t) t.src_columns*4;
              weights=t.vweight+(size_t) y;
              pitch=(size_t) t.dst_rows;
              taps=t.vcount[y];
            },
            [&](int i,const Q (&q)[4]) { store_pixel<Q,4>(inter+(size_t) i*4,q); });
          __syncthreads();
          // HorizontalFilter out of it
          const int w=xb-xa;
          filter((yb-ya)*w,
            [&](int i,const Q *&from,size_t &step,const double *&weights,size_t &pitch,int &taps)
            {
              const int r=i/w,x=xa+(i-r*w);
              from=inter+((size_t) r*(size_t) span+(size_t) (t.hstart[x]-cs))*4;
              step=4;
              weights=t.hweight+(size_t) x;
              pitch=(size_t) t.dst_columns;
              taps=t.hcount[x];
            },
            [&](int i,const Q (&q)[4])
            {
              const int r=i/w;
              store_pixel<Q,4>(dst+((size_t) (ya+r)*(size_t) t.dst_columns+(size_t) (xa+(i-r*w)))*4,q);
            });
        }
      xa=xb;
    }
  __syncthreads();
}

} // namespace mh
