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

namespace mh {

// ---------------------------------------------------------------- Accum
template<typename Q,int C,bool BLEND,class A,int R>
struct Accum
{
  typedef typename A::T T;
  T s[R][C];
  T g[R];

  struct In { T p[C]; T a; };

  __device__ __forceinline__ void init(T bias)
  {
#pragma unroll
    for (int r=0; r < R; r++)
      {
#pragma unroll
        for (int c=0; c < C; c++)
          s[r][c]=bias;
        g[r]=(T) 0;
      }
  }

  static __device__ __forceinline__ In prepare(const Q (&q)[C])
  {
    In in;
#pragma unroll
    for (int c=0; c < C; c++)
      in.p[c]=(T) q[c];
    in.a=(T) 0;
    if constexpr (BLEND)
      {
        // alpha=QuantumScale*GetPixelAlpha(): morphology.c:2766, :2965
        in.a=A::mul((T) kQS,in.p[C-1]);
        if constexpr (A::premultiply)
          {
#pragma unroll
            for (int c=0; c < C-1; c++)
              in.p[c]=A::mul(in.a,in.p[c]);
          }
      }
    return in;
  }

  __device__ __forceinline__ void tap(int r,T kv,const In &in)
  {
    if constexpr (BLEND)
      {
        if constexpr (A::premultiply)
          {
#pragma unroll
            for (int c=0; c < C; c++)
              s[r][c]=A::mac(s[r][c],kv,in.p[c]);
            g[r]=A::mac(g[r],kv,in.a);
          }
        else
          {
            // pixel+=alpha*(*k)*pixels[i]; gamma+=alpha*(*k);  morphology.c:2767-2768
            T w=A::mul(in.a,kv);
#pragma unroll
            for (int c=0; c < C-1; c++)
              s[r][c]=A::add(s[r][c],A::mul(w,in.p[c]));
            g[r]=A::add(g[r],w);
            s[r][C-1]=A::mac(s[r][C-1],kv,in.p[C-1]);      // alpha channel: no weighting
          }
      }
    else
      {
#pragma unroll
        for (int c=0; c < C; c++)
          s[r][c]=A::mac(s[r][c],kv,in.p[c]);              // morphology.c:2750
      }
  }

  // returns the number of channels that count as "changed" (morphology.c:2772, :3199)
  __device__ __forceinline__ unsigned finish(int r,const Q (&center)[C],uint32_t copy_mask,
    Q (&out)[C]) const
  {
    unsigned changed=0;
#pragma unroll
    for (int c=0; c < C; c++)
      {
        if ((copy_mask >> c) & 1u)
          {
            out[c]=center[c];
            continue;
          }
        double pixel=(double) s[r][c];
        if (fabs(pixel-(double) center[c]) >= kEps)
          changed++;
        if (BLEND && (c != C-1))
          {
            if constexpr (A::premultiply)
              pixel=(double) (s[r][c]*(T) perceptible_reciprocal((double) g[r]));
            else
              pixel=perceptible_reciprocal((double) g[r])*pixel;
          }
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
};

// --------------------------------------------------------------- column pass
template<typename Q,int C,bool BLEND,class A,int R,int WAVES>
__global__ __launch_bounds__(64*WAVES)
void conv_column_kernel(Conv1DArgs args)
{
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
          unsigned ch=acc.finish(r,center,args.copy_mask,out);
          if (x < W)
            {
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
          changed+=acc.finish(r,center,args.copy_mask,out);
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
  if (src.quantum == MH_QUANTUM_U16)
    {
      if (prec == MH_PRECISION_FAST)
        return dispatch_channels<uint16_t,Fast32,16>(src,dst,vertical,params,roles,changed);
      return dispatch_channels<uint16_t,Exact64,8>(src,dst,vertical,params,roles,changed);
    }
  // float Quantum always accumulates in double: an FP32 sum cannot stay
  // within 1 ULP of a float result.
  return dispatch_channels<float,Exact64,8>(src,dst,vertical,params,roles,changed);
}

} // namespace mh
