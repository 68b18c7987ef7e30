// 2-D ConvolveMorphology with a kernel that is an outer product (GaussianBlurImage's Gaussian:RxS,
// Square, Rectangle, any column x row product), BIT-IDENTICAL to the reference at the cost of two
// 1-D passes — for Q16 and for float Quantum, every layout.
//
// The reference walks the whole w x h window per sample (MagickCore/morphology.c:2919-2979:
// pixel += alpha*k*p, gamma += alpha*k over the reflected kernel, then gamma = 1/gamma, ClampTo-
// Quantum(gamma*pixel), :3192-3194): 361 cells for GaussianBlur 0x3, 6241 for 0x10 — 22 ms and
// 260 ms per 8192^2 frame in the generic kernel.  Here
//   1. separable_premultiply_kernel   P = (alpha*p .., alpha) as four DOUBLES per pixel (a product
//                                     of two Quantum values is exact in fp64), and the largest
//                                     |P_c| of the frame per channel (float Quantum);
//   2. two passes of the fp64 1-D kernels of convolve.hip over P (launch_conv1d_sums64): fused
//      multiply-adds, nothing rounded to a Quantum in between — S = sum_v col[v] sum_u row[u] P;
//   3. separable_finish_kernel        value = S_c/S_alpha (or S_c), rounded to the Quantum — unless
//      it lies closer to a rounding boundary than the two evaluations can differ: the kernel's cells
//      are the outer product only to 1e-13 of the largest cell (rank_one_factors, operators.cpp), the
//      summation orders differ by a few 1e-16 per term.  Those samples (a few per million) are
//      recomputed by conv2d_reference_sample in the reference's own order from the source frame.
// Edge policy: the reference clamps the window per axis (cache.c:2663-2679), so does each pass.
#include "mh_internal.hpp"
#include "device_common.hpp"
#include "tie_check.hpp"
#include "separable_args.hpp"
#include <cmath>
#include <cstdlib>
#include <vector>

namespace mh {

template<typename Q,int C,bool BLEND>
__global__ __launch_bounds__(256)
void separable_premultiply_kernel(SeparableArgs a)
{
  const Q *src=static_cast<const Q *>(a.src);
  const size_t n=(size_t) a.columns*a.rows;
  double most[4]={0.0,0.0,0.0,0.0};
  for (size_t i=(size_t) blockIdx.x*256u+threadIdx.x; i < n; i+=(size_t) gridDim.x*256u)
    {
      Q q[C];
      load_pixel<Q,C>(src+i*C,q);
      double p[4]={0.0,0.0,0.0,0.0};
      const double alpha=BLEND ? (double) q[C-1] : 1.0;
#pragma unroll
      for (int c=0; c < C; c++)
        {
          p[c]=(BLEND && (c != C-1)) ? alpha*(double) q[c] : (double) q[c];
          most[c]=__builtin_fmax(most[c],__builtin_fabs(p[c]));      // (NaN: skipped)
        }
      double *out=a.sums+i*4;
      reinterpret_cast<double2 *>(out)[0]=make_double2(p[0],p[1]);
      reinterpret_cast<double2 *>(out)[1]=make_double2(p[2],p[3]);
    }
  if constexpr (QuantumOps<Q>::is_float)
    {
      // one atomic per workgroup and channel, and only when it would raise the bound (8192 x 4
      // atomics on four words cost more than the pass itself)
      __shared__ double wave_most[4][4];
#pragma unroll
      for (int c=0; c < C; c++)
        {
          double m=most[c];
          for (int off=32; off > 0; off>>=1)
            m=__builtin_fmax(m,__shfl_xor(m,off,64));
          if ((threadIdx.x & 63) == 0)
            wave_most[threadIdx.x >> 6][c]=m;
        }
      __syncthreads();
      if (threadIdx.x < (unsigned) C)
        {
          const int c=(int) threadIdx.x;
          const double m=__builtin_fmax(__builtin_fmax(wave_most[0][c],wave_most[1][c]),
            __builtin_fmax(wave_most[2][c],wave_most[3][c]));
          // non-negative doubles order as their bit patterns
          const unsigned long long bits=(unsigned long long) __double_as_longlong(m);
          unsigned long long *word=reinterpret_cast<unsigned long long *>(a.bound+c);
          if (bits > __hip_atomic_load(word,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT))
            atomicMax(word,bits);
        }
    }
}

template<typename Q,int C,bool BLEND>
__global__ __launch_bounds__(256)
void separable_finish_kernel(SeparableArgs a)
{
  const Q *src=static_cast<const Q *>(a.src);
  Q *dst=static_cast<Q *>(a.dst);
  const size_t n=(size_t) a.columns*a.rows;
  unsigned recomputed=0;
  double error[4]={0.0,0.0,0.0,0.0};
#pragma unroll
  for (int c=0; c < C; c++)
    error[c]=a.error_unit*(QuantumOps<Q>::is_float ? a.bound[c] : a.fixed_bound[c]);
  // (the loop is uniform over the wave: the undecided samples are settled by all of its lanes)
  const int lane=(int) threadIdx.x & 63;
  for (size_t i0=(size_t) blockIdx.x*256u+(threadIdx.x & ~63u); i0 < n; i0+=(size_t) gridDim.x*256u)
    {
      const size_t i=i0+(size_t) lane < n ? i0+(size_t) lane : n-1;
      const bool mine=i0+(size_t) lane < n;
      const double2 s01=reinterpret_cast<const double2 *>(a.sums+i*4)[0];
      const double2 s23=reinterpret_cast<const double2 *>(a.sums+i*4)[1];
      double s[4]={s01.x,s01.y,s23.x,s23.y};
      if (a.delta != 0.0)
        {
          // + delta * (alpha*p .., alpha) of the one sample the extra cell sees
          const int y=(int) (i/(size_t) a.columns),x=(int) (i-(size_t) y*a.columns);
          int xx=x+a.delta_dx,yy=y+a.delta_dy;
          xx=xx < 0 ? 0 : (xx > a.columns-1 ? a.columns-1 : xx);
          yy=yy < 0 ? 0 : (yy > a.rows-1 ? a.rows-1 : yy);
          Q q[C];
          load_pixel<Q,C>(src+((size_t) yy*a.columns+(size_t) xx)*C,q);
          const double alpha=BLEND ? (double) q[C-1] : 1.0;
#pragma unroll
          for (int c=0; c < C; c++)
            s[c]=__builtin_fma(a.delta,(BLEND && (c != C-1)) ? alpha*(double) q[c] : (double) q[c],s[c]);
        }
      Q out[C];
      const uint32_t doubtful=settle_sums<Q,C,BLEND>(s,error,a.mixed_signs,out);
      unsigned long long pending=__ballot(mine && (doubtful != 0u));
      while (pending != 0ull)
        {
          const int who=__builtin_ctzll(pending);
          pending&=pending-1ull;
          const uint32_t which=(uint32_t) __builtin_amdgcn_readlane((int) doubtful,who);
          const size_t at=i0+(size_t) who;
          const int y=(int) (at/(size_t) a.columns),x=(int) (at-(size_t) y*a.columns);
#pragma unroll
          for (int c=0; c < C; c++)
            if ((which >> c) & 1u)
              {
                const Q settled=conv2d_reference_sample<Q,C,BLEND>(src,a.columns,a.rows,x,y,c,a.values,a.kw,a.kh,
                  a.shiftx,a.shifty,lane);
                if (lane == who)
                  {
                    out[c]=settled;
                    recomputed++;
                  }
              }
        }
      if (mine)
        store_pixel<Q,C>(dst+i*C,out);
    }
  if (a.recomputed != nullptr)
    {
      for (int off=32; off > 0; off>>=1)
        recomputed+=__shfl_xor(recomputed,off,64);
      if (((threadIdx.x & 63) == 0) && (recomputed != 0))
        atomicAdd(a.recomputed,(unsigned long long) recomputed);
    }
}

static unsigned long long *g_separable_recomputed[64]={};
static bool g_separable_count=false;

template<typename Q,int C,bool BLEND>
static MhStatus separable_typed(const View &src,SeparableArgs &a,const Conv1DParams &horizontal,
  const Conv1DParams &vertical,double *work)
{
  const size_t n=(size_t) a.columns*a.rows;
  size_t blocks=(n+255)/256;
  blocks=blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks);
  {
    ProfileScope prof("separable_premultiply",src.stream);
    hipLaunchKernelGGL((separable_premultiply_kernel<Q,C,BLEND>),dim3((unsigned) blocks),dim3(256),0,src.stream,a);
  }
  View sums=src,other=src;
  sums.channels=4;
  other.channels=4;
  sums.pixels=a.sums;
  other.pixels=work;
  MH_TRY(launch_conv1d_sums64(sums,other,false,horizontal));
  MH_TRY(launch_conv1d_sums64(other,sums,true,vertical));
  {
    ProfileScope prof("separable_finish",src.stream);
    hipLaunchKernelGGL((separable_finish_kernel<Q,C,BLEND>),dim3((unsigned) blocks),dim3(256),0,src.stream,a);
  }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

MhStatus launch_separable_exact(const View &src,const View &dst,const MhKernelInfo *kernel,
  const double *row,const double *column,const Roles &roles,bool *handled,int delta_x,int delta_y,
  double delta)
{
  *handled=false;
  const int kw=(int) kernel->width,kh=(int) kernel->height;
  // (the finish kernel re-reads src in the reference's order for undecided samples while other
  // workgroups store to dst: the two must be different buffers of one layout)
  if ((src.channels < 1) || (src.channels > 4) || (roles.copy_mask != 0) ||
      (src.pixels == dst.pixels) || (src.quantum != dst.quantum) || (src.channels != dst.channels) ||
      (src.columns != dst.columns) || (src.rows != dst.rows) || (kw > 255) || (kh > 255) ||
      (option("MAGICKHIP_NO_SEPARABLE_EXACT") != nullptr))
    return MH_OK;
  // (alpha-weighted: gray + alpha and RGBA; other layouts with an alpha trait keep the generic kernel)
  const bool blend=roles.blend && (roles.alpha == src.channels-1) && ((src.channels == 2) || (src.channels == 4));
  if (roles.blend && !blend)
    return MH_OK;
  // how far the separable evaluation can be from the reference's: the cells are the outer product
  // to 1e-13 of the largest cell (rank_one_factors), every one of the (kw+kh) fused multiply-adds
  // of a sample and of the reference's 3*kw*kh operations rounds to 2^-53
  double residual=0.0,magnitude=0.0,running=0.0,partials=0.0,total=0.0;
  bool negative=false,positive=false;
  for (int i=kw*kh-1; i >= 0; i--)                  // the reference's walk: from the last cell backwards
    {
      const double cell=kernel->values[i];
      if (std::isnan(cell))
        return MH_OK;
      total+=cell;
      negative=negative || (cell < 0.0);
      positive=positive || (cell > 0.0);
      // what the outer product (+ delta at its one cell) misses of this cell (rank_one_factors
      // admits up to 1e-13 of the largest cell; a Gaussian's cells are products to the last bit
      // or two), plus the rounding of the product that stands in for it
      double product=column[i/kw]*row[i % kw];
      double rounding=std::fabs(product)*2.220446049250313e-16;
      if ((delta != 0.0) && (i == delta_y*kw+delta_x))
        {
          rounding+=std::fabs(delta)*2.220446049250313e-16;
          product+=delta;
        }
      residual+=std::fabs(cell-product)+rounding;
      magnitude+=std::fabs(cell);
      // the reference's running sum after this cell is at most `running` * max|sample|: its
      // addition rounds to half an ulp of that
      running+=std::fabs(cell);
      partials+=running;
    }
  // reference: one rounding per addition (`partials`), two per term (alpha*k, *p); the two
  // separable passes: a fused multiply-add per tap on sums of at most `magnitude` * max|sample|
  // (+ |delta| for the extra cell).  Cells of both signs are fine: every bound is absolute, and
  // the finish kernel recomputes where an alpha sum has cancelled down to its error.
  const double unit=1.1102230246251565e-16;
  const double error_unit=2.0*(residual+unit*(partials+2.0*magnitude)+
    unit*((double) (kw+kh)+8.0)*(magnitude+std::fabs(delta)));
  const bool is_float=src.quantum != MH_QUANTUM_U16;
  if (!is_float && (error_unit*65535.0*65535.0 > 65535.0*1.0e-3))
    return MH_OK;                                  // (a kernel of tens of thousands of cells)
  // alpha-weighted frames under a kernel whose cells (nearly) cancel — EdgeImage's sums to zero:
  // sum(k*alpha) is all cancellation on any smooth alpha, PerceptibleReciprocal's clamp decides
  // every pixel and every pixel would go through the reference-order walk: the generic kernel's job
  if (blend && negative && positive && !(std::fabs(total) > 0.05*magnitude))
    return MH_OK;
  const size_t n=(size_t) src.columns*src.rows;
  // two launches (premultiply inside the row pass, this file's finish step inside the column
  // pass: convolve.hip) and one 32-byte-per-pixel intermediate where both axes have three taps
  const bool folded=(kw >= 3) && (kh >= 3) && (option("MAGICKHIP_NO_SEPARABLE_FOLD") == nullptr);
  Temp memory,table;
  // (folded: + the queue of undecided samples, at most 4 M entries; what overflows it is settled
  // inside the column pass)
  size_t queue_capacity=folded ? (n < ((size_t) 1 << 22) ? n : ((size_t) 1 << 22)) : 0;
  const long forced_capacity=option_long("MAGICKHIP_SEPARABLE_QUEUE",-1);      // (tests: the overflow path)
  if (folded && (forced_capacity >= 0) && ((size_t) forced_capacity < queue_capacity))
    queue_capacity=(size_t) forced_capacity;
  if (memory.alloc(src.device,(folded ? 1 : 2)*n*4*sizeof(double)+(4+1+queue_capacity)*sizeof(double),
        src.stream) != MH_OK)
    {
      // 64 bytes of fp64 sums per pixel do not fit next to the frames: the generic walk needs none
      (void) hipGetLastError();
      return MH_OK;
    }
  MH_TRY(upload_table(table,src.device,src.stream,kernel->values,(size_t) kw*kh*sizeof(double)));
  SeparableArgs a;
  a.src=src.pixels;
  a.dst=dst.pixels;
  a.sums=static_cast<double *>(memory.ptr);
  double *work=folded ? a.sums : a.sums+n*4;
  a.bound=work+n*4;
  a.queue_count=reinterpret_cast<unsigned *>(a.bound+4);
  a.queue=reinterpret_cast<unsigned long long *>(a.bound+5);
  a.queue_capacity=(unsigned) queue_capacity;
  a.columns=(int) src.columns;
  a.rows=(int) src.rows;
  a.values=table.as<double>();
  a.kw=kw;
  a.kh=kh;
  a.shiftx=kw-1-(int) kernel->x;
  a.shifty=kh-1-(int) kernel->y;
  a.error_unit=error_unit;
  for (int c=0; c < 4; c++)
    a.fixed_bound[c]=(blend && (c != src.channels-1)) ? 65535.0*65535.0 : 65535.0;
  a.delta=delta;
  // kernel cell (delta_y, delta_x) multiplies window position (kh-1-delta_y, kw-1-delta_x) of the
  // reflected walk: source (x + kernel->x - delta_x, y + kernel->y - delta_y)
  a.mixed_signs=(negative && positive) ? 1 : 0;
  a.delta_dx=(int) kernel->x-delta_x;
  a.delta_dy=(int) kernel->y-delta_y;
  a.recomputed=nullptr;
  if (g_separable_count && (src.device >= 0) && (src.device < 64))
    {
      if (g_separable_recomputed[src.device] == nullptr)
        {
          MH_HIP(hipMalloc(reinterpret_cast<void **>(&g_separable_recomputed[src.device]),sizeof(unsigned long long)));
          MH_HIP(hipMemsetAsync(g_separable_recomputed[src.device],0,sizeof(unsigned long long),src.stream));
        }
      a.recomputed=g_separable_recomputed[src.device];
    }
  MH_HIP(hipMemsetAsync(a.bound,0,5*sizeof(double),src.stream));     // and the queue's fill
  Conv1DParams horizontal,vertical;
  horizontal.taps=row;
  horizontal.ntaps=kw;
  horizontal.origin=(int) kernel->x;
  vertical.taps=column;
  vertical.ntaps=kh;
  vertical.origin=(int) kernel->y;
  if (folded)
    {
      MH_TRY(launch_separable_folded(src,a,horizontal,vertical,blend));
      *handled=true;
      return MH_OK;
    }
  MhStatus status=MH_OK;
#define MH_LAYOUT(QT) \
  switch (src.channels) \
  { \
    case 1: status=separable_typed<QT,1,false>(src,a,horizontal,vertical,work); break; \
    case 2: status=blend ? separable_typed<QT,2,true>(src,a,horizontal,vertical,work) : \
      separable_typed<QT,2,false>(src,a,horizontal,vertical,work); break; \
    case 3: status=separable_typed<QT,3,false>(src,a,horizontal,vertical,work); break; \
    default: status=blend ? separable_typed<QT,4,true>(src,a,horizontal,vertical,work) : \
      separable_typed<QT,4,false>(src,a,horizontal,vertical,work); break; \
  }
  if (is_float)
    { MH_LAYOUT(float) }
  else
    { MH_LAYOUT(uint16_t) }
#undef MH_LAYOUT
  MH_TRY(status);
  *handled=true;
  return MH_OK;
}

} // namespace mh

using namespace mh;

// Diagnostic: samples the separable EXACT path recomputed in the reference's order since the
// counter was last read (enable = 1 switches the counting on and reads, 0 reads and switches off).
extern "C" MH_API unsigned long long MhSeparableRecomputed(int enable)
{
  unsigned long long total=0;
  int current=-1;
  (void) hipGetDevice(&current);
  for (int d=0; d < 64; d++)
    if (g_separable_recomputed[d] != nullptr)
      {
        unsigned long long value=0;
        if (hipSetDevice(d) == hipSuccess)
          {
            (void) hipDeviceSynchronize();
            (void) hipMemcpy(&value,g_separable_recomputed[d],sizeof(value),hipMemcpyDeviceToHost);
            (void) hipMemset(g_separable_recomputed[d],0,sizeof(value));
          }
        total+=value;
      }
  if (current >= 0)
    (void) hipSetDevice(current);
  g_separable_count=enable != 0;
  return total;
}
