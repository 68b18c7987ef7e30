// General 2-D MorphologyPrimitive: arbitrary (NaN-masked) kernels for
// Convolve, Erode, Dilate, Erode/DilateIntensity, HitAndMiss, Thinning,
// Thicken and IterativeDistance.
//
// Reference semantics restated from the row path of MorphologyPrimitive,
// MagickCore/morphology.c:2811-3227 (origin handling :2623-2637).  The host
// flattens the kernel into the list of cells the CPU loop would actually
// touch, in the CPU's raster scan order (so floating-point sums associate
// identically), each with its offset from the output pixel and its value:
//   reflected methods (Convolve, Dilate, DilateIntensity, IterativeDistance):
//     cell (v,u) -> value[(h-1-v)*w+(w-1-u)], pixel (x-(w-kx-1)+u, y-(h-ky-1)+v)
//   direct methods (Erode, ErodeIntensity, HitAndMiss, Thinning, Thicken):
//     cell (v,u) -> value[v*w+u],             pixel (x-kx+u, y-ky+v)
// NaN cells are dropped; Erode keeps value>=0.5, Dilate value>0.5, etc.
//
// MI355X mapping: a workgroup stages a (64+kw-1) x (16+kh-1) edge-clamped
// tile of raw Quantum pixels in LDS with coalesced loads; lane = output
// column, each lane produces 4 output rows; the cell list is wave-uniform
// (scalar loads).  Min/max methods compare in the Quantum domain (exact).
#include "mh_internal.hpp"
#include "device_common.hpp"
#include <vector>

#include <cmath>
#include <cstdlib>
#include <algorithm>

namespace mh {

struct Cell
{
  int dx,dy;       // offset of the sample inside the LDS tile, relative to the output pixel's tile position
  double value;
};

enum MorphClass { MC_CONVOLVE,MC_ERODE,MC_DILATE,MC_HMT,MC_ERODE_INTENSITY,MC_DILATE_INTENSITY,MC_DISTANCE };

struct Morph2DArgs
{
  const void *src;
  void *dst;
  int columns,rows;
  int ncells;
  const Cell *cells;
  int left,top;              // how far the tile extends left/above the output block
  int tile_w,tile_h;
  double bias;
  double rescale;            // column path only: kernel->height/count, morphology.c:2775-2776
  uint32_t copy_mask;
  int hmt_mode;              // 0 HitAndMiss, 1 Thinning, 2 Thicken
  int linear,nonlinear,gray,intensity_method;
  unsigned long long *changed;
  const unsigned *only_if;   // nullptr, or: leave at once when the word is zero
};

constexpr int kTW=64;
constexpr int kTH=16;
constexpr int kRowsPerLane=4;

// Rec709Luma etc. for the *Intensity methods.  Only the gamma-free methods
// are evaluated here; others are rejected by the launcher.
template<typename Q,int C>
static __device__ __forceinline__ double morph_intensity(const Q (&q)[C],const Morph2DArgs &a)
{
  double red=(double) q[0];
  if (C == 1)
    return red;
  double green=(double) q[(C >= 3) && !a.gray ? 1 : 0];
  double blue=(double) q[(C >= 3) && !a.gray ? 2 : 0];
  switch (a.intensity_method)
  {
    case MH_INTENSITY_AVERAGE: return (red+green+blue)/3.0;
    case MH_INTENSITY_BRIGHTNESS:
    {
      double m=red > green ? red : green;
      return m > blue ? m : blue;
    }
    case MH_INTENSITY_LIGHTNESS:
    {
      double mn=red < green ? red : green;
      mn=mn < blue ? mn : blue;
      double mx=red > green ? red : green;
      mx=mx > blue ? mx : blue;
      return (mn+mx)/2.0;
    }
    case MH_INTENSITY_MS: return (red*red+green*green+blue*blue)/(3.0*kQR);
    case MH_INTENSITY_RMS: return sqrt(red*red+green*green+blue*blue)/sqrt(3.0);
    case MH_INTENSITY_REC601LUMA: return 0.298839*red+0.586811*green+0.114350*blue;
    default: break;
  }
  return 0.212656*red+0.715158*green+0.072186*blue;
}

template<typename Q,int C,bool BLEND,int MC>
__global__ __launch_bounds__(256)
void morph2d_kernel(Morph2DArgs args)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if ((args.only_if != nullptr) && (*args.only_if == 0u))
    return;
  Q *tile=reinterpret_cast<Q *>(smem_raw);
  const int W=args.columns,H=args.rows;
  const Q *src=static_cast<const Q *>(args.src);
  Q *dst=static_cast<Q *>(args.dst);
  const size_t pitch=(size_t) W*C;
  const int bx=(int) blockIdx.x*kTW,by=(int) blockIdx.y*kTH;
  const int TWID=args.tile_w,THGT=args.tile_h;

  // stage the tile (edge clamp, cache.c:2663-2679)
  for (int idx=(int) threadIdx.x; idx < TWID*THGT; idx+=256)
    {
      int ty=idx/TWID,tx=idx-ty*TWID;
      int sx=bx-args.left+tx,sy=by-args.top+ty;
      sx=sx < 0 ? 0 : (sx > W-1 ? W-1 : sx);
      sy=sy < 0 ? 0 : (sy > H-1 ? H-1 : sy);
      Q v[C];
      load_pixel<Q,C>(src+(size_t) sy*pitch+(size_t) sx*C,v);
      store_pixel<Q,C>(tile+(size_t) idx*C,v);
    }
  __syncthreads();

  const int lx=(int) (threadIdx.x & 63),ly0=(int) (threadIdx.x >> 6);
  const int x=bx+lx;
  unsigned changed=0;
#pragma unroll 1
  for (int rr=0; rr < kRowsPerLane; rr++)
    {
      const int ly=ly0+rr*4;
      const int y=by+ly;
      if ((x >= W) || (y >= H))
        continue;
      // position of the output pixel inside the tile
      const int px=lx+args.left,py=ly+args.top;
      Q center[C];
      load_pixel<Q,C>(tile+((size_t) py*TWID+px)*C,center);
      Q out[C];
      if constexpr (MC == MC_CONVOLVE)
        {
          double s[C],g=0.0;
#pragma unroll
          for (int c=0; c < C; c++)
            s[c]=args.bias;
          for (int i=0; i < args.ncells; i++)
            {
              const Cell cell=args.cells[i];
              Q q[C];
              load_pixel<Q,C>(tile+((size_t) (py+cell.dy)*TWID+(px+cell.dx))*C,q);
              if constexpr (BLEND)
                {
                  double alpha=kQS*(double) q[C-1];
                  double w=alpha*cell.value;
#pragma unroll
                  for (int c=0; c < C-1; c++)
                    s[c]=s[c]+w*(double) q[c];
                  g=g+w;
                  s[C-1]=s[C-1]+cell.value*(double) q[C-1];
                }
              else
                {
#pragma unroll
                  for (int c=0; c < C; c++)
                    s[c]=s[c]+cell.value*(double) q[c];
                }
            }
#pragma unroll
          for (int c=0; c < C; c++)
            {
              if ((args.copy_mask >> c) & 1u)
                {
                  out[c]=center[c];
                  continue;
                }
              if (fabs(s[c]-(double) center[c]) >= kEps)
                changed++;
              double gamma=(BLEND && (c != C-1)) ? perceptible_reciprocal(g) : 1.0;
              gamma=gamma*args.rescale;
              out[c]=QuantumOps<Q>::clamp(gamma*s[c]);
            }
        }
      else if constexpr ((MC == MC_ERODE) || (MC == MC_DILATE))
        {
          Q best[C];
#pragma unroll
          for (int c=0; c < C; c++)
            best[c]=(MC == MC_ERODE) ? center[c] : (Q) 0;     // morphology.c:2905-2912
          for (int i=0; i < args.ncells; i++)
            {
              const Cell cell=args.cells[i];
              Q q[C];
              load_pixel<Q,C>(tile+((size_t) (py+cell.dy)*TWID+(px+cell.dx))*C,q);
#pragma unroll
              for (int c=0; c < C; c++)
                {
                  if (MC == MC_ERODE)
                    {
                      if (q[c] < best[c]) best[c]=q[c];
                    }
                  else
                    {
                      if (q[c] > best[c]) best[c]=q[c];
                    }
                }
            }
#pragma unroll
          for (int c=0; c < C; c++)
            {
              if ((args.copy_mask >> c) & 1u)
                {
                  out[c]=center[c];
                  continue;
                }
              double pixel=(double) best[c];
              if (fabs(pixel-(double) center[c]) >= kEps)
                changed++;
              out[c]=QuantumOps<Q>::clamp(pixel);
            }
        }
      else if constexpr (MC == MC_HMT)
        {
          double mn[C],mx[C];
#pragma unroll
          for (int c=0; c < C; c++)
            {
              mn[c]=kQR;
              mx[c]=0.0;
            }
          for (int i=0; i < args.ncells; i++)
            {
              const Cell cell=args.cells[i];
              Q q[C];
              load_pixel<Q,C>(tile+((size_t) (py+cell.dy)*TWID+(px+cell.dx))*C,q);
              if (cell.value > 0.7)
                {
#pragma unroll
                  for (int c=0; c < C; c++)
                    if ((double) q[c] < mn[c]) mn[c]=(double) q[c];
                }
              else if (cell.value < 0.3)
                {
#pragma unroll
                  for (int c=0; c < C; c++)
                    if ((double) q[c] > mx[c]) mx[c]=(double) q[c];
                }
            }
#pragma unroll
          for (int c=0; c < C; c++)
            {
              if ((args.copy_mask >> c) & 1u)
                {
                  out[c]=center[c];
                  continue;
                }
              double m=mn[c]-mx[c];
              if (m < 0.0)
                m=0.0;
              double pixel=m;
              if (args.hmt_mode == 1)
                pixel=(double) center[c]-m;
              else if (args.hmt_mode == 2)
                pixel=(double) center[c]+m;
              if (fabs(pixel-(double) center[c]) >= kEps)
                changed++;
              out[c]=QuantumOps<Q>::clamp(pixel);
            }
        }
      else if constexpr ((MC == MC_ERODE_INTENSITY) || (MC == MC_DILATE_INTENSITY))
        {
          double best=(MC == MC_ERODE_INTENSITY) ? kQR : 0.0;
          bool found=false;
          Q chosen[C];
#pragma unroll
          for (int c=0; c < C; c++)
            chosen[c]=center[c];
          for (int i=0; i < args.ncells; i++)
            {
              const Cell cell=args.cells[i];
              Q q[C];
              load_pixel<Q,C>(tile+((size_t) (py+cell.dy)*TWID+(px+cell.dx))*C,q);
              double intensity=morph_intensity<Q,C>(q,args);
              bool take=(MC == MC_ERODE_INTENSITY) ? (intensity < best) : (intensity > best);
              if (take)
                {
                  best=intensity;
                  found=true;
#pragma unroll
                  for (int c=0; c < C; c++)
                    chosen[c]=q[c];
                }
            }
#pragma unroll
          for (int c=0; c < C; c++)
            {
              if ((args.copy_mask >> c) & 1u)
                {
                  out[c]=center[c];
                  continue;
                }
              if (found)
                {
                  out[c]=chosen[c];           // quantum_pixels path: no change count
                  continue;
                }
              // nothing selected: pixel keeps its initial value, morphology.c:2895-2912
              double pixel=(MC == MC_ERODE_INTENSITY) ? 0.0 : (double) center[c];
              if (fabs(pixel-(double) center[c]) >= kEps)
                changed++;
              out[c]=QuantumOps<Q>::clamp(pixel);
            }
        }
      else
        {
          // IterativeDistance, morphology.c:3171-3187
          double best[C];
#pragma unroll
          for (int c=0; c < C; c++)
            best[c]=(double) center[c];
          for (int i=0; i < args.ncells; i++)
            {
              const Cell cell=args.cells[i];
              Q q[C];
              load_pixel<Q,C>(tile+((size_t) (py+cell.dy)*TWID+(px+cell.dx))*C,q);
#pragma unroll
              for (int c=0; c < C; c++)
                if (((double) q[c]+cell.value) < best[c])
                  best[c]=(double) q[c]+cell.value;
            }
#pragma unroll
          for (int c=0; c < C; c++)
            {
              if ((args.copy_mask >> c) & 1u)
                {
                  out[c]=center[c];
                  continue;
                }
              if (fabs(best[c]-(double) center[c]) >= kEps)
                changed++;
              out[c]=QuantumOps<Q>::clamp(best[c]);
            }
        }
      store_pixel<Q,C>(dst+(size_t) y*pitch+(size_t) x*C,out);
    }
  if (args.changed != nullptr)
    {
      changed=wave_sum(changed);
      if (((threadIdx.x & 63) == 0) && (changed != 0))
        atomicAdd(args.changed,(unsigned long long) changed);
    }
}

template<typename Q,int C,bool BLEND>
static MhStatus launch_class(int mc,const Morph2DArgs &args,dim3 grid,size_t lds,hipStream_t stream)
{
#define MH_LAUNCH(MCV) \
  do { \
    if (lds > 64u*1024u) \
      MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&morph2d_kernel<Q,C,BLEND,MCV>), \
        hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds)); \
    hipLaunchKernelGGL((morph2d_kernel<Q,C,BLEND,MCV>),grid,dim3(256),lds,stream,args); \
  } while (0)
  switch (mc)
  {
    case MC_CONVOLVE: MH_LAUNCH(MC_CONVOLVE); break;
    case MC_ERODE: MH_LAUNCH(MC_ERODE); break;
    case MC_DILATE: MH_LAUNCH(MC_DILATE); break;
    case MC_HMT: MH_LAUNCH(MC_HMT); break;
    case MC_ERODE_INTENSITY: MH_LAUNCH(MC_ERODE_INTENSITY); break;
    case MC_DILATE_INTENSITY: MH_LAUNCH(MC_DILATE_INTENSITY); break;
    default: MH_LAUNCH(MC_DISTANCE); break;
  }
#undef MH_LAUNCH
  MH_HIP(hipGetLastError());
  return MH_OK;
}

template<typename Q>
static MhStatus launch_channels(int channels,bool blend,int mc,const Morph2DArgs &args,dim3 grid,
  size_t lds,hipStream_t stream)
{
  // only Convolve distinguishes the alpha-weighted variant
  const bool b=blend && (mc == MC_CONVOLVE);
  switch (channels)
  {
    case 1: return launch_class<Q,1,false>(mc,args,grid,lds,stream);
    case 2: return b ? launch_class<Q,2,true>(mc,args,grid,lds,stream) :
      launch_class<Q,2,false>(mc,args,grid,lds,stream);
    case 3: return launch_class<Q,3,false>(mc,args,grid,lds,stream);
    case 4: return b ? launch_class<Q,4,true>(mc,args,grid,lds,stream) :
      launch_class<Q,4,false>(mc,args,grid,lds,stream);
    default: break;
  }
  return fail(MH_UNSUPPORTED,"%d channels",channels);
}


// ------------------------------------------------ Erode / Dilate, convex flat kernels
// Disk, Diamond, Octagon, Square, Rectangle, Plus ...: every kernel row's active
// cells are ONE run [cx-h, cx+h] about a common centre column.  min/max are exact,
// so any evaluation order gives the reference's bits; instead of visiting every
// active cell (Disk:15 = 709) the kernel uses
//     result(x,y) = max_k  M_{h_k}(x+cx, y+dy_k),   M_h(x,y) = max_{|d|<=h} in(x+d, y)
// and builds the row-window maxima M_h incrementally over the kernel's DISTINCT
// half-widths h_0 < h_1 < ... (Disk:15: 10 levels):
//   raw tile (edge-clamped, coalesced loads) -> LDS
//   for each level l:  phase 1  plane(y,x) = max(plane(y,x), raw(y, x+cx+-d), h_{l-1} < d <= h_l)
//                      phase 2  every kernel row k of level l: out[r] = max(out[r], plane(y_r+dy_k, x))
// A 64 x 32 output tile costs ~2*hmax+1 raw reads per tile pixel plus one plane
// read per (kernel row, output) instead of one read per (active cell, output).
struct ConvexArgs
{
  const void *src;
  void *dst;
  int columns,rows;
  int cx;                     // centre column offset of the runs (dx of the run centres)
  int hmax;                   // widest half-width
  int top,bottom;             // kernel rows above / below the output row
  int nlevels;
  const int *level_h;         // [nlevels] ascending distinct half-widths
  const int *level_first;     // [nlevels+1] index into level_dy
  const int *level_dy;        // dy of every kernel row, grouped by level
  uint32_t copy_mask;
  unsigned long long *changed;
};

constexpr int kCTW=64;        // tile columns (= lanes)
constexpr int kCTR=32;        // tile rows; each of the kCWaves waves owns kCTR/kCWaves output rows
// waves per LDS tile: 8 (4 per SIMD with two tiles per CU) for small kernels, 16 for large ones
// (measured on 16384^2: Disk:15 7.6 / 6.1 / 5.5 ms with 4 / 8 / 16 waves, Disk:5 2.2 vs 2.8 ms with 8 vs 16)

template<typename Q,int C> struct PixelMinMax
{
  static __device__ __forceinline__ void apply(Q (&a)[C],const Q (&b)[C],bool take_max)
  {
    if constexpr ((sizeof(Q) == 2) && ((C & 1) == 0))
      {
        typedef unsigned short U2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int p=0; p < C/2; p++)
          {
            U2 va={(unsigned short) a[2*p],(unsigned short) a[2*p+1]};
            U2 vb={(unsigned short) b[2*p],(unsigned short) b[2*p+1]};
            U2 r=take_max ? __builtin_elementwise_max(va,vb) : __builtin_elementwise_min(va,vb);
            a[2*p]=(Q) r[0];
            a[2*p+1]=(Q) r[1];
          }
      }
    else
      {
#pragma unroll
        for (int c=0; c < C; c++)
          {
            if (take_max)
              { if (b[c] > a[c]) a[c]=b[c]; }          // morphology.c:3025-3026
            else
              { if (b[c] < a[c]) a[c]=b[c]; }          // morphology.c:2997-2998
          }
      }
  }
};

template<typename Q,int C,bool DILATE,int kCWaves>
__global__ __launch_bounds__(64*kCWaves)
void morph_convex_kernel(ConvexArgs args)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int W=args.columns,H=args.rows;
  const int hmax=args.hmax,top=args.top;
  const int tile_rows=kCTR+top+args.bottom;
  const int raw_w=kCTW+2*hmax;
  Q *raw=reinterpret_cast<Q *>(smem_raw);
  Q *plane=raw+(size_t) tile_rows*raw_w*C;
  const Q *src=static_cast<const Q *>(args.src);
  Q *dst=static_cast<Q *>(args.dst);
  const size_t pitch=(size_t) W*C;
  const int bx=(int) blockIdx.x*kCTW,by=(int) blockIdx.y*kCTR;
  const int lane=(int) (threadIdx.x & 63);
  const int wave=__builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));

  // raw tile: tile column t is image column bx + cx - hmax + t (edge clamp, cache.c:2663-2679)
  {
    constexpr int BATCH=6;
    const int items=tile_rows*raw_w;
    for (int i0=(int) threadIdx.x; i0 < items; i0+=64*kCWaves*BATCH)
      {
        Q v[BATCH][C];
#pragma unroll
        for (int k=0; k < BATCH; k++)
          {
            int idx=i0+64*kCWaves*k;
            idx=idx < items ? idx : items-1;
            const int ty=idx/raw_w,tx=idx-ty*raw_w;
            int sx=bx+args.cx-hmax+tx,sy=by-top+ty;
            sx=sx < 0 ? 0 : (sx > W-1 ? W-1 : sx);
            sy=sy < 0 ? 0 : (sy > H-1 ? H-1 : sy);
            load_pixel<Q,C>(src+(size_t) sy*pitch+(size_t) sx*C,v[k]);
          }
#pragma unroll
        for (int k=0; k < BATCH; k++)
          if (i0+64*kCWaves*k < items)
            store_pixel<Q,C>(raw+(size_t) (i0+64*kCWaves*k)*C,v[k]);
      }
  }
  // accumulators: Erode starts from the output pixel itself, Dilate from 0 (morphology.c:2905-2912)
  constexpr int R=kCTR/kCWaves;
  const int x=bx+lane;
  const int xc=x < W ? x : W-1;
  Q out[R][C],center[R][C];
#pragma unroll
  for (int r=0; r < R; r++)
    {
      int y=by+wave*R+r;
      y=y < H ? y : H-1;
      load_pixel<Q,C>(src+(size_t) y*pitch+(size_t) xc*C,center[r]);
#pragma unroll
      for (int c=0; c < C; c++)
        out[r][c]=DILATE ? (Q) 0 : center[r][c];
    }
  __syncthreads();

  int h_prev=-1;
  for (int l=0; l < args.nlevels; l++)
    {
      const int h=args.level_h[l];
      // phase 1: widen the row-window maxima of every tile row to half-width h.  Four
      // rows per step: their LDS reads are independent, so one latency covers all four.
      constexpr int RB=4;
      for (int t0=wave; t0 < tile_rows; t0+=kCWaves*RB)
        {
          Q m[RB][C];
          const Q *line[RB];
#pragma unroll
          for (int k=0; k < RB; k++)
            {
              int ty=t0+kCWaves*k;
              ty=ty < tile_rows ? ty : tile_rows-1;
              line[k]=raw+((size_t) ty*raw_w+(size_t) (lane+hmax))*C;
              if (h_prev < 0)
                load_pixel<Q,C>(line[k],m[k]);
              else
                load_pixel<Q,C>(plane+((size_t) ty*kCTW+lane)*C,m[k]);
            }
          for (int d=(h_prev < 0 ? 1 : h_prev+1); d <= h; d++)
            {
              Q a[RB][C],b[RB][C];
#pragma unroll
              for (int k=0; k < RB; k++)
                {
                  load_pixel<Q,C>(line[k]-(size_t) d*C,a[k]);
                  load_pixel<Q,C>(line[k]+(size_t) d*C,b[k]);
                }
#pragma unroll
              for (int k=0; k < RB; k++)
                {
                  PixelMinMax<Q,C>::apply(m[k],a[k],DILATE);
                  PixelMinMax<Q,C>::apply(m[k],b[k],DILATE);
                }
            }
#pragma unroll
          for (int k=0; k < RB; k++)
            {
              int ty=t0+kCWaves*k;
              if (ty < tile_rows)
                store_pixel<Q,C>(plane+((size_t) ty*kCTW+lane)*C,m[k]);
            }
        }
      __syncthreads();
      // phase 2: the kernel rows whose run has this half-width
      for (int i=args.level_first[l]; i < args.level_first[l+1]; i++)
        {
          const int dy=args.level_dy[i];
          const Q *col=plane+((size_t) (wave*R+top+dy)*kCTW+lane)*C;
#pragma unroll
          for (int r=0; r < R; r++)
            {
              Q v[C];
              load_pixel<Q,C>(col+(size_t) r*kCTW*C,v);
              PixelMinMax<Q,C>::apply(out[r],v,DILATE);
            }
        }
      __syncthreads();
      h_prev=h;
    }
  unsigned changed=0;
#pragma unroll
  for (int r=0; r < R; r++)
    {
      const int y=by+wave*R+r;
      if ((x < W) && (y < H))
        {
          Q res[C];
#pragma unroll
          for (int c=0; c < C; c++)
            {
              if ((args.copy_mask >> c) & 1u)
                {
                  res[c]=center[r][c];
                  continue;
                }
              double pixel=(double) out[r][c];
              if (fabs(pixel-(double) center[r][c]) >= kEps)
                changed++;
              res[c]=QuantumOps<Q>::clamp(pixel);
            }
          store_pixel<Q,C>(dst+(size_t) y*pitch+(size_t) x*C,res);
        }
    }
  if (args.changed != nullptr)
    {
      changed=wave_sum(changed);
      if ((lane == 0) && (changed != 0))
        atomicAdd(args.changed,(unsigned long long) changed);
    }
}

template<typename Q,int C,bool DILATE,int WAVES>
static MhStatus launch_convex_waves(const ConvexArgs &args,dim3 grid,size_t lds,hipStream_t stream)
{
  if (lds > 64u*1024u)
    MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&morph_convex_kernel<Q,C,DILATE,WAVES>),
      hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  hipLaunchKernelGGL((morph_convex_kernel<Q,C,DILATE,WAVES>),grid,dim3(64*WAVES),lds,stream,args);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

template<typename Q,int C>
static MhStatus launch_convex(bool dilate,const ConvexArgs &args,dim3 grid,size_t lds,hipStream_t stream)
{
  const bool large=(args.top+args.bottom) >= 16;
  if (dilate)
    return large ? launch_convex_waves<Q,C,true,16>(args,grid,lds,stream) :
      launch_convex_waves<Q,C,true,8>(args,grid,lds,stream);
  return large ? launch_convex_waves<Q,C,false,16>(args,grid,lds,stream) :
    launch_convex_waves<Q,C,false,8>(args,grid,lds,stream);
}

// ------------------------------------------------ Erode / Dilate, symmetric convex kernels
// A flat kernel whose rows are centred runs, symmetric about a centre row, with half-widths that do
// not grow away from it (Disk, Square, Rectangle, Diamond, Octagon, Plus ...) is the union of the
// centred rectangles  Rect(h_l, v_l),  h_0 < h_1 < ... < h_L  the distinct half-widths and v_l the
// largest |dy| whose row is at least h_l wide (v_0 > v_1 > ... > v_L).  With Row(h) = Row(h_0) +
// Row(h_1-h_0) + ... (Minkowski sums) and because dilation distributes over unions,
//     out = Row(h_0)[ Col(v_0) in  v  Row(h_1-h_0)[ Col(v_1) in  v  ... Row(h_L-h_{L-1})[ Col(v_L) in ] ] ]
// evaluated from the inside out:   C <- Col(v_l) in  (widened from Col(v_{l+1}) in: v_l grows),
//                                  S <- Row(h_l-h_{l-1})[ C v S ].
// 2 v_0 + 2 h_L + L+1 compares per pixel (Disk:15: 71 against the 709 cells), all min/max, so
// the result has the reference's bits (morphology.c:2980-3036) whatever the order.
//
// MI355X mapping: lane = two adjacent columns, a wave = 128 columns x SY rows held in VGPRs (C
// and S).  Col() reads the lane's own columns of the staged tile (ds_read_b128, conflict free,
// wave-uniform row offsets); Row(d) is d applications of Row(1), whose two neighbour columns
// that live in the next lane come over with one v_mov_b32_dpp wave_shr:1 / wave_shl:1 each — no
// LDS traffic and no barrier between the levels.  The columns within hmax of the wave's edges
// fill with garbage (one per Row(1)) and are not stored: a workgroup produces 128-2*hmax columns.
struct RectsArgs
{
  const void *src;            // Q16 or float Quantum, 4-byte multiples per pixel
  void *dst;
  int columns,rows;
  int cx,cy;                  // centre of the shape relative to the output pixel
  int hmax,vmax;              // half-width of the widest row, half-height
  int nlevels;
  int tiles_x,tiles_y,tiles_per_xcd;
  uint32_t copy_mask;
  unsigned long long *changed;
  unsigned char widen[64];    // [nlevels]  h_l - h_{l-1}  (h_{-1} = 0)
  unsigned char reach[64];    // [nlevels]  v_l
};

template<bool DILATE>
static __device__ __forceinline__ uint32_t pk_pick(uint32_t a,uint32_t b)
{
  typedef unsigned short U2 __attribute__((ext_vector_type(2)));
  const U2 va=__builtin_bit_cast(U2,a),vb=__builtin_bit_cast(U2,b);
  return __builtin_bit_cast(uint32_t,DILATE ? __builtin_elementwise_max(va,vb) : __builtin_elementwise_min(va,vb));
}

// one 32-bit word of samples: two Q16 channels (packed compare) or one float channel.  The float
// forms are v_max_f32 / v_min_f32: a NaN operand yields the other one, which is what the
// reference's `if (pixels[i] > pixel)` does with a NaN neighbour (morphology.c:2995-3030).
// (fmaxf / fminf: hipcc canonicalises both operands first — a v_max_f32 x,x each — and fuses pairs
// into v_max3_f32 / v_min3_f32.  The bare instructions through inline asm measured slower, 5.3
// against 4.8 ms for Dilate Disk:15 on 16384^2 float RGBA: the scheduler cannot see into them.)
template<typename Q,bool DILATE>
static __device__ __forceinline__ uint32_t word_pick(uint32_t a,uint32_t b)
{
  if constexpr (sizeof(Q) == 2)
    return pk_pick<DILATE>(a,b);
  else
    {
      const float fa=__builtin_bit_cast(float,a),fb=__builtin_bit_cast(float,b);
      return __builtin_bit_cast(uint32_t,DILATE ? __builtin_fmaxf(fa,fb) : __builtin_fminf(fa,fb));
    }
}

template<typename Q,bool DILATE>
static __device__ __forceinline__ uint32_t word_pick3(uint32_t a,uint32_t b,uint32_t c)
{
  return word_pick<Q,DILATE>(word_pick<Q,DILATE>(a,b),c);
}

// Q: uint16_t (2 or 4 channels) or float (1, 2 or 4 channels: the reference's default build is
// HDRI; min/max are exact in any type).  A lane holds SX columns = WPR words; RGBA float is one
// column per lane (a wave keeps 64-2*hmax of its 64 columns).
template<typename Q,int C,bool DILATE,int SY,int NWAVES>
__global__ __launch_bounds__(64*NWAVES)
void morph_rects_kernel(RectsArgs args)
{
  constexpr int PXB=C*(int) sizeof(Q);         // bytes per pixel
  static_assert((PXB % 4) == 0,"whole 32-bit words per pixel");
  constexpr int NW=PXB/4;                      // 32-bit words per pixel
  constexpr int SX=NW >= 4 ? 1 : 2;            // columns per lane
  constexpr int WPR=SX*NW;                     // words a lane holds per row
  constexpr bool kFloat=sizeof(Q) == 4;
  constexpr int TH=SY*NWAVES;                  // output rows per workgroup
  typedef uint32_t Group __attribute__((ext_vector_type(WPR)));
  typedef Group __attribute__((aligned(4))) LooseGroup;      // global memory: pixel alignment only
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Group *tile=reinterpret_cast<Group *>(smem_raw);           // [TH+2*vmax][64]
  const int W=args.columns,H=args.rows;
  const int hmax=args.hmax,vmax=args.vmax;
  const int tid=(int) threadIdx.x,lane=tid & 63;
  const int wave=__builtin_amdgcn_readfirstlane(tid >> 6);
  // consecutive workgroup ids go to different XCDs: give each XCD a contiguous run of tiles,
  // row-major, so that the halo a tile shares with its neighbours is in that XCD's L2
  const int id=((int) blockIdx.x & 7)*args.tiles_per_xcd+((int) blockIdx.x >> 3);
  if (id >= args.tiles_x*args.tiles_y)
    return;
  const int tile_y=id/args.tiles_x,tile_x=id-tile_y*args.tiles_x;
  const int valid_w=64*SX-2*hmax;
  const int bx=tile_x*valid_w,by=tile_y*TH;
  const int tile_rows=TH+2*vmax;

  // ---- stage the edge-clamped source window (cache.c:2663-2679): tile column t is image
  // column bx+cx-hmax+t, tile row q is image row by+cy-vmax+q
  {
    // a thread keeps its columns and walks down the rows NWAVES apart: the column clamp and the
    // in-frame test are done once, a row costs its clamp and one (wave-uniform, 64-bit) offset
    const int sx=bx+args.cx-hmax+SX*lane,sy0=by+args.cy-vmax;
    const bool inside=(sx >= 0) && (sx+SX-1 <= W-1);
    unsigned xoff[SX];
#pragma unroll
    for (int j=0; j < SX; j++)
      {
        int x=sx+j;
        x=x < 0 ? 0 : (x > W-1 ? W-1 : x);
        xoff[j]=(unsigned) x*(unsigned) PXB;
      }
    const unsigned row_bytes=(unsigned) W*(unsigned) PXB;
    const unsigned char *base=reinterpret_cast<const unsigned char *>(args.src);
    constexpr int BATCH=NWAVES >= 16 ? 5 : (NWAVES >= 12 ? 4 : 7);  // Disk:15: 13 (6 waves) or 7 (12 waves) rows per thread, two round trips
    for (int q0=wave; q0 < tile_rows; q0+=NWAVES*BATCH)
      {
        Group g[BATCH];
#pragma unroll
        for (int k=0; k < BATCH; k++)
          {
            int q=q0+NWAVES*k;
            q=q < tile_rows ? q : tile_rows-1;
            int sy=sy0+q;
            sy=sy < 0 ? 0 : (sy > H-1 ? H-1 : sy);
            const unsigned char *row=base+(size_t) sy*row_bytes;
            if (inside)
              g[k]=*reinterpret_cast<const LooseGroup *>(row+xoff[0]);
            else
              {
#pragma unroll
                for (int j=0; j < SX; j++)
#pragma unroll
                  for (int w=0; w < NW; w++)
                    g[k][j*NW+w]=*reinterpret_cast<const uint32_t *>(row+xoff[j]+4*w);
              }
          }
#pragma unroll
        for (int k=0; k < BATCH; k++)
          if (q0+NWAVES*k < tile_rows)
            tile[(q0+NWAVES*k)*64+lane]=g[k];
      }
  }
  __syncthreads();

  // ---- the wave's SY output rows: tile rows wave*SY+vmax+i
  const Group *centre=tile+(size_t) (wave*SY+vmax)*64+lane;
  uint32_t column[SY][WPR],spread[SY][WPR];    // C and S of the header
  // The rows a fold depth takes in slide: at depth k output row i needs tile rows i-k and i+k, and
  // row i-k = row (i+1)-(k+1) was read one depth earlier for output row i+1.  `upper` holds the
  // rows 0-j (j = k-SY+1 .. k) in slot j mod SY, `lower` the rows SY-1+j: a depth costs two row
  // reads instead of 2*SY (the LDS pipe, not the vector unit, set the pace of the fold), and
  // with the depths unrolled SY at a time every slot is a compile-time register.
  Group upper[SY],lower[SY];
#pragma unroll
  for (int i=0; i < SY; i++)
    {
      const Group g=centre[i*64];
      upper[(SY-i)%SY]=g;                      // row i = row 0-(-i)
      lower[(i+1)%SY]=g;                       // row i = row SY-1+(i-SY+1)
#pragma unroll
      for (int p=0; p < WPR; p++)
        {
          column[i][p]=g[p];
          spread[i][p]=DILATE ? 0u : (kFloat ? 0x7f800000u : 0xffffffffu);   // 0 / QuantumRange or +inf
        }
    }
  // lane l keeps level l's reach and widening: a v_readlane per level instead of a load from the
  // kernel arguments
  const int lane_reach=lane < args.nlevels ? (int) args.reach[lane] : 0;
  const int lane_widen=lane < args.nlevels ? (int) args.widen[lane] : 0;
  // every level whose reach the fold depth has arrived at: merge the column windows into the
  // spread and widen it (the levels' reaches grow as l falls)
  int level=args.nlevels-1;
  auto settle=[&](int depth)
  {
    while ((level >= 0) && (__builtin_amdgcn_readlane(lane_reach,level) <= depth))
      {
#pragma unroll
        for (int i=0; i < SY; i++)
#pragma unroll
          for (int p=0; p < WPR; p++)
            spread[i][p]=word_pick<Q,DILATE>(spread[i][p],column[i][p]);
        const int widen=__builtin_amdgcn_readlane(lane_widen,level);
        for (int step=0; step < widen; step++)
          {
            // Row(1): every column takes in its two neighbours.  Two columns a b per lane share
            // their pair maximum: a <- b' v (a v b), b <- (a v b) v a' (b', a': the neighbour lanes');
            // one column per lane: a <- a' v a v a''
#pragma unroll
            for (int i=0; i < SY; i++)
#pragma unroll
              for (int w=0; w < NW; w++)
                {
                  // wave_shr:1 — lane n reads lane n-1 (0x138); wave_shl:1 — lane n reads lane n+1 (0x130)
                  if constexpr (SX == 2)
                    {
                      const uint32_t a=spread[i][w],b=spread[i][NW+w];
                      const uint32_t left=(uint32_t) __builtin_amdgcn_mov_dpp((int) b,0x138,0xf,0xf,true);
                      const uint32_t right=(uint32_t) __builtin_amdgcn_mov_dpp((int) a,0x130,0xf,0xf,true);
                      const uint32_t both=word_pick<Q,DILATE>(a,b);
                      spread[i][w]=word_pick<Q,DILATE>(left,both);
                      spread[i][NW+w]=word_pick<Q,DILATE>(both,right);
                    }
                  else
                    {
                      const uint32_t a=spread[i][w];
                      const uint32_t left=(uint32_t) __builtin_amdgcn_mov_dpp((int) a,0x138,0xf,0xf,true);
                      const uint32_t right=(uint32_t) __builtin_amdgcn_mov_dpp((int) a,0x130,0xf,0xf,true);
                      spread[i][w]=word_pick3<Q,DILATE>(left,a,right);
                    }
                }
          }
        level--;
      }
  };
  settle(0);
  for (int k0=1; k0 <= vmax; k0+=SY)
    {
#pragma unroll
      for (int u=0; u < SY; u++)
        {
          const int k=k0+u;                    // k mod SY = (1+u) mod SY
          if (k > vmax)
            break;
          upper[(1+u)%SY]=centre[-k*64];
          lower[(1+u)%SY]=centre[(SY-1+k)*64];
#pragma unroll
          for (int i=0; i < SY; i++)
            {
              // row i-k is row 0-(k-i), row i+k is row SY-1+(k+i-SY+1)
              const Group &above=upper[(1+u+SY-i)%SY];
              const Group &below=lower[(2+u+i)%SY];
#pragma unroll
              for (int p=0; p < WPR; p++)
                column[i][p]=word_pick3<Q,DILATE>(column[i][p],above[p],below[p]);
            }
          settle(k);
        }
    }

  // ---- copy out: morphology.c:3180-3196 (channels without the update trait keep the source
  // value; `changed` counts the updated samples that differ from the source)
  unsigned changed=0;
  const int u0=SX*lane-hmax;                   // first of the lane's columns within the valid span
  const bool centred=(args.cx == 0) && (args.cy == 0);
  if (DILATE && (args.copy_mask == 0u) && (args.changed == nullptr))
    {
      // every channel updated, nobody counts: the maxima are the result
      const bool whole=(u0 >= 0) && (u0+SX-1 < valid_w) && (bx+u0+SX-1 < W);
#pragma unroll
      for (int i=0; i < SY; i++)
        {
          const int y=by+wave*SY+i;
          if (y >= H)
            break;
          unsigned char *out=reinterpret_cast<unsigned char *>(args.dst)+(size_t) y*((size_t) W*(size_t) PXB);
          Group result;
#pragma unroll
          for (int p=0; p < WPR; p++)
            result[p]=spread[i][p];
          if (whole)
            *reinterpret_cast<LooseGroup *>(out+(unsigned) (bx+u0)*(unsigned) PXB)=result;
          else
            {
#pragma unroll
              for (int j=0; j < SX; j++)
                if ((u0+j >= 0) && (u0+j < valid_w) && (bx+u0+j < W))
#pragma unroll
                  for (int w=0; w < NW; w++)
                    *reinterpret_cast<uint32_t *>(out+(unsigned) (bx+u0+j)*(unsigned) PXB+4*w)=result[j*NW+w];
            }
        }
      return;
    }
#pragma unroll
  for (int i=0; i < SY; i++)
    {
      const int y=by+wave*SY+i;
      if (y >= H)
        break;
      unsigned char *out_row=reinterpret_cast<unsigned char *>(args.dst)+(size_t) y*((size_t) W*(size_t) PXB);
      const unsigned char *in_row=reinterpret_cast<const unsigned char *>(args.src)+(size_t) y*((size_t) W*(size_t) PXB);
      bool ok[SX];
#pragma unroll
      for (int j=0; j < SX; j++)
        ok[j]=(u0+j >= 0) && (u0+j < valid_w) && (bx+u0+j < W);
      Group original;
      bool all=true;
#pragma unroll
      for (int j=0; j < SX; j++)
        all=all && ok[j];
      if (centred)
        original=centre[i*64];                 // the output pixel is the centre of its own window
      else if (all)
        original=*reinterpret_cast<const LooseGroup *>(in_row+(unsigned) (bx+u0)*(unsigned) PXB);
      else
        {
#pragma unroll
          for (int j=0; j < SX; j++)
#pragma unroll
            for (int w=0; w < NW; w++)
              original[j*NW+w]=ok[j] ? *reinterpret_cast<const uint32_t *>(in_row+(unsigned) (bx+u0+j)*(unsigned) PXB+4*w) : 0u;
        }
      Group result;
#pragma unroll
      for (int p=0; p < WPR; p++)
        {
          if constexpr (kFloat)
            {
              // one channel per word.  Erode starts from the output pixel itself
              // (morphology.c:2905-2912): a NaN there stays (no `<` is true against it)
              const int c=p % NW;
              const uint32_t mine_bits=original[p];    // (a copy: __builtin_bit_cast of the vector
              const float mine=__builtin_bit_cast(float,mine_bits);   // element itself reads element 0)
              uint32_t value=spread[i][p];
              if (!DILATE)
                value=mine != mine ? mine_bits : word_pick<Q,false>(value,mine_bits);
              const bool keep=((args.copy_mask >> c) & 1u) != 0u;
              result[p]=keep ? mine_bits : value;
              // morphology.c:3195: fabs(pixel-p[center+i]) >= MagickEpsilon
              if (!keep && ok[p/NW] &&
                  (fabs((double) __builtin_bit_cast(float,value)-(double) mine) >= kEps))
                changed++;
            }
          else
            {
              // Erode starts from the output pixel itself (morphology.c:2905-2912)
              const uint32_t value=DILATE ? spread[i][p] : pk_pick<false>(spread[i][p],original[p]);
              const int c0=2*(p % NW);             // channels of this word's halves
              uint32_t keep=0u;
              keep|=((args.copy_mask >> c0) & 1u) != 0u ? 0x0000ffffu : 0u;
              keep|=((args.copy_mask >> (c0+1)) & 1u) != 0u ? 0xffff0000u : 0u;
              result[p]=(original[p] & keep) | (value & ~keep);
              const uint32_t differs=(value ^ original[p]) & ~keep;
              if (ok[p/NW])
                changed+=((differs & 0xffffu) != 0u ? 1u : 0u)+((differs >> 16) != 0u ? 1u : 0u);
            }
        }
      if (all)
        *reinterpret_cast<LooseGroup *>(out_row+(unsigned) (bx+u0)*(unsigned) PXB)=result;
      else
        {
#pragma unroll
          for (int j=0; j < SX; j++)
            if (ok[j])
#pragma unroll
              for (int w=0; w < NW; w++)
                *reinterpret_cast<uint32_t *>(out_row+(unsigned) (bx+u0+j)*(unsigned) PXB+4*w)=result[j*NW+w];
        }
    }
  if (args.changed != nullptr)
    {
      changed=wave_sum(changed);
      if ((lane == 0) && (changed != 0))
        atomicAdd(args.changed,(unsigned long long) changed);
    }
}

template<typename Q,int C,bool DILATE,int SY,int WAVES>
static MhStatus launch_rects_typed(const RectsArgs &args,size_t lds,hipStream_t stream)
{
  const dim3 grid((unsigned) (8*args.tiles_per_xcd)),block(64*WAVES);
  MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&morph_rects_kernel<Q,C,DILATE,SY,WAVES>),
    hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  hipLaunchKernelGGL((morph_rects_kernel<Q,C,DILATE,SY,WAVES>),grid,block,lds,stream,args);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// shape: 0 = 6 waves x 8 rows, 1 = 12 waves x 4 rows, 3 = 16 waves x 3 rows (48 output rows
// each), 2 = 12 waves x 8 rows (Q16; float Quantum runs shape 1)
template<typename Q,int C>
static MhStatus launch_rects(bool dilate,int shape,const RectsArgs &args,size_t lds,hipStream_t stream)
{
  if constexpr (sizeof(Q) == 2)
    {
      if (shape == 3)
        return dilate ? launch_rects_typed<Q,C,true,3,16>(args,lds,stream) : launch_rects_typed<Q,C,false,3,16>(args,lds,stream);
      if (shape == 2)
        return dilate ? launch_rects_typed<Q,C,true,8,12>(args,lds,stream) : launch_rects_typed<Q,C,false,8,12>(args,lds,stream);
      if (shape == 0)
        return dilate ? launch_rects_typed<Q,C,true,8,6>(args,lds,stream) : launch_rects_typed<Q,C,false,8,6>(args,lds,stream);
    }
  return dilate ? launch_rects_typed<Q,C,true,4,12>(args,lds,stream) : launch_rects_typed<Q,C,false,4,12>(args,lds,stream);
}

// ------------------------------------------------ the same evaluation as a walk down a strip
// morph_rects_kernel spends a third of its vector instructions on columns it discards (a wave of
// 128 columns keeps 128-2*hmax) and on the v_mov_dpp that fetch a neighbour lane's edge column, it
// reads every staged row 2*SY times per output row quad, and it stages TH+2*vmax rows for TH rows
// of output.  This kernel
//   * gives a lane FOUR adjacent columns: a wave = 256 columns, 256-2*hmax kept; Row(1) costs 6
//     packed min/max + 2 cross-lane moves per word plane instead of 8 + 4 for the same four
//     columns (the pair maxima (a,b) and (c,d) are shared by the columns next to them);
//   * lets a wave slide over its rows: output rows c and c+1 at fold depth k need rows c-k, c+k and
//     c+1-k, c+1+k — two of the four were loaded for depth k-1, so a depth costs two row reads;
//   * lets a workgroup walk DOWN its strip through a ring of 2*TH+2*vmax rows in LDS: the rows two
//     successive tiles share stay where they are, and the TH new rows of the next tile are written
//     by global_load_lds_dwordx4 (memory -> LDS without passing through registers) while this tile
//     is evaluated.  Every source row is read once per strip: the frame is read 256/(256-2*hmax)
//     times instead of (128/(128-2*hmax))*(TH+2*vmax)/TH.
// One workgroup per CU (a ring row is 2 KiB for RGBA), one barrier per tile.
struct StripsArgs
{
  RectsArgs r;                // tiles_x = strips, tiles_y = ceil(rows/TH): steps of a whole strip
  int segments;               // vertical cuts of a strip: work items = strips*segments
  int steps_per_segment;
  int items_per_xcd;
};

template<int C,bool DILATE,int NWAVES>
__global__ __launch_bounds__(64*NWAVES)
void morph_strips_kernel(StripsArgs sargs)
{
  static_assert((C == 2) || (C == 4),"whole 32-bit words per pixel");
  const RectsArgs &args=sargs.r;
  constexpr int SX=4;                          // columns per lane
  constexpr int SY=2;                          // rows per wave
  constexpr int NW=C/2;                        // 32-bit words per pixel
  constexpr int WPR=SX*NW;                     // words a lane holds per row
  constexpr int HW=WPR/2;                      // ... per half (two columns): one LDS access
  constexpr int TH=SY*NWAVES;                  // output rows per step
  constexpr int kStoreDepth=4;                 // 1 + a multiple of 3
  typedef uint32_t Half __attribute__((ext_vector_type(HW)));
  typedef Half __attribute__((aligned(4))) LooseHalf;        // global memory: pixel alignment only
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Half *ring=reinterpret_cast<Half *>(smem_raw);             // [2*TH+2*vmax][2 halves][64 lanes]
  const int W=args.columns,H=args.rows;
  const int hmax=args.hmax,vmax=args.vmax,halo=2*vmax;
  const int R=2*TH+halo;                       // ring rows
  const int tid=(int) threadIdx.x,lane=tid & 63;
  const int wave=__builtin_amdgcn_readfirstlane(tid >> 6);
  const int item=((int) blockIdx.x & 7)*sargs.items_per_xcd+((int) blockIdx.x >> 3);
  if (item >= args.tiles_x*sargs.segments)
    return;
  const int segment=item/args.tiles_x,strip=item-segment*args.tiles_x;
  const int step_begin=segment*sargs.steps_per_segment;
  const int step_end=step_begin+sargs.steps_per_segment < args.tiles_y ? step_begin+sargs.steps_per_segment : args.tiles_y;
  const int valid_w=64*SX-2*hmax;
  const int bx=strip*valid_w;

  // ---- source access: tile column t is image column bx+cx-hmax+t, clamped (cache.c:2663-2679);
  // a lane keeps its four columns for the whole walk.  Lanes whose four columns are inside the
  // frame (all but a few of the first and the last strip) fetch straight into LDS; the others
  // through registers.
  const int sx=bx+args.cx-hmax+SX*lane;
  const bool inside=(sx >= 0) && (sx+SX-1 <= W-1);
  unsigned xoff[SX];
#pragma unroll
  for (int j=0; j < SX; j++)
    {
      int x=sx+j;
      x=x < 0 ? 0 : (x > W-1 ? W-1 : x);
      xoff[j]=(unsigned) x*(unsigned) (C*sizeof(uint16_t));
    }
  const unsigned row_bytes=(unsigned) W*(unsigned) (C*sizeof(uint16_t));
  const unsigned char *base=reinterpret_cast<const unsigned char *>(args.src);
  const unsigned lds_base=(unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) smem_raw;
  auto ring_at=[&](int row,int half) -> Half * { return ring+(size_t) (row*2+half)*64+lane; };
  // image row sy -> ring row `row`; `held`: the columns of a lane at the frame's edge
  auto fetch_row=[&](int sy,int row,Half (&held)[2])
  {
    sy=sy < 0 ? 0 : (sy > H-1 ? H-1 : sy);
    const unsigned at=(unsigned) sy*row_bytes;
    if (inside)
      {
        if constexpr (C == 4)
          {
            // 64 lanes x 16 bytes land at M0 + 16*lane: one (row, half) plane of the ring
            const unsigned to=(unsigned) __builtin_amdgcn_readfirstlane((int) (lds_base+(unsigned) row*2048u));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                         "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %2"
                         : : "s"(to),"v"(at+xoff[0]),"s"(base),"s"(to+1024u),"v"(at+xoff[2]) : "memory");
          }
        else
          {
            held[0]=*reinterpret_cast<const LooseHalf *>(base+at+xoff[0]);
            held[1]=*reinterpret_cast<const LooseHalf *>(base+at+xoff[2]);
          }
      }
    else
      {
#pragma unroll
        for (int j=0; j < SX; j++)
#pragma unroll
          for (int w=0; w < NW; w++)
            held[j >> 1][(j & 1)*NW+w]=*reinterpret_cast<const uint32_t *>(base+at+xoff[j]+4*w);
      }
  };
  auto commit_row=[&](int row,const Half (&held)[2])
  {
    if ((C != 4) || !inside)
      {
        *ring_at(row,0)=held[0];
        *ring_at(row,1)=held[1];
      }
  };
  auto wrapped=[&](int row) { return row >= R ? row-R : row; };

  // tile row q of step s is image row s*TH+cy-vmax+q and ring row (origin+q) mod R, `origin`
  // advancing by TH a step
  int origin=0;
  {
    const int sy0=step_begin*TH+args.cy-vmax;
    for (int q=wave; q < halo+TH; q+=NWAVES)
      {
        Half held[2];
        fetch_row(sy0+q,q,held);
        commit_row(q,held);
      }
  }
  Half held[SY][2];                            // the new rows of the next step, frame-edge lanes
  // lane l keeps level l's reach and widening (a v_readlane per level instead of a load from the
  // kernel arguments, whose latency nothing hides)
  const int lane_reach=lane < args.nlevels ? (int) args.reach[lane] : 0;
  const int lane_widen=lane < args.nlevels ? (int) args.widen[lane] : 0;
  const int u0=SX*lane-hmax;                   // first of the lane's columns within the valid span
  const bool centred=(args.cx == 0) && (args.cy == 0);
  bool ok[SX];
  bool all=true;
#pragma unroll
  for (int j=0; j < SX; j++)
    {
      ok[j]=(u0+j >= 0) && (u0+j < valid_w) && (bx+u0+j < W);
      all=all && ok[j];
    }
  unsigned changed=0;
  // the results of a step are stored at the beginning of the next one: the s_waitcnt vmcnt(0) that
  // ends a step (the fetched rows are in LDS) would otherwise wait for the stores just issued
  uint32_t result[SY][WPR];
  int result_by=-1;
  auto store_results=[&]()
  {
    if (result_by < 0)
      return;
#pragma unroll
    for (int i=0; i < SY; i++)
      {
        const int y=result_by+wave*SY+i;
        if (y >= H)
          break;
        unsigned char *out=reinterpret_cast<unsigned char *>(args.dst)+(unsigned) y*row_bytes;
        if (all)
          {
            Half lo,hi;
#pragma unroll
            for (int p=0; p < HW; p++)
              {
                lo[p]=result[i][p];
                hi[p]=result[i][HW+p];
              }
            unsigned char *at=out+(unsigned) (bx+u0)*(unsigned) (C*sizeof(uint16_t));
            *reinterpret_cast<LooseHalf *>(at)=lo;
            *reinterpret_cast<LooseHalf *>(at+sizeof(Half))=hi;
          }
        else
          {
#pragma unroll
            for (int j=0; j < SX; j++)
              if (ok[j])
#pragma unroll
                for (int w=0; w < NW; w++)
                  *reinterpret_cast<uint32_t *>(out+(unsigned) (bx+u0+j)*(unsigned) (C*sizeof(uint16_t))+4*w)=result[i][j*NW+w];
          }
      }
  };
  for (int step=step_begin; step < step_end; step++)
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the rows fetched for this step are in LDS
      __syncthreads();                                     // ... everybody's, and the previous tile is done with
      if (step+1 < step_end)
        {
          // the TH new rows of step+1: tile rows halo+TH .. halo+2*TH-1 of this step
          const int sy0=(step+1)*TH+args.cy-vmax+halo;
#pragma unroll
          for (int j=0; j < SY; j++)
            fetch_row(sy0+wave+NWAVES*j,wrapped(wrapped(origin+TH)+halo+wave+NWAVES*j),held[j]);
        }

      // ---- the wave's two output rows: tile rows centre, centre+1
      const int by=step*TH;
      const int centre=origin+vmax+wave*SY;      // < 2R: wrapped() per access
      uint32_t column[SY][WPR],spread[SY][WPR];    // C and S of the header above morph_rects_kernel
      // rows centre-k (`up`) and centre+1+k (`down`) of three successive depths k: the previous
      // one, the one being folded in and the next one, whose reads are already under way;
      // depth 0: the output rows themselves
      Half up[3][2],down[3][2];
      auto read_depth=[&](int k,Half (&to_up)[2],Half (&to_down)[2])
      {
        const int above=wrapped(centre-k),below=wrapped(wrapped(centre+1+k));     // centre-k >= origin >= 0
        to_up[0]=*ring_at(above,0);
        to_up[1]=*ring_at(above,1);
        to_down[0]=*ring_at(below,0);
        to_down[1]=*ring_at(below,1);
      };
      read_depth(0,up[0],down[0]);
      if (vmax >= 1)
        read_depth(1,up[1],down[1]);
#pragma unroll
      for (int p=0; p < WPR; p++)
        {
          column[0][p]=up[0][p/HW][p%HW];
          column[1][p]=down[0][p/HW][p%HW];
          spread[0][p]=DILATE ? 0u : 0xffffffffu;
          spread[1][p]=DILATE ? 0u : 0xffffffffu;
        }
      // depth k: row 0 takes centre-k (new `up`) and centre+k (previous `down`), row 1 takes
      // centre+1-k (previous `up`) and centre+1+k (new `down`)
      auto fold=[&](const Half (&new_up)[2],const Half (&new_down)[2],const Half (&old_up)[2],const Half (&old_down)[2])
      {
#pragma unroll
        for (int p=0; p < WPR; p++)
          {
            column[0][p]=pk_pick<DILATE>(pk_pick<DILATE>(column[0][p],new_up[p/HW][p%HW]),old_down[p/HW][p%HW]);
            column[1][p]=pk_pick<DILATE>(pk_pick<DILATE>(column[1][p],old_up[p/HW][p%HW]),new_down[p/HW][p%HW]);
          }
      };
      // every level whose reach the fold depth has arrived at: merge the column windows into the
      // spread and widen it (the levels' reaches grow as l falls)
      int level=args.nlevels-1;
      auto settle=[&](int depth)
      {
        while ((level >= 0) && (__builtin_amdgcn_readlane(lane_reach,level) <= depth))
          {
#pragma unroll
            for (int i=0; i < SY; i++)
#pragma unroll
              for (int p=0; p < WPR; p++)
                spread[i][p]=pk_pick<DILATE>(spread[i][p],column[i][p]);
            const int widen=__builtin_amdgcn_readlane(lane_widen,level);
            for (int stride=0; stride < widen; stride++)
              {
                // Row(1) over the lane's columns a b c d and the neighbours' d' (left) and a' (right):
                //   a <- d' v (a v b),  b <- (a v b) v c,  c <- b v (c v d),  d <- (c v d) v a'
#pragma unroll
                for (int i=0; i < SY; i++)
#pragma unroll
                  for (int w=0; w < NW; w++)
                    {
                      const uint32_t a=spread[i][w],b=spread[i][NW+w],c=spread[i][2*NW+w],d=spread[i][3*NW+w];
                      // wave_shr:1 — lane n reads lane n-1 (0x138); wave_shl:1 — lane n reads lane n+1 (0x130)
                      const uint32_t left=(uint32_t) __builtin_amdgcn_mov_dpp((int) d,0x138,0xf,0xf,true);
                      const uint32_t right=(uint32_t) __builtin_amdgcn_mov_dpp((int) a,0x130,0xf,0xf,true);
                      const uint32_t ab=pk_pick<DILATE>(a,b),cd=pk_pick<DILATE>(c,d);
                      spread[i][w]=pk_pick<DILATE>(left,ab);
                      spread[i][NW+w]=pk_pick<DILATE>(ab,c);
                      spread[i][2*NW+w]=pk_pick<DILATE>(b,cd);
                      spread[i][3*NW+w]=pk_pick<DILATE>(cd,right);
                    }
              }
            level--;
          }
      };
      // the fold depths in threes, so that which register set holds which depth's rows is known at
      // compile time
      settle(0);
      for (int k=1; k <= vmax; k+=3)
        {
          // the previous step's rows leave here, apart from the burst of fetches at the step's start
          // (measured: 2.25 -> 2.10 ms against storing them first thing)
          if (k == kStoreDepth)
            store_results();
          if (k+1 <= vmax)
            read_depth(k+1,up[2],down[2]);
          fold(up[1],down[1],up[0],down[0]);
          settle(k);
          if (k+1 <= vmax)
            {
              if (k+2 <= vmax)
                read_depth(k+2,up[0],down[0]);
              fold(up[2],down[2],up[1],down[1]);
              settle(k+1);
            }
          if (k+2 <= vmax)
            {
              if (k+3 <= vmax)
                read_depth(k+3,up[1],down[1]);
              fold(up[0],down[0],up[2],down[2]);
              settle(k+2);
            }
        }

      if (vmax < kStoreDepth)
        store_results();
      // ---- copy out: morphology.c:3180-3196 (channels without the update trait keep the source
      // value; `changed` counts the updated samples that differ from the source)
#pragma unroll
      for (int i=0; i < SY; i++)
        {
          const int y=by+wave*SY+i;
          if (y >= H)
            break;
          if (DILATE && (args.copy_mask == 0u) && (args.changed == nullptr))
            {
              // every channel updated, nobody counts: the maxima are the result
#pragma unroll
              for (int p=0; p < WPR; p++)
                result[i][p]=spread[i][p];
            }
          else
            {
              uint32_t original[WPR];
              if (centred)
                {
                  // the output pixel is the centre of its own window
                  const Half a=*ring_at(wrapped(centre+i),0),b=*ring_at(wrapped(centre+i),1);
#pragma unroll
                  for (int p=0; p < HW; p++)
                    {
                      original[p]=a[p];
                      original[HW+p]=b[p];
                    }
                }
              else
                {
                  const unsigned char *in=base+(unsigned) y*row_bytes;
#pragma unroll
                  for (int j=0; j < SX; j++)
#pragma unroll
                    for (int w=0; w < NW; w++)
                      original[j*NW+w]=ok[j] ? *reinterpret_cast<const uint32_t *>(in+(unsigned) (bx+u0+j)*(unsigned) (C*sizeof(uint16_t))+4*w) : 0u;
                }
#pragma unroll
              for (int p=0; p < WPR; p++)
                {
                  // Erode starts from the output pixel itself (morphology.c:2905-2912)
                  const uint32_t value=DILATE ? spread[i][p] : pk_pick<false>(spread[i][p],original[p]);
                  const int c0=2*(p % NW);             // channels of this word's halves
                  uint32_t keep=0u;
                  keep|=((args.copy_mask >> c0) & 1u) != 0u ? 0x0000ffffu : 0u;
                  keep|=((args.copy_mask >> (c0+1)) & 1u) != 0u ? 0xffff0000u : 0u;
                  result[i][p]=(original[p] & keep) | (value & ~keep);
                  const uint32_t differs=(value ^ original[p]) & ~keep;
                  if (ok[p/NW])
                    changed+=((differs & 0xffffu) != 0u ? 1u : 0u)+((differs >> 16) != 0u ? 1u : 0u);
                }
            }
        }
      result_by=by;
      if (step+1 < step_end)
        {
          // rows that came through registers (nobody reads these ring rows during this step)
#pragma unroll
          for (int j=0; j < SY; j++)
            commit_row(wrapped(wrapped(origin+TH)+halo+wave+NWAVES*j),held[j]);
        }
      origin=wrapped(origin+TH);
    }
  store_results();
  if (args.changed != nullptr)
    {
      changed=wave_sum(changed);
      if ((lane == 0) && (changed != 0))
        atomicAdd(args.changed,(unsigned long long) changed);
    }
}

template<int C,bool DILATE>
static MhStatus launch_strips_typed(const StripsArgs &args,size_t lds,hipStream_t stream)
{
  constexpr int WAVES=12;
  const dim3 grid(8u*(unsigned) args.items_per_xcd),block(64*WAVES);
  MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&morph_strips_kernel<C,DILATE,WAVES>),
    hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  hipLaunchKernelGGL((morph_strips_kernel<C,DILATE,WAVES>),grid,block,lds,stream,args);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

static MhStatus try_rects(const View &src,const View &dst,bool dilate,const std::vector<int> &half,
  int cx,int dy_min,const Roles &roles,unsigned long long *changed,bool *handled);

// A one-channel (gray) Q16 frame — the masks Erode and Dilate are mostly run on — has no 4- or 8-byte pixel for
// the union-of-rectangles kernel's lanes and took morph_convex (Disk:15 on 8192^2: 1.35 ms, an RGBA frame 0.51).
// Its rows cut into four bands ARE the four channels of a frame a quarter as tall (gray_bands_pack_kernel,
// pointwise.hip: the kernel's reach in extra rows between bands, the frame's own edge rows repeated above the first
// and below the last as cache.c:2663-2679 does); minima and maxima are taken per channel, so the result is the
// frame's own.  The change count (morphology.c:3199) is taken while unpacking, over the frame's own samples.
static MhStatus try_rects_gray_bands(const View &src,const View &dst,bool dilate,const std::vector<int> &half,
  int cx,int dy_min,const Roles &roles,unsigned long long *changed,bool *handled)
{
  *handled=false;
  const int span=(int) half.size();
  const int above=dy_min < 0 ? -dy_min : 0,below=dy_min+span-1 > 0 ? dy_min+span-1 : 0;
  const int halo=above > below ? above : below;
  if ((roles.copy_mask != 0) || (halo < 1) || !GrayBands::fits(src,halo,1l << 18))
    return MH_OK;
  GrayBands bands;
  MH_TRY(bands.pack(src,halo));
  bool inner=false;
  MH_TRY(try_rects(bands.packed,bands.result,dilate,half,cx,dy_min,GrayBands::plain_roles(),nullptr,&inner));
  if (!inner)
    return MH_OK;
  MH_TRY(bands.unpack(dst,src.pixels,changed));
  *handled=true;
  return MH_OK;
}

// A three-channel frame (RGB without alpha: what most photographs are) has 6- or 12-byte pixels and took morph_convex
// (Dilate Disk:15 on 8192^2 Q16: 4.05 ms, an RGBA frame 0.53).  With a fourth, empty channel it is an ordinary
// four-channel frame (rgb_pad_kernel, pointwise.hip); the change count is taken while dropping that channel again.
static MhStatus try_rects_rgb_padded(const View &src,const View &dst,bool dilate,const std::vector<int> &half,
  int cx,int dy_min,const Roles &roles,unsigned long long *changed,bool *handled)
{
  *handled=false;
  if (src.columns*src.rows < (size_t) option_long("MAGICKHIP_RGB_PAD_MIN_PIXELS",1l << 16))
    return MH_OK;
  View padded=src,result=src;
  padded.channels=result.channels=4;
  Temp padded_memory,result_memory;
  MH_TRY(padded_memory.alloc(src.device,padded.bytes(),src.stream));
  MH_TRY(result_memory.alloc(src.device,result.bytes(),src.stream));
  padded.pixels=padded_memory.ptr;
  result.pixels=result_memory.ptr;
  MH_TRY(launch_rgb_pad(src,padded));
  Roles four=roles;
  four.blend=false;
  four.alpha=-1;
  four.copy_mask=roles.copy_mask & 0x7u;
  four.update_mask=0xfu & ~four.copy_mask;
  bool inner=false;
  MH_TRY(try_rects(padded,result,dilate,half,cx,dy_min,four,nullptr,&inner));
  if (!inner)
    return MH_OK;
  MH_TRY(launch_rgb_unpad(result,dst,src.pixels,changed));
  *handled=true;
  return MH_OK;
}

// half[k]: half-width of kernel row dy_min+k (every row non-empty, runs centred on cx).  Handles
// the kernel when its rows are symmetric about the middle row and do not widen away from it.
static MhStatus try_rects(const View &src,const View &dst,bool dilate,const std::vector<int> &half,
  int cx,int dy_min,const Roles &roles,unsigned long long *changed,bool *handled)
{
  *handled=false;
  const int span=(int) half.size();
  const bool is_float=src.quantum != MH_QUANTUM_U16;
  if (!is_float && (src.channels == 1) && (option("MAGICKHIP_NO_RECTS") == nullptr) &&
      (option("MAGICKHIP_NO_GRAY_BANDS") == nullptr))
    return try_rects_gray_bands(src,dst,dilate,half,cx,dy_min,roles,changed,handled);
  if ((src.channels == 3) && (option("MAGICKHIP_NO_RECTS") == nullptr) && (option("MAGICKHIP_NO_RGB_PAD") == nullptr) &&
      (!is_float || (option("MAGICKHIP_NO_FLOAT_RECTS") == nullptr)))
    {
      // Q16: faster than morph_convex at every size and kernel measured (8192^2 Disk:15 4.02 -> 0.85 ms, Square:1 0.67
      // -> 0.60; 256^2 0.051 -> 0.039).  Float: padding moves 28 bytes a pixel twice (0.70 ms per 8192^2) and
      // morph_convex is quick on its small kernels (Disk:5 0.88 ms, padded 1.24) — only where its tile no longer fits
      // (try_convex: 80 KB; from Disk:8 on) and the frame would take the generic kernel (Disk:15: 15.2 ms, padded 1.97)
      const int v=span/2,h=half[(size_t) v];
      if (!is_float || option("MAGICKHIP_RGB_PAD_FLOAT_ALWAYS") != nullptr ||
          ((size_t) (kCTR+2*v)*(size_t) (2*kCTW+2*h)*12u > 80u*1024u))
        return try_rects_rgb_padded(src,dst,dilate,half,cx,dy_min,roles,changed,handled);
    }
  const bool layout=is_float ? ((src.channels == 1) || (src.channels == 2) || (src.channels == 4)) :
    ((src.channels == 2) || (src.channels == 4));
  if (!layout || ((span & 1) == 0) || (option("MAGICKHIP_NO_RECTS") != nullptr) ||
      (is_float && (option("MAGICKHIP_NO_FLOAT_RECTS") != nullptr)))
    return MH_OK;
  const int vmax=span/2;
  for (int d=0; d <= vmax; d++)
    {
      if (half[(size_t) (vmax+d)] != half[(size_t) (vmax-d)])
        return MH_OK;
      if ((d > 0) && (half[(size_t) (vmax+d)] > half[(size_t) (vmax+d-1)]))
        return MH_OK;
    }
  const int hmax=half[(size_t) vmax];
  // a lane holds one (16-byte pixels) or two columns, a wave 64 or 128 (morph_rects_kernel)
  const size_t pixel_bytes=(size_t) src.channels*(is_float ? sizeof(float) : sizeof(uint16_t));
  const int lane_columns=pixel_bytes >= 16 ? 1 : 2;
  const int wave_columns=64*lane_columns;
  const size_t row_lds=(size_t) wave_columns*pixel_bytes;
  int shape=1;
  if (const char *e=option("MAGICKHIP_RECTS_SHAPE"))
    shape=(atoi(e) >= 0) && (atoi(e) <= 3) ? atoi(e) : 0;
  if (is_float)
    shape=1;
  const int th=shape == 2 ? 96 : 48;
  const size_t lds=(size_t) (th+2*vmax)*row_lds;
  if ((wave_columns-2*hmax < 16) || (lds > 160u*1024u))
    return MH_OK;
  RectsArgs a;
  // levels: distinct half-widths ascending; reach = the outermost row at least that wide
  std::vector<int> widths(half.begin()+vmax,half.end());
  std::sort(widths.begin(),widths.end());
  widths.erase(std::unique(widths.begin(),widths.end()),widths.end());
  if (widths.size() > 64)
    return MH_OK;
  a.nlevels=(int) widths.size();
  for (size_t l=0; l < widths.size(); l++)
    {
      int reach=0;
      for (int d=0; d <= vmax; d++)
        if (half[(size_t) (vmax+d)] >= widths[l])
          reach=d;
      a.widen[l]=(unsigned char) (widths[l]-(l == 0 ? 0 : widths[l-1]));
      a.reach[l]=(unsigned char) reach;
    }
  a.src=src.pixels;
  a.dst=dst.pixels;
  a.columns=(int) src.columns;
  a.rows=(int) src.rows;
  a.cx=cx;
  a.cy=dy_min+vmax;
  a.hmax=hmax;
  a.vmax=vmax;
  a.copy_mask=roles.copy_mask;
  a.changed=changed;
  {
    // frames of several strips by several steps: the walk down 256-column strips.  Opt-in
    // (MAGICKHIP_STRIPS=1): on MI355X it reads the frame 1.13 times instead of 1.31 but takes
    // 2.10 ms where the tile kernel takes 1.95 (16384^2 RGBA, Disk:15) — three waves a SIMD hide
    // less than the tile kernel's six (profiles/r3_notes/dilate_experiments.txt).
    constexpr int kStripRows=24;
    const size_t strip_lds=(size_t) (2*kStripRows+2*vmax)*2u*row_lds;
    const int strip_w=256-2*hmax;
    if (!is_float && ((unsigned long long) src.columns*src.rows*pixel_bytes < (1ull << 32)) &&
        (strip_lds <= 160u*1024u) && ((int) src.columns >= 2*strip_w) &&
        ((int) src.rows >= 4*kStripRows) && (option("MAGICKHIP_STRIPS") != nullptr))
      {
        StripsArgs sa;
        sa.r=a;
        sa.r.tiles_x=((int) src.columns+strip_w-1)/strip_w;
        sa.r.tiles_y=((int) src.rows+kStripRows-1)/kStripRows;
        // cuts of a strip: the schedule (one workgroup per CU, 256 CUs) that finishes first; a
        // cut costs its 2*vmax rows of halo
        int best=1;
        double best_cost=1.0e300;
        for (int cuts=1; cuts <= sa.r.tiles_y; cuts++)
          {
            const int steps=(sa.r.tiles_y+cuts-1)/cuts;
            const int rounds=(sa.r.tiles_x*((sa.r.tiles_y+steps-1)/steps)+255)/256;
            const double cost=(double) rounds*((double) (steps+1)*kStripRows+2.0*vmax);
            if (cost < best_cost-1.0e-9)
              {
                best_cost=cost;
                best=cuts;
              }
          }
        if (const char *e=option("MAGICKHIP_STRIP_CUTS"))         // tests: walks of several steps on small frames
          best=(atoi(e) >= 1) && (atoi(e) <= sa.r.tiles_y) ? atoi(e) : best;
        sa.steps_per_segment=(sa.r.tiles_y+best-1)/best;
        sa.segments=(sa.r.tiles_y+sa.steps_per_segment-1)/sa.steps_per_segment;
        sa.items_per_xcd=(sa.r.tiles_x*sa.segments+7)/8;
        sa.r.tiles_per_xcd=sa.items_per_xcd;
        ProfileScope prof("morph_rects",src.stream);
        if (src.channels == 4)
          MH_TRY((dilate ? launch_strips_typed<4,true>(sa,strip_lds,src.stream) : launch_strips_typed<4,false>(sa,strip_lds,src.stream)));
        else
          MH_TRY((dilate ? launch_strips_typed<2,true>(sa,strip_lds,src.stream) : launch_strips_typed<2,false>(sa,strip_lds,src.stream)));
        *handled=true;
        return MH_OK;
      }
  }
  const int valid_w=wave_columns-2*hmax;
  a.tiles_x=((int) src.columns+valid_w-1)/valid_w;
  a.tiles_y=((int) src.rows+th-1)/th;
  a.tiles_per_xcd=(a.tiles_x*a.tiles_y+7)/8;
  a.copy_mask=roles.copy_mask;
  a.changed=changed;
  ProfileScope prof("morph_rects",src.stream);
  if (is_float)
    {
      if (src.channels == 4)
        MH_TRY((launch_rects<float,4>(dilate,shape,a,lds,src.stream)));
      else if (src.channels == 2)
        MH_TRY((launch_rects<float,2>(dilate,shape,a,lds,src.stream)));
      else
        MH_TRY((launch_rects<float,1>(dilate,shape,a,lds,src.stream)));
    }
  else if (src.channels == 4)
    MH_TRY((launch_rects<uint16_t,4>(dilate,shape,a,lds,src.stream)));
  else
    MH_TRY((launch_rects<uint16_t,2>(dilate,shape,a,lds,src.stream)));
  *handled=true;
  return MH_OK;
}

// Tries the convex fast path; *handled stays false when the kernel's active cells are
// not one centred run per row (Ring, Cross, user kernels ...) or the tile does not fit.
static MhStatus try_convex(const View &src,const View &dst,bool dilate,const std::vector<Cell> &cells,
  const Roles &roles,unsigned long long *changed,bool *handled)
{
  *handled=false;
  if (cells.size() < 9)                       // tiny kernels: the cell walk is already cheap
    return MH_OK;
  int dy_min=0x7fffffff,dy_max=-0x7fffffff;
  for (const Cell &c : cells)
    {
      dy_min=c.dy < dy_min ? c.dy : dy_min;
      dy_max=c.dy > dy_max ? c.dy : dy_max;
    }
  const int span=dy_max-dy_min+1;
  std::vector<int> lo((size_t) span,0x7fffffff),hi((size_t) span,-0x7fffffff),cnt((size_t) span,0);
  for (const Cell &c : cells)
    {
      size_t k=(size_t) (c.dy-dy_min);
      lo[k]=c.dx < lo[k] ? c.dx : lo[k];
      hi[k]=c.dx > hi[k] ? c.dx : hi[k];
      cnt[k]++;
    }
  int centre2=0x7fffffff,hmax=0;
  for (int k=0; k < span; k++)
    {
      if (cnt[(size_t) k] == 0)
        continue;
      if (cnt[(size_t) k] != hi[(size_t) k]-lo[(size_t) k]+1)
        return MH_OK;                                        // a gap in the row
      int c2=lo[(size_t) k]+hi[(size_t) k];
      if (centre2 == 0x7fffffff)
        centre2=c2;
      if ((c2 != centre2) || ((c2 & 1) != 0))
        return MH_OK;                                        // runs are not about one centre column
      int h=(hi[(size_t) k]-lo[(size_t) k])/2;
      hmax=h > hmax ? h : hmax;
    }
  const int cx=centre2/2;
  const int top=dy_min < 0 ? -dy_min : 0,bottom=dy_max > 0 ? dy_max : 0;
  if ((dy_min > 0) || (dy_max < 0) || (hmax > 48) || (span > 97))
    return MH_OK;
  {
    // symmetric shapes: the union-of-rectangles kernel
    std::vector<int> half((size_t) span,-1);
    bool full=true;
    for (int k=0; k < span; k++)
      {
        full=full && (cnt[(size_t) k] != 0);
        half[(size_t) k]=(hi[(size_t) k]-lo[(size_t) k])/2;
      }
    if (full && (src.columns < (1u << 30)) && (src.rows < (1u << 30)))
      {
        MH_TRY(try_rects(src,dst,dilate,half,cx,dy_min,roles,changed,handled));
        if (*handled)
          return MH_OK;
      }
  }
  const size_t px=(size_t) src.channels*(src.quantum == MH_QUANTUM_U16 ? 2u : 4u);
  const size_t tile_rows=(size_t) (kCTR+top+bottom);
  const size_t lds=tile_rows*((size_t) (kCTW+2*hmax)+(size_t) kCTW)*px;
  if (lds > 80u*1024u)                                       // keep two workgroups per CU
    return MH_OK;
  // distinct half-widths ascending, kernel rows grouped by level
  std::vector<int> level_h;
  for (int k=0; k < span; k++)
    if (cnt[(size_t) k] != 0)
      level_h.push_back((hi[(size_t) k]-lo[(size_t) k])/2);
  std::sort(level_h.begin(),level_h.end());
  level_h.erase(std::unique(level_h.begin(),level_h.end()),level_h.end());
  std::vector<int> level_first,level_dy;
  for (size_t l=0; l < level_h.size(); l++)
    {
      level_first.push_back((int) level_dy.size());
      for (int k=0; k < span; k++)
        if ((cnt[(size_t) k] != 0) && ((hi[(size_t) k]-lo[(size_t) k])/2 == level_h[l]))
          level_dy.push_back(k+dy_min);
    }
  level_first.push_back((int) level_dy.size());
  Temp d_h,d_first,d_dy;
  MH_TRY(upload_table(d_h,src.device,src.stream,level_h.data(),level_h.size()*sizeof(int)));
  MH_TRY(upload_table(d_first,src.device,src.stream,level_first.data(),level_first.size()*sizeof(int)));
  MH_TRY(upload_table(d_dy,src.device,src.stream,level_dy.data(),level_dy.size()*sizeof(int)));
  ConvexArgs a;
  a.src=src.pixels;
  a.dst=dst.pixels;
  a.columns=(int) src.columns;
  a.rows=(int) src.rows;
  a.cx=cx;
  a.hmax=hmax;
  a.top=top;
  a.bottom=bottom;
  a.nlevels=(int) level_h.size();
  a.level_h=d_h.as<int>();
  a.level_first=d_first.as<int>();
  a.level_dy=d_dy.as<int>();
  a.copy_mask=roles.copy_mask;
  a.changed=changed;
  dim3 grid((unsigned) ((src.columns+kCTW-1)/kCTW),(unsigned) ((src.rows+kCTR-1)/kCTR));
  ProfileScope prof("morph_convex",src.stream);
  MhStatus st=MH_OK;
#define MH_CONVEX(QT) \
  switch (src.channels) \
  { \
    case 1: st=launch_convex<QT,1>(dilate,a,grid,lds,src.stream); break; \
    case 2: st=launch_convex<QT,2>(dilate,a,grid,lds,src.stream); break; \
    case 3: st=launch_convex<QT,3>(dilate,a,grid,lds,src.stream); break; \
    case 4: st=launch_convex<QT,4>(dilate,a,grid,lds,src.stream); break; \
    default: return MH_OK; \
  }
  if (src.quantum == MH_QUANTUM_U16)
    { MH_CONVEX(uint16_t) }
  else
    { MH_CONVEX(float) }
#undef MH_CONVEX
  MH_TRY(st);
  *handled=true;
  return MH_OK;
}

MhStatus launch_morph2d(const View &src,const View &dst,const Morph2DParams &params,
  const Roles &roles,unsigned long long *changed)
{
  const MhKernelInfo *k=params.kernel;
  if ((src.columns != dst.columns) || (src.rows != dst.rows) ||
      (src.channels != dst.channels) || (src.quantum != dst.quantum))
    return fail(MH_BAD_ARGUMENT,"morphology: source/destination geometry mismatch");
  if (roles.blend && (roles.alpha != src.channels-1))
    return fail(MH_UNSUPPORTED,"alpha channel must be the last channel");
  const int w=(int) k->width,h=(int) k->height;
  if ((w < 1) || (h < 1) || (k->x < 0) || (k->y < 0) || (k->x >= w) || (k->y >= h))
    return fail(MH_BAD_ARGUMENT,"morphology: bad kernel geometry");
  int mc;
  bool reflected;
  int hmt_mode=0;
  switch (params.method)
  {
    case MH_MORPHOLOGY_CONVOLVE: mc=MC_CONVOLVE; reflected=true; break;
    case MH_MORPHOLOGY_DILATE: mc=MC_DILATE; reflected=true; break;
    case MH_MORPHOLOGY_DILATE_INTENSITY: mc=MC_DILATE_INTENSITY; reflected=true; break;
    case MH_MORPHOLOGY_ITERATIVE_DISTANCE: mc=MC_DISTANCE; reflected=true; break;
    case MH_MORPHOLOGY_ERODE: mc=MC_ERODE; reflected=false; break;
    case MH_MORPHOLOGY_ERODE_INTENSITY: mc=MC_ERODE_INTENSITY; reflected=false; break;
    case MH_MORPHOLOGY_HIT_AND_MISS: mc=MC_HMT; reflected=false; hmt_mode=0; break;
    case MH_MORPHOLOGY_THINNING: mc=MC_HMT; reflected=false; hmt_mode=1; break;
    case MH_MORPHOLOGY_THICKEN: mc=MC_HMT; reflected=false; hmt_mode=2; break;
    default:
      return fail(MH_BAD_ARGUMENT,"not a primitive morphology method");
  }
  const bool linear=(params.colorspace == MH_COLORSPACE_RGB) ||
    (params.colorspace == MH_COLORSPACE_LINEARGRAY);
  const bool nonlinear=(params.colorspace == MH_COLORSPACE_SRGB) ||
    (params.colorspace == MH_COLORSPACE_GRAY);
  if ((mc == MC_ERODE_INTENSITY) || (mc == MC_DILATE_INTENSITY))
    {
      // the gamma-dependent intensity methods are not evaluated in this kernel
      bool needs_gamma=false;
      switch (params.intensity)
      {
        case MH_INTENSITY_REC601LUMA: case MH_INTENSITY_REC709LUMA: case MH_INTENSITY_UNDEFINED:
          needs_gamma=linear; break;
        case MH_INTENSITY_REC601LUMINANCE: case MH_INTENSITY_REC709LUMINANCE:
          needs_gamma=true; break;
        default: break;
      }
      if (needs_gamma)
        return fail(MH_UNSUPPORTED,"intensity method needs a gamma transform");
    }
  // origin offsets, morphology.c:2623-2637
  const int ox=reflected ? w-(int) k->x-1 : (int) k->x;
  const int oy=reflected ? h-(int) k->y-1 : (int) k->y;
  std::vector<Cell> cells;
  cells.reserve((size_t) w*h);
  size_t non_nan=0;
  for (int v=0; v < h; v++)
    for (int u=0; u < w; u++)
      {
        double value=reflected ? k->values[(size_t) (h-1-v)*w+(size_t) (w-1-u)] :
          k->values[(size_t) v*w+(size_t) u];
        if (std::isnan(value))
          continue;
        non_nan++;
        bool keep=true;
        switch (mc)
        {
          case MC_ERODE: case MC_ERODE_INTENSITY: case MC_DILATE_INTENSITY: keep=value >= 0.5; break;
          case MC_DILATE: keep=value > 0.5; break;
          case MC_HMT: keep=(value > 0.7) || (value < 0.3); break;
          default: break;
        }
        if (!keep)
          continue;
        Cell c;
        c.dx=u-ox;
        c.dy=v-oy;
        c.value=value;
        cells.push_back(c);
      }
  if (((mc == MC_ERODE) || (mc == MC_DILATE)) && (option("MAGICKHIP_NO_CONVEX") == nullptr))
    {
      bool handled=false;
      MH_TRY(try_convex(src,dst,mc == MC_DILATE,cells,roles,changed,&handled));
      if (handled)
        return MH_OK;
    }
  Morph2DArgs args;
  args.src=src.pixels;
  args.dst=dst.pixels;
  args.columns=(int) src.columns;
  args.rows=(int) src.rows;
  args.ncells=(int) cells.size();
  args.left=ox;
  args.top=oy;
  args.tile_w=kTW+w-1;
  args.tile_h=kTH+h-1;
  args.bias=params.bias;
  args.rescale=1.0;
  if ((mc == MC_CONVOLVE) && (w == 1) && (non_nan != 0))
    args.rescale=(double) h/(double) non_nan;
  args.copy_mask=roles.copy_mask;
  args.only_if=params.only_if;
  args.hmt_mode=hmt_mode;
  args.linear=linear;
  args.nonlinear=nonlinear;
  args.gray=(params.colorspace == MH_COLORSPACE_GRAY) ||
    (params.colorspace == MH_COLORSPACE_LINEARGRAY) || (src.channels < 3);
  args.intensity_method=(int) params.intensity;
  args.changed=changed;
  Temp d_cells;
  Cell dummy{0,0,0.0};
  MH_TRY(upload_table(d_cells,src.device,src.stream,cells.empty() ? &dummy : cells.data(),
    (cells.empty() ? 1 : cells.size())*sizeof(Cell)));
  args.cells=d_cells.as<Cell>();
  const size_t px=(size_t) src.channels*(src.quantum == MH_QUANTUM_U16 ? 2u : 4u);
  const size_t lds=(size_t) args.tile_w*args.tile_h*px;
  if (lds > 160u*1024u)
    return fail(MH_UNSUPPORTED,"%dx%d kernel needs %zu bytes of LDS",w,h,lds);
  dim3 grid((unsigned) ((src.columns+kTW-1)/kTW),(unsigned) ((src.rows+kTH-1)/kTH));
  ProfileScope prof("morph2d",src.stream);
  if (src.quantum == MH_QUANTUM_U16)
    return launch_channels<uint16_t>(src.channels,roles.blend,mc,args,grid,lds,src.stream);
  return launch_channels<float>(src.channels,roles.blend,mc,args,grid,lds,src.stream);
}

// ---------------------------------------------------------------- MotionBlurImage
// effect.c:2347-2560: a 1-D kernel applied along a line of integer offsets, virtual pixels
// edge-clamped, alpha-weighted on Blend channels; fp64 in the reference's order
// (pixel += (k*alpha)*r; gamma += k*alpha).
struct MotionArgs
{
  const void *src;
  void *dst;
  int columns,rows;
  int width;
  const double *kernel;        // [width]
  const int *offset;           // [width][2] = x,y
  uint32_t copy_mask;
  int alpha;                   // alpha channel or -1
};

template<typename Q,int C,bool BLEND>
__global__ __launch_bounds__(256)
void motion_blur_kernel(MotionArgs a)
{
  const int x=(int) (blockIdx.x*blockDim.x+threadIdx.x),y=(int) blockIdx.y;
  if (x >= a.columns)
    return;
  const Q *src=static_cast<const Q *>(a.src);
  double pixel[C],gamma=0.0;
#pragma unroll
  for (int c=0; c < C; c++)
    pixel[c]=0.0;
  for (int j=0; j < a.width; j++)
    {
      int sx=x+a.offset[2*j],sy=y+a.offset[2*j+1];
      sx=sx < 0 ? 0 : (sx > a.columns-1 ? a.columns-1 : sx);      // cache.c:2663-2679
      sy=sy < 0 ? 0 : (sy > a.rows-1 ? a.rows-1 : sy);
      Q r[C];
      load_pixel<Q,C>(src+((size_t) sy*a.columns+sx)*C,r);
      const double k=a.kernel[j];
      if (BLEND)
        {
          const double alpha=kQS*(double) r[C-1];
          const double weight=k*alpha;
#pragma unroll
          for (int c=0; c < C-1; c++)
            pixel[c]+=weight*(double) r[c];
          gamma+=weight;
          pixel[C-1]+=k*(double) r[C-1];
        }
      else
        {
#pragma unroll
          for (int c=0; c < C; c++)
            pixel[c]+=k*(double) r[c];
        }
    }
  Q centre[C],out[C];
  load_pixel<Q,C>(src+((size_t) y*a.columns+x)*C,centre);
  const double g=BLEND ? perceptible_reciprocal(gamma) : 1.0;
#pragma unroll
  for (int c=0; c < C; c++)
    {
      if ((a.copy_mask >> c) & 1u)
        out[c]=centre[c];
      else if (BLEND && (c != C-1))
        out[c]=QuantumOps<Q>::clamp(g*pixel[c]);
      else
        out[c]=QuantumOps<Q>::clamp(pixel[c]);
    }
  store_pixel<Q,C>(static_cast<Q *>(a.dst)+((size_t) y*a.columns+x)*C,out);
}

template<typename Q>
static MhStatus motion_typed(const View &src,bool blend,const MotionArgs &a)
{
  dim3 grid((unsigned) ((a.columns+255)/256),(unsigned) a.rows),block(256);
  ProfileScope prof("motion_blur",src.stream);
  switch (src.channels)
  {
    case 1: hipLaunchKernelGGL((motion_blur_kernel<Q,1,false>),grid,block,0,src.stream,a); break;
    case 2:
      if (blend) hipLaunchKernelGGL((motion_blur_kernel<Q,2,true>),grid,block,0,src.stream,a);
      else hipLaunchKernelGGL((motion_blur_kernel<Q,2,false>),grid,block,0,src.stream,a);
      break;
    case 3: hipLaunchKernelGGL((motion_blur_kernel<Q,3,false>),grid,block,0,src.stream,a); break;
    default:
      if (blend) hipLaunchKernelGGL((motion_blur_kernel<Q,4,true>),grid,block,0,src.stream,a);
      else hipLaunchKernelGGL((motion_blur_kernel<Q,4,false>),grid,block,0,src.stream,a);
      break;
  }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

MhStatus launch_motion_blur(const View &src,const View &dst,const double *kernel,size_t width,
  const ptrdiff_t *offsets_xy,const Roles &roles)
{
  if ((width == 0) || (width > 65536))
    return fail(MH_BAD_ARGUMENT,"motion blur: bad kernel width");
  if (roles.blend && (roles.alpha != src.channels-1))
    return fail(MH_UNSUPPORTED,"alpha channel must be the last channel");
  std::vector<int> off(2*width);
  for (size_t j=0; j < 2*width; j++)
    off[j]=(int) offsets_xy[j];
  Temp d_kernel,d_offset;
  MH_TRY(upload_table(d_kernel,src.device,src.stream,kernel,width*sizeof(double)));
  MH_TRY(upload_table(d_offset,src.device,src.stream,off.data(),off.size()*sizeof(int)));
  MotionArgs a;
  a.src=src.pixels;
  a.dst=dst.pixels;
  a.columns=(int) src.columns;
  a.rows=(int) src.rows;
  a.width=(int) width;
  a.kernel=d_kernel.as<double>();
  a.offset=d_offset.as<int>();
  a.copy_mask=roles.copy_mask;
  a.alpha=roles.alpha;
  if (src.quantum == MH_QUANTUM_U16)
    return motion_typed<uint16_t>(src,roles.blend,a);
  return motion_typed<float>(src,roles.blend,a);
}

// ---------------------------------------------------------------- RotationalBlurImage
// effect.c:3209-3430: every pixel is the (alpha-weighted) mean of samples on the arc through it
// around the image centre; the sample stride grows towards the centre.  cos/sin tables come
// from the host (libm, as the reference); the coordinate expressions, the (ssize_t) casts and
// the accumulation order are the reference's.
struct RotationalArgs
{
  const void *src;
  void *dst;
  int columns,rows;
  int n;
  const double *cos_theta,*sin_theta;
  double center_x,center_y,blur_radius;
  uint32_t copy_mask;
  int has_alpha;               // the image has an alpha channel (last channel)
};

template<typename Q,int C>
__global__ __launch_bounds__(256)
void rotational_blur_kernel(RotationalArgs a)
{
  const int x=(int) (blockIdx.x*blockDim.x+threadIdx.x),y=(int) blockIdx.y;
  if (x >= a.columns)
    return;
  const Q *src=static_cast<const Q *>(a.src);
  const double cx=(double) x-a.center_x,cy=(double) y-a.center_y;
  // hypot of two half-integers: x*x+y*y is exact in fp64, so sqrt is the correctly rounded hypot
  const double radius=sqrt(cx*cx+cy*cy);
  size_t step=1;
  if (radius != 0)
    {
      step=(size_t) (a.blur_radius/radius);
      if (step == 0)
        step=1;
      else if (step >= (size_t) a.n)
        step=(size_t) a.n-1;
    }
  double plain[C],weighted[C],gamma_plain=0.0,gamma_alpha=0.0;
#pragma unroll
  for (int c=0; c < C; c++)
    {
      plain[c]=0.0;
      weighted[c]=0.0;
    }
  for (size_t j=0; j < (size_t) a.n; j+=step)
    {
      long sx=(long) (a.center_x+cx*a.cos_theta[j]-cy*a.sin_theta[j]+0.5);
      long sy=(long) (a.center_y+cx*a.sin_theta[j]+cy*a.cos_theta[j]+0.5);
      sx=sx < 0 ? 0 : (sx > a.columns-1 ? a.columns-1 : sx);      // cache.c:2663-2679
      sy=sy < 0 ? 0 : (sy > a.rows-1 ? a.rows-1 : sy);
      Q r[C];
      load_pixel<Q,C>(src+((size_t) sy*a.columns+(size_t) sx)*C,r);
      const double alpha=a.has_alpha ? kQS*(double) r[C-1] : 1.0;
#pragma unroll
      for (int c=0; c < C; c++)
        {
          plain[c]+=(double) r[c];
          weighted[c]+=alpha*(double) r[c];
        }
      gamma_plain+=1.0;
      gamma_alpha+=alpha;
    }
  Q centre[C],out[C];
  load_pixel<Q,C>(src+((size_t) y*a.columns+x)*C,centre);
  const double gp=perceptible_reciprocal(gamma_plain),ga=perceptible_reciprocal(gamma_alpha);
#pragma unroll
  for (int c=0; c < C; c++)
    {
      if ((a.copy_mask >> c) & 1u)
        out[c]=centre[c];
      else if (!a.has_alpha || (c == C-1))
        out[c]=QuantumOps<Q>::clamp(gp*plain[c]);
      else
        out[c]=QuantumOps<Q>::clamp(ga*weighted[c]);
    }
  store_pixel<Q,C>(static_cast<Q *>(a.dst)+((size_t) y*a.columns+x)*C,out);
}

MhStatus launch_rotational_blur(const View &src,const View &dst,const double *cos_theta,
  const double *sin_theta,size_t n,double blur_radius,const Roles &roles)
{
  Temp d_cos,d_sin;
  MH_TRY(upload_table(d_cos,src.device,src.stream,cos_theta,n*sizeof(double)));
  MH_TRY(upload_table(d_sin,src.device,src.stream,sin_theta,n*sizeof(double)));
  RotationalArgs a;
  a.src=src.pixels;
  a.dst=dst.pixels;
  a.columns=(int) src.columns;
  a.rows=(int) src.rows;
  a.n=(int) n;
  a.cos_theta=d_cos.as<double>();
  a.sin_theta=d_sin.as<double>();
  a.center_x=(double) (src.columns-1)/2.0;
  a.center_y=(double) (src.rows-1)/2.0;
  a.blur_radius=blur_radius;
  a.copy_mask=roles.copy_mask;
  a.has_alpha=roles.alpha >= 0 ? 1 : 0;
  if ((roles.alpha >= 0) && (roles.alpha != src.channels-1))
    return fail(MH_UNSUPPORTED,"alpha channel must be the last channel");
  dim3 grid((unsigned) ((a.columns+255)/256),(unsigned) a.rows),block(256);
  ProfileScope prof("rotational_blur",src.stream);
#define MH_CASE(QT) \
  switch (src.channels) { \
    case 1: hipLaunchKernelGGL((rotational_blur_kernel<QT,1>),grid,block,0,src.stream,a); break; \
    case 2: hipLaunchKernelGGL((rotational_blur_kernel<QT,2>),grid,block,0,src.stream,a); break; \
    case 3: hipLaunchKernelGGL((rotational_blur_kernel<QT,3>),grid,block,0,src.stream,a); break; \
    default: hipLaunchKernelGGL((rotational_blur_kernel<QT,4>),grid,block,0,src.stream,a); break; }
  if (src.quantum == MH_QUANTUM_U16)
    { MH_CASE(uint16_t) }
  else
    { MH_CASE(float) }
#undef MH_CASE
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// ---------------------------------------------------------------- LocalContrastImage
// effect.c:1760-2010, literally: a vertical then a horizontal pass of the same (lopsided)
// triangular weighting — weights 1..w on the first w samples, then w+1, w, ..., 3 on the
// next w-1 — over the float luma, with a float intermediate image of columns+2w whose
// padding mirrors the neighbouring columns; then R,G,B are scaled by
// (luma + (luma-blurred)*strength/100)/luma.
struct LocalContrastArgs
{
  const void *src;
  void *dst;
  float *inter;                // rows x (columns+2w)
  int columns,rows,channels,w;
  double total_weight,strength_scale;
  uint32_t copy_mask;
  int colour;                  // 3 for R,G,B[,A], 1 for gray[,A]
};

template<typename Q>
static __device__ __forceinline__ float luma_of(const Q *p,int colour)
{
  // GetPixelLuma, pixel-accessor.h:304-315, narrowed to float as the scanline buffer does
  const double r=(double) p[0],g=(double) p[colour >= 3 ? 1 : 0],b=(double) p[colour >= 3 ? 2 : 0];
  return (float) (0.212656*r+0.715158*g+0.072186*b);
}

static __device__ __forceinline__ double lopsided_sum(const float *pix,long stride,int w)
{
  double weight=1.0,sum=0;
  for (int i=0; i < w; i++)
    {
      sum+=weight*((double) *pix);
      pix+=stride;
      weight+=1.0;
    }
  for (int i=w+1; i < (2*w); i++)
    {
      sum+=weight*((double) *pix);
      pix+=stride;
      weight-=1.0;
    }
  return sum;
}

template<typename Q>
__global__ __launch_bounds__(256)
void local_contrast_luma_kernel(LocalContrastArgs a,float *luma)       // luma: (rows+2w) x columns
{
  const int x=(int) (blockIdx.x*blockDim.x+threadIdx.x),yy=(int) blockIdx.y;
  if (x >= a.columns)
    return;
  int y=yy-a.w;
  y=y < 0 ? 0 : (y > a.rows-1 ? a.rows-1 : y);                        // cache.c:2663-2679
  luma[(size_t) yy*a.columns+x]=luma_of<Q>(static_cast<const Q *>(a.src)+((size_t) y*a.columns+x)*a.channels,a.colour);
}

__global__ __launch_bounds__(256)
void local_contrast_vertical_kernel(LocalContrastArgs a,const float *luma)
{
  const int x=(int) (blockIdx.x*blockDim.x+threadIdx.x),y=(int) blockIdx.y;
  if (x >= a.columns)
    return;
  const size_t pitch=(size_t) a.columns+2*(size_t) a.w;
  const double sum=lopsided_sum(luma+(size_t) y*a.columns+x,a.columns,a.w);
  float *out=a.inter+(size_t) y*pitch+(size_t) x+(size_t) a.w;
  const float value=(float) (sum/a.total_weight);
  *out=value;
  if ((x <= a.w) && (x != 0))                                          // mirror into the padding
    *(out-(x*2))=value;
  if ((x > a.columns-a.w-2) && (x != a.columns-1))
    *(out+((size_t) (a.columns-x-1)*2))=value;
}

template<typename Q>
__global__ __launch_bounds__(256)
void local_contrast_horizontal_kernel(LocalContrastArgs a)
{
  const int x=(int) (blockIdx.x*blockDim.x+threadIdx.x),y=(int) blockIdx.y;
  if (x >= a.columns)
    return;
  const size_t pitch=(size_t) a.columns+2*(size_t) a.w;
  const double sum=lopsided_sum(a.inter+(size_t) y*pitch+x,1,a.w);
  const Q *p=static_cast<const Q *>(a.src)+((size_t) y*a.columns+x)*a.channels;
  Q *q=static_cast<Q *>(a.dst)+((size_t) y*a.columns+x)*a.channels;
  const double src_val=(double) luma_of<Q>(p,a.colour);
  double mult=(src_val-(sum/a.total_weight))*a.strength_scale;
  mult=(src_val+mult)/src_val;
  for (int c=0; c < a.channels; c++)
    {
      const bool colour_channel=c < a.colour;
      if (colour_channel && (((a.copy_mask >> c) & 1u) == 0))
        q[c]=QuantumOps<Q>::clamp((double) p[c]*mult);
      else
        q[c]=p[c];
    }
}

MhStatus launch_local_contrast(const View &src,const View &dst,double radius,double strength,
  const Roles &roles)
{
  LocalContrastArgs a;
  a.src=src.pixels;
  a.dst=dst.pixels;
  a.columns=(int) src.columns;
  a.rows=(int) src.rows;
  a.channels=src.channels;
  const long longest=(long) (src.columns > src.rows ? src.columns : src.rows);
  a.w=(int) (long) ((double) longest*0.002*fabs(radius));
  if ((a.w < 1) || (a.columns <= 2*a.w+2))
    return fail(MH_UNSUPPORTED,"LocalContrastImage: blur width %d on %d columns is left to the CPU",
      a.w,a.columns);
  a.total_weight=(double) (float) ((a.w+1)*(a.w+1));
  a.strength_scale=strength/100.0;
  a.copy_mask=roles.copy_mask;
  a.colour=src.channels-(roles.alpha >= 0 ? 1 : 0) >= 3 ? 3 : 1;
  Temp inter,luma;
  MH_TRY(inter.alloc(src.device,(size_t) a.rows*((size_t) a.columns+2*(size_t) a.w)*sizeof(float),src.stream));
  MH_TRY(luma.alloc(src.device,((size_t) a.rows+2*(size_t) a.w)*(size_t) a.columns*sizeof(float),src.stream));
  a.inter=inter.as<float>();
  const dim3 block(256);
  const unsigned gx=(unsigned) ((a.columns+255)/256);
  ProfileScope prof("local_contrast",src.stream);
  if (src.quantum == MH_QUANTUM_U16)
    hipLaunchKernelGGL((local_contrast_luma_kernel<uint16_t>),dim3(gx,(unsigned) (a.rows+2*a.w)),block,0,src.stream,a,luma.as<float>());
  else
    hipLaunchKernelGGL((local_contrast_luma_kernel<float>),dim3(gx,(unsigned) (a.rows+2*a.w)),block,0,src.stream,a,luma.as<float>());
  hipLaunchKernelGGL(local_contrast_vertical_kernel,dim3(gx,(unsigned) a.rows),block,0,src.stream,a,luma.as<float>());
  if (src.quantum == MH_QUANTUM_U16)
    hipLaunchKernelGGL((local_contrast_horizontal_kernel<uint16_t>),dim3(gx,(unsigned) a.rows),block,0,src.stream,a);
  else
    hipLaunchKernelGGL((local_contrast_horizontal_kernel<float>),dim3(gx,(unsigned) a.rows),block,0,src.stream,a);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// ---------------------------------------------------------------- DespeckleImage
// effect.c:1211-1306 (Hull) and :1308-1490: per channel, 16 Hull calls (4 directions x
// {+,-} offset x {raise, lower}), each two data-parallel sweeps between two zero-bordered
// (columns+2) x (rows+2) planes.  All channels are swept together (they never interact);
// the step is ScaleCharToQuantum(1) = 257, the guard ScaleCharToQuantum(2) = 514.
template<typename Q,int C>
__global__ __launch_bounds__(256)
void despeckle_load_kernel(const Q *src,Q *f,Q *g,int columns,int rows)
{
  const int px=(int) (blockIdx.x*blockDim.x+threadIdx.x),py=(int) blockIdx.y;     // padded coordinates
  if (px >= columns+2)
    return;
  Q v[C];
#pragma unroll
  for (int c=0; c < C; c++)
    v[c]=(Q) 0;
  Q zero[C];
#pragma unroll
  for (int c=0; c < C; c++)
    zero[c]=(Q) 0;
  if ((px >= 1) && (px <= columns) && (py >= 1) && (py <= rows))
    load_pixel<Q,C>(src+((size_t) (py-1)*columns+(px-1))*C,v);
  store_pixel<Q,C>(f+((size_t) py*(columns+2)+px)*C,v);
  store_pixel<Q,C>(g+((size_t) py*(columns+2)+px)*C,zero);
}

// SECOND=false: g = f raised/lowered where the neighbour at +offset in f calls for it;
// SECOND=true:  f = g raised/lowered where the neighbours at -offset and +offset in g do
template<typename Q,int C,bool SECOND>
__global__ __launch_bounds__(256)
void hull_kernel(const Q *from,Q *to,int columns,int rows,int offset,int polarity)
{
  const int x=(int) (blockIdx.x*blockDim.x+threadIdx.x),y=(int) blockIdx.y;
  if (x >= columns)
    return;
  const size_t i=((size_t) (y+1)*(columns+2)+(size_t) (x+1));
  Q centre[C],r[C],s[C],out[C];
  load_pixel<Q,C>(from+i*C,centre);
  load_pixel<Q,C>(from+(size_t) ((long) i+offset)*C,r);
  if (SECOND)
    load_pixel<Q,C>(from+(size_t) ((long) i-offset)*C,s);
#pragma unroll
  for (int c=0; c < C; c++)
    {
      double v=(double) centre[c];
      if (polarity > 0)
        {
          if (SECOND ? (((double) s[c] >= (v+514.0)) && ((double) r[c] > v)) : ((double) r[c] >= (v+514.0)))
            v+=257.0;
        }
      else
        {
          if (SECOND ? (((double) s[c] <= (v-514.0)) && ((double) r[c] < v)) : ((double) r[c] <= (v-514.0)))
            v-=257.0;
        }
      out[c]=(Q) v;
    }
  store_pixel<Q,C>(to+i*C,out);
}

template<typename Q,int C>
__global__ __launch_bounds__(256)
void despeckle_store_kernel(const Q *f,const Q *src,Q *dst,int columns,int rows,uint32_t copy_mask)
{
  const int x=(int) (blockIdx.x*blockDim.x+threadIdx.x),y=(int) blockIdx.y;
  if (x >= columns)
    return;
  Q v[C],original[C];
  load_pixel<Q,C>(f+((size_t) (y+1)*(columns+2)+(x+1))*C,v);
  load_pixel<Q,C>(src+((size_t) y*columns+x)*C,original);
#pragma unroll
  for (int c=0; c < C; c++)
    if ((copy_mask >> c) & 1u)
      v[c]=original[c];
  store_pixel<Q,C>(dst+((size_t) y*columns+x)*C,v);
}

template<typename Q,int C>
static MhStatus despeckle_typed(const View &src,const View &dst,const Roles &roles)
{
  const int W=(int) src.columns,H=(int) src.rows;
  const size_t plane=(size_t) (W+2)*(size_t) (H+2)*C*sizeof(Q);
  Temp tf,tg;
  MH_TRY(tf.alloc(src.device,plane,src.stream));
  MH_TRY(tg.alloc(src.device,plane,src.stream));
  Q *f=tf.as<Q>(),*g=tg.as<Q>();
  const dim3 block(256);
  ProfileScope prof("despeckle",src.stream);
  hipLaunchKernelGGL((despeckle_load_kernel<Q,C>),dim3((unsigned) ((W+2+255)/256),(unsigned) (H+2)),block,0,
    src.stream,static_cast<const Q *>(src.pixels),f,g,W,H);
  const dim3 grid((unsigned) ((W+255)/256),(unsigned) H);
  static const int X[4]={0,1,1,-1},Y[4]={1,0,1,1};
  for (int k=0; k < 4; k++)
    {
      const int offset=Y[k]*(W+2)+X[k];
      const int sequence[4][2]={{offset,1},{-offset,1},{-offset,-1},{offset,-1}};
      for (const auto &call : sequence)
        {
          hipLaunchKernelGGL((hull_kernel<Q,C,false>),grid,block,0,src.stream,f,g,W,H,call[0],call[1]);
          hipLaunchKernelGGL((hull_kernel<Q,C,true>),grid,block,0,src.stream,g,f,W,H,call[0],call[1]);
        }
    }
  hipLaunchKernelGGL((despeckle_store_kernel<Q,C>),grid,block,0,src.stream,f,static_cast<const Q *>(src.pixels),
    static_cast<Q *>(dst.pixels),W,H,roles.copy_mask);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

MhStatus launch_despeckle(const View &src,const View &dst,const Roles &roles)
{
#define MH_CASE(QT) \
  switch (src.channels) { \
    case 1: return despeckle_typed<QT,1>(src,dst,roles); \
    case 2: return despeckle_typed<QT,2>(src,dst,roles); \
    case 3: return despeckle_typed<QT,3>(src,dst,roles); \
    default: return despeckle_typed<QT,4>(src,dst,roles); }
  if (src.quantum == MH_QUANTUM_U16)
    { MH_CASE(uint16_t) }
  MH_CASE(float)
#undef MH_CASE
}

// ---------------------------------------------------------------- WaveletDenoiseImage
// visual-effects.c:3478-3760: five levels of the a-trous "hat" transform (rows, then columns)
// on float planes of the colour channels, soft thresholding of each detail band, and the
// reconstruction.  All arithmetic is the reference's float arithmetic, expression by
// expression (HatTransform's three index ranges in closed form).
static __device__ __forceinline__ float hat_value(const float *line,long stride,int i,int extent,int scale)
{
  const float centre=line[(long) i*stride];
  if (i < scale)
    return 0.25f*(centre+centre+line[(long) (scale-i)*stride]+line[(long) (scale+i)*stride]);
  if (i < extent-scale)
    return 0.25f*(2.0f*centre+line[(long) (i-scale)*stride]+line[(long) (i+scale)*stride]);
  return 0.25f*(centre+centre+line[(long) (i-scale)*stride]+line[(long) (2*extent-2-scale-i)*stride]);
}

template<typename Q>
__global__ __launch_bounds__(256)
void wavelet_load_kernel(const Q *src,float *plane,size_t npixels,int channels,int colour)
{
  const size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x;
  if (i >= npixels)
    return;
  for (int c=0; c < colour; c++)
    plane[i*colour+c]=(float) src[i*channels+c];
}

// ROWS: out(x,y) from the row y of `in`; else from the column x
template<bool ROWS>
__global__ __launch_bounds__(256)
void wavelet_hat_kernel(const float *in,float *out,int columns,int rows,int colour,int scale)
{
  const int xc=(int) (blockIdx.x*blockDim.x+threadIdx.x),y=(int) blockIdx.y;     // xc = x*colour+c
  if (xc >= columns*colour)
    return;
  const int x=xc/colour,c=xc-x*colour;
  const size_t pitch=(size_t) columns*colour;
  float v;
  if (ROWS)
    v=hat_value(in+(size_t) y*pitch+c,colour,x,columns,scale);
  else
    v=hat_value(in+(size_t) x*colour+c,(long) pitch,y,rows,scale);
  out[(size_t) y*pitch+xc]=v;
}

__global__ __launch_bounds__(256)
void wavelet_threshold_kernel(float *high,const float *low,float *base,size_t count,double magnitude,
  double softness,int accumulate)
{
  const size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x;
  if (i >= count)
    return;
  float h=high[i];
  h-=low[i];
  if ((double) h < -magnitude)
    h+=(float) (magnitude-softness*magnitude);
  else if ((double) h > magnitude)
    h-=(float) (magnitude-softness*magnitude);
  else
    h*=(float) softness;
  high[i]=h;
  if (accumulate != 0)
    base[i]+=h;
}

template<typename Q>
__global__ __launch_bounds__(256)
void wavelet_store_kernel(const Q *src,Q *dst,const float *base,const float *low,size_t npixels,int channels,
  int colour,uint32_t copy_mask)
{
  const size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x;
  if (i >= npixels)
    return;
  for (int c=0; c < channels; c++)
    {
      // every colour channel is rebuilt, whatever its traits (visual-effects.c:3590-3598 only
      // skips channels that are not R, G or B)
      if (c < colour)
        dst[i*channels+c]=QuantumOps<Q>::clamp((double) base[i*colour+c]+(double) low[i*colour+c]);
      else
        dst[i*channels+c]=src[i*channels+c];
    }
}

MhStatus launch_wavelet_denoise(const View &src,const View &dst,double threshold,double softness,
  const Roles &roles)
{
  const int W=(int) src.columns,H=(int) src.rows;
  if ((W < 33) || (H < 33))
    return fail(MH_UNSUPPORTED,"WaveletDenoiseImage: %dx%d is below two spans of the coarsest level",W,H);
  const int colour=src.channels-(roles.alpha >= 0 ? 1 : 0) >= 3 ? 3 : 1;
  const size_t n=(size_t) W*H,count=n*(size_t) colour;
  Temp planes[4];
  float *plane[4];
  for (int k=0; k < 4; k++)
    {
      MH_TRY(planes[k].alloc(src.device,count*sizeof(float),src.stream));
      plane[k]=planes[k].as<float>();
    }
  static const float noise_levels[]={0.8002f,0.2735f,0.1202f,0.0585f,0.0291f,0.0152f,0.0080f,0.0044f};
  const dim3 block(256);
  const unsigned pixel_blocks=(unsigned) ((n+255)/256),value_blocks=(unsigned) ((count+255)/256);
  ProfileScope prof("wavelet_denoise",src.stream);
  if (src.quantum == MH_QUANTUM_U16)
    hipLaunchKernelGGL((wavelet_load_kernel<uint16_t>),dim3(pixel_blocks),block,0,src.stream,
      static_cast<const uint16_t *>(src.pixels),plane[0],n,src.channels,colour);
  else
    hipLaunchKernelGGL((wavelet_load_kernel<float>),dim3(pixel_blocks),block,0,src.stream,
      static_cast<const float *>(src.pixels),plane[0],n,src.channels,colour);
  const dim3 grid((unsigned) ((W*colour+255)/256),(unsigned) H);
  int high=0,low=1;
  for (int level=0; level < 5; level++)
    {
      low=(level & 1)+1;                             // low_pass = number_pixels*((level & 0x01)+1)
      hipLaunchKernelGGL((wavelet_hat_kernel<true>),grid,block,0,src.stream,plane[high],plane[3],W,H,colour,
        1 << level);
      hipLaunchKernelGGL((wavelet_hat_kernel<false>),grid,block,0,src.stream,plane[3],plane[low],W,H,colour,
        1 << level);
      const double magnitude=threshold*(double) noise_levels[level];
      hipLaunchKernelGGL(wavelet_threshold_kernel,dim3(value_blocks),block,0,src.stream,plane[high],plane[low],
        plane[0],count,magnitude,softness,high != 0 ? 1 : 0);
      high=low;
    }
  if (src.quantum == MH_QUANTUM_U16)
    hipLaunchKernelGGL((wavelet_store_kernel<uint16_t>),dim3(pixel_blocks),block,0,src.stream,
      static_cast<const uint16_t *>(src.pixels),static_cast<uint16_t *>(dst.pixels),plane[0],plane[low],n,
      src.channels,colour,roles.copy_mask);
  else
    hipLaunchKernelGGL((wavelet_store_kernel<float>),dim3(pixel_blocks),block,0,src.stream,
      static_cast<const float *>(src.pixels),static_cast<float *>(dst.pixels),plane[0],plane[low],n,
      src.channels,colour,roles.copy_mask);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

} // namespace mh
