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

#include <cmath>

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

} // namespace mh
