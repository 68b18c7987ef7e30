// Any 2-D ConvolveMorphology — float or Q16 Quantum, one to four channels, alpha-weighted or plain,
// any cells (NaN = no cell) — BIT-IDENTICAL to the reference at a fraction of the generic cell walk:
// fused multiply-adds over alpha-premultiplied doubles, four output rows a lane, and a tie check.
//
// The reference (MagickCore/morphology.c:2919-2979, epilogue :3192-3198) forms, per output sample
// and cell, alpha = QuantumScale*a, w = alpha*k, t = w*p, pixel += t, gamma += w — five separately
// rounded fp64 operations and the conversions, 22 000 vector instructions per RGBA pixel under a
// Disk:15 in the generic kernel (morph2d_kernel: 180 ms per 16384^2 float frame).  The value it
// rounds at the end is, up to those roundings, S_c/S_alpha with S = sum k*P over the window, P =
// alpha*p (exact in fp64: a product of two Quantum values) or alpha or the plain sample.  This
// kernel computes the S with ONE fused multiply-add per cell and channel from doubles staged once
// per tile, rounds S_c/S_alpha (or S_c) to the Quantum — and recomputes, in the reference's own
// order and by the whole wave (tie_check.hpp), the samples whose value lies closer to a rounding
// boundary than the two evaluations can differ: the host's bound (the reference's roundings along
// its walk + the fused chain's, convolve_separable.hip has the derivation) times the largest |P|
// of the tile.  A tile that holds a non-finite sample (float Quantum) is recomputed entirely: a
// zero stands in for a NaN cell here, and 0*inf is not "no cell".
//
// MI355X mapping.  A workgroup of four waves owns 64 x 16 outputs and stages the (16+kh-1) x
// (64+kw-1) source window as PIXEL SLOTS of four 4-byte samples (69 KB for 31 x 31: two workgroups
// a CU; consecutive lanes read consecutive slots: no bank conflicts).  Lane = output column, wave =
// four output rows: for a kernel column u the lane walks DOWN its window column once, and every
// pixel it reads — one ds_read_b128 — feeds four multiply-adds per channel: output row r takes it
// with cell v-r.  16 multiply-adds against two LDS reads (the pixel, the cell: broadcast) and four
// conversions (seven vector instructions for an alpha-weighted float frame, whose alpha*p is formed
// here), the fp64 pipe is the bound: 16384^2 x 4 x 709 FMAs = 19 ms at 64 a clock and CU.
#include "mh_internal.hpp"
#include "device_common.hpp"
#include "tie_check.hpp"
#include <vector>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace mh {

struct Conv2DTieArgs
{
  const void *src;
  void *dst;
  int columns,rows;
  int kw,kh;
  int shiftx,shifty;          // output (x,y) reads source (x-shiftx+u, y-shifty+v)
  const double *cells;        // [kh][kw] in window order (the reflected walk), NaN cells as 0 (device)
  const int *spans;           // [kw][2]: first and one past the last window row with a non-zero cell of column u
  const double *values;       // the kernel's cells as the reference walks them (device; NaN = no cell)
  int tile_w,tile_h;          // 64+kw-1 (rounded up to even), 16+kh-1
  double error_unit;          // |fused - reference| <= error_unit * max|P_c| of the tile
  int mixed_signs;
  unsigned long long *recomputed;
  const unsigned *only_if;    // nullptr, or: leave at once when the word is zero
};

constexpr int kTieW=64;       // output columns per workgroup = lanes
constexpr int kTieH=16;       // output rows per workgroup: four waves x four rows

template<typename Q,int C,bool BLEND>
__global__ __launch_bounds__(256)
void conv2d_tie_kernel(Conv2DTieArgs args)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if ((args.only_if != nullptr) && (*args.only_if == 0u))
    return;
  const Q *src=static_cast<const Q *>(args.src);
  Q *dst=static_cast<Q *>(args.dst);
  const int W=args.columns,H=args.rows;
  const int TW=args.tile_w,TH=args.tile_h;
  const int plane=TW*TH;
  // the staged samples, four bytes each and a pixel's channels side by side (one ds_read_b128 — or
  // b64 for one and two channels — hands a lane the whole pixel): alpha*p of a Q16 frame as the
  // integer below 2^32 it is, a float frame's samples as the floats they are — alpha*p of two floats
  // needs a double, so an alpha-weighted float frame stages (p.., alpha) and multiplies in the walk,
  // in fp64, exactly.  (First version: channel planes, the alpha-weighted float frame as DOUBLES —
  // 138 KB for 31 x 31, one workgroup of four waves a CU, five LDS reads per 16 multiply-adds:
  // Disk:15 on 4096^2 5.7 ms.)
  constexpr int SLOT=C <= 2 ? 2 : 4;
  typedef uint32_t Slot __attribute__((ext_vector_type(SLOT)));
  Slot *tile=reinterpret_cast<Slot *>(smem_raw);                   // [TH][TW]
  double *cells=reinterpret_cast<double *>(smem_raw+(((size_t) plane*sizeof(Slot)+15u) & ~(size_t) 15u));   // [kh][kw]
  double *wave_most=cells+args.kw*args.kh;                          // [4 waves][4 channels]
  int *spans=reinterpret_cast<int *>(wave_most+16);                 // [kw][2]
  const int tid=(int) threadIdx.x,lane=tid & 63,wave=tid >> 6;
  const int x0=(int) blockIdx.x*kTieW,y0=(int) blockIdx.y*kTieH;

  for (int i=tid; i < args.kw*args.kh; i+=256)
    cells[i]=args.cells[i];
  for (int i=tid; i < 2*args.kw; i+=256)
    spans[i]=args.spans[i];
  // ---- the window, edge-clamped (cache.c:2663-2679), premultiplied, as doubles; the largest |P|
  // per channel on the way (a NaN or an infinity makes it non-finite)
  double most[4]={0.0,0.0,0.0,0.0};
  for (int i=tid; i < plane; i+=256)
    {
      const int row=i/TW,col=i-row*TW;
      int y=y0-args.shifty+row,x=x0-args.shiftx+col;
      y=y < 0 ? 0 : (y > H-1 ? H-1 : y);
      x=x < 0 ? 0 : (x > W-1 ? W-1 : x);
      Q q[C];
      load_pixel<Q,C>(src+((size_t) y*W+(size_t) x)*C,q);
      const double alpha=BLEND ? (double) q[C-1] : 1.0;
      Slot staged;
#pragma unroll
      for (int c=0; c < SLOT; c++)
        staged[c]=0u;
#pragma unroll
      for (int c=0; c < C; c++)
        {
          double p;
          if constexpr (sizeof(Q) == 2)
            {
              const uint32_t product=(BLEND && (c != C-1)) ? (uint32_t) q[C-1]*(uint32_t) q[c] : (uint32_t) q[c];
              staged[c]=product;
              p=(double) product;
            }
          else
            {
              p=(BLEND && (c != C-1)) ? alpha*(double) q[c] : (double) q[c];
              staged[c]=__float_as_uint((float) q[c]);
            }
          const double magnitude=__builtin_fabs(p);
          most[c]=(magnitude > most[c]) || !(magnitude == magnitude) ? magnitude : most[c];
        }
      tile[i]=staged;
    }
#pragma unroll
  for (int c=0; c < C; c++)
    {
      double m=most[c];
      for (int off=32; off > 0; off>>=1)
        {
          const double other=__shfl_xor(m,off,64);
          m=(other > m) || !(other == other) ? other : m;
        }
      if (lane == 0)
        wave_most[wave*4+c]=m;
    }
  __syncthreads();
  double error[4]={0.0,0.0,0.0,0.0};
  bool finite=true;
#pragma unroll
  for (int c=0; c < C; c++)
    {
      double m=wave_most[c];
#pragma unroll
      for (int w=1; w < 4; w++)
        {
          const double other=wave_most[w*4+c];
          m=(other > m) || !(other == other) ? other : m;
        }
      finite=finite && (m <= 1.7976931348623157e308);
      error[c]=args.error_unit*m;
    }

  // ---- lane = output column x0+lane, wave = output rows y0+4*wave .. +3
  double acc[4][C];
#pragma unroll
  for (int r=0; r < 4; r++)
#pragma unroll
    for (int c=0; c < C; c++)
      acc[r][c]=0.0;
  const Slot *column=tile+(4*wave)*TW+lane;
  for (int u=0; u < args.kw; u++)
    {
      // down the window column x+u: the sample of window row v (of output row 0) is the sample of
      // window row v-r of output row r.  Only the rows between the column's first and last cell
      // (a disk's columns are short at its sides), and three more for the rows behind.
      const int first=spans[2*u],last=spans[2*u+1];
      double k1=0.0,k2=0.0,k3=0.0;               // the cells of rows v-1, v-2, v-3
      const Slot *at=column+u+first*TW;
      const double *cell=cells+u;
#pragma unroll 4
      for (int v=first; v < last+3; v++)
        {
          const double k0=v < last ? cell[v*args.kw] : 0.0;
          const Slot raw=*at;
          double p[C];
          if constexpr (sizeof(Q) == 2)
            {
#pragma unroll
              for (int c=0; c < C; c++)
                p[c]=(double) raw[c];
            }
          else
            {
#pragma unroll
              for (int c=0; c < C; c++)
                p[c]=(double) __uint_as_float(raw[c]);
              if constexpr (BLEND)
                {
#pragma unroll
                  for (int c=0; c < C-1; c++)
                    p[c]=p[C-1]*p[c];              // alpha*p: exact in fp64
                }
            }
#pragma unroll
          for (int c=0; c < C; c++)
            {
              acc[0][c]=__builtin_fma(k0,p[c],acc[0][c]);
              acc[1][c]=__builtin_fma(k1,p[c],acc[1][c]);
              acc[2][c]=__builtin_fma(k2,p[c],acc[2][c]);
              acc[3][c]=__builtin_fma(k3,p[c],acc[3][c]);
            }
          k3=k2;
          k2=k1;
          k1=k0;
          at+=TW;
        }
    }
  // ---- the Quantum of each sum, or the reference's own walk where the bound cannot tell
  const int x=x0+lane;
  uint32_t doubtful=0;                           // bit 4r+c
#pragma unroll
  for (int r=0; r < 4; r++)
    {
      const int y=y0+4*wave+r;
      double s[4]={0.0,0.0,0.0,0.0};
#pragma unroll
      for (int c=0; c < C; c++)
        s[c]=acc[r][c];
      Q out[C];
      uint32_t undecided=settle_sums<Q,C,BLEND>(s,error,args.mixed_signs,out);
      if (!finite)
        undecided=(1u << C)-1u;
      if ((x < W) && (y < H))
        {
          doubtful|=undecided << (4*r);
          store_pixel<Q,C>(dst+((size_t) y*W+(size_t) x)*C,out);
        }
    }
  unsigned recomputed=0;
  unsigned long long pending=__ballot(doubtful != 0u);
  while (pending != 0ull)
    {
      const int who=__builtin_ctzll(pending);
      pending&=pending-1ull;
      uint32_t which=(uint32_t) __builtin_amdgcn_readlane((int) doubtful,who);
      const int xx=x0+who;
      while (which != 0u)
        {
          const int bit=__builtin_ctz(which);
          which&=which-1u;
          const int yy=y0+4*wave+(bit >> 2),c=bit & 3;
          const Q settled=conv2d_reference_sample<Q,C,BLEND>(src,W,H,xx,yy,c,args.values,args.kw,args.kh,
            args.shiftx,args.shifty,lane);
          if (lane == who)
            {
              dst[((size_t) yy*W+(size_t) xx)*C+c]=settled;
              recomputed++;
            }
        }
    }
  if (args.recomputed != nullptr)
    {
      for (int off=32; off > 0; off>>=1)
        recomputed+=__shfl_xor(recomputed,off,64);
      if ((lane == 0) && (recomputed != 0))
        atomicAdd(args.recomputed,(unsigned long long) recomputed);
    }
}

static unsigned long long *g_tie2d_recomputed[64]={};
static bool g_tie2d_count=false;

template<typename Q,int C,bool BLEND>
static MhStatus launch_conv2d_tie_typed(const View &src,Conv2DTieArgs &args,size_t lds)
{
  MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2d_tie_kernel<Q,C,BLEND>),
    hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  ProfileScope prof("conv2d_tie",src.stream);
  const dim3 grid((unsigned) ((args.columns+kTieW-1)/kTieW),(unsigned) ((args.rows+kTieH-1)/kTieH));
  hipLaunchKernelGGL((conv2d_tie_kernel<Q,C,BLEND>),grid,dim3(256),lds,src.stream,args);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// w x h Convolve (bias 0, every channel updated) of any layout and Quantum type.  *handled stays
// false (nothing launched) when the kernel or the frame does not qualify.  only_if: a device word;
// the kernel leaves at once when it is zero (the fallback behind convolve2d_exact.hip's float form).
MhStatus launch_conv2d_tie(const View &src,const View &dst,const MhKernelInfo *kernel,const Roles &roles,
  bool *handled,const unsigned *only_if)
{
  *handled=false;
  const int kw=(int) kernel->width,kh=(int) kernel->height;
  if ((src.channels < 1) || (src.channels > 4) || (roles.copy_mask != 0) || (src.quantum != dst.quantum) ||
      (src.channels != dst.channels) || (src.columns != dst.columns) || (src.rows != dst.rows) ||
      (src.pixels == dst.pixels) || (kw < 2) || (kh < 1) || (kw*kh < 25) ||     // kw == 1: see below (kernel->x < 0) || (kernel->y < 0) ||
      (kernel->x >= kw) || (kernel->y >= kh) || (src.columns >= (1u << 30)) ||
      ((src.rows+kTieH-1)/kTieH > 65535u) ||                 // (gridDim.y)
      (option("MAGICKHIP_NO_TIE_2D") != nullptr))
    return MH_OK;
  // (a one-column kernel is the reference's width == 1 fast path, morphology.c:2654-2807, which
  // scales gamma by height / count when NaN cells leave fewer than `height` terms — :2775-2776.
  // NaN-free columns run in launch_conv1d; columns WITH NaN cells keep the generic kernel, which
  // applies that factor.)
  // (alpha-weighted: gray + alpha and RGBA; other layouts with an alpha trait keep the generic kernel)
  const bool blend=roles.blend && (roles.alpha == src.channels-1) && ((src.channels == 2) || (src.channels == 4));
  if (roles.blend && !blend)
    return MH_OK;
  const int tile_w=(kTieW+kw-1+1) & ~1,tile_h=kTieH+kh-1;
  const bool is_float=src.quantum != MH_QUANTUM_U16;
  const size_t slot_bytes=src.channels <= 2 ? 8u : 16u;                     // conv2d_tie_kernel's Slot
  const size_t lds=(((size_t) tile_w*tile_h*slot_bytes+15u) & ~(size_t) 15u)+
    ((size_t) kw*kh+16)*sizeof(double)+(size_t) 2*kw*sizeof(int);
  if (lds > 160u*1024u)
    return MH_OK;
  // the bound of convolve_separable.hip without its outer-product term: the reference rounds every
  // term twice (three times when alpha-weighted) and every addition once, to half an ulp of its
  // running sum — the prefix sums of |cell| along its walk; the fused chain rounds once per cell
  double magnitude=0.0,running=0.0,partials=0.0,total=0.0;
  bool negative=false,positive=false;
  int count=0;
  for (int i=kw*kh-1; i >= 0; i--)
    {
      const double cell=kernel->values[i];
      if (std::isnan(cell))
        continue;
      if (!std::isfinite(cell))
        return MH_OK;
      total+=cell;
      negative=negative || (cell < 0.0);
      positive=positive || (cell > 0.0);
      magnitude+=std::fabs(cell);
      running+=std::fabs(cell);
      partials+=running;
      count++;
    }
  if (count == 0)
    return MH_OK;
  const double unit=1.1102230246251565e-16;
  const double error_unit=2.0*unit*(partials+4.0*magnitude+((double) count+8.0)*magnitude);
  if (!is_float && (error_unit*65535.0*65535.0 > 65535.0*1.0e-3))
    return MH_OK;
  // alpha-weighted frames under cells that (nearly) cancel: PerceptibleReciprocal's clamp decides
  // every pixel, every pixel would take the reference-order walk — the generic kernel's job
  if (blend && negative && positive && !(std::fabs(total) > 0.05*magnitude))
    return MH_OK;
  // window order: cell (v,u) of the window carries values[(kh-1-v)*kw+(kw-1-u)] (the reflected walk)
  std::vector<double> window((size_t) kw*kh,0.0);
  for (int v=0; v < kh; v++)
    for (int u=0; u < kw; u++)
      {
        const double cell=kernel->values[(size_t) (kh-1-v)*kw+(size_t) (kw-1-u)];
        window[(size_t) v*kw+(size_t) u]=std::isnan(cell) ? 0.0 : cell;
      }
  std::vector<int> spans((size_t) 2*kw,0);
  for (int u=0; u < kw; u++)
    {
      int first=kh,last=0;
      for (int v=0; v < kh; v++)
        if (window[(size_t) v*kw+(size_t) u] != 0.0)
          {
            first=v < first ? v : first;
            last=v+1;
          }
      spans[(size_t) 2*u]=first < last ? first : 0;
      spans[(size_t) 2*u+1]=first < last ? last : -3;        // (an empty column: no rows at all)
    }
  TableBundle tables;
  const size_t t_spans=tables.add(spans.data(),spans.size()*sizeof(int));
  const size_t t_cells=tables.add(window.data(),window.size()*sizeof(double));
  const size_t t_values=tables.add(kernel->values,(size_t) kw*kh*sizeof(double));
  MH_TRY(tables.upload(src.device,src.stream));
  Conv2DTieArgs args;
  args.src=src.pixels;
  args.dst=dst.pixels;
  args.columns=(int) src.columns;
  args.rows=(int) src.rows;
  args.kw=kw;
  args.kh=kh;
  args.shiftx=kw-1-(int) kernel->x;
  args.shifty=kh-1-(int) kernel->y;
  args.cells=tables.at<double>(t_cells);
  args.spans=tables.at<int>(t_spans);
  args.values=tables.at<double>(t_values);
  args.tile_w=tile_w;
  args.tile_h=tile_h;
  args.error_unit=error_unit;
  args.mixed_signs=(negative && positive) ? 1 : 0;
  args.only_if=only_if;
  args.recomputed=nullptr;
  if (g_tie2d_count && (src.device >= 0) && (src.device < 64))
    {
      if (g_tie2d_recomputed[src.device] == nullptr)
        {
          MH_HIP(hipMalloc(reinterpret_cast<void **>(&g_tie2d_recomputed[src.device]),sizeof(unsigned long long)));
          MH_HIP(hipMemsetAsync(g_tie2d_recomputed[src.device],0,sizeof(unsigned long long),src.stream));
        }
      args.recomputed=g_tie2d_recomputed[src.device];
    }
  *handled=true;
#define MH_LAYOUT(QT) \
  switch (src.channels) \
  { \
    case 1: return launch_conv2d_tie_typed<QT,1,false>(src,args,lds); \
    case 2: return blend ? launch_conv2d_tie_typed<QT,2,true>(src,args,lds) : launch_conv2d_tie_typed<QT,2,false>(src,args,lds); \
    case 3: return launch_conv2d_tie_typed<QT,3,false>(src,args,lds); \
    default: return blend ? launch_conv2d_tie_typed<QT,4,true>(src,args,lds) : launch_conv2d_tie_typed<QT,4,false>(src,args,lds); \
  }
  if (is_float)
    { MH_LAYOUT(float) }
  MH_LAYOUT(uint16_t)
#undef MH_LAYOUT
}

} // namespace mh

using namespace mh;

// Diagnostic: samples the fused 2-D convolve recomputed in the reference's order since the counter
// was last read (enable = 1 switches the counting on and reads, 0 reads and switches off).
extern "C" MH_API unsigned long long MhConvolve2DTieRecomputed(int enable)
{
  unsigned long long total=0;
  int current=-1;
  (void) hipGetDevice(&current);
  for (int d=0; d < 64; d++)
    if (g_tie2d_recomputed[d] != nullptr)
      {
        unsigned long long value=0;
        if (hipSetDevice(d) == hipSuccess)
          {
            (void) hipDeviceSynchronize();
            (void) hipMemcpy(&value,g_tie2d_recomputed[d],sizeof(value),hipMemcpyDeviceToHost);
            (void) hipMemset(g_tie2d_recomputed[d],0,sizeof(value));
          }
        total+=value;
      }
  if (current >= 0)
    (void) hipSetDevice(current);
  g_tie2d_count=enable != 0;
  return total;
}
