// Erode / Dilate with a large symmetric convex flat kernel (Disk:15, Octagon:10 ...; Q16, two or
// four channels; MagickCore/morphology.c:2980-3036 under the edge policy of cache.c:2663-2679):
// the van Herk / Gil-Werman route — running maxima along the rows, then the combination of the
// kernel's rows — as a walk down the frame in which nothing is computed twice vertically.
//
// With h(dy) the half-width of kernel row dy (|dy| <= V <= 15, h <= 15):
//     out(x,y) = max over dy of P_h(dy)(x, y+dy),     P_h(x,r) = max over |dx| <= h of in(x+dx, r).
// A wave owns 64 output columns and walks down the source rows r of a chunk.  Per source row:
//   running maxima   the row segment B[0..93] (the wave's columns and 15 either side) goes to the
//                    wave's LDS; D_k[i] = max B[i .. i+2^k-1], k = 1..4, by doubling
//                    (D_k[i] = max(D_{k-1}[i], D_{k-1}[i+2^{k-1}]): one shifted LDS read and one
//                    maximum per level for each of the row's two 64-entry halves);
//   a kernel row     P_h(x) = max(D_k[x+15-h], D_k[x+15+h-2^k+1]) with 2^k <= 2h+1 < 2^(k+1): two
//                    LDS reads and one maximum for every DISTINCT |dy| (rows dy and -dy share it);
//   accumulation     source row r is row dy of the window of centre row c = r-dy: 31 pending
//                    centre rows live in registers (slot = c mod 31) and take one maximum each;
//                    the row whose last contribution this was (c = r-15) is stored and its slot
//                    reset.  The walk is unrolled over the 31 phases of r mod 31, so every slot is
//                    a compile-time register.
// 4x2 + 16 + 31 = 55 maxima per 64 pixels-rows and word against the union-of-rectangles kernel's
// 71 per pixel of which a third is discarded (morph_rects_kernel: its Row() steps spoil the
// columns at the wave's edges); no workgroup barrier; the frame is read 1.47 times through L2
// (the halo columns; the four waves of a workgroup are neighbours) and written once.
// min / max are exact whatever the order: the reference's bits.
#include "mh_internal.hpp"
#include "device_common.hpp"
#include <vector>

namespace mh {

namespace {

constexpr int kReach=15;                       // largest |dy| and half-width
constexpr int kSlots=2*kReach+1;               // pending centre rows
constexpr int kPlane=128;                      // entries of a D_k plane (94 used, reads run over the end)

struct WalkArgs
{
  const void *src;
  void *dst;
  int columns,rows;
  int cx,cy;                                   // centre of the shape relative to the output pixel
  int vmax;                                    // kernel rows |dy| <= vmax
  int strips,chunks,rows_per_chunk;            // rows_per_chunk+30 is a multiple of 31
  uint32_t copy_mask;
  unsigned long long *changed;
  unsigned char half[kReach+1];                // h(|dy|)
  unsigned char level[kReach+1];               // k(|dy|): 2^k <= 2h+1 < 2^(k+1)
};

template<bool DILATE>
static __device__ __forceinline__ uint32_t pick16(uint32_t a,uint32_t b)
{
  typedef unsigned short U2 __attribute__((ext_vector_type(2)));
  const U2 va=__builtin_bit_cast(U2,a),vb=__builtin_bit_cast(U2,b);
  return __builtin_bit_cast(uint32_t,DILATE ? __builtin_elementwise_max(va,vb) : __builtin_elementwise_min(va,vb));
}

// NW: 32-bit words per pixel (1: two Q16 channels, 2: four).  NEED_ORIGINAL: the output pixel
// itself is read (Erode starts from it, morphology.c:2905-2912; copied channels; the change count)
template<int NW,bool DILATE,bool NEED_ORIGINAL>
__global__ __launch_bounds__(256,3)            // 31 pending rows + 32 addresses: 168 registers, three waves a SIMD
void morph_walk_kernel(WalkArgs args)
{
  typedef uint32_t Pixel __attribute__((ext_vector_type(NW)));
  typedef Pixel __attribute__((aligned(4))) LoosePixel;
  constexpr int PXB=4*NW;
  constexpr uint32_t kIdentity=DILATE ? 0u : 0xffffffffu;
  constexpr int WAVE_BYTES=6*kPlane*PXB;                    // D_0..D_4 and a plane of the identity
  __shared__ __attribute__((aligned(16))) unsigned char smem[4*WAVE_BYTES];

  const int lane=(int) threadIdx.x & 63;
  const int wave=__builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
  // the four waves of a workgroup are neighbouring strips of the same rows: the halo columns a
  // wave reads beside its own are its neighbours' own (L1 / L2)
  const int groups_x=(args.strips+3)/4;
  const int chunk=(int) blockIdx.x/groups_x,group=(int) blockIdx.x-chunk*groups_x;
  const int strip=4*group+wave;
  if (strip >= args.strips)
    return;
  const int W=args.columns,H=args.rows;
  const int x0=64*strip,y0=chunk*args.rows_per_chunk;
  int y1=y0+args.rows_per_chunk;
  y1=y1 < H ? y1 : H;
  unsigned char *mine=smem+wave*WAVE_BYTES;
  Pixel *plane=reinterpret_cast<Pixel *>(mine);              // D_k[i] = plane[k*kPlane+i]

  // the two halves of the row segment: entry i = lane and i = 64+lane is image column x0+cx-15+i
  unsigned column_bytes[2];
#pragma unroll
  for (int p=0; p < 2; p++)
    {
      int x=x0+args.cx-kReach+64*p+lane;
      x=x < 0 ? 0 : (x > W-1 ? W-1 : x);
      column_bytes[p]=(unsigned) x*(unsigned) PXB;
    }
  const size_t row_bytes=(size_t) W*(size_t) PXB;
  const unsigned char *base=static_cast<const unsigned char *>(args.src);
  auto fetch=[&](int r,Pixel (&into)[2])
  {
    r=r < 0 ? 0 : (r > H-1 ? H-1 : r);
    const unsigned char *row=base+(size_t) r*row_bytes;
#pragma unroll
    for (int p=0; p < 2; p++)
      into[p]=*reinterpret_cast<const LoosePixel *>(row+column_bytes[p]);
  };
  const int x=x0+lane;
  const unsigned own_bytes=(unsigned) (x < W ? x : W-1)*(unsigned) PXB;
  auto fetch_original=[&](int y,Pixel &into)
  {
    y=y < 0 ? 0 : (y > H-1 ? H-1 : y);
    into=*reinterpret_cast<const LoosePixel *>(base+(size_t) y*row_bytes+own_bytes);
  };
  // where kernel row |dy| = d finds its two running maxima (LDS byte addresses of this lane)
  const Pixel *first[kReach+1],*second[kReach+1];
#pragma unroll
  for (int d=0; d <= kReach; d++)
    {
      const int h=args.half[d],k=args.level[d];
      first[d]=plane+k*kPlane+lane+kReach-h;
      second[d]=plane+k*kPlane+lane+kReach+h-(1 << k)+1;
      // rows the kernel does not have read the identity: no branch in the walk (a taken branch
      // costs more than the three instructions it would skip)
      if (d > args.vmax)
        first[d]=second[d]=plane+5*kPlane+lane;
    }
  {
    Pixel identity;
#pragma unroll
    for (int w=0; w < NW; w++)
      identity[w]=kIdentity;
    plane[5*kPlane+lane]=identity;
  }
  Pixel acc[kSlots];
#pragma unroll
  for (int s=0; s < kSlots; s++)
#pragma unroll
    for (int w=0; w < NW; w++)
      acc[s][w]=kIdentity;
  unsigned changed=0u;

  // source rows r = r_first+t, t = 0 .. rows_per_chunk+29; the centre row completed at t is
  // r-15 = output row y0+t-30
  const int r_first=y0+args.cy-kReach;
  const int total=args.rows_per_chunk+2*kReach;              // a multiple of 31
  Pixel ahead[2],original;
  fetch(r_first,ahead);
  if constexpr (NEED_ORIGINAL)
    fetch_original(y0-2*kReach,original);
  {
    // (a use in front of the walk: it is entered with no load in flight, see resize_stream.hip)
    uint32_t any=ahead[0][0] | ahead[1][0];
    if constexpr (NEED_ORIGINAL)
      any|=original[0];
    asm volatile("" :: "v"(any));
  }
  for (int tb=0; tb < total; tb+=kSlots)
    {
#pragma unroll
      for (int j=0; j < kSlots; j++)
        {
          const int t=tb+j,r=r_first+t;
          Pixel now[2]={ahead[0],ahead[1]};
          Pixel mine_original;
          if constexpr (NEED_ORIGINAL)
            mine_original=original;
          // the next row and the next output pixel are requested in every phase, in front of this
          // phase's store: the wait for them lets exactly one store stay in flight
          fetch(r+1,ahead);
          const int y=y0+t-2*kReach;
          if constexpr (NEED_ORIGINAL)
            fetch_original(y+1,original);
          // ---- running maxima of the row segment by doubling (LDS operations of one wave
          // execute in order)
          plane[lane]=now[0];
          plane[64+lane]=now[1];
          asm volatile("" ::: "memory");
#pragma unroll
          for (int k=1; k <= 4; k++)
            {
              const int shift=1 << (k-1);
#pragma unroll
              for (int p=0; p < 2; p++)
                {
                  const Pixel other=plane[(k-1)*kPlane+64*p+lane+shift];
#pragma unroll
                  for (int w=0; w < NW; w++)
                    now[p][w]=pick16<DILATE>(now[p][w],other[w]);
                  plane[k*kPlane+64*p+lane]=now[p];
                }
              asm volatile("" ::: "memory");
            }
          // ---- the kernel's rows: row dy of the window of centre row r-dy
#pragma unroll
          for (int d=0; d <= kReach; d++)
              {
                const Pixel a=*first[d],b=*second[d];
                Pixel p;
#pragma unroll
                for (int w=0; w < NW; w++)
                  p[w]=pick16<DILATE>(a[w],b[w]);
                constexpr int kNone=-1;
                const int up=(j-d+kReach+kSlots) % kSlots;                  // centre row r-d
                const int down=d == 0 ? kNone : (j+d+kReach) % kSlots;      // centre row r+d
#pragma unroll
                for (int w=0; w < NW; w++)
                  acc[up][w]=pick16<DILATE>(acc[up][w],p[w]);
                if (down != kNone)
#pragma unroll
                  for (int w=0; w < NW; w++)
                    acc[down][w]=pick16<DILATE>(acc[down][w],p[w]);
              }
          asm volatile("" ::: "memory");
          // ---- centre row r-15 is complete: output row y, slot j
          Pixel result=acc[j];
#pragma unroll
          for (int w=0; w < NW; w++)
            acc[j][w]=kIdentity;
          const bool live=(t >= 2*kReach) && (y < y1) && (x < W);
          if constexpr (NEED_ORIGINAL)
            {
#pragma unroll
              for (int w=0; w < NW; w++)
                {
                  const uint32_t value=DILATE ? result[w] : pick16<false>(result[w],mine_original[w]);
                  uint32_t keep=0u;
                  keep|=((args.copy_mask >> (2*w)) & 1u) != 0u ? 0x0000ffffu : 0u;
                  keep|=((args.copy_mask >> (2*w+1)) & 1u) != 0u ? 0xffff0000u : 0u;
                  result[w]=(mine_original[w] & keep) | (value & ~keep);
                  const uint32_t differs=(value ^ mine_original[w]) & ~keep;
                  if (live)
                    changed+=((differs & 0xffffu) != 0u ? 1u : 0u)+((differs >> 16) != 0u ? 1u : 0u);
                }
            }
          // (no branch around the store: dead lanes and rows get an offset beyond the row's
          // descriptor and the hardware drops them)
          int row=y < 0 ? 0 : (y > H-1 ? H-1 : y);
          const __amdgpu_buffer_rsrc_t drow=__builtin_amdgcn_make_buffer_rsrc(
            static_cast<unsigned char *>(args.dst)+(size_t) row*row_bytes,0,(int) row_bytes,0x00020000);
          const unsigned offset=live ? (unsigned) x*(unsigned) PXB : 0xffffffffu;
          typedef unsigned words2 __attribute__((ext_vector_type(2)));
          if constexpr (NW == 2)
            __builtin_amdgcn_raw_buffer_store_b64(words2{result[0],result[1]},drow,offset,0,0);
          else
            __builtin_amdgcn_raw_buffer_store_b32(result[0],drow,offset,0,0);
        }
    }
  if (NEED_ORIGINAL && (args.changed != nullptr))
    {
      changed=wave_sum(changed);
      if ((lane == 0) && (changed != 0u))
        atomicAdd(args.changed,(unsigned long long) changed);
    }
}

template<int NW>
static MhStatus launch_walk_words(const WalkArgs &a,bool dilate,bool need_original,hipStream_t stream)
{
  const dim3 grid((unsigned) (((a.strips+3)/4)*a.chunks)),block(256);
#define MH_WALK(D,O) hipLaunchKernelGGL((morph_walk_kernel<NW,D,O>),grid,block,0,stream,a)
  if (dilate)
    {
      if (need_original) MH_WALK(true,true); else MH_WALK(true,false);
    }
  else
    MH_WALK(false,true);
#undef MH_WALK
  MH_HIP(hipGetLastError());
  return MH_OK;
}

} // namespace

// half[k]: half-width of kernel row dy_min+k (every row non-empty, runs centred on cx, symmetric
// about the middle row: try_rects has checked).  *handled = false: not this kernel's case.
MhStatus launch_morph_walk(const View &src,const View &dst,bool dilate,const std::vector<int> &half,
  int cx,int dy_min,const Roles &roles,unsigned long long *changed,bool *handled)
{
  *handled=false;
  const int span=(int) half.size(),vmax=span/2;
  if ((src.quantum != MH_QUANTUM_U16) || ((src.channels != 2) && (src.channels != 4)) ||
      ((span & 1) == 0) || (option("MAGICKHIP_NO_MORPH_WALK") != nullptr))
    return MH_OK;
  // large kernels (the union-of-rectangles kernel keeps the small ones: it has no 30-row lead-in)
  int hmax=0;
  for (int k=0; k < span; k++)
    hmax=half[(size_t) k] > hmax ? half[(size_t) k] : hmax;
  if ((vmax > kReach) || (hmax > kReach) || (vmax < 6) || (src.rows < 64) || (src.columns < 64))
    return MH_OK;
  const size_t pixel_bytes=(size_t) src.channels*sizeof(uint16_t);
  if ((size_t) src.columns*pixel_bytes >= (1ull << 31))
    return MH_OK;                                // (32-bit byte offsets inside a row)
  WalkArgs a;
  a.src=src.pixels;
  a.dst=dst.pixels;
  a.columns=(int) src.columns;
  a.rows=(int) src.rows;
  a.cx=cx;
  a.cy=dy_min+vmax;
  a.vmax=vmax;
  a.copy_mask=roles.copy_mask;
  a.changed=changed;
  for (int d=0; d <= kReach; d++)
    {
      const int h=d <= vmax ? half[(size_t) (vmax+d)] : 0;
      int k=0;
      while ((2 << k) <= 2*h+1)
        k++;
      a.half[d]=(unsigned char) h;
      a.level[d]=(unsigned char) k;
    }
  a.strips=((int) src.columns+63)/64;
  // rows a wave walks: 31*n-30 (the walk is unrolled over the 31 phases of its pending rows);
  // enough items for a few rounds of the chip's 16 waves a CU
  int n=(int) option_long("MAGICKHIP_MORPH_WALK_PERIODS",10);
  n=n < 2 ? 2 : (n > 64 ? 64 : n);
  a.rows_per_chunk=kSlots*n-2*kReach;
  a.chunks=((int) src.rows+a.rows_per_chunk-1)/a.rows_per_chunk;
  const bool centred=(a.cx == 0) && (a.cy == 0);
  (void) centred;
  const bool need_original=!dilate || (roles.copy_mask != 0) || (changed != nullptr);
  ProfileScope prof("morph_walk",src.stream);
  *handled=true;
  if (src.channels == 4)
    return launch_walk_words<2>(a,dilate,need_original,src.stream);
  return launch_walk_words<1>(a,dilate,need_original,src.stream);
}

} // namespace mh
