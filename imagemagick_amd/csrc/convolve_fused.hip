// BlurImage's two passes in ONE launch (FAST precision, Q16 RGBA: alpha-weighted colour or
// four plain channels, RGB).  MagickCore/effect.c:765-796 -> morphology.c:2811-2979 (row kernel)
// -> :2654-2807 (column kernel) with the Quantum-rounded intermediate of :4012-4022.
//
// The two launches of convolve_mfma.hip move the intermediate frame through HBM once in each
// direction: half of all bytes of a blur.  Here a workgroup owns a strip of 64 pixel columns
// and walks it downwards; the row pass's Quantum-rounded results go straight into an LDS ring
// in the column pass's operand format, and the column pass runs out of that ring.  HBM sees
// the source once (plus the K-1 halo columns of a strip, which neighbouring strips fetch at the
// same time: L2 / Infinity-Cache hits) and the result once.
//
// Both passes are the banded (Toeplitz) matrix product of convolve_mfma.hip with hi/lo-split f16
// operands (same precision argument, same epilogue; mfma_common.hpp).  The first version of this
// file used v_mfma_f32_32x32x16_f16 tiles with one wave per SIMD (0.67 ms per 8192^2 blur: every
// dependent instruction waited out the ALU latency; removed in round 2c); the kernel below uses
// 16x16x32 tiles and sixteen waves.
#include "mh_internal.hpp"
#include "device_common.hpp"
#include "mfma_common.hpp"
#include <cstdlib>
#include <type_traits>
#include <cstdio>
#include <string>
#include <vector>

namespace mh {

struct BlurFusedArgs
{
  const uint16_t *src;
  uint16_t *dst;
  int columns,rows;
  int ntaps;
  int shift;                 // K-1-origin: offset of the first input sample (both axes)
  const float *taps;         // float[K], taps[v] multiplies input o-shift+v
  const double *taps64;      // the same as doubles (exact_alpha_level)
  int strips;                // ceil(columns/64)
  int segments;              // vertical cuts of a strip
  int blocks;                // ceil(rows/32) output blocks per strip
  int blocks_per_segment;
  int items_per_xcd;         // ceil(strips*segments/8)
  float gain;                // UnsharpMaskImage's epilogue (blur_fused16_kernel<.., UNSHARP>)
  int threshold;             // ceil(QuantumRange*threshold), see unsharp_sample
  unsigned long long *trace; // diagnostic build (-DMH_FUSED_TRACE) only
};

// Diagnostic build only (-DMH_FUSED_TRACE, tools/trace_fused_blur.py): waves 0, 4, 8 and 12 of
// the first four workgroups stamp the shader clock at the phase boundaries of 48 steady-state
// iterations: trace[block][wave>>2][iteration][mark].
// Diagnostic builds only (-DMH_FUSED_KNOCK=bits, tools/gpu_knock.sh): parts of the iteration are
// skipped at run time (behind a test the compiler cannot fold) to see what each costs.  The
// results are wrong.  1 result stores, 2 source fetches after the first, 4 staging conversion,
// 8 column pass, 16 row pass MFMAs, 32 row epilogue arithmetic, 64 the whole row pass
#ifdef MH_FUSED_KNOCK
#define MH_KNOCKED(bit) ((((MH_FUSED_KNOCK) & (bit)) != 0) && (args.threshold == 0))
#else
#define MH_KNOCKED(bit) false
#endif
#ifdef MH_FUSED_TRACE
#define MH_FTRACE_MARK(id) \
  do { \
    if (traced && (g >= 64) && (g < 112)) \
      { \
        const unsigned long long now=__builtin_readcyclecounter(); \
        if (lane == 0) \
          args.trace[((((int) blockIdx.x*4+(wave >> 2))*48)+(g-64))*10+(id)]=now; \
      } \
  } while (0)
#else
#define MH_FTRACE_MARK(id) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// The walk on v_mfma_f32_16x16x32_f16 tiles with SIXTEEN waves per workgroup.
//
// A 32x32x16 form keeps one wave on each SIMD (its operand registers leave room for no more),
// and the counters showed what that costs (profiles/r2b_*): the matrix pipe busy 22 % of the
// time, 28 % spent at barriers and waitcnts, and half of all cycles going to the VALU stream
// (conversions, epilogues, address arithmetic) issuing at ~7.6 cycles per instruction.  16x16x32
// tiles need a third of the operand registers (a 32-sample chunk per instruction: three chunks
// cover the 79 taps' 94-sample band against seven 16-sample chunks for a 32-output tile), so a
// wave fits in 128 VGPRs and four of them share a SIMD:
//
//   row pass     entries e = 4*row + channel    (16 = 4 rows x 4 channels), 16 outputs along x:
//                16 tiles per 16-row group, one per wave.  D (row = 4*(lane>>4)+reg, col =
//                lane&15) leaves a lane the four channels of ONE pixel: the division by the alpha
//                sum, the Quantum rounding and the exact small-alpha test are lane-local.  A 4x4
//                transpose between registers and the four 16-lane rows (2 v_permlane32_swap + 2
//                v_permlane16_swap) then gives each lane four consecutive rows of one channel —
//                one 8-byte ring store per plane.  (Round 2b had e = 4*channel+row: no
//                transpose, but the alpha sums came over with four ds_bpermute and every lane
//                repeated the division: 10 % slower.)
//   column pass  entries e = 4*column + channel (16 = 4 columns x 4 channels), 16 outputs along
//                y: 16 tiles per 16-row block; a lane ends up with the four channels of one pixel
//                (lane-local division).  The tiles belong to the waves that do NOT stage (two or
//                three apiece, below), and the pixels go through a 16 x 64 LDS tile so that,
//                after the barrier, wave w stores row w as one contiguous 512-byte segment (in the
//                shadow of the row pass's MFMA chain) instead of 64 separate 8-byte writes.
//
// Ring: a 16-output column tile reads NG = 2*NC groups of 16 rows; the ring holds one more
// (NR = NG+1), so the column pass of block g-NG only needs groups the PREVIOUS iterations
// wrote and runs in the same barrier interval as the staging of group g:
//   iteration g:  staging waves: stage group g, fetch group g+1 | tile waves: column pass of
//                 block g-NG -> out_tile | X | row pass of group g -> ring slot g mod NR, store of
//                 out_tile's rows | Y
// What the knock-out builds say (-DMH_FUSED_KNOCK, tools/gpu_ab2.sh; profiles/r2_notes): the costs
// of the parts ADD UP (result stores 10 %, source fetches 10 %, staging conversion 13 %, column
// pass 18 %, row pass 33 %, barriers and loop 16 %): with four waves on a SIMD and two barriers
// per group the iteration is a chain of latencies, not a throughput limit.
// (fused16_read_degree / fused16_layout: mfma_common.hpp)
// (Fused16Geometry: mfma_common.hpp — shared with convolve_fused_exact.hip)
typedef float floatx4 __attribute__((ext_vector_type(4)));

template<int NC,int MODE,bool UNSHARP>
__global__ __launch_bounds__(1024)
void blur_fused16_kernel(BlurFusedArgs args)
{
  // MFMA_PLAIN3 (RGB, 6-byte pixels) runs as four plain channels whose fourth is zero: only the
  // pixel loads and stores differ
  constexpr int PX=MODE == MFMA_PLAIN3 ? 3 : 4;          // u16 per pixel in memory
  constexpr int SAMPLES=MODE == MFMA_PLAIN3 ? MFMA_PLAIN4 : MODE;
  // (6-byte pixels are 2-byte aligned; global memory takes dword accesses at that alignment)
  typedef unsigned __attribute__((aligned(2))) LooseDword;
  auto load_pixel16=[&](const uint16_t *at) -> uint2
  {
    if constexpr (MODE == MFMA_PLAIN3)
      return make_uint2(*reinterpret_cast<const LooseDword *>(at),(unsigned) at[2]);
    else
      return *reinterpret_cast<const uint2 *>(at);
  };
  auto store_pixel16=[&](uint16_t *at,uint2 value)
  {
    if constexpr (MODE == MFMA_PLAIN3)
      {
        *reinterpret_cast<LooseDword *>(at)=value.x;
        at[2]=(uint16_t) value.y;
      }
    else
      *reinterpret_cast<uint2 *>(at)=value;
  };
  typedef Fused16Geometry<NC> G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *ring_hi=reinterpret_cast<_Float16 *>(smem_raw);
  _Float16 *ring_lo=ring_hi+G::RING_PLANE;
  _Float16 *stage_hi=ring_lo+G::RING_PLANE;
  _Float16 *stage_lo=stage_hi+G::STAGE_PLANE;
  uint2 *out_tile=reinterpret_cast<uint2 *>(smem_raw+G::planes_bytes);
  const int tid=(int) threadIdx.x,lane=tid & 63;
  const int wave=__builtin_amdgcn_readfirstlane(tid >> 6);
  const int n=lane & 15,kq=lane >> 4;
  const int K=args.ntaps;
  const int W=args.columns,H=args.rows;

  const int items=args.strips*args.segments;
  const int item=((int) blockIdx.x & 7)*args.items_per_xcd+((int) blockIdx.x >> 3);
  if (item >= items)
    return;
  const int segment=item/args.strips,strip=item-segment*args.strips;
  const int x0=G::COLS*strip;
  const int block_begin=segment*args.blocks_per_segment;
  const int block_end=block_begin+args.blocks_per_segment < args.blocks ?
    block_begin+args.blocks_per_segment : args.blocks;
  const int nblocks=block_end-block_begin;     // blocks of 16 output rows
  const int out_begin=G::GROUP*block_begin;
  const int in0=out_begin-args.shift;
  const int xin0=x0-args.shift;
  const int ngroups=nblocks+G::NG-1;
#ifdef MH_FUSED_TRACE
  const bool traced=(args.trace != nullptr) && (blockIdx.x < 4) && ((wave & 3) == 0);
#endif

  // ---- Toeplitz operands: T[c][i] = 256*tap[32c+8*kq+i-n]
  half8 t_hi[NC],t_lo[NC];
  {
    float *tap_lds=reinterpret_cast<float *>(stage_hi);
    for (int j=tid; j < K; j+=1024)
      tap_lds[j]=args.taps[j];
    __syncthreads();
#pragma unroll
    for (int c=0; c < NC; c++)
#pragma unroll
      for (int i=0; i < 8; i++)
        {
          int j=32*c+8*kq+i-n;
          const bool inside=(j >= 0) && (j < K);
          j=inside ? j : 0;
          const float t=inside ? 256.0f*tap_lds[j] : 0.0f;
          _Float16 h,l;
          split_f16(t,h,l);
          t_hi[c][i]=h;
          t_lo[c][i]=l;
        }
    __syncthreads();                             // tap_lds is the staging plane
  }

  // ---- staging: thread -> (row, 4 consecutive columns) of the 16 x XS source window
  const bool stager=tid < G::FETCH_GROUPS;     // wave-uniform (FETCH_GROUPS is a multiple of 64)
  const int srow=tid/G::GROUPS_PER_ROW,sxg=tid-srow*G::GROUPS_PER_ROW;
  uint2 raw[4];
  auto fetch=[&](int g)
  {
    if (stager)
      {
        int y=in0+G::GROUP*g+srow;
        y=y < 0 ? 0 : (y > H-1 ? H-1 : y);       // the intermediate's edge clamp (cache.c:2663-2679)
        const int xs=xin0+4*sxg;
        if ((MODE == MFMA_PLAIN3) && (xs >= 0) && (xs+3 <= W-1))
          {
            // four RGB pixels = 24 contiguous bytes: three 8-byte loads, re-cut into pixels
            const uint16_t *at=args.src+pixel_index(y,W,xs)*3;
            const LooseDword *words=reinterpret_cast<const LooseDword *>(at);
            const uint2 a=make_uint2(words[0],words[1]),b=make_uint2(words[2],words[3]),c=make_uint2(words[4],words[5]);
            raw[0]=make_uint2(a.x,a.y & 0xffffu);
            raw[1]=make_uint2((a.y >> 16) | (b.x << 16),b.x >> 16);
            raw[2]=make_uint2(b.y,c.x & 0xffffu);
            raw[3]=make_uint2((c.x >> 16) | (c.y << 16),c.y >> 16);
          }
        else if ((MODE != MFMA_PLAIN3) && (xs >= 0) && (xs+3 <= W-1))
          {
            // four 8-byte pixels = 32 contiguous bytes: two 16-byte loads, one address
            // (pixel alignment only: 8 bytes)
            typedef unsigned LooseQuad __attribute__((ext_vector_type(4),aligned(8)));
            const LooseQuad *at=reinterpret_cast<const LooseQuad *>(args.src+pixel_index(y,W,xs)*4);
            const LooseQuad a=at[0],b=at[1];
            raw[0]=make_uint2(a[0],a[1]);
            raw[1]=make_uint2(a[2],a[3]);
            raw[2]=make_uint2(b[0],b[1]);
            raw[3]=make_uint2(b[2],b[3]);
          }
        else
          {
#pragma unroll
            for (int i=0; i < 4; i++)
              {
                int x=xs+i;
                x=x < 0 ? 0 : (x > W-1 ? W-1 : x);
                raw[i]=load_pixel16(args.src+pixel_index(y,W,x)*PX);
              }
          }
      }
  };
  auto stage=[&]()
  {
    if (stager)
      {
        f32x2 v[4][2];
        quantum_to_samples<SAMPLES>(raw,v);
#pragma unroll
        for (int c=0; c < 4; c++)
          {
            uint2 hi,lo;
            split_f16_pair(v[c][0],hi.x,lo.x);
            split_f16_pair(v[c][1],hi.y,lo.y);
            const int at=c*G::CHR+srow*G::SR+4*sxg;
            *reinterpret_cast<uint2 *>(stage_hi+at)=hi;
            *reinterpret_cast<uint2 *>(stage_lo+at)=lo;
          }
      }
  };

  // row pass: wave = row quad (4 rows) x output tile (16 columns); entry e = 4*channel+row
  const int rq=wave & 3,ot=wave >> 2;
  // entries e = 4*row + channel: D then hands a lane the four channels of ONE pixel, so the
  // division by the alpha sum is lane-local (as in the column pass)
  const int row_entry=(n & 3)*G::CHR+(4*rq+(n >> 2))*G::SR+16*ot+8*kq;
  // column pass: wave = column quad; entry e = 4*column+channel; a 32-row chunk spans two ring
  // groups: k 0..15 (kq 0,1) in the first, k 16..31 (kq 2,3) in the next
  // The staging waves (tid < FETCH_GROUPS: 9 of 16 for 79 taps) convert and stage; the column
  // pass's sixteen tiles (four columns each) belong to the OTHER waves, two or three apiece.  A
  // staging wave's chain per iteration is then wait - convert - stage - fetch and a tile wave's
  // two or three short multiplies, side by side on every SIMD (wave w runs on SIMD w & 3: each
  // SIMD holds two or three staging waves and one or two tile waves).  Before, every wave owned
  // one tile and the staging waves started theirs after staging: the others waited for them.
  static_assert((G::FETCH_GROUPS % 64) == 0,"whole staging waves");
  constexpr int TILE_WAVES=16-G::FETCH_GROUPS/64;
  constexpr int CT=(16+TILE_WAVES-1)/TILE_WAVES;
  const int tile_wave=wave-(16-TILE_WAVES);    // < 0: a staging wave
  const int ctiles=tile_wave < 0 ? 0 : 16/TILE_WAVES+(tile_wave < 16 % TILE_WAVES ? 1 : 0);
  const int ctile0=tile_wave < 0 ? 0 : tile_wave*(16/TILE_WAVES)+(tile_wave < 16 % TILE_WAVES ? tile_wave : 16 % TILE_WAVES);
  constexpr int GROUP_STRIDE=2*G::OB;
  const int col_entry=(n & 3)*G::CHC+(4*ctile0+(n >> 2))*8+(kq & 1)*G::OB;
  const int ring_entry=kq*G::CHC+(rq >> 1)*G::OB+(16*ot+n)*8+4*(rq & 1);   // the row pass's ring store
  int ring_group=0;                            // g mod NR (wave-uniform)
  // The column pass leaves its 16 x 64 pixels in out_tile; after barrier X wave w stores row w:
  // 64 lanes x 8 bytes = one contiguous 512-byte segment (the tiles' own lanes hold 16 ROWS of
  // 4 pixels each: 64 separate 8-byte writes per store instruction, 1024 partial-line write
  // requests per iteration and CU, and the memory pipe's queue backed up into the staging waves'
  // fetches).  UNSHARP: the unblurred pixel (effect.c:4364-4369) of that lane, fetched at the
  // top of the iteration (the strip's rows left the CU 32*NC rows ago: an L2 / MALL hit).
  uint2 original=make_uint2(0u,0u);
  auto fetch_original=[&](int block)
  {
    const int x=x0+lane,y=out_begin+G::GROUP*block+wave;
    if ((block >= 0) && (block < nblocks) && (x < W) && (y < H))
      original=load_pixel16(args.src+pixel_index(y,W,x)*PX);
  };

  fetch(0);
  for (int g=0; g <= ngroups; g++)
    {
      if constexpr (UNSHARP)
        fetch_original(g-G::NG);
      MH_FTRACE_MARK(0);
      if (g < ngroups)
        {
#ifdef MH_FUSED_TRACE
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          MH_FTRACE_MARK(1);
#endif
          if (!(MH_KNOCKED(4) && (g > 1)))
            stage();
#ifdef MH_FUSED_TRACE
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          MH_FTRACE_MARK(2);
#endif
          if ((g+1 < ngroups) && !(MH_KNOCKED(2) && (g > 1)))
            fetch(g+1);
        }
      MH_FTRACE_MARK(3);
      if ((g >= G::NG) && !MH_KNOCKED(8))
        {
          // ---- column pass of output rows out_begin+16*(g-NG) .. +16
          // block mod NR = (g+1) mod NR (NR = NG+1): the oldest group the ring still holds
          const int first=ring_group+1 == G::NR ? 0 : ring_group+1;
          // the ring group of chunk c for this lane: (first + 2c + (kq>>1)) mod NR for a value
          // below 2*NR: min with the wrapped difference
          int chunk_at[NC];
#pragma unroll
          for (int c=0; c < NC; c++)
            {
              const unsigned wide=(unsigned) (first+2*c+(kq >> 1));
              const unsigned group=wide < wide-(unsigned) G::NR ? wide : wide-(unsigned) G::NR;
              chunk_at[c]=col_entry+GROUP_STRIDE*(int) group;
            }
          // the wave's tiles side by side: their MFMA chains are independent and interleave
          auto column_tiles=[&](auto count)
          {
            constexpr int N=decltype(count)::value;
            floatx4 acc[N > 0 ? N : 1];
#pragma unroll
            for (int t=0; t < N; t++)
              acc[t]=floatx4{0.0f,0.0f,0.0f,0.0f};
#pragma unroll
            for (int c=0; c < NC; c++)
              {
                half8 a_hi[N > 0 ? N : 1],a_lo[N > 0 ? N : 1];
#pragma unroll
                for (int t=0; t < N; t++)
                  {
                    a_hi[t]=*reinterpret_cast<const half8 *>(ring_hi+chunk_at[c]+4*t*G::SC);
                    a_lo[t]=*reinterpret_cast<const half8 *>(ring_lo+chunk_at[c]+4*t*G::SC);
                  }
#pragma unroll
                for (int t=0; t < N; t++)
                  acc[t]=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[t],t_hi[c],acc[t],0,0,0);
#pragma unroll
                for (int t=0; t < N; t++)
                  acc[t]=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[t],t_hi[c],acc[t],0,0,0);
#pragma unroll
                for (int t=0; t < N; t++)
                  acc[t]=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[t],t_lo[c],acc[t],0,0,0);
              }
            // lane (n, kq): the four channels (registers) of pixel (column 4*tile+kq, row n)
#pragma unroll
            for (int t=0; t < N; t++)
              out_tile[n*G::OUT_STRIDE+4*(ctile0+t)+kq]=sums_to_quantum<SAMPLES>(acc[t][0],acc[t][1],acc[t][2],acc[t][3]);
          };
          if (ctiles == CT)
            column_tiles(std::integral_constant<int,CT>{});
          else if (ctiles == CT-1)
            column_tiles(std::integral_constant<int,CT-1>{});
        }
      MH_FTRACE_MARK(4);
      __syncthreads();                           // X: staged; the column pass's pixels are in out_tile
      auto store_row=[&]()
      {
        if (g >= G::NG)
          {
            uint2 result=out_tile[wave*G::OUT_STRIDE+lane];
            if constexpr (UNSHARP)
              result=unsharp_pixel(original,result,args.gain,args.threshold);
            const int x=x0+lane,y=out_begin+G::GROUP*(g-G::NG)+wave;
            if ((x < W) && (y < H) && !MH_KNOCKED(1))
              store_pixel16(args.dst+pixel_index(y,W,x)*PX,result);
          }
      };
      if (g == ngroups)
        {
          store_row();
          break;
        }
      MH_FTRACE_MARK(5);
      // ---- row pass of ring group g
      if (!MH_KNOCKED(64))
      {
        half8 a_hi[NC],a_lo[NC];
#pragma unroll
        for (int c=0; c < NC; c++)
          {
            a_hi[c]=*reinterpret_cast<const half8 *>(stage_hi+row_entry+32*c);
            a_lo[c]=*reinterpret_cast<const half8 *>(stage_lo+row_entry+32*c);
          }
        floatx4 acc={0.0f,0.0f,0.0f,0.0f};
        if (MH_KNOCKED(16))
          acc=floatx4{(float) a_hi[0][0],(float) a_lo[1 % NC][1],(float) a_hi[(NC-1)][2],1024.0f+(float) a_lo[0][3]};
        else
          {
#pragma unroll
        for (int c=0; c < NC; c++)
          {
            acc=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[c],t_hi[c],acc,0,0,0);
            acc=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[c],t_hi[c],acc,0,0,0);
            acc=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[c],t_lo[c],acc,0,0,0);
          }
          }
        // the store of the column pass's row in the shadow of the MFMA chain (its LDS read and
        // address arithmetic need no matrix result)
        store_row();
#ifdef MH_FUSED_TRACE
        asm volatile("s_nop 0" :: "v"(acc[0]),"v"(acc[3]));      // the chain has completed
        MH_FTRACE_MARK(6);
#endif
        // lane (n, kq): the four channels (registers) of pixel (column x0+16*ot+n, row 4*rq+kq).
        // Quantum-rounded colour = 65536*S_c/S_a and alpha = S_a/128 (sums_to_quantum); the
        // column pass's samples: alpha*colour*2^-17 and alpha/2 (plain: level/2).
        float v[4];
        if (MH_KNOCKED(32))
          {
            v[0]=acc[0]; v[1]=acc[1]; v[2]=acc[2]; v[3]=acc[3];
          }
        else
        {
          const uint2 q=sums_to_quantum<SAMPLES>(acc[0],acc[1],acc[2],acc[3]);
          const f32x2 c01={(float) (q.x & 0xffffu),(float) (q.x >> 16)};
          const f32x2 c23={(float) (q.y & 0xffffu),(float) (q.y >> 16)};
          if constexpr (MODE == MFMA_BLEND4)
            {
              // the row pass's alpha becomes a weight: exact where it is small and the f32 sum
              // cannot decide the level (mfma_common.hpp)
              float alpha=c23[1];
              if ((acc[3] < kSmallAlpha*128.0f) && alpha_sum_is_ambiguous(acc[3]))
                {
                  const int x=x0+16*ot+n;
                  int y=in0+G::GROUP*g+4*rq+kq;
                  y=y < 0 ? 0 : (y > H-1 ? H-1 : y);
                  if (x < W)
                    alpha=(float) exact_alpha_level(args.src,pixel_index(y,W,0),1,W,x-args.shift,args.taps64,K);
                }
              const float weight=alpha*(0.5f/65536.0f);
              const f32x2 v01=c01*f32x2{weight,weight};
              v[0]=v01[0]; v[1]=v01[1];
              v[2]=c23[0]*weight;
              v[3]=alpha*0.5f;
            }
          else
            {
              const f32x2 v01=c01*0.5f,v23=c23*0.5f;
              v[0]=v01[0]; v[1]=v01[1]; v[2]=v23[0]; v[3]=v23[1];
            }
        }
        // 4x4 transpose between the registers (channels) and the four 16-lane rows (pixel rows):
        // v_permlane32_swap on (0,2),(1,3), v_permlane16_swap on (0,1),(2,3).  Afterwards lane
        // (n, kq) holds channel kq of rows 4*rq+0..3 — four consecutive rows for one 8-byte ring
        // store per plane.
        // (inline asm: hipcc of ROCm 7.2 loses the second result of a v_permlane16_swap builtin
        // that follows a v_permlane32_swap builtin; s_nop 1 = the two wait states a swap needs
        // after a VALU write of its operands)
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"
                     "s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3"
                     : "+v"(v[0]),"+v"(v[1]),"+v"(v[2]),"+v"(v[3]));
        uint2 hi,lo;
        split_f16_pair(f32x2{v[0],v[1]},hi.x,lo.x);
        split_f16_pair(f32x2{v[2],v[3]},hi.y,lo.y);
        const int at=ring_entry+ring_group*GROUP_STRIDE;
        *reinterpret_cast<uint2 *>(ring_hi+at)=hi;
        *reinterpret_cast<uint2 *>(ring_lo+at)=lo;
      }
#ifdef MH_FUSED_TRACE
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      MH_FTRACE_MARK(7);
#endif
      __syncthreads();                           // Y: ring group g complete, staging reads done
      MH_FTRACE_MARK(8);
      ring_group=ring_group+1 == G::NR ? 0 : ring_group+1;
    }
}

template<int NC,int MODE,bool UNSHARP>
static MhStatus launch_fused16_typed(const View &src,BlurFusedArgs &args)
{
  typedef Fused16Geometry<NC> G;
  args.strips=(args.columns+G::COLS-1)/G::COLS;
  args.blocks=(args.rows+G::GROUP-1)/G::GROUP;
  // Cut the strips so that every CU gets a work item.  A segment recomputes NG-1 ring groups
  // (the K-1 halo rows of its first block), so it stays at least 16 blocks long.
  const int cus=compute_units(src.device);
  const int max_segments=args.blocks/16 > 1 ? args.blocks/16 : 1;
  int segments=(cus+args.strips-1)/args.strips;
  segments=segments < 1 ? 1 : (segments > max_segments ? max_segments : segments);
  if (const char *e=option("MAGICKHIP_FUSED_SEGMENTS"))
    segments=atoi(e) < 1 ? 1 : (atoi(e) > args.blocks ? args.blocks : atoi(e));
  args.blocks_per_segment=(args.blocks+segments-1)/segments;
  args.segments=(args.blocks+args.blocks_per_segment-1)/args.blocks_per_segment;   // no empty segment
  const int items=args.strips*args.segments;
  args.items_per_xcd=(items+7)/8;
  const size_t lds=G::lds_bytes;
  MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&blur_fused16_kernel<NC,MODE,UNSHARP>),
    hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
#ifdef MH_FUSED_TRACE
  const char *trace_path=option("MAGICKHIP_FUSED_TRACE");
  const size_t trace_bytes=4u*4u*48u*10u*sizeof(unsigned long long);
  if (trace_path != nullptr)
    {
      MH_HIP(hipMalloc(reinterpret_cast<void **>(&args.trace),trace_bytes));
      MH_HIP(hipMemsetAsync(args.trace,0,trace_bytes,src.stream));
    }
#endif
  {
    ProfileScope prof(UNSHARP ? "unsharp_fused" : "blur_fused",src.stream);
    hipLaunchKernelGGL((blur_fused16_kernel<NC,MODE,UNSHARP>),dim3((unsigned) (8*args.items_per_xcd)),dim3(1024),lds,
      src.stream,args);
    MH_HIP(hipGetLastError());
  }
#ifdef MH_FUSED_TRACE
  if (args.trace != nullptr)
    {
      std::vector<unsigned long long> host(trace_bytes/sizeof(unsigned long long));
      MH_HIP(hipMemcpyAsync(host.data(),args.trace,trace_bytes,hipMemcpyDeviceToHost,src.stream));
      MH_HIP(hipStreamSynchronize(src.stream));
      MH_HIP(hipFree(args.trace));
      if (FILE *f=fopen(trace_path,"wb"))
        {
          fwrite(host.data(),1,trace_bytes,f);
          fclose(f);
        }
    }
#endif
  return MH_OK;
}

template<int NC>
static MhStatus launch_fused16(const View &src,BlurFusedArgs &args,bool blend,bool unsharp)
{
  if (src.channels == 3)
    return unsharp ? launch_fused16_typed<NC,MFMA_PLAIN3,true>(src,args) :
      launch_fused16_typed<NC,MFMA_PLAIN3,false>(src,args);
  if (unsharp)
    return blend ? launch_fused16_typed<NC,MFMA_BLEND4,true>(src,args) :
      launch_fused16_typed<NC,MFMA_PLAIN4,true>(src,args);
  return blend ? launch_fused16_typed<NC,MFMA_BLEND4,false>(src,args) :
    launch_fused16_typed<NC,MFMA_PLAIN4,false>(src,args);
}

// unsharp: UnsharpMaskImage(gain, threshold) of src with this blur, in the same launch (the
// column pass's copy-out applies effect.c:4364-4369 against the unblurred pixel).
MhStatus launch_blur_fused(const View &src,const View &dst,const float *taps_device,
  const double *taps64_device,int ntaps,int shift,bool blend,bool *handled,bool unsharp,double gain,
  double threshold)
{
  *handled=false;
  if ((src.quantum != MH_QUANTUM_U16) || (dst.quantum != MH_QUANTUM_U16) ||
      ((src.channels != 4) && ((src.channels != 3) || blend)) ||
      (dst.channels != src.channels) || (src.columns != dst.columns) || (src.rows != dst.rows) || (ntaps < 2))
    return MH_OK;
  if ((src.columns >= (1u << 24)) || (src.rows >= (1u << 24)) ||
      ((unsigned long long) src.columns*src.rows >= (1ull << 32)))
    return MH_OK;                                // pixel_index()
  if (ntaps > 81)
    return MH_OK;                                // three 32-sample chunks hold 94 band slots
  BlurFusedArgs args;
  args.trace=nullptr;
  args.src=static_cast<const uint16_t *>(src.pixels);
  args.dst=static_cast<uint16_t *>(dst.pixels);
  args.columns=(int) src.columns;
  args.rows=(int) src.rows;
  args.ntaps=ntaps;
  args.shift=shift;
  args.taps=taps_device;
  args.taps64=taps64_device;
  args.gain=(float) gain;
  {
    const double level=std::ceil(65535.0*threshold);
    args.threshold=level > 131072.0 ? 131072 : (level < 0.0 ? 0 : (int) level);
  }
  *handled=true;
  const int nc=(ntaps+15+31)/32;                 // 16 outputs + K-1 halo, in 32-sample chunks
  if (nc == 1)
    return launch_fused16<1>(src,args,blend,unsharp);
  if (nc == 2)
    return launch_fused16<2>(src,args,blend,unsharp);
  if (nc == 3)
    return launch_fused16<3>(src,args,blend,unsharp);
  *handled=false;                                // wider kernels: row pass + (fused) column pass
  return MH_OK;
}

} // namespace mh
