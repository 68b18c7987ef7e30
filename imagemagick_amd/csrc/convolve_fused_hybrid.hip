// BlurImage's two passes in ONE launch, MH_PRECISION_FAST (Q16; RGBA with alpha-weighted colour,
// four plain channels, RGB): f16 matrix-core sums for the colour, EXACT integer sums for the one
// thing the +-1 contract needs exactly — the row pass's alpha.
// MagickCore/effect.c:765-796 -> morphology.c:2811-2979 (row kernel) -> :2654-2807 (column kernel),
// the Quantum-rounded intermediate of :4012-4022.
//
// Why this is within +-1 level of the reference on any content:
//
//   The reference's column pass forms  O_c = round( sum_j k_j A_j C_j / sum_j k_j A_j )  over the
//   ROUNDED row-pass results A_j = round(alpha sum), C_j = round(R_j), R_j = N_j/D_j.
//   * A_j is a WEIGHT.  A weight that is one level off moves O_c by (C_j - O_c)/A per sample —
//     many levels where alpha is small or the colour varies.  So A_j has to be the reference's
//     own level: the alpha sums of the row pass are formed as exact integers on the i8 matrix
//     cores, with the certificate and the recomputation of convolve_fused_exact.hip
//     (blur_exact_common.hpp) — bit-identical to the reference's intermediate alpha.
//   * C_j enters linearly under non-negative weights: replacing C_j by the UNROUNDED value R_j
//     (+ the f16 path's error eps) moves the real value the column pass rounds by a weighted mean
//     of (R_j + eps_j - round(R_j)), i.e. by at most 0.5 + eps.  With the column pass's own error
//     delta the two real values differ by less than 1 — and |a - b| < 1 implies
//     |round(a) - round(b)| <= 1.  No tie structure of the frame can line the differences up to 2
//     (round 2's kernel ROUNDED its approximate intermediate: +-1 per sample, +-2 after the second
//     pass on checkerboards), because nothing is rounded here that the reference does not round.
//   * plain channels (no alpha weighting): the same argument with A_j = 1 — no exact part at all.
//   eps + delta: f16 hi/lo operands (22 bits), f32 accumulation: measured < 0.05 level — while every tap's operand
//   carries its bits: the operands are scale*tap with scale the power of two that puts the kernel's LARGEST tap
//   just under 2^15 (f16_tap_scale), and the launcher declines a kernel with a tap below 2^-19 of the largest
//   (f16_taps_resolved): where such a tap is the only one that meets an opaque sample it is the whole result.
//
// Per 16 x 64 pixels the row pass is 12 x 9 v_mfma_f32_16x16x32_f16 (three colour channels x four 16-column
// tiles, entry = row: no tile entry idles on the alpha the integer path supplies) + 4 x 18 v_mfma_i32_16x16x64_i8
// (four waves, one 16-row x 16-column alpha tile each: nine digit products, two chunks) against 16 x 28 i8
// instructions for the all-exact row pass of convolve_fused_exact.hip, and its epilogue is f32 — one multiply by
// the alpha waves' weight per value instead of ~68 fp64-rate instructions per pixel.
//
// The walk (strips of 64 columns, groups of 16 rows, a ring of the unrounded row-pass result in LDS, two barrier
// intervals an iteration, XCD-aware item order) is described at the loop; DESIGN.md section 4.1.1 has the LDS
// budget and what bounds the kernel.
#include "blur_exact_common.hpp"
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <type_traits>

namespace mh {

// worst number of ds_read_b128 lines of one lane group that share a 16-byte slot, for the alpha
// tile's byte-plane operand: entry e = lane&15 = row, k quarter = lane>>4 -> the next 16 columns
static constexpr int alpha_plane_degree(int stride)
{
  int worst=1;
  for (int g=0; g < 4; g++)
    {
      int count[16]={0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
      for (int i=0; i < 16; i++)
        {
          const int base=(g & 1) == 0 ? (i < 4 ? i : (i < 8 ? i+8 : i+12)) : (i < 8 ? i+4 : (i < 12 ? i+8 : i+16));
          const int lane=base+32*(g >> 1);
          const int e=lane & 15,kq=lane >> 4;
          const int slot=((e*stride+16*kq) % 256)/16;
          count[slot]++;
          worst=count[slot] > worst ? count[slot] : worst;
        }
    }
  return worst;
}

static constexpr int alpha_plane_stride(int extent)
{
  int best=(extent+15) & ~15,best_degree=99;
  for (int stride=(extent+15) & ~15; stride <= extent+96; stride+=16)
    if (alpha_plane_degree(stride) < best_degree)
      {
        best_degree=alpha_plane_degree(stride);
        best=stride;
      }
  return best;
}

template<int NC,bool BLEND>
struct HybridGeometry
{
  typedef Fused16Geometry<NC> F;
  static constexpr int COLS=64,GROUP=16;
  static constexpr int NG=2*NC,NR=NG+1;
  static constexpr int NX=(NC+1)/2;            // 64-slot chunks of the alpha tile's band
  static constexpr int XS=F::XS;
  // f16 staging planes: three colour channels (alpha-weighted frames: the alpha sums are the
  // integer ones) or four plain channels
  static constexpr int STAGE_CHANNELS=BLEND ? 3 : 4;
  static constexpr int STAGE_PLANE=STAGE_CHANNELS*F::CHR;           // halves
  static constexpr int ring_bytes=(int) (2*F::RING_PLANE*sizeof(_Float16));
  static constexpr int stage_bytes=(int) (2*STAGE_PLANE*sizeof(_Float16));
  // the alpha levels of the staged window as two signed-byte planes [row][column] (low, high)
  static constexpr int ASTRIDE=alpha_plane_stride(XS);
  static constexpr int alpha_bytes=BLEND ? 2*GROUP*ASTRIDE : 0;
  // the colour quotients' weights A/(scale*D) (f32) of a group's pixels on their way from the alpha waves to the
  // row waves: [column][16 rows], a lane's four rows one 16-byte slot (swizzled with the column: weight_slot)
  static constexpr int sum_bytes=BLEND ? (int) (COLS*GROUP*sizeof(float)) : 0;
  static constexpr int OUT_STRIDE=COLS+1;
  static constexpr int out_bytes=(int) (GROUP*OUT_STRIDE*sizeof(uint2));
  // the taps' digits for the alpha tiles' Toeplitz operands: digit j of tap v at [j][v+15], v =
  // -15..95 (zeros outside the kernel), in four copies shifted by 0..3 bytes so that every lane's
  // 16-byte window (it begins at tap 16*kq-n) starts on a dword of one of them.  The operands
  // are read from here for every group: sixty registers of per-lane constants do not fit beside
  // the f16 operands (the compiler spilled them to scratch and reloaded them inside the walk).
  static constexpr int DLP=116;
  static constexpr int digit_bytes=BLEND ? 4*kExactDigits*DLP : 0;
  static constexpr size_t lds_bytes=(size_t) ring_bytes+stage_bytes+alpha_bytes+sum_bytes+out_bytes+digit_bytes;
  static_assert(lds_bytes <= 163840,"more than the 160 KiB of a CU");
  static constexpr int GROUPS_PER_ROW=XS/4;
  static constexpr int FETCH_GROUPS=GROUP*GROUPS_PER_ROW;
  static_assert(FETCH_GROUPS <= 1024,"one staging round");
  static_assert((FETCH_GROUPS % 64) == 0,"whole staging waves");
  static_assert((alpha_bytes % 16) == 0,"aligned planes");
};

// Diagnostic builds only (-DMH_HYBRID_KNOCK, tools/gpu_hybrid_knock.sh): parts of the alpha path are
// skipped at run time (bits of MAGICKHIP_HYBRID_KNOCK, handed over in args.threshold) to see what
// each costs.  The results are wrong.  1 the alpha tiles' digit loads and products, 2 the alpha
// epilogue of interval A, 4 the colour epilogue's reads of the exact alpha, 8 exact_sums,
// 16 the staging of the alpha byte planes; 32 / 64 / 128: plain-channel arithmetic in the staging /
// the colour epilogue / the column pass's epilogue of an alpha-weighted frame; 256: every fetch reads
// the strip's first group (cache-resident: no memory latency, no read traffic); 512: no stores
#ifdef MH_HYBRID_KNOCK
#define MH_HKNOCKED(bit) ((args.threshold & (bit)) != 0)
#else
#define MH_HKNOCKED(bit) false
#endif

// Diagnostic build only (-DMH_HYBRID_TRACE, tools/trace_hybrid_blur.py): every wave of the first four
// workgroups stamps the shader clock at the phase boundaries of 48 steady-state iterations:
// trace[block][wave][iteration][mark].
#ifdef MH_HYBRID_TRACE
#define MH_HTRACE_MARK(id) \
  do { \
    if (traced && (g >= 64) && (g < 112)) \
      { \
        const unsigned long long now=__builtin_readcyclecounter(); \
        if (lane == 0) \
          args.trace[((((int) blockIdx.x*16+wave)*48)+(g-64))*12+(id)]=now; \
      } \
  } while (0)
#else
#define MH_HTRACE_MARK(id) do { } while (0)
#endif

template<int NC,int MODE>
__global__ __launch_bounds__(1024)
void blur_fused_hybrid_kernel(BlurExactArgs args)
{
  // MFMA_PLAIN3 (RGB, 6-byte pixels) runs as four plain channels whose fourth is zero: only the
  // pixel loads and stores differ
  constexpr int PX=MODE == MFMA_PLAIN3 ? 3 : 4;          // u16 per pixel in memory
  constexpr int SAMPLES=MODE == MFMA_PLAIN3 ? MFMA_PLAIN4 : MODE;
  constexpr bool BLEND=SAMPLES == MFMA_BLEND4;
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
  typedef HybridGeometry<NC,BLEND> G;
  typedef Fused16Geometry<NC> F;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *ring_hi=reinterpret_cast<_Float16 *>(smem_raw);
  _Float16 *ring_lo=ring_hi+F::RING_PLANE;
  _Float16 *stage_hi=reinterpret_cast<_Float16 *>(smem_raw+G::ring_bytes);
  _Float16 *stage_lo=stage_hi+G::STAGE_PLANE;
  unsigned char *alpha_plane=smem_raw+G::ring_bytes+G::stage_bytes;      // [2][GROUP][ASTRIDE]
  float *alpha_weight=reinterpret_cast<float *>(alpha_plane+G::alpha_bytes); // [COLS][GROUP]
  uint2 *out_tile=reinterpret_cast<uint2 *>(alpha_plane+G::alpha_bytes+G::sum_bytes);
  unsigned char *digit_table=alpha_plane+G::alpha_bytes+G::sum_bytes+G::out_bytes;   // [4][kExactDigits][DLP]
  const int tid=(int) threadIdx.x,lane=tid & 63;
  const int wave=__builtin_amdgcn_readfirstlane(tid >> 6);
  const int n=lane & 15,kq=lane >> 4;
  const int K=args.ntaps;
  const int W=args.columns,H=args.rows;

  const int items=args.strips*args.segments;
  const int item=((int) blockIdx.x & 7)*args.items_per_xcd+((int) blockIdx.x >> 3);
  if (item >= items)
    return;
#ifdef MH_HYBRID_TRACE
  const bool traced=(args.trace != nullptr) && (blockIdx.x < 4);
#endif
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

  // the exact alpha tiles: waves 12..15 (one per SIMD; not stagers), tile t = wave & 3 = 16 rows x columns
  // 16*t..+15; entry e = row, so D hands lane (n, kq) the rows 4*kq..4*kq+3 (registers) of column 16*t+n
  const bool alpha_wave=BLEND && (wave >= 12);
  const int alpha_tile=wave & 3;
  const int alpha_entry=n*G::ASTRIDE+16*alpha_tile+16*kq;
  // this lane's window of the digit table: taps 16*kq-n .. +15 (and 64+8*kq-n .. +7 of the second chunk)
  const int digit_copy=(3-n) & 3;
  const int digit_entry=digit_copy*(kExactDigits*G::DLP)+16*kq+15-n-digit_copy;

  // ---- The matrix instructions' constant operands, in REGISTERS for the whole walk, one array for both kinds
  // of wave (a wave is one or the other; two arrays would both be live across the loop — 54 registers — and the
  // kernel sits at the 128 of four waves a SIMD):
  //   every wave but the alpha waves: the Toeplitz operands of the f16 products (both passes),
  //     T[c][i] = tap_scale*tap[32c+8*kq+i-n], hi/lo split: operand[c] (hi), operand[NC+c] (lo)
  //   alpha waves: the taps' digits for the alpha tiles' Toeplitz operands — digit j of this lane's 16-byte
  //     window: operand[j]; of the 8-byte window of the second chunk (kernels of more than 49 taps):
  //     operand[5] (digits 0, 1) and the three pixel registers of a staging thread (digits 2..4: an alpha wave
  //     does not stage).  Rounds 4-5 re-read them from an LDS table in every iteration (thirty 4-byte reads a
  //     lane): eleven waits for the LDS in front of the integer chain, ~1000 of the 1500 cycles it took — the
  //     longest thing in interval B.  The alpha waves therefore run no column tiles (those need the Toeplitz
  //     operands); the tile waves share all sixteen.
  constexpr int NOPERANDS=!BLEND ? 2*NC : (2*NC > (G::NX == 2 ? 6 : 5) ? 2*NC : (G::NX == 2 ? 6 : 5));
  intx4 operand[NOPERANDS];
  uint2 raw[4];
  auto toeplitz_hi=[&](int c) { return __builtin_bit_cast(half8,operand[c]); };
  auto toeplitz_lo=[&](int c) { return __builtin_bit_cast(half8,operand[NC+c]); };
  {
    float *tap_lds=reinterpret_cast<float *>(stage_hi);
    for (int j=tid; j < K; j+=1024)
      tap_lds[j]=args.taps[j];
    if constexpr (BLEND)
      for (int at=tid; at < G::digit_bytes; at+=1024)
        {
          const int copy=at/(kExactDigits*G::DLP),rest=at-copy*(kExactDigits*G::DLP);
          const int j=rest/G::DLP,v=rest-j*G::DLP+copy-15;
          digit_table[at]=(unsigned char) (((v >= 0) && (v < K)) ? args.digits[j*kExactDigitPitch+v] : (signed char) 0);
        }
    __syncthreads();
#pragma unroll
    for (int i=0; i < NOPERANDS; i++)
      operand[i]=intx4{0,0,0,0};
#pragma unroll
    for (int i=0; i < 4; i++)
      raw[i]=make_uint2(0u,0u);
    if (alpha_wave)
      {
        if constexpr (BLEND)
          {
#pragma unroll
            for (int j=0; j < kExactDigits; j++)
              {
                const unsigned *window=reinterpret_cast<const unsigned *>(digit_table+digit_entry+j*G::DLP);
                operand[j]=intx4{(int) window[0],(int) window[1],(int) window[2],(int) window[3]};
              }
            if constexpr (G::NX == 2)
              {
#pragma unroll
                for (int j=0; j < kExactDigits; j++)
                  {
                    const unsigned *window=reinterpret_cast<const unsigned *>(digit_table+digit_entry+64-8*kq+j*G::DLP);
                    if (j < 2)
                      {
                        operand[NOPERANDS-1][2*j]=(int) window[0];
                        operand[NOPERANDS-1][2*j+1]=(int) window[1];
                      }
                    else
                      raw[j-2]=make_uint2(window[0],window[1]);
                  }
              }
          }
      }
    else
      {
#pragma unroll
        for (int c=0; c < NC; c++)
          {
            half8 t_hi,t_lo;
#pragma unroll
            for (int i=0; i < 8; i++)
              {
                int j=32*c+8*kq+i-n;
                const bool inside=(j >= 0) && (j < K);
                j=inside ? j : 0;
                const float tap=inside ? args.tap_scale*tap_lds[j] : 0.0f;
                _Float16 h,l;
                split_f16(tap,h,l);
                t_hi[i]=h;
                t_lo[i]=l;
              }
            operand[c]=__builtin_bit_cast(intx4,t_hi);
            operand[NC+c]=__builtin_bit_cast(intx4,t_lo);
          }
      }
    __syncthreads();                             // tap_lds is the staging plane
  }

  // ---- staging: thread -> (row, 4 consecutive columns) of the 16 x XS source window
  // (the walk recomputes a stager's row and column group from its thread index in every iteration
  // — a handful of vector instructions — instead of keeping them, the addresses derived from
  // them and the other loop-invariant indices in registers: the kernel sits at the 128 registers
  // of a wave, and what the compiler spills it reloads from scratch INSIDE the walk, each reload
  // behind an s_waitcnt vmcnt(0) that also waits for the prefetched pixels: 9 such reloads cost
  // 0.25 ms per 8192^2 frame.)
  const bool stager=tid < G::FETCH_GROUPS;     // wave-uniform (FETCH_GROUPS is a multiple of 64)
  auto fetch=[&](int g,int srow,int sxg)
  {
    if (stager)
      {
        int y=in0+G::GROUP*(MH_HKNOCKED(256) ? 0 : g)+srow;
        y=y < 0 ? 0 : (y > H-1 ? H-1 : y);       // the intermediate's edge clamp (cache.c:2663-2679)
        const int xs=xin0+4*sxg;
        if ((MODE == MFMA_PLAIN3) && (xs >= 0) && (xs+3 <= W-1))
          {
            // four RGB pixels = 24 contiguous bytes, re-cut into pixels
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
            // (the strips at the left and right image edges only; opaque to the optimiser so that
            // the four clamped columns are not kept in registers across the whole walk)
            int edge=xs;
            asm volatile("" : "+v"(edge));
#pragma unroll
            for (int i=0; i < 4; i++)
              {
                int x=edge+i;
                x=x < 0 ? 0 : (x > W-1 ? W-1 : x);
                raw[i]=load_pixel16(args.src+pixel_index(y,W,x)*PX);
              }
          }
      }
  };
  // raw Quantum pixels -> the f16 operand planes (alpha*p*2^-17 per colour channel, or level/2 of
  // a plain channel: hi/lo split) and, alpha-weighted frames, the alpha levels as two byte planes
  auto stage_group=[&](int srow,int sxg)
  {
    if (stager)
      {
        f32x2 v[4][2];
        if (MH_HKNOCKED(32))
          quantum_to_samples<MFMA_PLAIN4>(raw,v);
        else
          quantum_to_samples<SAMPLES>(raw,v);
        // (one address + immediate offsets)
        unsigned char *to=smem_raw+G::ring_bytes+2*(srow*F::SR+4*sxg);
        // (RGB frames: the fourth plane stays as it is — its row tiles are not run, what the column pass makes of it
        // stays in the entries of the channel that is not stored)
        constexpr int STAGED=MODE == MFMA_PLAIN3 ? 3 : G::STAGE_CHANNELS;
#pragma unroll
        for (int c=0; c < STAGED; c++)
          {
            uint2 hi,lo;
            split_f16_pair(v[c][0],hi.x,lo.x);
            split_f16_pair(v[c][1],hi.y,lo.y);
            *reinterpret_cast<uint2 *>(to+2*c*F::CHR)=hi;
            *reinterpret_cast<uint2 *>(to+2*(G::STAGE_PLANE+c*F::CHR))=lo;
          }
        if constexpr (BLEND)
          if (!MH_HKNOCKED(16))
          {
            // bytes 2,3 of the second word of each pixel = the alpha level, paired over the four positions
            const unsigned hy01=__builtin_amdgcn_perm(raw[1].y,raw[0].y,0x07030602u),hy23=__builtin_amdgcn_perm(raw[3].y,raw[2].y,0x07030602u);
            unsigned char *line=smem_raw+G::ring_bytes+G::stage_bytes+srow*G::ASTRIDE+4*sxg;
            *reinterpret_cast<unsigned *>(line)=__builtin_amdgcn_perm(hy23,hy01,0x05040100u) ^ 0x80808080u;
            *reinterpret_cast<unsigned *>(line+G::GROUP*G::ASTRIDE)=__builtin_amdgcn_perm(hy23,hy01,0x07060302u) ^ 0x80808080u;
          }
      }
  };

  // row pass: wave = ONE channel x output tile (16 columns) x the group's 16 rows; entry e = row, so D hands
  // lane (n, kq) rows 4*kq..4*kq+3 of column 16*ot+n — four consecutive rows of one ring column, the column
  // pass's 8-byte unit, with no transpose.  Alpha-weighted frames: three channels x four tiles = waves 0..11
  // (three a SIMD); the alpha waves 12..15 run the integer chain only.  (Rounds 4-5 made the entries
  // 4 rows x 4 channels: on an alpha-weighted frame the alpha entry of every tile idled — a quarter of the row
  // pass's matrix instructions and operand reads — and the alpha waves, last at both barriers, carried a row
  // tile on top of their chain.)  The 16 lanes of a read group are 16 rows, SR halves apart: 16 different
  // bank slots (alpha_plane_degree(2*SR) == 1).
  static_assert(alpha_plane_degree(2*F::SR) == 1,"the row operands' reads are conflict-free");
  const int rc=wave >> 2,ot=wave & 3;
  const bool row_wave=!(BLEND || (MODE == MFMA_PLAIN3)) || (wave < 12);   // wave-uniform (three channels: twelve row tiles)
  const int row_entry=rc*F::CHR+n*F::SR+16*ot+8*kq;
  // column pass: the sixteen tiles of a block belong to the waves that neither stage nor run the alpha tiles,
  // three at a time (the operands of three tiles in flight are what the registers hold).  Where that leaves
  // tiles over (81 taps, alpha-weighted: nine staging waves, four alpha waves, three tile waves = nine tiles) they
  // go to staging waves 0..3 — the oldest wave of each SIMD, served first, done with its staging after 1750 of
  // the interval's 2800 cycles — behind their staging.
  constexpr int TILE_WAVES=16-G::FETCH_GROUPS/64-(BLEND ? 4 : 0);
  static_assert(TILE_WAVES >= 1,"somebody runs the column pass");
  constexpr int CT=3;
  constexpr int SPILLED=16 > CT*TILE_WAVES ? 16-CT*TILE_WAVES : 0;   // tiles the tile waves cannot take
  static_assert(SPILLED <= 4*CT,"four staging waves take the rest");
  const int tile_wave=wave-G::FETCH_GROUPS/64; // < 0: a staging wave; >= TILE_WAVES: an alpha wave
  int ctiles=0,ctile0=0;
  if ((tile_wave >= 0) && (tile_wave < TILE_WAVES))
    {
      if constexpr (SPILLED == 0)
        {
          ctiles=16/TILE_WAVES+(tile_wave < 16 % TILE_WAVES ? 1 : 0);
          ctile0=tile_wave*(16/TILE_WAVES)+(tile_wave < 16 % TILE_WAVES ? tile_wave : 16 % TILE_WAVES);
        }
      else
        {
          ctiles=CT;
          ctile0=CT*tile_wave;
        }
    }
  else if ((SPILLED > 0) && (wave < 4))
    {
      ctiles=SPILLED/4+(wave < SPILLED % 4 ? 1 : 0);
      ctile0=CT*TILE_WAVES+wave*(SPILLED/4)+(wave < SPILLED % 4 ? wave : SPILLED % 4);
    }
  constexpr int GROUP_STRIDE=2*F::OB;          // f16 ring: halves per 16-row group
  const int col_entry16=(n & 3)*F::CHC+(4*ctile0+(n >> 2))*8+(kq & 1)*F::OB;
  // the row epilogue's ring store: lane (n, kq) holds rows 4*kq..+3 of channel rc, column 16*ot+n
  const int ring_entry16=rc*F::CHC+(kq >> 1)*F::OB+(16*ot+n)*8+4*(kq & 1);
  // ... the alpha wave's: rows 4*kq..+3 of the alpha channel, column 16*t+n
  const int ring_alpha16=3*F::CHC+(kq >> 1)*F::OB+(16*alpha_tile+n)*8+4*(kq & 1);
  // the weights of rows 4*kq..+3 of column 16*t+n: 16 bytes, the four row quads of a column swizzled with the
  // column so that the 16 lanes of a b128 group (16 columns) fall into 16 different bank slots
  auto weight_slot=[&](int tile) { return (16*tile+n)*G::GROUP+4*(kq ^ ((n >> 2) & 3)); };
  int ring_group=0;                            // g mod NR (wave-uniform)
  unsigned recomputed=0u;

  // ---- The walk.  Iteration g:
  //   interval A: stage group g, fetch g+1 (staging waves) | column pass of block cb = g-NG-1 -> out_tile (tile
  //               waves) | alpha waves: the integer chain of group g-1 on the byte planes they read in interval B
  //               of the iteration before (matrix pipe, otherwise busy with the tile waves' column tiles only),
  //               the exact sums -> levels (+ the few recomputed) -> ring slot (alpha channel) and the colour
  //               quotients' weights (alpha_weight)
  //   interval B: f16 row chain of group g (row waves): matrix pipe  ||  colour epilogue of group g-1: f32 sums x
  //               the weights of interval A -> ring slot  ||  alpha waves: group g's byte-plane operands into
  //               registers  ||  every wave: the store of one of block cb's rows
  // What crosses a barrier: a row wave's f32 sums (4 floats); an alpha wave's plane operands (12 registers).
  // (Rounds 4-5 ran the integer chain in interval B and carried its five class tiles across barrier Y: the
  // youngest waves of their SIMDs, the alpha waves issued their 18 matrix instructions behind the 27 of the row
  // waves, and the interval lasted until they were through.)
  floatx4 sums_row={0.0f,0.0f,0.0f,0.0f};
  intx4 plane_low={0,0,0,0},plane_high={0,0,0,0};
  long plane_low2=0,plane_high2=0;
  auto store_row=[&](int block,int lane)
  {
    if ((block >= 0) && (block < nblocks))
      {
        const uint2 result=out_tile[wave*G::OUT_STRIDE+lane];
        const int x=x0+lane,y=out_begin+G::GROUP*block+wave;
        if ((x < W) && (y < H) && !MH_HKNOCKED(512))
          store_pixel16(args.dst+pixel_index(y,W,x)*PX,result);
      }
  };
  fetch(0,tid/G::GROUPS_PER_ROW,tid % G::GROUPS_PER_ROW);
  // (s_setprio 3 for the alpha waves — the youngest waves of their SIMDs, with the longest chain of interval A —
  // only changes who waits: 0.528 against 0.522 ms, profiles/r6_notes/hybrid_blur_steps.txt)
  for (int g=0; g <= ngroups+1; g++)
    {
      int opaque_tid=tid;
      asm volatile("" : "+v"(opaque_tid));       // not loop-invariant to the optimiser (see `stager`)
      const int srow=opaque_tid/G::GROUPS_PER_ROW,sxg=opaque_tid-srow*G::GROUPS_PER_ROW;
      const int cb=g-G::NG-1;                    // the column pass's block of this iteration
      const int first=ring_group;                // ring slot of group cb: (g-NG-1) mod NR = g mod NR
      const int previous=ring_group == 0 ? G::NR-1 : ring_group-1;   // ring slot of group g-1
      MH_HTRACE_MARK(0);
      if (g < ngroups)
        {
          stage_group(srow,sxg);
          MH_HTRACE_MARK(1);
          if (g+1 < ngroups)
            fetch(g+1,srow,sxg);
        }
      MH_HTRACE_MARK(2);
      // ======================================================================== interval A
      if constexpr (BLEND)
        if (alpha_wave && !MH_HKNOCKED(2))
          {
            // ---- the exact alpha sums of this wave's 16 x 16 tile of group g-1.  Sample = level*2^16: byte
            // planes 2 (low) and 3 (high); the products b_3 x d_j (class j) and b_2 x d_j (class j-1; b_2 x d_0
            // is dropped: part of the error bound), blur_exact_common.hpp
            intx4 tiles[5];
#pragma unroll
            for (int c=0; c < 5; c++)
              tiles[c]=intx4{0,0,0,0};
            if (!MH_HKNOCKED(1))
              {
                // five tiles, then four: an instruction's tile was last written five instructions
                // earlier (a dependent v_mfma waits for its predecessor's passes)
#pragma unroll
                for (int j=0; j < kExactDigits; j++)
                  tiles[j]=digit_product(plane_high,operand[j],tiles[j]);
#pragma unroll
                for (int j=1; j < kExactDigits; j++)
                  tiles[j-1]=digit_product(plane_low,operand[j],tiles[j-1]);
                if constexpr (G::NX == 2)
                  {
                    auto join=[](unsigned lo,unsigned hi) { return (long) (((unsigned long) hi << 32) | (unsigned long) lo); };
                    const long digit[kExactDigits]={
                      join((unsigned) operand[NOPERANDS-1][0],(unsigned) operand[NOPERANDS-1][1]),
                      join((unsigned) operand[NOPERANDS-1][2],(unsigned) operand[NOPERANDS-1][3]),
                      join(raw[0].x,raw[0].y),join(raw[1].x,raw[1].y),join(raw[2].x,raw[2].y)};
#pragma unroll
                    for (int j=0; j < kExactDigits; j++)
                      tiles[j]=digit_product(plane_high2,digit[j],tiles[j]);
#pragma unroll
                    for (int j=1; j < kExactDigits; j++)
                      tiles[j-1]=digit_product(plane_low2,digit[j],tiles[j-1]);
                  }
              }
            // ---- the alpha levels of group g-1 from the class tiles
            double sums_alpha[4];
            if (!MH_HKNOCKED(8))
              exact_sums(tiles,args.offset,sums_alpha);
            else
              sums_alpha[0]=sums_alpha[1]=sums_alpha[2]=sums_alpha[3]=0.0;
            unsigned q[4];
            const bool doubtful=exact_levels<false>(sums_alpha,args,q);
            const int x=x0+16*alpha_tile+n;
            {
              // the alpha levels of sample v of the windows of lane `from`'s four rows
              auto fetch_alpha=[&](int from,int v,unsigned (&level)[4])
              {
                int xx=x0+16*alpha_tile+(from & 15)-args.shift+v;
                xx=xx < 0 ? 0 : (xx > W-1 ? W-1 : xx);
#pragma unroll
                for (int r=0; r < 4; r++)
                  {
                    int yy=in0+G::GROUP*(g-1)+4*(from >> 4)+r;
                    yy=yy < 0 ? 0 : (yy > H-1 ? H-1 : yy);
                    level[r]=(unsigned) args.src[pixel_index(yy,W,xx)*4+3];
                  }
              };
              recomputed+=settle_doubtful_pixels<false,4>(doubtful && (g >= 1) && (g-1 < ngroups) && (x < W),lane,
                args.taps64,K,fetch_alpha,q);
            }
            // the levels as the column pass's alpha samples (level/2: hi + lo is exact)
            const float half_level[4]={0.5f*(float) q[0],0.5f*(float) q[1],0.5f*(float) q[2],0.5f*(float) q[3]};
            uint2 hi,lo;
            split_f16_pair(f32x2{half_level[0],half_level[1]},hi.x,lo.x);
            split_f16_pair(f32x2{half_level[2],half_level[3]},hi.y,lo.y);
            const int at=ring_alpha16+previous*GROUP_STRIDE;     // group g-1's slot
            *reinterpret_cast<uint2 *>(ring_hi+at)=hi;
            *reinterpret_cast<uint2 *>(ring_lo+at)=lo;
            // ... and the weight every colour sum of the pixel is multiplied with: the column pass's sample is
            // A*R_c*2^-17 with A = the exact level, R_c = sum(k*alpha*p)/sum(k*alpha) = 2^17*S_c/(scale*D), D the
            // REAL alpha sum in levels.  An all-transparent window: D = 0 and A = 0 -> 0*inf = NaN, which the
            // row epilogue's v_max_f32 turns into the 0 of PerceptibleReciprocal's clamp times a zero pixel sum
            // (morphology.c:2974-2977)
            floatx4 weight;
#pragma unroll
            for (int r=0; r < 4; r++)
              weight[r]=half_level[r]*args.two_over_scale*__builtin_amdgcn_rcpf((float) (sums_alpha[r]*args.alpha_scale));
            *reinterpret_cast<floatx4 *>(alpha_weight+weight_slot(alpha_tile))=weight;
          }
      MH_HTRACE_MARK(3);
      // ---- f16 column pass of block cb, whole (products, division, rounding) -> out_tile
      if ((cb >= 0) && (cb < nblocks))
        {
          int chunk_at[NC];
#pragma unroll
          for (int c=0; c < NC; c++)
            {
              const unsigned wide=(unsigned) (first+2*c+(kq >> 1));
              const unsigned group=wide < wide-(unsigned) G::NR ? wide : wide-(unsigned) G::NR;
              chunk_at[c]=col_entry16+GROUP_STRIDE*(int) group;
            }
          auto column_tiles=[&](auto count,int done)
          {
            constexpr int N=decltype(count)::value;
            floatx4 acc[N];
#pragma unroll
            for (int i=0; i < N; i++)
              acc[i]=floatx4{0.0f,0.0f,0.0f,0.0f};
#pragma unroll
            for (int c=0; c < NC; c++)
              {
                half8 a_hi[N],a_lo[N];
#pragma unroll
                for (int i=0; i < N; i++)
                  {
                    a_hi[i]=*reinterpret_cast<const half8 *>(ring_hi+chunk_at[c]+4*(done+i)*F::SC);
                    a_lo[i]=*reinterpret_cast<const half8 *>(ring_lo+chunk_at[c]+4*(done+i)*F::SC);
                  }
#pragma unroll
                for (int i=0; i < N; i++)
                  acc[i]=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[i],toeplitz_hi(c),acc[i],0,0,0);
#pragma unroll
                for (int i=0; i < N; i++)
                  acc[i]=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[i],toeplitz_hi(c),acc[i],0,0,0);
#pragma unroll
                for (int i=0; i < N; i++)
                  acc[i]=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[i],toeplitz_lo(c),acc[i],0,0,0);
              }
#pragma unroll
            for (int i=0; i < N; i++)
              out_tile[n*G::OUT_STRIDE+4*(ctile0+done+i)+kq]=MH_HKNOCKED(128) ?
                sums_to_quantum<MFMA_PLAIN4>(acc[i][0],acc[i][1],acc[i][2],acc[i][3],args.quantum_unit) :
                sums_to_quantum<SAMPLES>(acc[i][0],acc[i][1],acc[i][2],acc[i][3],args.quantum_unit);
          };
          for (int done=0; done < ctiles; done+=CT)
            {
              const int left=ctiles-done;
              if (left >= CT)
                column_tiles(std::integral_constant<int,CT>{},done);
              else if (left == 2)
                column_tiles(std::integral_constant<int,2>{},done);
              else
                column_tiles(std::integral_constant<int,1>{},done);
            }
        }
      MH_HTRACE_MARK(4);
      __syncthreads();                           // X: group g staged, the alpha of group g-1 published
      MH_HTRACE_MARK(5);
      // ======================================================================== interval B
      {
        // the weights of this lane's four pixels of group g-1, read first: the chains below hide the LDS latency
        // (at the head of the epilogue's dependent chain it cost 0.04 ms per frame)
        floatx4 weight={1.0f,1.0f,1.0f,1.0f};
        if constexpr (BLEND)
          if (row_wave && !MH_HKNOCKED(4))
            weight=*reinterpret_cast<const floatx4 *>(alpha_weight+weight_slot(ot));
        if constexpr (BLEND)
          if (alpha_wave)
            {
              // group g's byte-plane operands, for the chain of the next interval A (the planes are rewritten there)
              plane_low=*reinterpret_cast<const intx4 *>(alpha_plane+alpha_entry);
              plane_high=*reinterpret_cast<const intx4 *>(alpha_plane+G::GROUP*G::ASTRIDE+alpha_entry);
              if constexpr (G::NX == 2)
                {
                  const int at=alpha_entry+64-8*kq;          // columns 64+8*kq .. +7
                  plane_low2=*reinterpret_cast<const long *>(alpha_plane+at);
                  plane_high2=*reinterpret_cast<const long *>(alpha_plane+G::GROUP*G::ASTRIDE+at);
                }
            }
        MH_HTRACE_MARK(6);
        // ---- f16 row chain of group g (one channel, 16 rows x 16 columns) and, beside it, the colour epilogue
        // of group g-1: the column pass's samples, UNROUNDED:
        //   alpha-weighted: A*R_c*2^-17 = S_c*weight (the alpha waves' weight, interval A)
        //   plain:          level/2 = S/scale
        if (row_wave)
          {
            floatx4 acc={0.0f,0.0f,0.0f,0.0f};
#pragma unroll
            for (int c=0; c < NC; c++)
              {
                const half8 a_hi=*reinterpret_cast<const half8 *>(stage_hi+row_entry+32*c);
                const half8 a_lo=*reinterpret_cast<const half8 *>(stage_lo+row_entry+32*c);
                acc=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi,toeplitz_hi(c),acc,0,0,0);
                acc=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo,toeplitz_hi(c),acc,0,0,0);
                acc=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi,toeplitz_lo(c),acc,0,0,0);
              }
            float v[4];
            if (BLEND && !MH_HKNOCKED(64))
              {
#pragma unroll
                for (int r=0; r < 4; r++)
                  v[r]=__builtin_fmaxf(sums_row[r]*weight[r],0.0f);
              }
            else
              {
                const float back=0.5f*args.two_over_scale;
#pragma unroll
                for (int r=0; r < 4; r++)
                  v[r]=sums_row[r]*back;
              }
            uint2 hi,lo;
            split_f16_pair(f32x2{v[0],v[1]},hi.x,lo.x);
            split_f16_pair(f32x2{v[2],v[3]},hi.y,lo.y);
            const int at=ring_entry16+previous*GROUP_STRIDE;
            *reinterpret_cast<uint2 *>(ring_hi+at)=hi;
            *reinterpret_cast<uint2 *>(ring_lo+at)=lo;
            // (the wait states between the chain and the first vector read of its tile, whatever the
            // block layout: hipcc pads them per basic block — see settle_tiles)
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc));
            sums_row=acc;
          }
        // the store of the column pass's rows
        store_row(cb,opaque_tid & 63);
      }
      MH_HTRACE_MARK(7);
      __syncthreads();                           // Y: ring group g-1 complete, out_tile read, staging reads done
      MH_HTRACE_MARK(8);
      ring_group=ring_group+1 == G::NR ? 0 : ring_group+1;
    }
  if ((args.recomputed != nullptr) && (recomputed != 0u) && (lane == 0))     // a wave-uniform count
    atomicAdd(args.recomputed,(unsigned long long) recomputed);
}

template<int NC,int MODE>
static MhStatus launch_hybrid_typed(const View &src,BlurExactArgs &args)
{
  constexpr bool BLEND=MODE == MFMA_BLEND4;
  typedef HybridGeometry<NC,BLEND> G;
  args.strips=(args.columns+G::COLS-1)/G::COLS;
  args.blocks=(args.rows+G::GROUP-1)/G::GROUP;
  // Cut the strips so that every CU gets a work item.  A segment recomputes NG-1 ring groups
  // (the K-1 halo rows of its first block), so it stays at least 16 blocks long.
  const int cus=compute_units(src.device);
  const int max_segments=args.blocks/16 > 1 ? args.blocks/16 : 1;
  int segments=(cus+args.strips-1)/args.strips;
  segments=segments < 1 ? 1 : (segments > max_segments ? max_segments : segments);
  const int forced_segments=(int) option_long("MAGICKHIP_FUSED_SEGMENTS",0);
  if (forced_segments > 0)
    segments=forced_segments > args.blocks ? args.blocks : forced_segments;
  args.blocks_per_segment=(args.blocks+segments-1)/segments;
  args.segments=(args.blocks+args.blocks_per_segment-1)/args.blocks_per_segment;   // no empty segment
  const int items=args.strips*args.segments;
  args.items_per_xcd=(items+7)/8;
  const size_t lds=G::lds_bytes;
  static bool attribute_set[64]={};              // once per kernel and device
  const int slot=src.device >= 0 && src.device < 64 ? src.device : 0;
  if (!attribute_set[slot])
    {
      MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&blur_fused_hybrid_kernel<NC,MODE>),
        hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
      attribute_set[slot]=true;
    }
#ifdef MH_HYBRID_TRACE
  const char *trace_path=option("MAGICKHIP_HYBRID_TRACE");
  const size_t trace_bytes=4u*16u*48u*12u*sizeof(unsigned long long);
  if (trace_path != nullptr)
    {
      MH_HIP(hipMalloc(reinterpret_cast<void **>(&args.trace),trace_bytes));
      MH_HIP(hipMemsetAsync(args.trace,0,trace_bytes,src.stream));
    }
#endif
  {
    ProfileScope prof("blur_fused_hybrid",src.stream);
    hipLaunchKernelGGL((blur_fused_hybrid_kernel<NC,MODE>),dim3((unsigned) (8*args.items_per_xcd)),dim3(1024),lds,
      src.stream,args);
    MH_HIP(hipGetLastError());
  }
#ifdef MH_HYBRID_TRACE
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
      args.trace=nullptr;
    }
#endif
  return MH_OK;
}

template<int NC>
static MhStatus launch_hybrid_modes(const View &src,BlurExactArgs &args,bool blend)
{
  if (src.channels == 3)
    return launch_hybrid_typed<NC,MFMA_PLAIN3>(src,args);
  return blend ? launch_hybrid_typed<NC,MFMA_BLEND4>(src,args) : launch_hybrid_typed<NC,MFMA_PLAIN4>(src,args);
}

// taps: host doubles in the reversed walk of morphology.c:2746 (taps[v] multiplies the input at
// o-shift+v), all positive.  *handled = false: the shape or the taps are outside the kernel's
// reach, nothing was launched.
MhStatus launch_blur_fused_hybrid(const View &src,const View &dst,const double *taps,int ntaps,int shift,
  bool blend,bool *handled)
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
    return MH_OK;                                // three 32-row ring chunks hold 94 band slots
  // the digits and the certificate of the alpha sums (plain frames use the f16 taps only, but the
  // plan's conditions — positive, finite taps — are what the f16 error argument assumes too)
  const ExactTapPlan plan=plan_exact_taps(taps,ntaps);
  if (!plan.ok)
    return MH_OK;
  // ... and every tap's f16 terms must carry the precision the +-1 argument counts on (f16_taps_resolved)
  if (!f16_taps_resolved(taps,ntaps))
    return MH_OK;
  ExactDeviceTaps device;
  MH_TRY(upload_exact_taps(src,taps,ntaps,plan,&device));
  BlurExactArgs args;
  args.tap_scale=f16_tap_scale(taps,ntaps);
  args.two_over_scale=2.0f/args.tap_scale;
  args.quantum_unit=(float) (2.0/((double) args.tap_scale*65535.0));
  args.src=static_cast<const uint16_t *>(src.pixels);
  args.dst=static_cast<uint16_t *>(dst.pixels);
  args.columns=(int) src.columns;
  args.rows=(int) src.rows;
  args.ntaps=ntaps;
  args.shift=shift;
  args.taps64=device.taps64;
  args.taps=device.taps;
  args.digits=device.digits;
  args.offset=plan.offset_plain;                 // the alpha tiles: samples level*2^16
  args.alpha_scale=plan.alpha_scale;
  args.colour_window=plan.colour_window;
  args.alpha_half_window=0.5-plan.alpha_window_plain;
  args.alpha_floor=plan.alpha_floor;
  args.gain=0.0f;
  args.threshold=(int) option_long("MAGICKHIP_HYBRID_KNOCK",0);       // diagnostic builds only (MH_HKNOCKED)
  args.recomputed=exact_recomputed_counter(src.device);
  args.give_up=nullptr;
  args.trace=nullptr;
  *handled=true;
  const int nc=(ntaps+15+31)/32;                 // 16 outputs + K-1 halo, in 32-sample chunks
  if (nc == 1)
    return launch_hybrid_modes<1>(src,args,blend);
  if (nc == 2)
    return launch_hybrid_modes<2>(src,args,blend);
  if (nc == 3)
    return launch_hybrid_modes<3>(src,args,blend);
  *handled=false;
  return MH_OK;
}

} // namespace mh
