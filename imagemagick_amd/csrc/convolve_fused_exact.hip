// BlurImage's two passes in ONE launch with an EXACT-INTEGER row pass on the matrix cores
// (Q16; RGBA with alpha-weighted colour, four plain channels, RGB).
// MagickCore/effect.c:765-796 -> morphology.c:2811-2979 (row kernel) -> :2654-2807 (column
// kernel), the Quantum-rounded intermediate of :4012-4022.
//
// The reference forms every sum in fp64 (pixel += alpha*k*p, gamma += alpha*k, one rounding per
// operation) and rounds gamma*pixel to a Quantum level.  The level only depends on the REAL value
// of sum(k*alpha*p)/sum(k*alpha) unless that value lies within the reference's own rounding error
// (~1e-9 level) of a rounding tie.  This kernel therefore computes the real value to a known
// error bound with integer arithmetic that cannot round — v_mfma_i32_16x16x64_i8 — and recomputes,
// in the reference's operation order, the few results that bound cannot decide:
//
//   samples  x = alpha*p (an exact integer below 2^32) or alpha*2^16 / p*2^16: four bytes
//            b_0..b_3, stored as b^0x80 = the signed byte b-128 (the offset 128*sum(digit) is a
//            constant per launch: every output's window holds all K taps);
//   taps     q = rint(k*2^F) as five balanced signed 8-bit digits d_0..d_4 (host side; F = 43 for
//            the 79-tap sigma = 10 Gaussian), |k - q*2^-F| <= 2^-(F+1);
//   products b_i x d_j of equal weight 2^(8(i+j)) accumulate in the same i32 tile; the classes
//            i+j <= 2 are dropped (their sum is below 0.06 of 2^32: part of the error bound), so a
//            64-slot chunk of the band costs 14 instructions (9 without alpha weighting) and five
//            accumulator tiles, combined exactly in fp64 in the epilogue (|M| < 2^53);
//   bound    |N - N~| <= sum|k - q 2^-F| * 65535^2 + (dropped classes), the same for the alpha
//            sum: the level of 65536*N~/D~ is certain unless its fraction lies within
//            w = 65536*(E_N+E_D)/D~ + 4e-9 of a tie (w = 2e-6 for an opaque window, growing with
//            1/alpha) — about seven samples per million on random RGBA.  Those are recomputed by
//            the lane itself with conv1d_reference_sample (row pass: from the source frame) or
//            its twin over the ring (column pass: from the exact intermediate).
//
// Both passes are exact: the result is BIT-IDENTICAL to the reference's BlurImage / UnsharpMaskImage
// (MH_PRECISION_EXACT; UnsharpMaskImage in both modes).  MH_PRECISION_FAST BlurImage is
// convolve_fused_hybrid.hip, which shares the alpha sums' arithmetic and certificate with this kernel.
// (Rounds 3-5 also kept a variant with this row pass and an f16 column pass; nothing called it after the
// hybrid kernel: removed in round 6.)
//
// The walk (strips of 64 columns, groups of 16 rows, a ring of the intermediate in LDS, two barrier
// intervals an iteration, XCD-aware item order) is described at the loop.
#include "blur_exact_common.hpp"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <type_traits>
#include <vector>

namespace mh {

// worst number of ds_read_b128 lines of a lane group that share a 16-byte slot, for the row
// pass's byte-plane operand: entry e = lane&15 -> channel e&3, row e>>2; k quarter = lane>>4
// reads the next 16 bytes of the line
static constexpr int exact_stage_degree(int SRX,int PAD)
{
  const int CH=16*SRX+PAD;
  int worst=1;
  for (int g=0; g < 4; g++)
    {
      int count[16]={0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
      for (int i=0; i < 16; i++)
        {
          const int base=(g & 1) == 0 ? (i < 4 ? i : (i < 8 ? i+8 : i+12)) : (i < 8 ? i+4 : (i < 12 ? i+8 : i+16));
          const int lane=base+32*(g >> 1);
          const int e=lane & 15,kq=lane >> 4;
          const int bytes=(e & 3)*CH+(e >> 2)*SRX+16*kq;
          const int slot=(bytes % 256)/16;
          count[slot]++;
          worst=count[slot] > worst ? count[slot] : worst;
        }
    }
  return worst;
}

static constexpr int exact_stage_pad(int SRX,int max_pad)
{
  int best=0,best_degree=99;
  for (int PAD=0; PAD <= max_pad; PAD+=16)
    {
      const int degree=exact_stage_degree(SRX,PAD);
      if (degree < best_degree)
        {
          best_degree=degree;
          best=PAD;
        }
    }
  return best;
}

template<int NC>
struct ExactGeometry
{
  typedef Fused16Geometry<NC> F;
  static constexpr int COLS=64,GROUP=16;
  static constexpr int NG=2*NC,NR=NG+1;
  static constexpr int NX=(NC+1)/2;            // 64-slot chunks of the band (16 outputs + K-1)
  static constexpr int XS=F::XS;               // staged columns = bytes per line of a plane
  static constexpr int SRX=XS;
  static constexpr int STAGE_PAD=exact_stage_pad(SRX,160);
  static constexpr int CHS=GROUP*SRX+STAGE_PAD;          // bytes per channel of a stage plane
  static constexpr int STAGE_PLANE=4*CHS;
  static constexpr int stage_bytes=4*STAGE_PLANE;
  // exact ring: byte plane [channel][ring group][column][16 rows]; a column-pass lane's 16 rows
  // are one 16-byte unit, and with a channel stride of 4 units mod 16 the 16 lanes of a
  // ds_read_b128 group (4 columns x 4 channels) land in 16 different slots
  static constexpr int CHU=(NR*COLS+4)*16;               // bytes per channel of a ring plane
  static constexpr int RINGX_PLANE=4*CHU;
  static constexpr int ring_bytes=4*RINGX_PLANE;
  static constexpr int OUT_STRIDE=COLS+1;
  static constexpr size_t lds_bytes=(size_t) ring_bytes+stage_bytes+(size_t) GROUP*OUT_STRIDE*sizeof(uint2);
  static_assert(lds_bytes <= 163840,"more than the 160 KiB of a CU");
  static constexpr int GROUPS_PER_ROW=XS/4;
  static constexpr int FETCH_GROUPS=GROUP*GROUPS_PER_ROW;
  static_assert(FETCH_GROUPS <= 1024,"one staging round");
  static_assert((FETCH_GROUPS % 64) == 0,"whole staging waves");
  static_assert((SRX % 16) == 0,"16-byte operand reads");
};

template<int NC,int MODE,bool UNSHARP>
__global__ __launch_bounds__(1024)
void blur_fused_exact_kernel(BlurExactArgs args)
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
    else if constexpr (UNSHARP)
      {
        // UnsharpMask reads every source pixel twice — for the window, and 94 rows later for the
        // epilogue: results that stream THROUGH the L2 push those lines out before their second use
        // (PMC: 1.34x the compulsory traffic on 16384^2).  Non-temporal stores: 4.82 -> 4.46 ms.
        typedef unsigned NativePair __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(NativePair{value.x,value.y},reinterpret_cast<NativePair *>(at));
      }
    else
      *reinterpret_cast<uint2 *>(at)=value;
  };
  typedef ExactGeometry<NC> G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char *ring=smem_raw;                          // byte planes
  unsigned char *stage=smem_raw+G::ring_bytes;
  uint2 *out_tile=reinterpret_cast<uint2 *>(stage+G::stage_bytes);
  const int tid=(int) threadIdx.x,lane=tid & 63;
  const int wave=__builtin_amdgcn_readfirstlane(tid >> 6);
  const int n=lane & 15,kq=lane >> 4;
  const int K=args.ntaps;
  const int W=args.columns,H=args.rows;
#ifdef MH_EXACT_PRIO
  // the four waves of a SIMD (w, w+4, w+8, w+12) take the matrix pipe in a fixed order, so that
  // one wave's epilogue runs beside the next one's products instead of all four doing the same
  switch (wave >> 2)
    {
      case 0: __builtin_amdgcn_s_setprio(3); break;
      case 1: __builtin_amdgcn_s_setprio(2); break;
      case 2: __builtin_amdgcn_s_setprio(1); break;
      default: __builtin_amdgcn_s_setprio(0); break;
    }
#endif

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
#ifdef MH_EXACT_TRACE
  const bool traced=(args.trace != nullptr) && (blockIdx.x < 4) && ((wave & 3) == 0);
#endif

  // ---- Toeplitz operands.  Digit j of chunk c for output n, k quarter kq: bytes b = 0..15 hold
  // d_j[64c+16kq+b-n] (0 outside the kernel).  Whatever order the instruction gives the 64 slots
  // of a chunk, it is the same for both operands: slot (kq, b) of the samples is position
  // 64c+16kq+b.
  intx4 t0[kExactDigits];                       // slots 0..63: 16 per k quarter
  SecondOperand t1[kExactDigits];               // slots from 64: 8 or 16 per k quarter (NX = 2 only)
  {
    constexpr int DL=176;                        // digit line: tap v at [16+v], zeros around
    signed char *digit_lds=reinterpret_cast<signed char *>(stage);
    for (int at=tid; at < kExactDigits*DL; at+=1024)
      {
        const int j=at/DL,v=at-j*DL-16;
        digit_lds[at]=((v >= 0) && (v < K)) ? args.digits[j*kExactDigitPitch+v] : (signed char) 0;
      }
    __syncthreads();
    auto packed=[&](const signed char *from) -> unsigned
    {
      return (unsigned) (unsigned char) from[0] | ((unsigned) (unsigned char) from[1] << 8) |
        ((unsigned) (unsigned char) from[2] << 16) | ((unsigned) (unsigned char) from[3] << 24);
    };
#pragma unroll
    for (int j=0; j < kExactDigits; j++)
      {
        const signed char *from=digit_lds+j*DL+16+16*kq-n;
#pragma unroll
        for (int w=0; w < 4; w++)
          t0[j][w]=(int) packed(from+4*w);
        const signed char *next=digit_lds+j*DL+16+64+kSecondBytes*kq-n;
#ifndef MH_EXACT_K64
        t1[j]=G::NX == 2 ? (long) (((unsigned long) packed(next+4) << 32) | (unsigned long) packed(next)) : 0l;
#else
#pragma unroll
        for (int w=0; w < 4; w++)
          t1[j][w]=G::NX == 2 ? (int) packed(next+4*w) : 0;
#endif
      }
    __syncthreads();                             // digit_lds is the staging plane
    // every byte plane starts as 0x80 = the signed sample byte of zero: the planes 0 and 1 of an
    // alpha / plain channel (sample = level * 2^16) are never written again
    {
      const int words=(int) (G::ring_bytes+G::stage_bytes)/4;
      unsigned *fill=reinterpret_cast<unsigned *>(ring);
      for (int at=tid; at < words; at+=1024)
        fill[at]=0x80808080u;
    }
    __syncthreads();
  }

  // ---- staging: thread -> (row, 4 consecutive columns) of the 16 x XS source window
  const bool stager=tid < G::FETCH_GROUPS;     // wave-uniform (FETCH_GROUPS is a multiple of 64)
  // (the walk recomputes a stager's row and column group from its thread index in every iteration — a handful of
  // vector instructions — instead of keeping them and what is derived from them in registers: the kernel sits at
  // the 128 registers of a wave, and what the compiler spills it reloads from scratch INSIDE the walk, in front of
  // the fetch, behind an s_waitcnt vmcnt(0): the alpha-weighted 79-tap kernel did, until round 6)
  uint2 raw[4];
  auto fetch=[&](int g,int srow,int sxg)
  {
    if (stager)
      {
        int y=in0+G::GROUP*g+srow;
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
  // raw Quantum pixels -> signed byte planes.  BLEND: colour sample alpha*p (four bytes), alpha
  // sample alpha*2^16 (planes 2, 3); plain: p*2^16 for every channel.
  auto stage_group=[&](int srow,int sxg)
  {
    if (stager)
      {
        unsigned char *line=stage+srow*G::SRX+4*sxg;
        auto put=[&](int plane,int channel,unsigned bytes)
        {
          *reinterpret_cast<unsigned *>(line+plane*G::STAGE_PLANE+channel*G::CHS)=bytes ^ 0x80808080u;
        };
        // bytes 0,1 / 2,3 of the two words of each pixel, paired over the four positions
        const unsigned lx01=__builtin_amdgcn_perm(raw[1].x,raw[0].x,0x05010400u),lx23=__builtin_amdgcn_perm(raw[3].x,raw[2].x,0x05010400u);
        const unsigned hx01=__builtin_amdgcn_perm(raw[1].x,raw[0].x,0x07030602u),hx23=__builtin_amdgcn_perm(raw[3].x,raw[2].x,0x07030602u);
        const unsigned ly01=__builtin_amdgcn_perm(raw[1].y,raw[0].y,0x05010400u),ly23=__builtin_amdgcn_perm(raw[3].y,raw[2].y,0x05010400u);
        const unsigned hy01=__builtin_amdgcn_perm(raw[1].y,raw[0].y,0x07030602u),hy23=__builtin_amdgcn_perm(raw[3].y,raw[2].y,0x07030602u);
        if constexpr (BLEND)
          {
            unsigned x[3][4];
#pragma unroll
            for (int i=0; i < 4; i++)
              {
                const unsigned alpha=raw[i].y >> 16;
                x[0][i]=__umul24(raw[i].x & 0xffffu,alpha);
                x[1][i]=__umul24(raw[i].x >> 16,alpha);
                x[2][i]=__umul24(raw[i].y & 0xffffu,alpha);
              }
#pragma unroll
            for (int c=0; c < 3; c++)
              {
                unsigned p[4];
                byte_planes(x[c],p);
#pragma unroll
                for (int i=0; i < 4; i++)
                  put(i,c,p[i]);
              }
            put(2,3,__builtin_amdgcn_perm(hy23,hy01,0x05040100u));
            put(3,3,__builtin_amdgcn_perm(hy23,hy01,0x07060302u));
          }
        else
          {
            put(2,0,__builtin_amdgcn_perm(lx23,lx01,0x05040100u));
            put(3,0,__builtin_amdgcn_perm(lx23,lx01,0x07060302u));
            put(2,1,__builtin_amdgcn_perm(hx23,hx01,0x05040100u));
            put(3,1,__builtin_amdgcn_perm(hx23,hx01,0x07060302u));
            put(2,2,__builtin_amdgcn_perm(ly23,ly01,0x05040100u));
            put(3,2,__builtin_amdgcn_perm(ly23,ly01,0x07060302u));
            put(2,3,__builtin_amdgcn_perm(hy23,hy01,0x05040100u));
            put(3,3,__builtin_amdgcn_perm(hy23,hy01,0x07060302u));
          }
      }
  };

  // row pass: wave = row quad (4 rows) x output tile (16 columns); entry e = 4*row + channel, so
  // D hands a lane the four channels of ONE pixel (lane-local division and rounding)
  const int rq=wave & 3,ot=wave >> 2;
  const int row_entry=(n & 3)*G::CHS+(4*rq+(n >> 2))*G::SRX+16*ot+16*kq;
  // column pass: entry e = 4*column + channel, 16 outputs along y; every wave owns one tile (four columns); a
  // 64-row chunk spans four ring groups, one per k quarter
  const int col_entryx=(n & 3)*G::CHU+(4*wave+(n >> 2))*16;                       // + group*1024
  const int ring_entryx=kq*G::CHU+(16*ot+n)*16+4*rq;                              // + group*1024
  int ring_group=0;                            // g mod NR (wave-uniform)
  // Who stores the rows of a finished block (interval A of the next iteration), and when.  The staging waves
  // begin interval A with s_waitcnt vmcnt(0) for the pixels they fetched an iteration ago, and that counter also
  // counts stores and younger loads: what is issued in front of it is waited for in full.  STORE_MODE 0: every wave
  // stores row `wave` first thing (rounds 3-5); 1: the same, behind the staging; 2: the waves that do not stage
  // store all sixteen rows.  One box, 8192^2, 79 taps, 40 launches each, modes 0 / 1 / 2
  // (profiles/r6_notes/exact_store_modes.txt): RGBA 0.899 / 0.881 / 0.895 ms, UnsharpMask 0.994 / 1.008 / 1.205,
  // four plain channels 0.710 / 0.696 / 0.689, RGB 0.783 / 0.773 / 0.778 — two or three per cent either way: the
  // waves that do not stage are the youngest of their SIMDs and the last through both intervals, so handing them three
  // rows each only pays where staging is short (plain frames), and UnsharpMask's second read of the source pixel
  // prefers to be issued early.
#ifndef MH_EXACT_STORE_MODE
#define MH_EXACT_STORE_MODE (-1)
#endif
  constexpr int STORE_MODE=MH_EXACT_STORE_MODE >= 0 ? MH_EXACT_STORE_MODE :
    (UNSHARP ? 0 : (BLEND ? 1 : (MODE == MFMA_PLAIN3 ? 1 : 2)));
  constexpr int STAGE_WAVES=G::FETCH_GROUPS/64;
  constexpr bool SPREAD_STORES=(STORE_MODE == 2) && (STAGE_WAVES < 16);
  constexpr int STORE_WAVES=SPREAD_STORES ? 16-STAGE_WAVES : 16;
  constexpr int STORE_ROWS=(16+STORE_WAVES-1)/STORE_WAVES;            // rows of a block per storing wave
  const int store_wave=SPREAD_STORES ? wave-STAGE_WAVES : wave;       // < 0: stores nothing
  uint2 original[STORE_ROWS];
#pragma unroll
  for (int r=0; r < STORE_ROWS; r++)
    original[r]=make_uint2(0u,0u);
  auto fetch_original=[&](int block)
  {
    if ((block >= 0) && (block < nblocks) && (store_wave >= 0))
      {
#pragma unroll
        for (int r=0; r < STORE_ROWS; r++)
          {
            const int row=store_wave+r*STORE_WAVES;
            const int x=x0+lane,y=out_begin+G::GROUP*block+row;
            if ((row < G::GROUP) && (x < W) && (y < H))
              original[r]=load_pixel16(args.src+pixel_index(y,W,x)*PX);
          }
      }
  };
  unsigned recomputed=0u;

  // ---- The walk, software-pipelined over the two barrier intervals of an iteration so that every
  // interval pairs one matrix chain with the INDEPENDENT epilogue of the other pass (the chain of
  // a wave and the epilogue that consumes it never share an interval: four waves of a SIMD that
  // all run chain -> epilogue in lockstep leave the matrix pipe idle during the epilogues and the
  // vector pipe idle during the chains — measured 41 % matrix-pipe time, 40 % of the wave cycles
  // parked).  Iteration g:
  //   interval A: stage group g, fetch g+1 | store the rows of block cb-1 | column chain of block
  //               cb = g-NG-1 (ring groups cb..cb+NG-1, all written before) -> sums
  //               || row epilogue of group g-1 (sums -> levels -> ring slot)
  //   interval B: row chain of group g -> sums  || column epilogue of block cb -> out_tile
  // Sums (4 doubles per pass) are what crosses a barrier.  The ring holds NR = NG+1 groups:
  // g-NG-1 .. g-1.  Out-of-range iterations (pipeline fill and drain) run on whatever the planes
  // hold and their results are not stored.
  double sums_row[4]={0.0,0.0,0.0,0.0},sums_col[4]={0.0,0.0,0.0,0.0};
  auto init_tiles=[&](intx4 (&acc)[5])
  {
#pragma unroll
    for (int c=0; c < 5; c++)
      acc[c]=intx4{0,0,0,0};
  };
  auto store_row=[&](int block)
  {
    if ((block >= 0) && (block < nblocks) && (store_wave >= 0))
      {
        // (addresses derived from an opaque copy of the lane: loop-invariant ones are hoisted into registers the
        // kernel does not have, spilled, and reloaded in the walk — see `stager`)
        int lane=tid & 63;
        asm volatile("" : "+v"(lane));
#pragma unroll
        for (int r=0; r < STORE_ROWS; r++)
          {
            const int row=store_wave+r*STORE_WAVES;
            if (row < G::GROUP)                  // wave-uniform
              {
                uint2 result=out_tile[row*G::OUT_STRIDE+lane];
                if constexpr (UNSHARP)
                  result=unsharp_pixel(original[r],result,args.gain,args.threshold);
                const int x=x0+lane,y=out_begin+G::GROUP*block+row;
                if ((x < W) && (y < H))
                  store_pixel16(args.dst+pixel_index(y,W,x)*PX,result);
              }
          }
      }
  };
  fetch(0,tid/G::GROUPS_PER_ROW,tid % G::GROUPS_PER_ROW);
  __shared__ unsigned stop_word;
  if (tid == 0)
    stop_word=0u;
  // the lane that looks at the give_up word: on a wave that does not stage (see MH_EXACT_STORE_MODE)
  constexpr int WORD_LANE=SPREAD_STORES ? 1023 : 0;
  for (int g=0; g <= ngroups+1; g++)
    {
      const int cb=g-G::NG-1;                    // the column pass's block of this iteration
      const int first=ring_group;                // ring slot of group cb: (g-NG-1) mod NR = g mod NR
      const int previous=ring_group == 0 ? G::NR-1 : ring_group-1;   // ring slot of group g-1
      unsigned seen_word=0u;
      auto issue_rest=[&]()
      {
        if ((args.give_up != nullptr) && (tid == WORD_LANE))
          seen_word=__hip_atomic_load(args.give_up,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
        // the rows of block cb-1 (out_tile was written in the previous interval B)
        store_row(cb-1);
        if constexpr (UNSHARP)
          fetch_original(cb);
      };
      if constexpr (STORE_MODE == 0)
        issue_rest();
      MH_XTRACE_MARK(0);
      // (STORE_MODE != 0: the staging, which waits for the pixels fetched an iteration ago, ahead of the store, the
      // give_up word and UnsharpMask's source pixel)
      if ((g < ngroups) && stager)
        {
          int opaque_tid=tid;
          asm volatile("" : "+v"(opaque_tid));   // not loop-invariant to the optimiser (see `stager`)
          const int srow=opaque_tid/G::GROUPS_PER_ROW,sxg=opaque_tid-srow*G::GROUPS_PER_ROW;
#ifdef MH_EXACT_TRACE
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          MH_XTRACE_MARK(1);
#endif
          stage_group(srow,sxg);
#ifdef MH_EXACT_TRACE
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          MH_XTRACE_MARK(2);
#endif
          if (g+1 < ngroups)
            fetch(g+1,srow,sxg);
        }
      // BlurExactArgs::give_up: one lane looks at the word, the whole workgroup sees its copy behind
      // barrier X and leaves behind barrier Y
      if constexpr (STORE_MODE != 0)
        issue_rest();
      MH_XTRACE_MARK(3);
      // ======================================================================== interval A
      {
        intx4 acc[5];
        init_tiles(acc);
        {
          // ring group of this lane's 16 rows of the first chunk: (first + kq) mod NR
          unsigned group=(unsigned) (first+kq);
          group=group >= (unsigned) G::NR ? group-(unsigned) G::NR : group;
          const unsigned char *from=ring+col_entryx+(int) group*(G::COLS*16);
          intx4 a[4];
#pragma unroll
          for (int i=BLEND ? 0 : 2; i < 4; i++)
            a[i]=*reinterpret_cast<const intx4 *>(from+i*G::RINGX_PLANE);
          exact_products<!BLEND,false>(a,t0,acc);
        }
        if constexpr (G::NX == 2)
          {
            // second chunk
#ifndef MH_EXACT_K64
            // rows 64..95: 8 rows per lane, groups first+4 and first+5
            unsigned group=(unsigned) (first+4+(kq >> 1));
            group=group >= (unsigned) G::NR ? group-(unsigned) G::NR : group;
            const unsigned char *from=ring+col_entryx+(int) group*(G::COLS*16)+8*(kq & 1);
#else
            // rows 64..127: group (first + 4 + kq) mod NR.  Beyond the NG groups of the band the
            // digits are zero: whatever the slot holds is multiplied by 0
            unsigned group=(unsigned) (first+4+kq);
            group=group >= (unsigned) G::NR ? group-(unsigned) G::NR : group;
            group=group >= (unsigned) G::NR ? group-(unsigned) G::NR : group;
            const unsigned char *from=ring+col_entryx+(int) group*(G::COLS*16);
#endif
            SecondOperand a[4];
#pragma unroll
            for (int i=BLEND ? 0 : 2; i < 4; i++)
              a[i]=*reinterpret_cast<const SecondOperand *>(from+i*G::RINGX_PLANE);
            exact_products<!BLEND,true>(a,t1,acc);
          }
        // ---- row epilogue of group g-1 (independent of the chain above)
        {
          unsigned q[4];
          const bool doubtful=exact_levels<BLEND>(sums_row,args,q);
          exact_sums(acc,args.offset,sums_col);
          const int x=x0+16*ot+n;
          {
            // source pixel (y clamped like the intermediate's rows) of lane `from`, sample v
            auto fetch=[&](int from,int v,unsigned (&level)[4])
            {
              int yy=in0+G::GROUP*(g-1)+4*rq+(from >> 4);
              yy=yy < 0 ? 0 : (yy > H-1 ? H-1 : yy);
              int xx=x0+16*ot+(from & 15)-args.shift+v;
              xx=xx < 0 ? 0 : (xx > W-1 ? W-1 : xx);
              const uint2 pixel=load_pixel16(args.src+pixel_index(yy,W,xx)*PX);
              level[0]=pixel.x & 0xffffu; level[1]=pixel.x >> 16; level[2]=pixel.y & 0xffffu;
              level[3]=PX == 4 ? pixel.y >> 16 : 0u;
            };
            recomputed+=settle_doubtful_pixels<BLEND,PX>(doubtful && (g >= 1) && (g-1 < ngroups) && (x < W),lane,
              args.taps64,K,fetch,q);
          }
          // the column pass's samples of this pixel, as signed bytes
          unsigned v[4];
          if constexpr (BLEND)
            {
              v[0]=__umul24(q[0],q[3]);
              v[1]=__umul24(q[1],q[3]);
              v[2]=__umul24(q[2],q[3]);
              v[3]=q[3] << 16;
            }
          else
            {
              v[0]=q[0] << 16; v[1]=q[1] << 16; v[2]=q[2] << 16; v[3]=q[3] << 16;
            }
          // 4x4 transpose between the registers (channels) and the four 16-lane rows (pixel rows)
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"
                       "s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3"
                       : "+v"(v[0]),"+v"(v[1]),"+v"(v[2]),"+v"(v[3]));
          // lane (n, kq): channel kq of rows 4*rq+0..3 -> one dword per byte plane
          unsigned p[4];
          byte_planes(v,p);
          unsigned char *to=ring+ring_entryx+previous*(G::COLS*16);
#pragma unroll
          for (int i=0; i < 4; i++)
            *reinterpret_cast<unsigned *>(to+i*G::RINGX_PLANE)=p[i] ^ 0x80808080u;
        }
      }
#ifdef MH_EXACT_TRACE
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      MH_XTRACE_MARK(4);
      if ((args.give_up != nullptr) && (tid == WORD_LANE))
        stop_word=seen_word;
      __syncthreads();                           // X: group g staged, ring group g-1 complete
      MH_XTRACE_MARK(5);
      // ======================================================================== interval B
      {
        // ---- row chain of group g
        intx4 acc[5];
        init_tiles(acc);
        {
          intx4 a[4];
#pragma unroll
          for (int i=BLEND ? 0 : 2; i < 4; i++)
            a[i]=*reinterpret_cast<const intx4 *>(stage+i*G::STAGE_PLANE+row_entry);
          exact_products<!BLEND,false>(a,t0,acc);
        }
        if constexpr (G::NX == 2)
          {
            SecondOperand a[4];
#pragma unroll
            for (int i=BLEND ? 0 : 2; i < 4; i++)
              a[i]=*reinterpret_cast<const SecondOperand *>(stage+i*G::STAGE_PLANE+row_entry+64-(16-kSecondBytes)*kq);
            exact_products<!BLEND,true>(a,t1,acc);
          }
        {
          // ---- column epilogue of block cb (independent of the chain above)
          // lane (n, kq): the four channels (registers) of pixel (column 4*wave+kq, row n)
          unsigned q[4];
          const bool doubtful=exact_levels<BLEND>(sums_col,args,q);
          exact_sums(acc,args.offset,sums_row);
          const int x=x0+4*wave+kq,y=out_begin+G::GROUP*cb+n;
          {
            // the exact intermediate out of the ring: pixel column 4*wave + (from>>4), output row
            // from&15 of the block -> ring row (from&15)+v.  Samples alpha*p (colour; p is the
            // exact quotient) and alpha*2^16 / p*2^16.
            auto fetch=[&](int from,int v,unsigned (&level)[4])
            {
              const int row=(from & 15)+v;
              int group=first+(row >> 4);
              group=group >= G::NR ? group-G::NR : group;
              group=group >= G::NR ? group-G::NR : group;
              const unsigned char *at=ring+(group*G::COLS+4*wave+(from >> 4))*16+(row & 15);
              unsigned sample[4];
#pragma unroll
              for (int c=0; c < 4; c++)
                {
                  const unsigned char *p=at+c*G::CHU;
                  sample[c]=(((BLEND && (c != 3)) ? ((unsigned) p[0] | ((unsigned) p[G::RINGX_PLANE] << 8)) : 0x8080u) |
                    ((unsigned) p[2*G::RINGX_PLANE] << 16) | ((unsigned) p[3*G::RINGX_PLANE] << 24)) ^ 0x80808080u;
                }
              if constexpr (BLEND)
                {
                  const unsigned a=sample[3] >> 16;
                  const double inverse=a != 0u ? 1.0/(double) a : 0.0;
#pragma unroll
                  for (int c=0; c < 3; c++)
                    level[c]=(unsigned) ((double) sample[c]*inverse+0.5);   // alpha*p / alpha
                  level[3]=a;
                }
              else
                {
#pragma unroll
                  for (int c=0; c < 4; c++)
                    level[c]=sample[c] >> 16;
                }
            };
            recomputed+=settle_doubtful_pixels<BLEND,4>(doubtful && (cb >= 0) && (x < W) && (y < H),lane,
              args.taps64,K,fetch,q);
          }
          out_tile[n*G::OUT_STRIDE+4*wave+kq]=make_uint2(q[0] | (q[1] << 16),q[2] | (q[3] << 16));
        }
      }
#ifdef MH_EXACT_TRACE
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      MH_XTRACE_MARK(6);
      bool stop=false;
      if (args.give_up != nullptr)
        {
          stop=stop_word == args.give_up_token;  // (written before barrier X, rewritten after barrier Y)
          if ((recomputed > 32u+(unsigned) g) && (lane == 0))
            __hip_atomic_store(args.give_up,args.give_up_token,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
        }
      __syncthreads();                           // Y: out_tile complete, staging and ring reads done
      MH_XTRACE_MARK(7);
      if (stop)
        return;                                  // the passes behind this kernel write the frame
      ring_group=ring_group+1 == G::NR ? 0 : ring_group+1;
    }
  store_row(nblocks-1);
  if ((args.recomputed != nullptr) && (recomputed != 0u) && (lane == 0))     // a wave-uniform count
    atomicAdd(args.recomputed,(unsigned long long) recomputed);
}

// MhExactBlurRecomputed: one device counter per device, allocated on the first enable
static unsigned long long *g_recomputed[64]={};
static bool g_count_recomputed=false;

unsigned long long *exact_recomputed_counter(int device)
{
  return (g_count_recomputed && (device >= 0) && (device < 64)) ? g_recomputed[device] : nullptr;
}

template<int NC,int MODE,bool UNSHARP>
static MhStatus launch_exact_typed(const View &src,BlurExactArgs &args)
{
  typedef ExactGeometry<NC> G;
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
  // once per kernel and device (the attribute is per device)
  static bool attribute_set[64]={};
  const int slot=src.device >= 0 && src.device < 64 ? src.device : 0;
  if (!attribute_set[slot])
    {
      MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&blur_fused_exact_kernel<NC,MODE,UNSHARP>),
        hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
      attribute_set[slot]=true;
    }
#ifdef MH_EXACT_TRACE
  const char *trace_path=option("MAGICKHIP_EXACT_TRACE");
  const size_t trace_bytes=4u*4u*48u*12u*sizeof(unsigned long long);
  if (trace_path != nullptr)
    {
      MH_HIP(hipMalloc(reinterpret_cast<void **>(&args.trace),trace_bytes));
      MH_HIP(hipMemsetAsync(args.trace,0,trace_bytes,src.stream));
    }
#endif
  {
  ProfileScope prof(UNSHARP ? "unsharp_fused_exact" : "blur_fused_exact",src.stream);
  hipLaunchKernelGGL((blur_fused_exact_kernel<NC,MODE,UNSHARP>),dim3((unsigned) (8*args.items_per_xcd)),
    dim3(1024),lds,src.stream,args);
  MH_HIP(hipGetLastError());
  }
#ifdef MH_EXACT_TRACE
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
static MhStatus launch_exact_modes(const View &src,BlurExactArgs &args,bool blend,bool unsharp)
{
  if (src.channels == 3)
    return unsharp ? launch_exact_typed<NC,MFMA_PLAIN3,true>(src,args) :
      launch_exact_typed<NC,MFMA_PLAIN3,false>(src,args);
  if (unsharp)
    return blend ? launch_exact_typed<NC,MFMA_BLEND4,true>(src,args) :
      launch_exact_typed<NC,MFMA_PLAIN4,true>(src,args);
  return blend ? launch_exact_typed<NC,MFMA_BLEND4,false>(src,args) :
    launch_exact_typed<NC,MFMA_PLAIN4,false>(src,args);
}

// taps: host doubles in the reversed walk of morphology.c:2746 (taps[v] multiplies the input at
// o-shift+v).
// *handled = false: the shape or the taps are outside the kernel's reach, nothing was launched.
MhStatus launch_blur_fused_exact(const View &src,const View &dst,const double *taps,int ntaps,int shift,
  bool blend,bool *handled,bool unsharp,double gain,double threshold,
  unsigned long long *recomputed_device,unsigned *give_up,unsigned give_up_token)
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
  const ExactTapPlan plan=plan_exact_taps(taps,ntaps);
  if (!plan.ok)
    return MH_OK;
  ExactDeviceTaps device;
  MH_TRY(upload_exact_taps(src,taps,ntaps,plan,&device));
  BlurExactArgs args;
  args.src=static_cast<const uint16_t *>(src.pixels);
  args.dst=static_cast<uint16_t *>(dst.pixels);
  args.columns=(int) src.columns;
  args.rows=(int) src.rows;
  args.ntaps=ntaps;
  args.shift=shift;
  args.taps64=device.taps64;
  args.taps=device.taps;
  args.digits=device.digits;
  args.offset=blend ? plan.offset_blend : plan.offset_plain;
  args.alpha_scale=plan.alpha_scale;
  args.colour_window=plan.colour_window;
  args.alpha_half_window=0.5-(blend ? plan.alpha_window_blend : plan.alpha_window_plain);
  args.alpha_floor=plan.alpha_floor;
  args.tap_scale=args.two_over_scale=args.quantum_unit=0.0f;       // (the hybrid kernel's)
  args.gain=(float) gain;
  {
    const double level=std::ceil(65535.0*threshold);
    args.threshold=level > 131072.0 ? 131072 : (level < 0.0 ? 0 : (int) level);
  }
  args.recomputed=recomputed_device;
  args.give_up=give_up;
  args.give_up_token=give_up_token;
  args.trace=nullptr;
  if ((args.recomputed == nullptr) && g_count_recomputed && (src.device >= 0) && (src.device < 64))
    args.recomputed=g_recomputed[src.device];
  *handled=true;
  const int nc=(ntaps+15+31)/32;                 // 16 outputs + K-1 halo, in 32-sample chunks
  if (nc == 1)
    return launch_exact_modes<1>(src,args,blend,unsharp);
  if (nc == 2)
    return launch_exact_modes<2>(src,args,blend,unsharp);
  if (nc == 3)
    return launch_exact_modes<3>(src,args,blend,unsharp);
  *handled=false;
  return MH_OK;
}

} // namespace mh

extern "C" MH_API unsigned long long MhExactBlurRecomputed(int enable)
{
  using namespace mh;
  int device=0;
  if ((hipGetDevice(&device) != hipSuccess) || (device < 0) || (device >= 64))
    return 0ull;
  unsigned long long count=0ull;
  if (g_recomputed[device] != nullptr)
    {
      if (hipDeviceSynchronize() == hipSuccess)
        (void) hipMemcpy(&count,g_recomputed[device],sizeof(count),hipMemcpyDeviceToHost);
      (void) hipMemset(g_recomputed[device],0,sizeof(count));
    }
  else if (enable != 0)
    {
      if (hipMalloc(reinterpret_cast<void **>(&g_recomputed[device]),sizeof(count)) != hipSuccess)
        g_recomputed[device]=nullptr;
      else
        (void) hipMemset(g_recomputed[device],0,sizeof(count));
    }
  g_count_recomputed=enable != 0;
  return count;
}
