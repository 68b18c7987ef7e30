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
// COLX = true  (MH_PRECISION_EXACT): both passes exact -> the result is BIT-IDENTICAL to the
//              reference's BlurImage / UnsharpMaskImage;
// COLX = false (MH_PRECISION_FAST): exact row pass — the intermediate is the reference's own
//              intermediate, bit for bit — and the f16 column pass of convolve_fused.hip, whose
//              result is within +-1 level of the reference's column pass on the same input: the
//              two-pass composite is +-1 BY CONSTRUCTION, on any content.
//
// The walk, the ring and the roles of the waves are those of convolve_fused.hip.
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

template<int NC,bool COLX>
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
  static constexpr int ring_bytes=COLX ? 4*RINGX_PLANE : (int) (2*F::RING_PLANE*sizeof(_Float16));
  static constexpr int OUT_STRIDE=COLS+1;
  static constexpr size_t lds_bytes=(size_t) ring_bytes+stage_bytes+(size_t) GROUP*OUT_STRIDE*sizeof(uint2);
  static_assert(lds_bytes <= 163840,"more than the 160 KiB of a CU");
  static constexpr int GROUPS_PER_ROW=XS/4;
  static constexpr int FETCH_GROUPS=GROUP*GROUPS_PER_ROW;
  static_assert(FETCH_GROUPS <= 1024,"one staging round");
  static_assert((FETCH_GROUPS % 64) == 0,"whole staging waves");
  static_assert((SRX % 16) == 0,"16-byte operand reads");
};

template<int NC,int MODE,bool UNSHARP,bool COLX>
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
  typedef ExactGeometry<NC,COLX> G;
  typedef Fused16Geometry<NC> F;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char *ring=smem_raw;                          // COLX: byte planes
  _Float16 *ring_hi=reinterpret_cast<_Float16 *>(smem_raw);   // !COLX: the f16 operand planes
  _Float16 *ring_lo=ring_hi+F::RING_PLANE;
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
  // 64c+16kq+b.  COLX = false: also the f16 operands of the column pass (convolve_fused.hip).
  intx4 t0[kExactDigits];                       // slots 0..63: 16 per k quarter
  SecondOperand t1[kExactDigits];               // slots from 64: 8 or 16 per k quarter (NX = 2 only)
  half8 t_hi[COLX ? 1 : NC],t_lo[COLX ? 1 : NC];
  {
    constexpr int DL=176;                        // digit line: tap v at [16+v], zeros around
    signed char *digit_lds=reinterpret_cast<signed char *>(stage);
    float *tap_lds=reinterpret_cast<float *>(stage+kExactDigits*DL);
    for (int at=tid; at < kExactDigits*DL; at+=1024)
      {
        const int j=at/DL,v=at-j*DL-16;
        digit_lds[at]=((v >= 0) && (v < K)) ? args.digits[j*kExactDigitPitch+v] : (signed char) 0;
      }
    if constexpr (!COLX)
      for (int j=tid; j < K; j+=1024)
        tap_lds[j]=args.taps[j];
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
    if constexpr (!COLX)
      {
#pragma unroll
        for (int c=0; c < NC; c++)
#pragma unroll
          for (int i=0; i < 8; i++)
            {
              int j=32*c+8*kq+i-n;
              const bool inside=(j >= 0) && (j < K);
              j=inside ? j : 0;
              const float tap=inside ? 256.0f*tap_lds[j] : 0.0f;
              _Float16 h,l;
              split_f16(tap,h,l);
              t_hi[c][i]=h;
              t_lo[c][i]=l;
            }
      }
    __syncthreads();                             // digit_lds is the staging plane
    // every byte plane starts as 0x80 = the signed sample byte of zero: the planes 0 and 1 of an
    // alpha / plain channel (sample = level * 2^16) are never written again
    {
      const int words=(int) ((COLX ? G::ring_bytes : 0)+G::stage_bytes)/4;
      unsigned *fill=reinterpret_cast<unsigned *>(COLX ? ring : stage);
      for (int at=tid; at < words; at+=1024)
        fill[at]=0x80808080u;
    }
    __syncthreads();
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
  auto stage_group=[&]()
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
  // column pass: entry e = 4*column + channel, 16 outputs along y
  //   COLX: every wave owns one tile (four columns);  a 64-row chunk spans four ring groups, one
  //         per k quarter
  //   else: the tiles belong to the waves that do not stage (convolve_fused.hip)
  constexpr int TILE_WAVES=COLX ? 16 : 16-G::FETCH_GROUPS/64;
  constexpr int CT=(16+TILE_WAVES-1)/TILE_WAVES;
  const int tile_wave=wave-(16-TILE_WAVES);    // < 0: a staging wave
  const int ctiles=tile_wave < 0 ? 0 : 16/TILE_WAVES+(tile_wave < 16 % TILE_WAVES ? 1 : 0);
  const int ctile0=tile_wave < 0 ? 0 : tile_wave*(16/TILE_WAVES)+(tile_wave < 16 % TILE_WAVES ? tile_wave : 16 % TILE_WAVES);
  constexpr int GROUP_STRIDE=2*F::OB;          // f16 ring: halves per 16-row group
  const int col_entry16=(n & 3)*F::CHC+(4*ctile0+(n >> 2))*8+(kq & 1)*F::OB;
  const int ring_entry16=kq*F::CHC+(rq >> 1)*F::OB+(16*ot+n)*8+4*(rq & 1);
  const int col_entryx=(n & 3)*G::CHU+(4*wave+(n >> 2))*16;                       // + group*1024
  const int ring_entryx=kq*G::CHU+(16*ot+n)*16+4*rq;                              // + group*1024
  int ring_group=0;                            // g mod NR (wave-uniform)
  uint2 original=make_uint2(0u,0u);
  auto fetch_original=[&](int block)
  {
    const int x=x0+lane,y=out_begin+G::GROUP*block+wave;
    if ((block >= 0) && (block < nblocks) && (x < W) && (y < H))
      {
        original=load_pixel16(args.src+pixel_index(y,W,x)*PX);
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
    if ((block >= 0) && (block < nblocks))
      {
        uint2 result=out_tile[wave*G::OUT_STRIDE+lane];
        if constexpr (UNSHARP)
          result=unsharp_pixel(original,result,args.gain,args.threshold);
        const int x=x0+lane,y=out_begin+G::GROUP*block+wave;
        if ((x < W) && (y < H))
          store_pixel16(args.dst+pixel_index(y,W,x)*PX,result);
      }
  };
  fetch(0);
  __shared__ unsigned stop_word;
  if (tid == 0)
    stop_word=0u;
  for (int g=0; g <= ngroups+1; g++)
    {
      // BlurExactArgs::give_up: one lane looks at the word, the whole workgroup sees its copy behind
      // barrier X and leaves behind barrier Y
      unsigned seen_word=0u;
      if ((args.give_up != nullptr) && (tid == 0))
        seen_word=__hip_atomic_load(args.give_up,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
      const int cb=g-G::NG-1;                    // the column pass's block of this iteration
      const int first=ring_group;                // ring slot of group cb: (g-NG-1) mod NR = g mod NR
      const int previous=ring_group == 0 ? G::NR-1 : ring_group-1;   // ring slot of group g-1
      if constexpr (COLX)
        {
          // the rows of block cb-1 (out_tile was written in the previous interval B)
          store_row(cb-1);
          if constexpr (UNSHARP)
            fetch_original(cb);
        }
      else if constexpr (UNSHARP)
        fetch_original(cb);
      MH_XTRACE_MARK(0);
      if (g < ngroups)
        {
#ifdef MH_EXACT_TRACE
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          MH_XTRACE_MARK(1);
#endif
          stage_group();
#ifdef MH_EXACT_TRACE
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          MH_XTRACE_MARK(2);
#endif
          if (g+1 < ngroups)
            fetch(g+1);
        }
      // ---- (COLX = false) row epilogue of group g-1: exact levels -> the f16 column pass's samples.
      // Runs in interval A.  (-DMH_HYBRID_EPILOGUE_IN_B puts it beside the row chain of group g in
      // interval B, which otherwise holds nothing but matrix instructions; legal — ring group g-1 is
      // then complete at barrier Y and first read in interval A of iteration g+1 — and measured
      // 2 % SLOWER, 0.724 against 0.707 ms: vector instructions of one wave do not hide behind the
      // matrix instructions of another beyond what the issue model of tools/ubench says.)
      auto row_epilogue16=[&]()
      {
            unsigned q[4];
            const bool doubtful=exact_levels<BLEND>(sums_row,args,q);
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
            // alpha*colour*2^-17 and alpha/2 (plain: level/2)
            float v[4];
            const f32x2 c01={(float) q[0],(float) q[1]};
            const f32x2 c23={(float) q[2],(float) q[3]};
            if constexpr (BLEND)
              {
                const float alpha=c23[1];
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
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"
                         "s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3"
                         : "+v"(v[0]),"+v"(v[1]),"+v"(v[2]),"+v"(v[3]));
            uint2 hi,lo;
            split_f16_pair(f32x2{v[0],v[1]},hi.x,lo.x);
            split_f16_pair(f32x2{v[2],v[3]},hi.y,lo.y);
            const int at=ring_entry16+previous*GROUP_STRIDE;
            *reinterpret_cast<uint2 *>(ring_hi+at)=hi;
            *reinterpret_cast<uint2 *>(ring_lo+at)=lo;
                };
      MH_XTRACE_MARK(3);
      // ======================================================================== interval A
      if constexpr (COLX)
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
      else
        {
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
              auto column_tiles=[&](auto count)
              {
                constexpr int N=decltype(count)::value;
                floatx4 acc[N > 0 ? N : 1];
#pragma unroll
                for (int i=0; i < N; i++)
                  acc[i]=floatx4{0.0f,0.0f,0.0f,0.0f};
#pragma unroll
                for (int c=0; c < NC; c++)
                  {
                    half8 a_hi[N > 0 ? N : 1],a_lo[N > 0 ? N : 1];
#pragma unroll
                    for (int i=0; i < N; i++)
                      {
                        a_hi[i]=*reinterpret_cast<const half8 *>(ring_hi+chunk_at[c]+4*i*F::SC);
                        a_lo[i]=*reinterpret_cast<const half8 *>(ring_lo+chunk_at[c]+4*i*F::SC);
                      }
#pragma unroll
                    for (int i=0; i < N; i++)
                      acc[i]=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[i],t_hi[c],acc[i],0,0,0);
#pragma unroll
                    for (int i=0; i < N; i++)
                      acc[i]=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[i],t_hi[c],acc[i],0,0,0);
#pragma unroll
                    for (int i=0; i < N; i++)
                      acc[i]=__builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[i],t_lo[c],acc[i],0,0,0);
                  }
#pragma unroll
                for (int i=0; i < N; i++)
                  out_tile[n*G::OUT_STRIDE+4*(ctile0+i)+kq]=sums_to_quantum<SAMPLES>(acc[i][0],acc[i][1],acc[i][2],acc[i][3]);
              };
              if (ctiles == CT)
                column_tiles(std::integral_constant<int,CT>{});
              else if (ctiles == CT-1)
                column_tiles(std::integral_constant<int,CT-1>{});
            }
#ifndef MH_HYBRID_EPILOGUE_IN_B
          row_epilogue16();
#endif
        }
#ifdef MH_EXACT_TRACE
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      MH_XTRACE_MARK(4);
      if ((args.give_up != nullptr) && (tid == 0))
        stop_word=seen_word;
      __syncthreads();                           // X: group g staged, ring group g-1 complete
      MH_XTRACE_MARK(5);
      // ======================================================================== interval B
      {
#ifdef MH_HYBRID_EPILOGUE_IN_B
        if constexpr (!COLX)
          row_epilogue16();
#endif
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
        if constexpr (COLX)
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
        else
          {
            // the store of the column pass's row in the shadow of the matrix chain
            store_row(cb);
            exact_sums(acc,args.offset,sums_row);
          }
      }
#ifdef MH_EXACT_TRACE
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      MH_XTRACE_MARK(6);
      bool stop=false;
      if (args.give_up != nullptr)
        {
          stop=stop_word != 0u;                  // (written before barrier X, rewritten after barrier Y)
          if ((recomputed > 32u+(unsigned) g) && (lane == 0))
            __hip_atomic_store(args.give_up,1u,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
        }
      __syncthreads();                           // Y: out_tile complete, staging and ring reads done
      MH_XTRACE_MARK(7);
      if (stop)
        return;                                  // the passes behind this kernel write the frame
      ring_group=ring_group+1 == G::NR ? 0 : ring_group+1;
    }
  if constexpr (COLX)
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

template<int NC,int MODE,bool UNSHARP,bool COLX>
static MhStatus launch_exact_typed(const View &src,BlurExactArgs &args)
{
  typedef ExactGeometry<NC,COLX> G;
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
      MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&blur_fused_exact_kernel<NC,MODE,UNSHARP,COLX>),
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
  ProfileScope prof(UNSHARP ? (COLX ? "unsharp_fused_exact" : "unsharp_fused_exact_row") :
    (COLX ? "blur_fused_exact" : "blur_fused_exact_row"),src.stream);
  hipLaunchKernelGGL((blur_fused_exact_kernel<NC,MODE,UNSHARP,COLX>),dim3((unsigned) (8*args.items_per_xcd)),
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

template<int NC,bool COLX>
static MhStatus launch_exact_modes(const View &src,BlurExactArgs &args,bool blend,bool unsharp)
{
  if (src.channels == 3)
    return unsharp ? launch_exact_typed<NC,MFMA_PLAIN3,true,COLX>(src,args) :
      launch_exact_typed<NC,MFMA_PLAIN3,false,COLX>(src,args);
  if (unsharp)
    return blend ? launch_exact_typed<NC,MFMA_BLEND4,true,COLX>(src,args) :
      launch_exact_typed<NC,MFMA_PLAIN4,true,COLX>(src,args);
  return blend ? launch_exact_typed<NC,MFMA_BLEND4,false,COLX>(src,args) :
    launch_exact_typed<NC,MFMA_PLAIN4,false,COLX>(src,args);
}

// taps: host doubles in the reversed walk of morphology.c:2746 (taps[v] multiplies the input at
// o-shift+v).  exact_column: both passes exact (bit-identical result) or the f16 column pass.
// *handled = false: the shape or the taps are outside the kernel's reach, nothing was launched.
MhStatus launch_blur_fused_exact(const View &src,const View &dst,const double *taps,int ntaps,int shift,
  bool blend,bool exact_column,bool *handled,bool unsharp,double gain,double threshold,
  unsigned long long *recomputed_device,unsigned *give_up)
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
  args.tap_scale=256.0f;                         // (the f16 column pass, COLX = false, keeps its fixed factor)
  args.two_over_scale=1.0f/128.0f;
  args.quantum_unit=1.0f/(128.0f*65535.0f);
  args.gain=(float) gain;
  {
    const double level=std::ceil(65535.0*threshold);
    args.threshold=level > 131072.0 ? 131072 : (level < 0.0 ? 0 : (int) level);
  }
  args.recomputed=recomputed_device;
  args.give_up=give_up;
  args.trace=nullptr;
  if ((args.recomputed == nullptr) && g_count_recomputed && (src.device >= 0) && (src.device < 64))
    args.recomputed=g_recomputed[src.device];
  *handled=true;
  const int nc=(ntaps+15+31)/32;                 // 16 outputs + K-1 halo, in 32-sample chunks
  if (exact_column)
    {
      if (nc == 1)
        return launch_exact_modes<1,true>(src,args,blend,unsharp);
      if (nc == 2)
        return launch_exact_modes<2,true>(src,args,blend,unsharp);
      if (nc == 3)
        return launch_exact_modes<3,true>(src,args,blend,unsharp);
    }
  else
    {
      if (nc == 1)
        return launch_exact_modes<1,false>(src,args,blend,unsharp);
      if (nc == 2)
        return launch_exact_modes<2,false>(src,args,blend,unsharp);
      if (nc == 3)
        return launch_exact_modes<3,false>(src,args,blend,unsharp);
    }
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
