// Separable-convolution pass on the matrix cores (FAST precision, Q16: RGBA with
// alpha-weighted colour channels — BlurImage's case, MagickCore/morphology.c:2654-2979 —,
// RGB and four plain channels; template parameter MODE).  The same kernel also runs
//   * the two passes of a separated 2-D kernel (GaussianBlurImage), which hand the undivided
//     f32 sums from the row pass to the column pass (IO = MFMA_TO_SUMS / MFMA_FROM_SUMS), and
//   * UnsharpMaskImage's column pass with its threshold/gain epilogue fused into the copy-out
//     (IO = MFMA_UNSHARP, effect.c:4364-4369).
//
// A K-tap 1-D convolution of a tile is a banded (Toeplitz) matrix product
//
//     out[e][n] = sum_k  data[e][k] * T[k][n],      T[k][n] = tap[k-n]  (0 <= k-n < K)
//
// with e = (pixel column or row, channel), n = output position along the filter axis and
// k = input position.  At 79 taps the vector-ALU formulation needs 316 f32 multiply-adds
// per pixel and pass and is issue-bound at ~35 % of the vector peak while
// HBM idles at 17 %; the f16 MFMA pipe has 16x the f32 FMA rate, so the same sums are
// formed there and the pass becomes a stream again.
//
// Precision.  The reference evaluates gamma*sum(k*alpha*p) with gamma=1/sum(k*alpha) in
// fp64 and rounds to Q16; FAST promises +-1 level.  f16 carries 11 significant bits, so
// every operand is split into two f16 terms (22 bits, relative error 2^-22):
//     v  = alpha*p*2^-17 (colour) or alpha/2 (alpha channel)   -> v_hi + v_lo
//     t  = 256*tap                                             -> t_hi + t_lo
// and the product is accumulated in f32 as v_hi*t_hi + v_lo*t_hi + v_hi*t_lo (the dropped
// v_lo*t_lo term is 2^-22 of the result).  f16 x f16 products are exact in f32; the f32
// accumulation is the same kind of error the vector FAST path has.  The scale factors are
// powers of two and cancel exactly in the epilogue.
//
// Shape.  v_mfma_f32_32x32x16_f16: A = data (32 entries e x 16 k), B = Toeplitz taps
// (16 k x 32 n), D (32 e x 32 n) with D row = (reg&3)+8*(reg>>2)+4*(lane>>5), col = lane&31:
// with e = 4*pixel+channel the four channels of a pixel sit in four consecutive registers
// of ONE lane, so the gamma division is lane-local.  The tap operands depend only on
// (lane, k-chunk) and stay in registers for the lifetime of a persistent workgroup.
//
// Streaming.  A workgroup walks a strip of 16 units (pixel columns for the column pass,
// rows for the row pass) along the filter axis in steps of 64 outputs.  The converted
// samples live in an LDS ring of R = 16*NQ+32 axis positions; a step stages only the 64
// positions that are new (every input sample is fetched and converted once, not once per
// tile it overlaps — the first, tile-shaped version of this kernel spent 2.25x the staging
// work on halos and measured 0.53 ms per pass), multiplies, and leaves.
#include "mh_internal.hpp"
#include "device_common.hpp"
#include "mfma_common.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

namespace mh {

struct ConvMfmaArgs
{
  const uint16_t *src;
  uint16_t *dst;
  unsigned long long *trace;   // diagnostic builds only
  const uint16_t *orig;        // MFMA_UNSHARP: the unblurred frame
  float gain;                  // MFMA_UNSHARP
  int threshold;               // MFMA_UNSHARP: ceil(QuantumRange*threshold), see unsharp_pair
  int columns,rows;
  int ntaps;
  int shift;                 // K-1-origin: offset of the first input sample
  const float *taps;         // float[K], taps[v] multiplies input o-shift+v
  const double *taps64;      // the same taps as doubles: exact_alpha_level (row pass, blend mode)
  // the f16 operands are tap_scale*tap (f16_tap_scale, mfma_common.hpp; 256 where the sums leave or enter
  // the pass as floats: MFMA_TO_SUMS / MFMA_FROM_SUMS); two_over_scale = 2/tap_scale, quantum_unit = that / 65535
  float tap_scale,two_over_scale,quantum_unit;
  int strips,segments,steps_per_segment,steps;   // strips of UNITS units, steps of STEP outputs
};

// Strip shape, both passes: 16 units x 64 outputs per step; 4 waves = 2 unit groups x 2
// output groups.  (32 columns x 32 outputs was tried for the column pass to get 256-byte row
// segments, and 8 rows x 128 outputs for the row pass to get 1 KB row segments and a 28 KB
// ring: neither is faster.)
template<bool VERTICAL> struct StripShape
{
  static constexpr int UNITS=16;
  static constexpr int STEP=64;
};

// LDS: two planes (hi, lo) of f16, plane[channel][unit 0..15][ring slot 0..R), the filter
// axis contiguous: line stride S halves, channel stride CH = 16*S+PAD halves.  Bank rules
// (MI355X_MICROARCH.md, LDS):
//  * ds_read_b128 (the operand lines) is served in four 16-lane groups {0-3,12-15,20-27},
//    {4-11,16-19,28-31}, ... over 64 banks: the 4 units x 4 channels of a group must hit 16
//    different 16-byte slots of a 256-byte bank row (reads_conflict_free below checks a
//    candidate at compile time).  (S = R+8 with PAD = 8 measured 77 % of all LDS cycles as
//    bank conflicts, 0.41 ms per pass.)
//  * ds_write_b64 (staging) is served in contiguous 16-lane groups over 32 banks.  The row
//    pass writes 128 contiguous bytes per group whatever S is; the column pass writes 2 units
//    x 64 bytes, conflict-free when the line stride is 64 bytes mod 128.
// Row pass: the smallest conflict-free-read layout — S = R, PAD = 8 when R = 32 mod 64,
// else S = R+8, PAD = 32.  Column pass: the smallest S >= R that is 32 mod 64 halves, PAD = 8
// (conflict-free reads and writes).  A smaller ring means more workgroups per CU: 24 KB for
// up to 33 taps, 39 KB (row) for the 79 taps of sigma = 10.
static constexpr bool reads_conflict_free(int S,int PAD)
{
  const int group[2][4]={{0,3,5,6},{1,2,4,7}};
  const int CH=16*S+PAD;
  for (int mg=0; mg < 2; mg++)
    for (int g=0; g < 2; g++)
      {
        unsigned seen=0;
        for (int i=0; i < 4; i++)
          for (int c=0; c < 4; c++)
            {
              const int bytes=(c*CH+(8*mg+group[g][i])*S)*2;
              const unsigned slot=1u << ((bytes % 256)/16);
              if ((seen & slot) != 0)
                return false;
              seen|=slot;
            }
      }
  return true;
}

template<bool VERTICAL,int NQ>
struct MfmaGeometry
{
  static constexpr int UNITS=StripShape<VERTICAL>::UNITS,STEP=StripShape<VERTICAL>::STEP;
  static constexpr int R=16*NQ+STEP-32;        // ring positions: STEP outputs + K-1 halo, 16-aligned
  static constexpr int S=VERTICAL ? ((R-32+63)/64)*64+32 : (R % 64 == 32 ? R : R+8);
  static constexpr int PAD=VERTICAL || (R % 64 == 32) ? 8 : 32;
  static_assert((S >= R) && (S % 8 == 0) && reads_conflict_free(S,PAD),"ring layout with LDS bank conflicts");
  static constexpr int CH=UNITS*S+PAD;         // halves per channel
  static constexpr int PLANE=4*CH;             // halves per plane
  static constexpr int OUT_STRIDE=(UNITS+2)*4; // u16 per output row: 144 bytes, 16-byte aligned,
                                               // 16 rows spread over 8 bank groups instead of 1
  static constexpr int OUT=STEP*OUT_STRIDE;    // u16 of the column pass's output tile
  static constexpr size_t ring_bytes=(size_t) 2*PLANE*sizeof(_Float16);
  static constexpr size_t out_bytes=(size_t) OUT*sizeof(uint16_t);
};

// VERTICAL: units are pixel columns, the filter axis runs down the rows (column pass).
// 4 waves: wave w multiplies unit group w&1 (8 units = 32 entries) by output group w>>1
// (32 outputs) of the step.
// MODE: an MfmaMode (mfma_common.hpp).

// Diagnostic build only (-DMH_MFMA_TRACE, tools/trace_blur_steps.py): wave 0 of a few workgroups
// records the shader clock at the phase boundaries of its first steps.
#ifdef MH_MFMA_TRACE
#define MH_TRACE_MARK(id) \
  do { \
    if ((trace != nullptr) && (trace_step < 48)) \
      { \
        const unsigned long long now=__builtin_readcyclecounter(); \
        if (lane == 0) \
          trace[trace_step*8+(id)]=now; \
      } \
  } while (0)
#else
#define MH_TRACE_MARK(id) do { } while (0)
#endif

struct __attribute__((packed,aligned(2))) Rgb16 { uint16_t c[3]; };

// IO (enum MfmaIo, mh_internal.hpp): what a pass reads and writes.

template<bool VERTICAL,int NQ,int MODE,int IO>
__global__ __launch_bounds__(256)
void conv_mfma_kernel(ConvMfmaArgs args)
{
  static_assert((IO == MFMA_Q16) || ((IO == MFMA_TO_SUMS) && !VERTICAL) ||
    (((IO == MFMA_FROM_SUMS) || (IO == MFMA_UNSHARP)) && VERTICAL),
    "sums are written by a row pass and read by a column pass; the unsharp epilogue is a column pass's");
  static_assert((IO != MFMA_UNSHARP) || (MODE != MFMA_PLAIN3),"the fused unsharp copy-out handles 8-byte pixels");
  // (the sums always have four floats per pixel; for RGB the fourth is zero)
  typedef typename std::conditional<IO == MFMA_FROM_SUMS,uint4,uint2>::type Raw;
  constexpr int PX=MODE == MFMA_PLAIN3 ? 3 : 4;    // u16 per pixel in memory
  typedef MfmaGeometry<VERTICAL,NQ> G;
  constexpr int R=G::R,kStripUnits=G::UNITS,kStepOutputs=G::STEP;
  constexpr int MG=kStripUnits/8;              // unit groups; output groups = 4/MG
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16 *plane_hi=reinterpret_cast<_Float16 *>(smem_raw);
  _Float16 *plane_lo=plane_hi+G::PLANE;
  uint16_t *tile_out=reinterpret_cast<uint16_t *>(plane_lo+G::PLANE);      // [64][16][4], column pass
  const int tid=(int) threadIdx.x,lane=tid & 63;
  const int wave=__builtin_amdgcn_readfirstlane(tid >> 6);
  const int n=lane & 31,half=lane >> 5;
  const int mg=wave % MG,ng=wave/MG;
  const int K=args.ntaps;
  const int W=args.columns,H=args.rows;

  // ---- Toeplitz operands: T[q][i] = tap_scale*tap[16q+8*half+i-n]; taps staged through LDS
  half8 t_hi[NQ],t_lo[NQ];
  {
    float *tap_lds=reinterpret_cast<float *>(smem_raw);
    for (int j=tid; j < K; j+=256)
      tap_lds[j]=args.taps[j];
    __syncthreads();
#pragma unroll
    for (int q=0; q < NQ; q++)
#pragma unroll
      for (int i=0; i < 8; i++)
        {
          const int j=16*q+8*half+i-n;
          const float t=((j >= 0) && (j < K)) ? args.tap_scale*tap_lds[j] : 0.0f;
          _Float16 h,l;
          split_f16(t,h,l);
          t_hi[q][i]=h;
          t_lo[q][i]=l;
        }
  }

  // staging role of this thread: 4 consecutive axis positions (`group`) of one unit
  int stage_unit,stage_group;
  if (VERTICAL)
    {
      // ds_write_b64 is served in contiguous 16-lane groups over 32 banks: 2 units x 8 groups
      // per 16 lanes write 2 x 64 contiguous bytes, 128 bytes (mod 256) apart
      stage_unit=8*(wave & 1)+(lane >> 3);
      stage_group=8*(wave >> 1)+(lane & 7);
    }
  else
    {
      stage_unit=tid/(kStepOutputs/4);           // adjacent lanes: adjacent x groups of a row
      stage_group=tid % (kStepOutputs/4);
    }
  Raw raw[4];
  // MFMA_UNSHARP: the unblurred pixels of one step's outputs, in the copy-out's thread layout
  // (always a valid 16-byte pair: clamped, W >= 2)
  constexpr int kOriginals=IO == MFMA_UNSHARP ? (G::STEP*G::UNITS/2)/256 : 1;
  uint4 original[kOriginals];
#pragma unroll
  for (int round=0; round < kOriginals; round++)
    original[round]=make_uint4(0u,0u,0u,0u);
  auto fetch_original=[&](int unit0,int first_output)
  {
    if constexpr (IO == MFMA_UNSHARP)
      {
#pragma unroll
        for (int round=0; round < kOriginals; round++)
          {
            const int u=tid+256*round;
            const int row=u/(G::UNITS/2),pair=u % (G::UNITS/2);
            int x=unit0+2*pair,y=first_output+row;
            x=x > args.columns-2 ? args.columns-2 : x;
            y=y > args.rows-1 ? args.rows-1 : y;
            original[round]=*reinterpret_cast<const uint4 *>(args.orig+pixel_index(y,args.columns,x)*4);
          }
      }
  };
  // fetch the 4 samples at axis positions pos..pos+3 of this thread's unit (edge clamp,
  // cache.c:2663-2679)
  auto fetch=[&](Raw (&buf)[4],int unit0,int pos)
  {
#pragma unroll
    for (int i=0; i < 4; i++)
      {
        int x=VERTICAL ? unit0+stage_unit : pos+i;
        int y=VERTICAL ? pos+i : unit0+stage_unit;
        x=x < 0 ? 0 : (x > W-1 ? W-1 : x);
        y=y < 0 ? 0 : (y > H-1 ? H-1 : y);
        const size_t at=pixel_index(y,W,x);
        if constexpr (IO == MFMA_FROM_SUMS)
          buf[i]=*reinterpret_cast<const uint4 *>(reinterpret_cast<const float *>(args.src)+at*4);
        else if constexpr (MODE == MFMA_PLAIN3)
          {
            Rgb16 p;
            __builtin_memcpy(&p,args.src+at*3,sizeof(p));
            buf[i]=make_uint2((unsigned) p.c[0] | ((unsigned) p.c[1] << 16),(unsigned) p.c[2]);
          }
        else
          buf[i]=*reinterpret_cast<const uint2 *>(args.src+at*4);
      }
  };
  // convert raw[] and write it to ring slots slot..slot+3 (slot multiple of 4, no wrap inside).
  // Values are handled as pairs of neighbouring positions: packed f32 multiplies, one
  // v_cvt_pkrtz_f16_f32 per pair for the hi halves and one v_cvt_pk_f16_f32 for the lo halves.
  auto stage=[&](const Raw (&buf)[4],int slot)
  {
    f32x2 v[4][2];
#pragma unroll
    for (int j=0; j < 2; j++)
      {
        if constexpr (IO == MFMA_FROM_SUMS)
          {
            // the row pass left sums of 256*tap*v with v < 32768: 1/256 brings them back under
            // the f16 range; the factor is common to colour and alpha sums (it cancels in the
            // division) and equals the taps' own factor 256 (so the epilogues stay as they are)
            const float scale=1.0f/256.0f;
            const uint4 r0=buf[2*j],r1=buf[2*j+1];
            v[0][j]=f32x2{__builtin_bit_cast(float,r0.x),__builtin_bit_cast(float,r1.x)}*scale;
            v[1][j]=f32x2{__builtin_bit_cast(float,r0.y),__builtin_bit_cast(float,r1.y)}*scale;
            v[2][j]=f32x2{__builtin_bit_cast(float,r0.z),__builtin_bit_cast(float,r1.z)}*scale;
            v[3][j]=f32x2{__builtin_bit_cast(float,r0.w),__builtin_bit_cast(float,r1.w)}*scale;
            continue;
          }
        const unsigned r0x=buf[2*j].x,r0y=buf[2*j].y,r1x=buf[2*j+1].x,r1y=buf[2*j+1].y;
        const f32x2 c0={(float) (r0x & 0xffffu),(float) (r1x & 0xffffu)};
        const f32x2 c1={(float) (r0x >> 16),(float) (r1x >> 16)};
        const f32x2 c2={(float) (r0y & 0xffffu),(float) (r1y & 0xffffu)};
        const f32x2 c3={(float) (r0y >> 16),(float) (r1y >> 16)};
        if (MODE == MFMA_BLEND4)
          {
            const f32x2 weight=c3*(0.5f/65536.0f);
            v[0][j]=c0*weight;
            v[1][j]=c1*weight;
            v[2][j]=c2*weight;
            v[3][j]=c3*0.5f;
          }
        else
          {
            // scaled by 1/2 like the alpha entry of the blend mode: 65535 stays inside f16
            v[0][j]=c0*0.5f;
            v[1][j]=c1*0.5f;
            v[2][j]=c2*0.5f;
            v[3][j]=c3*0.5f;
          }
      }
#pragma unroll
    for (int c=0; c < 4; c++)
      {
        uint2 hi,lo;
        split_f16_pair(v[c][0],hi.x,lo.x);
        split_f16_pair(v[c][1],hi.y,lo.y);
        const int at=c*G::CH+stage_unit*G::S+slot;
        *reinterpret_cast<uint2 *>(plane_hi+at)=hi;
        *reinterpret_cast<uint2 *>(plane_lo+at)=lo;
      }
  };

  const int entry=(n & 3)*G::CH+(8*mg+(n >> 2))*G::S+8*half;    // this lane's operand line
#ifdef MH_MFMA_TRACE
  unsigned long long *trace=nullptr;
  int trace_step=0;
  if ((args.trace != nullptr) && (wave == 0) && ((blockIdx.x % 97) == 0) && (blockIdx.x/97 < 8))
    trace=args.trace+(size_t) (blockIdx.x/97)*48*8;
#endif
  const int items=args.strips*args.segments;
  for (int item=(int) blockIdx.x; item < items; item+=(int) gridDim.x)
    {
      const int strip=item/args.segments,segment=item-strip*args.segments;
      const int unit0=kStripUnits*strip;
      const int step_begin=segment*args.steps_per_segment;
      const int step_end=step_begin+args.steps_per_segment < args.steps ?
        step_begin+args.steps_per_segment : args.steps;
      const int out_begin=kStepOutputs*step_begin;
      const int in0=out_begin-args.shift;        // axis position held by ring slot 0
      __syncthreads();                           // ring and tap_lds readers of the last item are done
      // ---- prologue: positions [in0, in0+R-64) -> slots [0, R-64)
      for (int g0=0; g0 < (R-kStepOutputs)/4; g0+=kStepOutputs/4)
        {
          const int group=g0+stage_group;
          if (group < (R-kStepOutputs)/4)
            {
              fetch(raw,unit0,in0+4*group);
              stage(raw,4*group);
            }
        }
      // first step's new positions [in0+R-64, in0+R)
      fetch(raw,unit0,in0+R-kStepOutputs+4*stage_group);
      stage(raw,R-kStepOutputs+4*stage_group);
      __syncthreads();                           // the first step's samples are in the ring
      // The 64 new positions of step j+1 are fetched a whole step ahead, right after step j's
      // were staged.  Order matters: loads and stores share one counter (vmcnt) and complete out
      // of order with respect to each other, so once a store is pending a load can only be waited
      // for with vmcnt(0) — each step therefore issues its stores AFTER it has consumed the
      // previous fetch and issued the next one, so that wait only ever sees old stores.
      if (step_begin+1 < step_end)
        fetch(raw,unit0,in0+R+4*stage_group);
      fetch_original(unit0,out_begin);
      int base=0;                                // ring slot of the step's first input position
      for (int step=step_begin; step < step_end; step++)
        {
          const int out0=kStepOutputs*step;
          const bool has_next=step+1 < step_end;
          MH_TRACE_MARK(0);
          // MFMA_UNSHARP: this step's unblurred pixels were fetched a step ago, together with the
          // samples of the next step; keep them aside, the registers are refilled below
          uint4 mine[kOriginals];
#pragma unroll
          for (int round=0; round < kOriginals; round++)
            mine[round]=original[round];
          // ---- multiply
          floatx16 acc;
#pragma unroll
          for (int r=0; r < 16; r++)
            acc[r]=0.0f;
          int chunk=base+32*ng;                  // wave-uniform
          chunk=chunk >= R ? chunk-R : chunk;
#pragma unroll
          for (int q=0; q < NQ; q++)
            {
              const half8 a_hi=*reinterpret_cast<const half8 *>(plane_hi+entry+chunk);
              const half8 a_lo=*reinterpret_cast<const half8 *>(plane_lo+entry+chunk);
              acc=__builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi,t_hi[q],acc,0,0,0);
              acc=__builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo,t_hi[q],acc,0,0,0);
              acc=__builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi,t_lo[q],acc,0,0,0);
              chunk+=16;
              chunk=chunk >= R ? chunk-R : chunk;
            }
          // ---- epilogue: lane holds 4 pixels (reg>>2) x 4 channels (reg&3) of output n
          uint2 result[4];
          if constexpr (IO == MFMA_TO_SUMS)
            {
              // the undivided sums, 16 bytes per pixel, 512 contiguous bytes per 32 lanes
#pragma unroll
              for (int pg=0; pg < 4; pg++)
                {
                  const int x=out0+32*ng+n,y=unit0+8*mg+2*pg+half;
                  if ((x < W) && (y < H))
                    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(args.dst)+pixel_index(y,W,x)*4)=
                      make_float4(acc[4*pg+0],acc[4*pg+1],acc[4*pg+2],acc[4*pg+3]);
                  result[pg]=make_uint2(0u,0u);
                }
            }
          else
          {
#pragma unroll
          for (int pg=0; pg < 4; pg++)
            {
              // (with the taps' factor 256) S_c = 2^-9 * sum k*alpha*p, S_a = 128 * sum k*alpha:
              //   gamma*pixel = sum(k*alpha*p)/sum(k*alpha) = 65536 * S_c / S_a
              // v_rcp_f32(0) = inf and 0*inf = NaN convert to 0: PerceptibleReciprocal's clamp
              // for an all-transparent window (as the vector FAST epilogue)
              // plain modes: S_c = 128 * sum k*p
              // v_cvt_pknorm_u16_f32 rounds 65535*x to the nearest level, clamps to [0,65535],
              // maps NaN to 0 and packs two results: the whole quantisation in one instruction
              const float sa=acc[4*pg+3];
              const float unit=args.quantum_unit;
              const float inv=MODE == MFMA_BLEND4 ? __builtin_amdgcn_rcpf(sa)*(65536.0f/65535.0f) : unit;
              const f32x2 scale01={inv,inv};
              const f32x2 scale23={inv,MODE == MFMA_BLEND4 ? unit : inv};
              const f32x2 p01=f32x2{acc[4*pg+0],acc[4*pg+1]}*scale01;
              const f32x2 p23=f32x2{acc[4*pg+2],sa}*scale23;
              const pknorm2 lo2=__builtin_amdgcn_cvt_pknorm_u16(p01[0],p01[1]);
              const pknorm2 hi2=__builtin_amdgcn_cvt_pknorm_u16(p23[0],p23[1]);
              result[pg]=make_uint2(__builtin_bit_cast(unsigned,lo2),__builtin_bit_cast(unsigned,hi2));
              if constexpr ((MODE == MFMA_BLEND4) && !VERTICAL && (IO == MFMA_Q16))
                {
                  // the row pass's alpha becomes a weight in the column pass: exact where it is small
                  // and the f32 sum cannot decide the level (mfma_common.hpp)
                  if ((args.taps64 != nullptr) && alpha_sum_is_ambiguous(sa*args.two_over_scale))
                    {
                      const int x=out0+32*ng+n,y=unit0+8*mg+2*pg+half;
                      if ((x < W) && (y < H))
                        {
                          const unsigned level=exact_alpha_level(args.src,pixel_index(y,W,0),1,W,x-args.shift,
                            args.taps64,K);
                          result[pg].y=(result[pg].y & 0xffffu) | (level << 16);
                        }
                    }
                }
              if (VERTICAL)
                {
                  const int unit_out=8*mg+2*pg+half;         // D row = (reg&3)+8*(reg>>2)+4*half
                  const int pos_out=32*ng+n;
                  uint16_t *to=tile_out+(size_t) pos_out*G::OUT_STRIDE+unit_out*PX;
                  if (MODE == MFMA_PLAIN3)
                    {
                      to[0]=(uint16_t) (result[pg].x & 0xffffu);
                      to[1]=(uint16_t) (result[pg].x >> 16);
                      to[2]=(uint16_t) (result[pg].y & 0xffffu);
                    }
                  else
                    *reinterpret_cast<uint2 *>(to)=result[pg];
                }
            }
          }
#ifdef MH_MFMA_TRACE
          asm volatile("s_nop 0" :: "v"(result[0].x),"v"(result[3].y));      // the epilogue has issued
#endif
          MH_TRACE_MARK(1);
          __syncthreads();                       // B2: this step's ring slots may be overwritten
          MH_TRACE_MARK(2);
          if (has_next)
            {
              int slot=base+4*stage_group;       // positions in0+R+64j+4g -> slots (64j+4g) mod R
              slot=slot >= R ? slot-R : slot;
#ifdef MH_MFMA_TRACE
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              MH_TRACE_MARK(3);
#endif
              stage(raw,slot);
#ifdef MH_MFMA_TRACE
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              MH_TRACE_MARK(4);
#endif
              if (step+2 < step_end)
                fetch(raw,unit0,in0+(out0-out_begin)+R+kStepOutputs+4*stage_group);
              fetch_original(unit0,out0+kStepOutputs);
            }
          MH_TRACE_MARK(5);
          // ---- stores
          if (VERTICAL && (MODE == MFMA_PLAIN3))
            {
              // STEP rows of UNITS*3 samples (96 bytes).  Rows start 8-byte aligned when the
              // row pitch 6*W is a multiple of 8
              constexpr int ROW=kStripUnits*3;
              uint16_t *row0=args.dst+pixel_index(out0,W,unit0)*3;
              const int valid=(W-unit0)*3 < ROW ? (W-unit0)*3 : ROW;       // samples inside the image
              if ((W & 3) == 0)
                {
#pragma unroll
                  for (int round=0; round < (kStepOutputs*ROW/4)/256; round++)
                    {
                      const int u=tid+256*round;
                      const int row=u/(ROW/4),e=4*(u % (ROW/4));
                      if ((out0+row < H) && (e < valid))        // valid is a multiple of 4 here
                        *reinterpret_cast<uint2 *>(row0+pixel_index(row,W,0)*3+e)=
                          *reinterpret_cast<const uint2 *>(tile_out+(size_t) row*G::OUT_STRIDE+e);
                    }
                }
              else
                {
#pragma unroll 4
                  for (int round=0; round < (kStepOutputs*ROW)/256; round++)
                    {
                      const int u=tid+256*round;
                      const int row=u/ROW,e=u % ROW;
                      if ((out0+row < H) && (e < valid))
                        row0[pixel_index(row,W,0)*3+e]=tile_out[(size_t) row*G::OUT_STRIDE+e];
                    }
                }
            }
          else if (VERTICAL)
            {
              // coalesced copy-out: STEP rows of UNITS pixels, 16 bytes (2 pixels) per thread
#pragma unroll
              for (int round=0; round < (kStepOutputs*kStripUnits/2)/256; round++)
                {
                  const int u=tid+256*round;
                  const int row=u/(kStripUnits/2),pair=u % (kStripUnits/2);
                  const int x=unit0+2*pair,y=out0+row;
                  const uint16_t *from=tile_out+(size_t) row*G::OUT_STRIDE+2*pair*4;
                  uint16_t *to=args.dst+pixel_index(y,W,x)*4;
                  uint4 value=*reinterpret_cast<const uint4 *>(from);
                  if constexpr (IO == MFMA_UNSHARP)
                    {
                      uint4 p=mine[round];
                      if (x > W-2)               // the last pixel of an odd row: second half of the clamped pair
                        p=make_uint4(p.z,p.w,p.z,p.w);
                      const uint2 first=unsharp_pixel(make_uint2(p.x,p.y),make_uint2(value.x,value.y),
                        args.gain,args.threshold);
                      const uint2 second=unsharp_pixel(make_uint2(p.z,p.w),make_uint2(value.z,value.w),
                        args.gain,args.threshold);
                      value=make_uint4(first.x,first.y,second.x,second.y);
                    }
                  if (y < H)
                    {
                      if (x+1 < W)
                        *reinterpret_cast<uint4 *>(to)=value;
                      else if (x < W)
                        *reinterpret_cast<uint2 *>(to)=make_uint2(value.x,value.y);
                    }
                }
            }
          else if constexpr (IO != MFMA_TO_SUMS)     // (the sums went out with the epilogue)
            {
#pragma unroll
              for (int pg=0; pg < 4; pg++)
                {
                  const int x=out0+32*ng+n,y=unit0+8*mg+2*pg+half;
                  if ((x < W) && (y < H))
                    {
                      if (MODE == MFMA_PLAIN3)
                        {
                          Rgb16 p;
                          p.c[0]=(uint16_t) (result[pg].x & 0xffffu);
                          p.c[1]=(uint16_t) (result[pg].x >> 16);
                          p.c[2]=(uint16_t) (result[pg].y & 0xffffu);
                          __builtin_memcpy(args.dst+pixel_index(y,W,x)*3,&p,sizeof(p));
                        }
                      else
                        *reinterpret_cast<uint2 *>(args.dst+pixel_index(y,W,x)*4)=result[pg];
                    }
                }
            }
          base+=kStepOutputs;
          base=base >= R ? base-R : base;
          MH_TRACE_MARK(6);
          __syncthreads();                       // next step's samples are in the ring; tile_out is free
          MH_TRACE_MARK(7);
#ifdef MH_MFMA_TRACE
          trace_step++;
#endif
        }
    }
}

template<bool VERTICAL,int NQ,int MODE,int IO>
static MhStatus launch_mfma_typed(const View &src,ConvMfmaArgs &args)
{
  typedef MfmaGeometry<VERTICAL,NQ> G;
  constexpr int kStripUnits=G::UNITS,kStepOutputs=G::STEP;
  const int units=VERTICAL ? args.columns : args.rows;
  const int axis=VERTICAL ? args.rows : args.columns;
  args.strips=(units+kStripUnits-1)/kStripUnits;
  args.steps=(axis+kStepOutputs-1)/kStepOutputs;
  const size_t lds=G::ring_bytes+(VERTICAL ? G::out_bytes : 0);
  MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<VERTICAL,NQ,MODE,IO>),
    hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  // The grid is persistent (every workgroup walks its share of the items), so it must not
  // exceed what is resident at once: LDS and registers decide, asked once per instantiation
  static int resident=0;
  if (resident == 0)
    {
      int n=0;
      MH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n,
        reinterpret_cast<const void *>(&conv_mfma_kernel<VERTICAL,NQ,MODE,IO>),256,lds));
      resident=n < 1 ? 1 : n;
    }
  // three per CU even where four fit (row pass up to 79 taps, both passes up to 33): measured
  // 0.515 / 0.559 ms (3 / 4 per CU, sigma 4) and 0.538 / 0.543 ms (sigma 10) — the passes run
  // against the power limit, more waves in flight only lower the clock
  int per_cu=resident > 3 ? 3 : resident;
  if (const char *e=option("MAGICKHIP_MFMA_PER_CU"))
    per_cu=atoi(e) < 1 ? 1 : (atoi(e) < per_cu ? atoi(e) : per_cu);
  const int nblocks=compute_units(src.device)*per_cu;
  // Cut the strips into segments so that the work items divide evenly among the resident
  // workgroups.  A segment re-stages R-64 positions, so it stays at least 8 steps long.
  const int max_segments=args.steps/8 > 1 ? args.steps/8 : 1;
  int segments=(nblocks+args.strips-1)/args.strips;
  segments=segments < 1 ? 1 : (segments > max_segments ? max_segments : segments);
  for (int s=segments; (s <= segments+8) && (s <= max_segments); s++)
    if (((long) args.strips*s) % nblocks == 0)
      {
        segments=s;
        break;
      }
  if ((((long) args.strips*segments) % nblocks != 0) && ((long) args.strips*segments < 4L*nblocks))
    {
      // no even split: at least four items per workgroup bound the imbalance to 25 %
      int s=(int) ((4L*nblocks+args.strips-1)/args.strips);
      segments=s > max_segments ? max_segments : s;
    }
  args.segments=segments;
  args.steps_per_segment=(args.steps+segments-1)/segments;
#ifdef MH_MFMA_TRACE
  args.trace=nullptr;
  const char *trace_path=option("MAGICKHIP_MFMA_TRACE");
  const size_t trace_bytes=8u*48u*8u*sizeof(unsigned long long);
  if (trace_path != nullptr)
    {
      MH_HIP(hipMalloc(reinterpret_cast<void **>(&args.trace),trace_bytes));
      MH_HIP(hipMemsetAsync(args.trace,0,trace_bytes,src.stream));
    }
#else
  args.trace=nullptr;
#endif
  ProfileScope prof(VERTICAL ? "conv_column" : "conv_row",src.stream);
  hipLaunchKernelGGL((conv_mfma_kernel<VERTICAL,NQ,MODE,IO>),dim3((unsigned) nblocks),dim3(256),lds,
    src.stream,args);
  MH_HIP(hipGetLastError());
#ifdef MH_MFMA_TRACE
  if (args.trace != nullptr)
    {
      std::vector<unsigned long long> host(trace_bytes/sizeof(unsigned long long));
      MH_HIP(hipMemcpyAsync(host.data(),args.trace,trace_bytes,hipMemcpyDeviceToHost,src.stream));
      MH_HIP(hipStreamSynchronize(src.stream));
      MH_HIP(hipFree(args.trace));
      std::string path=std::string(trace_path)+(VERTICAL ? ".column" : ".row");
      if (FILE *f=fopen(path.c_str(),"wb"))
        {
          fwrite(host.data(),1,trace_bytes,f);
          fclose(f);
        }
    }
#endif
  return MH_OK;
}

// *handled = false: shape outside this kernel's reach, nothing launched
// io: MFMA_Q16 (src and dst Quantum), MFMA_TO_SUMS (row pass: src Quantum, dst float sums) or
// MFMA_FROM_SUMS (column pass: src float sums, dst Quantum); the geometry is that of src
MhStatus launch_conv1d_mfma(const View &src,const View &dst,bool vertical,const float *taps_device,
  int ntaps,int shift,bool blend,int io,bool *handled,const View *unsharp_original,double gain,
  double threshold,const double *taps64_device,float tap_scale)
{
  *handled=false;
  if (!(tap_scale > 0.0f))
    return MH_OK;
  if ((io == MFMA_TO_SUMS) || (io == MFMA_FROM_SUMS))
    tap_scale=256.0f;                            // (the float sums between the two passes carry this factor)
  if ((io == MFMA_UNSHARP) && ((unsharp_original == nullptr) || !vertical || (src.channels != 4) ||
      (dst.channels != 4) ||
      (src.columns < 2) || (unsharp_original->quantum != MH_QUANTUM_U16)))
    return MH_OK;
  const View &quantum_side=io == MFMA_FROM_SUMS ? dst : src;
  if ((quantum_side.quantum != MH_QUANTUM_U16) || (ntaps < 2))
    return MH_OK;
  if ((io == MFMA_TO_SUMS) || (io == MFMA_FROM_SUMS))
    {
      const View &sums_side=io == MFMA_FROM_SUMS ? src : dst;
      if ((sums_side.quantum != MH_QUANTUM_F32) || (sums_side.channels != 4) ||
          (vertical != (io == MFMA_FROM_SUMS)))
        return MH_OK;
    }
  if ((quantum_side.channels != 4) && ((quantum_side.channels != 3) || blend))
    return MH_OK;
  if ((src.columns >= (1u << 24)) || (src.rows >= (1u << 24)) ||
      ((unsigned long long) src.columns*src.rows >= (1ull << 32)))
    return MH_OK;                                // pixel_index()
  const int mode=blend ? MFMA_BLEND4 : (quantum_side.channels == 4 ? MFMA_PLAIN4 : MFMA_PLAIN3);
  const int nq=(ntaps+31+15)/16;                 // 32 outputs + K-1 halo, in 16-sample chunks
  if (nq > 9)
    return MH_OK;
  ConvMfmaArgs args;
  args.src=static_cast<const uint16_t *>(src.pixels);
  args.dst=static_cast<uint16_t *>(dst.pixels);
  args.columns=(int) src.columns;
  args.rows=(int) src.rows;
  args.ntaps=ntaps;
  args.shift=shift;
  args.taps=taps_device;
  args.taps64=taps64_device;
  args.tap_scale=tap_scale;
  args.two_over_scale=2.0f/tap_scale;
  args.quantum_unit=(float) (2.0/((double) tap_scale*65535.0));
  args.orig=nullptr;
  args.gain=0.0f;
  args.threshold=0;
  if (io == MFMA_UNSHARP)
    {
      args.orig=static_cast<const uint16_t *>(unsharp_original->pixels);
      args.gain=(float) gain;
      const double level=std::ceil(65535.0*threshold);
      args.threshold=level > 131072.0 ? 131072 : (level < 0.0 ? 0 : (int) level);
    }
  *handled=true;
#define MH_NQ(NQV) \
  case NQV: \
    if (io == MFMA_TO_SUMS) \
      return mode == MFMA_BLEND4 ? launch_mfma_typed<false,NQV,MFMA_BLEND4,MFMA_TO_SUMS>(src,args) : \
        (mode == MFMA_PLAIN4 ? launch_mfma_typed<false,NQV,MFMA_PLAIN4,MFMA_TO_SUMS>(src,args) : \
         launch_mfma_typed<false,NQV,MFMA_PLAIN3,MFMA_TO_SUMS>(src,args)); \
    if (io == MFMA_UNSHARP) \
      return mode == MFMA_BLEND4 ? launch_mfma_typed<true,NQV,MFMA_BLEND4,MFMA_UNSHARP>(src,args) : \
        launch_mfma_typed<true,NQV,MFMA_PLAIN4,MFMA_UNSHARP>(src,args); \
    if (io == MFMA_FROM_SUMS) \
      return mode == MFMA_BLEND4 ? launch_mfma_typed<true,NQV,MFMA_BLEND4,MFMA_FROM_SUMS>(src,args) : \
        (mode == MFMA_PLAIN4 ? launch_mfma_typed<true,NQV,MFMA_PLAIN4,MFMA_FROM_SUMS>(src,args) : \
         launch_mfma_typed<true,NQV,MFMA_PLAIN3,MFMA_FROM_SUMS>(src,args)); \
    if (mode == MFMA_BLEND4) \
      return vertical ? launch_mfma_typed<true,NQV,MFMA_BLEND4,MFMA_Q16>(src,args) : \
        launch_mfma_typed<false,NQV,MFMA_BLEND4,MFMA_Q16>(src,args); \
    if (mode == MFMA_PLAIN4) \
      return vertical ? launch_mfma_typed<true,NQV,MFMA_PLAIN4,MFMA_Q16>(src,args) : \
        launch_mfma_typed<false,NQV,MFMA_PLAIN4,MFMA_Q16>(src,args); \
    return vertical ? launch_mfma_typed<true,NQV,MFMA_PLAIN3,MFMA_Q16>(src,args) : \
      launch_mfma_typed<false,NQV,MFMA_PLAIN3,MFMA_Q16>(src,args);
  switch (nq < 3 ? 3 : nq)
  {
    MH_NQ(3) MH_NQ(4) MH_NQ(5) MH_NQ(6) MH_NQ(7) MH_NQ(8) MH_NQ(9)
    default: break;
  }
#undef MH_NQ
  return fail(MH_UNSUPPORTED,"conv1d (matrix cores): kernel length outside the dispatch table");
}

} // namespace mh
