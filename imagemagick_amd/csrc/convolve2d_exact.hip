// Non-separable 2-D ConvolveMorphology whose cells are small integer multiples of one unit — every
// flat shape kernel (Disk, Diamond, Octagon, Square, Rectangle, Plus, Cross, Ring, Peaks ... after
// `convolve:scale`), binomial and hand-written integer kernels — as EXACT INTEGER sums on the i8
// matrix cores: BIT-IDENTICAL to the reference in both precision modes (Q16; RGBA with
// alpha-weighted colour, four plain channels, RGB).
//
// Reference: the reflected-kernel loops of MorphologyPrimitive, MagickCore/morphology.c:2919-2979
// (pixel += alpha*k*p and gamma += alpha*k per non-NaN cell, each operation rounded to fp64), the
// epilogue :3192-3198 (gamma = PerceptibleReciprocal(gamma), ClampToQuantum(gamma*pixel)).  SURVEY
// section 8d names C5's `ConvolveMorphology Disk:15` as the MAC-bound variant of the path.
//
// Arithmetic.  With cells k_i = m_i*u (m_i an integer of at most seven bits, u the unit; the host
// finds both and carries sum|k_i - m_i*u| in the error bound) the REAL value of the reference's
// sum is u * sum m_i*P_i, P = alpha*p (a 32-bit integer, four bytes) or p (two bytes), and
// sum m_i*P_i is an integer: byte plane b of the samples times the cell integers accumulates in an
// i32 tile without rounding (v_mfma_i32_16x16x64_i8; the samples are stored as b-128 because the
// instruction is signed x signed, the constant 128*sum(m) comes back in the epilogue), two or four
// tiles per output combine exactly in fp64.  The level of that real value is the reference's
// unless it lies within the reference's own rounding error (a few 1e-9 level: tie_check.hpp) of a
// rounding boundary; those samples are recomputed by the whole wave in the reference's order.
// A Disk:15 has 709 cells of 1/709: S/709 is never within 7e-4 of n+1/2, nothing is recomputed;
// a kernel with an even cell sum has true ties on one sample in sum(m), all recomputed.
// (The f16 form of convolve2d_mfma.hip splits samples and taps into hi/lo halves: three products
// of 32 slots where this needs two — or four, alpha-weighted — of 64, and is within one level.)
//
// Formulation and walk: those of convolve2d_mfma.hip.  One kernel row is a banded (Toeplitz)
// product, the kh rows accumulate into the same tiles; a workgroup (8 waves) walks down a strip of
// 64 output columns, 32 rows a step, through a ring of 32+kh-1 source rows in LDS whose 32 new rows
// are fetched into registers while the products of the current step run.
//   * wave = 8 output rows x 32 output columns: v_mfma_i32_32x32x32_i8 with the data as the
//     32-entry operand (entry e = 4*row + channel) and the cells as the 32-column Toeplitz operand;
//     the band of 32 outputs and kw cells is NC = 2 (kw <= 33) or 3 (kw <= 65) chunks of 32 slots.
//     D hands a lane the four channels of four pixels, so the division by the alpha sum is
//     lane-local.  (A first version used v_mfma_i32_16x16x64_i8 with four 16-column tiles a wave:
//     17 LDS reads of 16 bytes a lane per 16 products, the LDS as busy as the matrix pipe — 5.9 ms of
//     products for Disk:15 on 16384^2 RGBA where the instructions alone take 4.1.  Here a product is
//     twice the size: 10 reads per 8.)
//   * LDS: byte planes [plane][channel][ring row][128 columns]; a data operand is one
//     ds_read_b128 (16 consecutive columns of one plane, channel, row).  The 16 lanes the LDS
//     serves together hold 4 channels x 4 rows {r, r+3, r+5, r+6} — all residues mod 4 — at the same
//     16-byte block: with a channel stride of 32 (mod 256), a row stride of 128 and the block
//     index XORed with bit 1 of the ring row they land in 16 different 16-byte slots (the ring has
//     a multiple of four rows so that the wrap keeps the residues);
//   * the Toeplitz operand of a lane (output column n, slot block h of chunk c: cells
//     32c+16h+i-n, i = 0..15, of the kernel row) is one of kw+17 windows of 16 bytes of the
//     zero-padded kernel row: a host-built table [kernel row][window][16], one ds_read_b128 each.
// The operands of the next kernel row are read while the products of this one run; the reads are
// issued through asm and waited for by count (see the product loop).
#include "mh_internal.hpp"
#include "device_common.hpp"
#include "mfma_common.hpp"
#include "tie_check.hpp"
#include <vector>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace mh {

struct Conv2DXArgs
{
  const void *src;
  void *dst;
  int columns,rows;
  int kw,kh;                  // kernel size
  int shiftx,shifty;          // output (x,y) reads source (x-shiftx+u, y-shifty+v)
  const signed char *taps;    // [kh+1][kw+17 windows][16]: the Toeplitz operands, the cell integers; a zero row
  const double *values;       // the kernel's cells as the reference walks them (device; NaN = no cell)
  int window_rows;            // 32+kh-1: source rows of a step
  int stage_rows;             // rows of the ring: window_rows rounded up to a multiple of four
  int plane;                  // bytes per (byte plane, channel): stage_rows*128 + 32
  int strips,groups;          // 64-column strips, 32-row steps of a whole strip
  int segments,steps_per_segment,items_per_xcd;   // vertical cuts of a strip: work items = strips*segments
  double unit;                // cells = integers * unit
  int offset;                 // 128*sum(m)*257: the signed-byte constant of a pair of byte planes
  double error[4];            // how far unit*sum can be from the reference's running sums, per channel
  double relative;            // > 0 (no negative cell): ... as a fraction of the sum itself instead
  unsigned long long *recomputed;
  unsigned *not_integral;     // float Quantum: set by whoever meets a sample that is not an integer of 0..65535
};

constexpr int kCXCols=64;     // output columns per strip
constexpr int kCXStride=128;  // bytes per ring row of a plane

typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));

// ds_read_b128 the compiler does not count: the caller waits with s_waitcnt lgkmcnt(n)
template<int OFFSET>
static __device__ __forceinline__ void lds_read128(intx4 &into,unsigned address)
{
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(into) : "v"(address),"n"(OFFSET));
}
template<int COUNT>
static __device__ __forceinline__ void lds_wait()
{
  asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(COUNT < 15 ? COUNT : 15) : "memory");
}

// The Quantum levels of one pixel from its exact integer sums M (alpha-weighted: M[c] = sum m*alpha*p
// for the colour channels, M[3] = sum m*alpha; plain: M[c] = sum m*p): what ClampToQuantum makes of
// unit*M[c], or of M[c]/M[3] (unit and QuantumScale cancel; gamma = PerceptibleReciprocal(sum
// alpha*k) clamps below 1e-12, which the host has ruled out for an alpha sum of one level, and an
// alpha sum of zero — every alpha the kernel sees is zero — gives 0 like 1e12 * 0 does).  Returns
// the channels whose value lies within the reference's rounding error of a rounding boundary.
template<int PX,bool BLEND>
static __device__ __forceinline__ uint32_t integer_sums_to_levels(const double (&M)[4],const Conv2DXArgs &args,
  uint16_t (&out)[PX])
{
  double inverse=0.0;
  if constexpr (BLEND)
    {
      const double d=M[3];
      double r=__builtin_amdgcn_rcp(d);
      double e=__builtin_fma(-d,r,1.0);
      r=__builtin_fma(r,e,r);
      e=__builtin_fma(-d,r,1.0);
      r=__builtin_fma(r,e,r);
      inverse=d > 0.0 ? r : 0.0;
    }
  const bool relative=args.relative > 0.0;
  uint32_t doubtful=0;
#pragma unroll
  for (int c=0; c < PX; c++)
    {
      const bool weighted=BLEND && (c != PX-1);
      const double value=weighted ? M[c]*inverse : args.unit*M[c];
      // cells of one sign: every partial sum of the reference is below its last and its rounding
      // errors scale with the sum itself (a nearly transparent window is as well determined as
      // an opaque one; numerator and denominator each contribute); cells of both signs (plain
      // channels only): an absolute bound.  1e-9: this evaluation's own roundings.
      const double bound=relative ? __builtin_fma(value,weighted ? 2.0*args.relative : args.relative,1.0e-9) :
        args.error[c]+1.0e-9;
      const double shifted=value+0.5;
      const double fraction=__builtin_amdgcn_fract(shifted);
      const double distance=__builtin_fmin(fraction,1.0-fraction);
      // ClampToQuantum (quantum.h:86-97): the only boundaries are the n+1/2 inside the range
      const bool inside=(value > -1.0) && (value < 65536.0);
      const unsigned level=(unsigned) __builtin_fmax(shifted,0.0);   // v_cvt_u32_f64 truncates
      out[c]=(uint16_t) (level < 65535u ? level : 65535u);
      if (inside && !(distance > bound))
        doubtful|=1u << c;
    }
  return doubtful;
}

template<typename Q,int MODE,int NC,int ROWS>
__global__ __launch_bounds__(512)
void conv2d_exact_kernel(Conv2DXArgs args)
{
  // ROWS output rows per step, eight waves: 32 = four row groups x two column tiles, a wave does
  // 8 rows x 32 columns; 64 (plain layouts, when the taller ring fits) = eight row groups, a wave
  // does 8 rows x both tiles — the middle 32-column block of a row serves both tiles and the cells
  // are read once for both: 8 LDS reads per 8 products where the 32-row form of a plain layout
  // needs 12 (its two byte planes give the cell reads half the products to spread over), which
  // kept its LDS 75 % busy: 3.5 ms of products against 2.0 of instructions.  (Also measured:
  // 16-row steps with two workgroups per CU out of step with each other — the same time.)
  constexpr int NT=512;                                    // threads
  constexpr int TILES=ROWS == 64 ? 2 : 1;                  // column tiles per wave
  constexpr int BLOCKS=NC+TILES-1;                         // 32-column data blocks per plane and row
  // Q = float: a float-Quantum frame whose samples are all integers of 0..65535 (what an 8- or 16-bit
  // file decodes to) has the same exact integer sums; the results are floats, the boundaries those
  // of the float rounding (tie_check.hpp).  The staging checks every sample; the first one that is
  // not such an integer raises args.not_integral, every workgroup leaves at its next step and the
  // caller's generic kernel — launched behind this one, returning at once otherwise — does the frame.
  constexpr bool kFloat=sizeof(Q) == 4;
  constexpr bool BLEND=MODE == MFMA_BLEND4;
  constexpr int NP=BLEND ? 4 : 2;                          // byte planes of a sample
  constexpr int PX=MODE == MFMA_PLAIN3 ? 3 : 4;            // u16 per pixel in memory
  constexpr int STAGED=32*(NC+1);                          // columns staged: 64 outputs + 32*(NC-1) of halo
  typedef unsigned __attribute__((aligned(2))) LooseDword;
  typedef typename std::conditional<kFloat,uint4,uint2>::type Raw;      // one pixel as loaded
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Q *src=static_cast<const Q *>(args.src);
  Q *dst=static_cast<Q *>(args.dst);
  const int CH=args.plane;
  unsigned char *stage=smem_raw;                           // [NP][4][R][128]
  unsigned char *taps_lds=stage+NP*4*CH;                   // [kh+1][kw+17][16], the last row zero
  const unsigned lds_base=(unsigned) (size_t) (__attribute__((address_space(3))) unsigned char *) smem_raw;
  const int tid=(int) threadIdx.x,lane=tid & 63;
  const int wave=__builtin_amdgcn_readfirstlane(tid >> 6);
  const int W=args.columns,H=args.rows;
  const int items=args.strips*args.segments;
  const int item=((int) blockIdx.x & 7)*args.items_per_xcd+((int) blockIdx.x >> 3);
  if (item >= items)
    return;
  const int segment=item/args.strips,strip=item-segment*args.strips;
  const int step_begin=segment*args.steps_per_segment;
  const int step_end=step_begin+args.steps_per_segment < args.groups ? step_begin+args.steps_per_segment : args.groups;
  const int x0=kCXCols*strip;
  const int xin0=x0-args.shiftx;
  const int R=args.stage_rows,NEEDED=args.window_rows;
  const int windows=args.kw+17;
  const int flag_at=(args.kh+1)*windows*16;       // one word behind the cell table

  for (int idx=tid; idx < (args.kh+1)*windows; idx+=NT)
    reinterpret_cast<uint4 *>(taps_lds)[idx]=reinterpret_cast<const uint4 *>(args.taps)[idx];
  // ---- source rows, edge-clamped (cache.c:2663-2679), as quads of four pixels: item idx = (row,
  // quad) of a block of rows that starts at image row `first`
  constexpr int QUADS=STAGED/4;
  auto load_quad=[&](int first,int idx,Raw (&raw)[4])
  {
    const int row=idx/QUADS,quad=idx-row*QUADS;
    int y=first+row;
    y=y < 0 ? 0 : (y > H-1 ? H-1 : y);
#pragma unroll
    for (int i=0; i < 4; i++)
      {
        int x=xin0+4*quad+i;
        x=x < 0 ? 0 : (x > W-1 ? W-1 : x);
        const Q *at=src+pixel_index(y,W,x)*PX;
        if constexpr (kFloat)
          {
            if constexpr (MODE == MFMA_PLAIN3)
              raw[i]=make_uint4(__float_as_uint(at[0]),__float_as_uint(at[1]),__float_as_uint(at[2]),0u);
            else
              raw[i]=*reinterpret_cast<const uint4 *>(at);
          }
        else if constexpr (MODE == MFMA_PLAIN3)
          raw[i]=make_uint2(*reinterpret_cast<const LooseDword *>(at),(unsigned) at[2]);
        else
          raw[i]=*reinterpret_cast<const uint2 *>(at);
      }
  };
  // ... as byte planes of alpha*p (alpha itself for the alpha channel) or p, each byte b stored as
  // b-128, written to ring row (ring_first + row) mod R; the 16-byte block of a column is XORed
  // with bit 1 of the ring row
  auto store_quad=[&](int ring_first,int idx,const Raw (&raw)[4])
  {
    const int row=idx/QUADS,quad=idx-row*QUADS;
    int ring_row=ring_first+row;
    ring_row=ring_row >= R ? ring_row-R : ring_row;
    unsigned sample[4][4];                                  // [channel][pixel]
    bool integral=true;
#pragma unroll
    for (int i=0; i < 4; i++)
      {
        unsigned c0,c1,c2,c3;
        if constexpr (kFloat)
          {
            // (v_cvt_u32_f32 truncates and saturates; NaN -> 0; -0.0 passes as 0, which it is
            // to every sum)
            const float f0=__uint_as_float(raw[i].x),f1=__uint_as_float(raw[i].y);
            const float f2=__uint_as_float(raw[i].z),f3=__uint_as_float(raw[i].w);
            c0=(unsigned) f0; c1=(unsigned) f1; c2=(unsigned) f2; c3=(unsigned) f3;
            integral=integral && ((float) c0 == f0) && ((float) c1 == f1) && ((float) c2 == f2) && ((float) c3 == f3) &&
              ((c0 | c1 | c2 | c3) <= 0xffffu);
          }
        else
          {
            c0=raw[i].x & 0xffffu; c1=raw[i].x >> 16; c2=raw[i].y & 0xffffu; c3=raw[i].y >> 16;
          }
        sample[0][i]=BLEND ? __umul24(c0,c3) : c0;
        sample[1][i]=BLEND ? __umul24(c1,c3) : c1;
        sample[2][i]=BLEND ? __umul24(c2,c3) : c2;
        sample[3][i]=c3;
      }
    const int at=ring_row*kCXStride+(((quad >> 2) ^ ((ring_row >> 1) & 1)) << 4)+4*(quad & 3);
#pragma unroll
    for (int c=0; c < 4; c++)
      {
        unsigned p[4];
        byte_planes(sample[c],p);
#pragma unroll
        for (int b=0; b < NP; b++)
          *reinterpret_cast<unsigned *>(stage+(b*4+c)*CH+at)=p[b] ^ 0x80808080u;
      }
    if constexpr (kFloat)
      if (!integral)
        *args.not_integral=1u;
  };
  // the whole window of the first step: every load of a thread's batch is in flight before the
  // first conversion
  {
    constexpr int ITEMS=4;
    const int total=NEEDED*QUADS;
    const int first=ROWS*step_begin-args.shifty;
    for (int i0=tid; i0 < total; i0+=NT*ITEMS)
      {
        Raw raw[ITEMS][4];
#pragma unroll
        for (int k=0; k < ITEMS; k++)
          {
            const int idx=i0+NT*k;
            load_quad(first,idx < total ? idx : total-1,raw[k]);
          }
#pragma unroll
        for (int k=0; k < ITEMS; k++)
          if (i0+NT*k < total)
            store_quad(0,i0+NT*k,raw[k]);
      }
  }
  __syncthreads();

  // ---- wave = row group rg (8 rows), column tile T (32 columns); lane (e, h): entry e = 4*row +
  // channel of the data operand / output column n = e of the cell operand, slot block h
  const int e=lane & 31,h=lane >> 5;
  const int rg=wave % (ROWS/8),T=TILES*(wave/(ROWS/8));   // row group, first column tile
  const int a_row=8*rg+(e >> 2);                 // window row of kernel row 0
  const unsigned a_column=lds_base+(unsigned) ((e & 3)*CH);
  const int a_block=2*T+h;                       // 16-byte block of chunk 0 (chunk c: + 2c)
  // window o = 32c+16h-n of the zero-padded kernel row, o = -16 .. kw (beyond: all zero)
  unsigned cell_at[NC];
#pragma unroll
  for (int c=0; c < NC; c++)
    {
      int o=32*c+16*h-e;
      o=o < -16 ? -16 : (o > args.kw ? args.kw : o);
      cell_at[c]=lds_base+(unsigned) (NP*4*CH+(o+16)*16);
    }
  const int cell_row=windows*16;
  constexpr int NEW_ITEMS=(ROWS*QUADS+NT-1)/NT;
  static_assert(NEW_ITEMS <= 4,"four pixel quads per thread");
  unsigned recomputed=0;
  int origin=0;                                  // ring row of the window's first row
  for (int step=step_begin; step < step_end; step++)
    {
      const int y0=ROWS*step;
      Raw ahead[NEW_ITEMS][4];
      if (step+1 < step_end)
        {
#pragma unroll
          for (int k=0; k < NEW_ITEMS; k++)
            {
              const int idx=tid+NT*k;
              load_quad(y0-args.shifty+NEEDED,idx < ROWS*QUADS ? idx : ROWS*QUADS-1,ahead[k]);
            }
        }
      intx16 acc[NP][TILES];
#pragma unroll
      for (int b=0; b < NP; b++)
#pragma unroll
        for (int t=0; t < TILES; t++)
#pragma unroll
          for (int i=0; i < 16; i++)
            acc[b][t][i]=0;
      // ---- the products.  A pass is four groups of NC products (one byte plane of one kernel row);
      // alpha-weighted: the four planes of one kernel row, plain: the two planes of two kernel
      // rows.  The operands of group g of the NEXT pass are read into the registers group g has just
      // consumed, so every read has three groups of products (200 cycles) to arrive and one set of
      // operand registers does.  The reads are issued through asm and waited for by count: hipcc's
      // own bookkeeping ends every iteration of such a loop with lgkmcnt(0) — the LDS latency in
      // front of every kernel row.  (To the compiler an asm's result exists from the asm on: the
      // loop must stay free of copies and spills of the registers these reads fill — hence two
      // sets of cell registers used in turn instead of a "next" copied to a "current"; check the
      // ISA after changing it: make asm FILE=convolve2d_exact.)
      constexpr int ROWS_PER_PASS=BLEND ? 1 : 2;
      constexpr int CELL_READS=NC*ROWS_PER_PASS;
      intx4 data[4][BLOCKS];
      auto row_address=[&](int v) -> unsigned
      {
        int ring_row=origin+a_row+v;               // < 2R
        ring_row=ring_row >= R ? ring_row-R : ring_row;
        return a_column+(unsigned) (ring_row*kCXStride+((a_block ^ ((ring_row >> 1) & 1)) << 4));
      };
      // (row kh of the table is zero: the odd last row of a plain pass multiplies nothing)
      auto read_cells=[&](int v,intx4 (&into)[NC])
      {
        const unsigned row=(unsigned) ((v < args.kh ? v : args.kh)*cell_row);
#pragma unroll
        for (int c=0; c < NC; c++)
          lds_read128<0>(into[c],cell_at[c]+row);
      };
      auto read_group=[&](int g,const unsigned (&row_at)[ROWS_PER_PASS])
      {
        const unsigned at=BLEND ? row_at[0]+(unsigned) (g*4*CH) : row_at[g >> 1]+(unsigned) ((g & 1)*4*CH);
        lds_read128<0>(data[g][0],at);
        lds_read128<32>(data[g][1],at);
        if constexpr (BLOCKS >= 3)
          lds_read128<64>(data[g][2],at);
        if constexpr (BLOCKS >= 4)
          lds_read128<96>(data[g][3],at);
      };
      auto multiply_group=[&](int g,const intx4 (&cells)[ROWS_PER_PASS][NC])
      {
        // (tile t, chunk c: the data block t+c)
        const int b=BLEND ? g : (g & 1);
#pragma unroll
        for (int c=0; c < NC; c++)
#pragma unroll
          for (int t=0; t < TILES; t++)
            acc[b][t]=__builtin_amdgcn_mfma_i32_32x32x32_i8(data[g][t+c],cells[BLEND ? 0 : (g >> 1)][c],acc[b][t],0,0,0);
      };
      auto addresses=[&](int v,unsigned (&row_at)[ROWS_PER_PASS])
      {
#pragma unroll
        for (int r=0; r < ROWS_PER_PASS; r++)
          row_at[r]=row_address(v+r < args.kh ? v+r : args.kh-1);
      };
      intx4 cells_even[ROWS_PER_PASS][NC],cells_odd[ROWS_PER_PASS][NC];
      {
        unsigned row_at[ROWS_PER_PASS];
        addresses(0,row_at);
#pragma unroll
        for (int r=0; r < ROWS_PER_PASS; r++)
          read_cells(r,cells_even[r]);
#pragma unroll
        for (int g=0; g < 4; g++)
          read_group(g,row_at);
      }
      auto pass=[&](int v,const intx4 (&cells)[ROWS_PER_PASS][NC],intx4 (&next_cells)[ROWS_PER_PASS][NC])
      {
        const int next=v+ROWS_PER_PASS < args.kh ? v+ROWS_PER_PASS : v;    // (the last pass reads its own rows again: no branch)
        unsigned row_at[ROWS_PER_PASS];
        addresses(next,row_at);
#pragma unroll
        for (int g=0; g < 4; g++)
          {
            __builtin_amdgcn_sched_barrier(0);
            // the cells and group g's operands are the oldest reads in flight; younger: the
            // three groups behind it and, after group 0, the next pass's cells
            if (g == 0)
              lds_wait<3*BLOCKS>();
            else
              lds_wait<3*BLOCKS+CELL_READS>();
            __builtin_amdgcn_sched_barrier(0);
            multiply_group(g,cells);
            __builtin_amdgcn_sched_barrier(0);
            if (g == 0)
              {
#pragma unroll
                for (int r=0; r < ROWS_PER_PASS; r++)
                  read_cells(next+r,next_cells[r]);
              }
            read_group(g,row_at);
          }
      };
#ifdef MH_CX_KNOCK
      if ((MH_CX_KNOCK & 1) == 0)
#endif
        {
          // pairs of passes, then the odd one (a loop that leaves between the two copies the
          // tiles from one register set to another every time round)
          const int passes=(args.kh+ROWS_PER_PASS-1)/ROWS_PER_PASS;
          int v=0;
          for (int p=0; p+1 < passes; p+=2)
            {
              pass(v,cells_even,cells_odd);
              pass(v+ROWS_PER_PASS,cells_odd,cells_even);
              v+=2*ROWS_PER_PASS;
            }
          if (passes & 1)
            pass(v,cells_even,cells_odd);
        }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the last pass's reads land in registers that are about to be reused
      __builtin_amdgcn_sched_barrier(0);
      // (hipcc pads the wait states between a v_mfma and the first read of its tile per basic
      // block; the loop's exit is a branch: convolve_fused_exact.hip, settle_tiles)
#pragma unroll
      for (int b=0; b < NP; b++)
#pragma unroll
        for (int t=0; t < TILES; t++)
          asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc[b][t]));
      // The two waves of a SIMD (w and w+4) do their epilogues at different points of the step: w
      // before the step's barriers, w+4 behind them — the vector work of one then runs beside the
      // products of the other instead of both idling the matrix pipe at the same time (the tiles
      // of w+4 stay in their registers across the staging, which needs none of the operands').
      const bool deferred=wave >= 4;
      auto epilogue=[&]()
      {
        // ---- D: register 4q+c of lane (n, h) = channel c of pixel (row 8*rg+2q+h, column 32T+n)
        uint32_t doubtful=0;                         // bit 16t+4q+c: channel c of pixel q of tile t
  #pragma unroll
        for (int t=0; t < TILES; t++)
  #pragma unroll
        for (int q=0; q < 4; q++)
          {
            const int x=x0+32*(T+t)+e;
            const int y=y0+8*rg+2*q+h;
            double M[4];
  #pragma unroll
            for (int c=0; c < 4; c++)
              {
                // |acc| <= 128*sum|m| and 257*128*sum|m| <= 2^30 (the host checks): plane + 256*plane'
                // + the signed-byte constant stays in i32
                const int low=acc[0][t][4*q+c]+256*acc[1][t][4*q+c]+args.offset;
                M[c]=(double) low;
                if constexpr (BLEND)
                  {
                    const int high=acc[2][t][4*q+c]+256*acc[3][t][4*q+c]+args.offset;
                    M[c]=__builtin_fma((double) high,65536.0,M[c]);
                  }
              }
            Q out[PX];
  #if defined(MH_CX_KNOCK) && (MH_CX_KNOCK & 2)
            uint32_t undecided=0;
  #pragma unroll
            for (int c=0; c < PX; c++)
              out[c]=(Q) (acc[0][t][4*q+c]+acc[NP-1][t][4*q+c]);
  #else
            uint32_t undecided;
            if constexpr (kFloat)
              {
                // the float nearest to unit*M or M_c/M_alpha, unless that lies within the reference's
                // rounding error of the midpoint of two floats (settle_sums, tie_check.hpp)
                double sums[4],error[4];
  #pragma unroll
                for (int c=0; c < 4; c++)
                  {
                    sums[c]=args.unit*M[c];
                    error[c]=args.relative > 0.0 ? args.relative*__builtin_fabs(sums[c]) : args.error[c];
                  }
                undecided=settle_sums<float,PX,BLEND>(sums,error,0,out);
              }
            else
              undecided=integer_sums_to_levels<PX,BLEND>(M,args,out);
  #endif
            if ((y < H) && (x < W))
              {
                doubtful|=undecided << (16*t+4*q);
                Q *at=dst+pixel_index(y,W,x)*PX;
                if constexpr (kFloat)
                  {
                    if constexpr (MODE == MFMA_PLAIN3)
                      {
                        at[0]=out[0];
                        at[1]=out[1];
                        at[2]=out[2];
                      }
                    else
                      *reinterpret_cast<float4 *>(at)=make_float4(out[0],out[1],out[2],out[3]);
                  }
                else if constexpr (MODE == MFMA_PLAIN3)
                  {
                    *reinterpret_cast<LooseDword *>(at)=(unsigned) out[0] | ((unsigned) out[1] << 16);
                    at[2]=out[2];
                  }
                else
                  *reinterpret_cast<uint2 *>(at)=make_uint2((unsigned) out[0] | ((unsigned) out[1] << 16),
                    (unsigned) out[2] | ((unsigned) out[3] << 16));
              }
          }
        // ---- the samples the bound could not decide: the whole wave walks the reference's loop for
        // each and the lane that owns it overwrites what it stored above
        unsigned long long pending=__ballot(doubtful != 0u);
        while (pending != 0ull)
          {
            const int who=__builtin_ctzll(pending);
            pending&=pending-1ull;
            uint32_t which=(uint32_t) __builtin_amdgcn_readlane((int) doubtful,who);
            while (which != 0u)
              {
                const int bit=__builtin_ctz(which);
                which&=which-1u;
                const int xx=x0+32*(T+(bit >> 4))+(who & 31);
                const int yy=y0+8*rg+2*((bit >> 2) & 3)+(who >> 5),c=bit & 3;
                const Q settled=conv2d_reference_sample<Q,PX,BLEND>(src,W,H,xx,yy,c,args.values,
                  args.kw,args.kh,args.shiftx,args.shifty,lane);
                if (lane == who)
                  {
                    dst[pixel_index(yy,W,xx)*PX+c]=settled;
                    recomputed++;
                  }
              }
          }
      };
      if (!deferred || (step+1 >= step_end))
        epilogue();
      if (step+1 < step_end)
        {
          if constexpr (kFloat)
            if (tid == 0)
              *reinterpret_cast<volatile unsigned *>(taps_lds+flag_at)=
                __hip_atomic_load(args.not_integral,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
          __syncthreads();                           // every read of the 32 oldest rows is done
          if constexpr (kFloat)
            if (*reinterpret_cast<volatile unsigned *>(taps_lds+flag_at) != 0u)
              break;                                 // (the same word for every thread: all leave)
          // the new rows follow the window: ring rows origin+NEEDED .. +31 (mod R) — the slack of
          // the rounded-up ring and the oldest rows
          int first=origin+NEEDED;
          first=first >= R ? first-R : first;
#if defined(MH_CX_KNOCK) && (MH_CX_KNOCK & 4)
          if (args.kh == 1000)
#endif
#pragma unroll
          for (int k=0; k < NEW_ITEMS; k++)
            if (tid+NT*k < ROWS*QUADS)
              store_quad(first,tid+NT*k,ahead[k]);
          origin=origin+ROWS >= R ? origin+ROWS-R : origin+ROWS;
          __syncthreads();
          if (deferred)
            epilogue();
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

static unsigned long long *g_conv2d_recomputed[64]={};
static bool g_conv2d_count=false;

template<typename Q,int MODE,int NC,int ROWS>
static MhStatus launch_conv2d_exact_typed(const View &src,Conv2DXArgs &args,size_t lds)
{
  MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv2d_exact_kernel<Q,MODE,NC,ROWS>),
    hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  ProfileScope prof("conv2d_exact",src.stream);
  hipLaunchKernelGGL((conv2d_exact_kernel<Q,MODE,NC,ROWS>),dim3((unsigned) (8*args.items_per_xcd)),dim3(512),lds,
    src.stream,args);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// cells = integers * unit?  The unit is the smallest cell, or that over 2..8 (cells 2s and 3s).
static bool integer_cells(const MhKernelInfo *kernel,std::vector<int> &m,double *unit)
{
  const size_t n=kernel->width*kernel->height;
  double smallest=0.0;
  for (size_t i=0; i < n; i++)
    {
      const double cell=kernel->values[i];
      if (std::isnan(cell) || (cell == 0.0))
        continue;
      if (!std::isfinite(cell))
        return false;
      if ((smallest == 0.0) || (std::fabs(cell) < smallest))
        smallest=std::fabs(cell);
    }
  if (!(smallest > 0.0))
    return false;
  m.assign(n,0);
  for (int divisor=1; divisor <= 8; divisor++)
    {
      const double u=smallest/(double) divisor;
      bool fits=true;
      for (size_t i=0; (i < n) && fits; i++)
        {
          const double cell=kernel->values[i];
          if (std::isnan(cell))
            {
              m[i]=0;
              continue;
            }
          const double q=cell/u,nearest=std::nearbyint(q);
          fits=(std::fabs(nearest) <= 127.0) && (std::fabs(q-nearest) <= 1.0e-9);
          m[i]=fits ? (int) nearest : 0;
        }
      if (fits)
        {
          *unit=u;
          return true;
        }
    }
  return false;
}

// w x h Convolve of an RGBA (alpha-weighted colour, alpha last), four-plain-channel or RGB frame
// with integer-multiple cells.  *handled stays false (nothing launched) when the kernel or the
// frame does not qualify.  Float Quantum: `flag` receives a device word that the kernel raises
// when the frame is not made of integers of 0..65535 — the caller then has to run the generic
// kernel behind this one, conditional on that word (Morph2DParams::only_if).
MhStatus launch_conv2d_exact(const View &src,const View &dst,const MhKernelInfo *kernel,bool blend,
  bool *handled,Temp *flag)
{
  *handled=false;
  const bool is_float=src.quantum != MH_QUANTUM_U16;
  if ((src.quantum != dst.quantum) || (is_float && ((src.quantum != MH_QUANTUM_F32) || (flag == nullptr))) ||
      ((src.channels != 4) && ((src.channels != 3) || blend)) ||
      (dst.channels != src.channels) || (src.columns != dst.columns) || (src.rows != dst.rows) ||
      (src.pixels == dst.pixels))
    return MH_OK;
  if ((src.columns >= (1u << 24)) || (src.rows >= (1u << 24)) ||
      ((unsigned long long) src.columns*src.rows >= (1ull << 32)))
    return MH_OK;                                // pixel_index()
  const int kw=(int) kernel->width,kh=(int) kernel->height;
  // below 5 x 5 the generic kernel's 25 taps are cheaper than a staged window
  if ((kw < 2) || (kh < 2) || (kw*kh < 25) || (kw > 65) || (kh > 128) ||
      (kernel->x < 0) || (kernel->y < 0) || (kernel->x >= kw) || (kernel->y >= kh))
    return MH_OK;
  std::vector<int> m;
  double unit=0.0;
  if (!integer_cells(kernel,m,&unit))
    return MH_OK;
  // the error bound of convolve_separable.hip with the integer form standing in for the outer
  // product: what m*unit misses of each cell; the reference rounds every term twice (three times
  // when alpha-weighted) and every addition once, to half an ulp of its running sum
  double residual=0.0,magnitude=0.0,running=0.0,partials=0.0,worst_ratio=0.0;
  int cells=0;
  long long total=0,absolute=0;
  bool negative=false,positive=false;
  for (int i=kw*kh-1; i >= 0; i--)
    {
      const double cell=kernel->values[i];
      if (std::isnan(cell))
        continue;
      total+=m[(size_t) i];
      absolute+=std::abs(m[(size_t) i]);
      negative=negative || (m[(size_t) i] < 0);
      positive=positive || (m[(size_t) i] > 0);
      residual+=std::fabs(cell-(double) m[(size_t) i]*unit)+std::fabs(cell)*4.440892098500626e-16;
      if (cell != 0.0)
        worst_ratio=std::fmax(worst_ratio,std::fabs(cell-(double) m[(size_t) i]*unit)/std::fabs(cell)+
          4.440892098500626e-16);
      cells++;
      magnitude+=std::fabs(cell);
      running+=std::fabs(cell);
      partials+=running;
    }
  // alpha-weighted sums with cells of both signs stay on the fp64 kernels (cancellation); an i32
  // tile holds 128*sum|m|, the epilogue adds 256 times another and the signed-byte constant
  if ((blend && negative) || (absolute == 0) || (absolute > 32000))
    return MH_OK;
  // gamma = PerceptibleReciprocal(sum alpha*k): an alpha sum of one level must be above its clamp
  if (blend && !(unit/65535.0 >= 1.000001e-12))
    return MH_OK;
  const double ulp=1.1102230246251565e-16;
  const double error_unit=2.0*(residual+ulp*(partials+12.0*magnitude));
  if (error_unit*65535.0*65535.0 > 65535.0*1.0e-3)
    return MH_OK;
  const int planes=blend ? 4 : 2;
  const int nc=(32+kw-1+31)/32;                  // 32 outputs + kw-1 of halo, in 32-slot chunks
  const int windows=kw+17;
  // plain layouts: 64-row steps (a wave does both column tiles) when the taller ring fits the LDS
  // (two chunks only: with three, the operands of two tiles do not fit beside the prefetched rows)
  int step_rows=(blend || (nc != 2) || (option("MAGICKHIP_CONV2D_ROWS32") != nullptr)) ? 32 : 64;
  int window_rows=0,ring_rows=0,plane_bytes=0;
  size_t lds=0;
  for (;;)
    {
      window_rows=step_rows+kh-1;
      ring_rows=(window_rows+3) & ~3;
      // channel stride = 32 mod 256, a multiple of four rows: conv2d_exact_kernel's operand reads
      plane_bytes=ring_rows*kCXStride+32;
      lds=(size_t) planes*4*plane_bytes+(size_t) (kh+1)*windows*16+16;
      if ((step_rows == 32) || (lds <= 160u*1024u))
        break;
      step_rows=32;
    }
  if ((nc < 2) || (nc > 3) || (lds > 160u*1024u))
    return MH_OK;
  // the reflected walk of morphology.c:2925: cell (v,u) of the window carries values[(kh-1-v)*kw+
  // (kw-1-u)]; window o = w-16 of kernel row v: cells u = o .. o+15, zero outside the row
  std::vector<signed char> table((size_t) (kh+1)*windows*16,0);         // (+ a row of zeros)
  for (int v=0; v < kh; v++)
    for (int w=0; w < windows; w++)
      for (int i=0; i < 16; i++)
        {
          const int u=w-16+i;
          if ((u >= 0) && (u < kw))
            table[((size_t) v*windows+(size_t) w)*16+(size_t) i]=
              (signed char) m[(size_t) (kh-1-v)*kw+(size_t) (kw-1-u)];
        }
  TableBundle tables;
  const size_t t_cells=tables.add(table.data(),table.size());
  const size_t t_values=tables.add(kernel->values,(size_t) kw*kh*sizeof(double));
  MH_TRY(tables.upload(src.device,src.stream));
  Conv2DXArgs args;
  args.src=src.pixels;
  args.dst=dst.pixels;
  args.not_integral=nullptr;
  if (is_float)
    {
      MH_TRY(flag->alloc(src.device,sizeof(unsigned),src.stream));
      MH_HIP(hipMemsetAsync(flag->ptr,0,sizeof(unsigned),src.stream));
      args.not_integral=flag->as<unsigned>();
    }
  args.columns=(int) src.columns;
  args.rows=(int) src.rows;
  args.kw=kw;
  args.kh=kh;
  args.shiftx=kw-1-(int) kernel->x;
  args.shifty=kh-1-(int) kernel->y;
  args.taps=tables.at<signed char>(t_cells);
  args.values=tables.at<double>(t_values);
  args.window_rows=window_rows;
  args.stage_rows=ring_rows;
  args.plane=plane_bytes;
  args.unit=unit;
  args.offset=(int) (128*257*total);
  for (int c=0; c < 4; c++)
    args.error[c]=error_unit*((blend && (c != 3)) ? 65535.0*65535.0 : 65535.0);
  // one sign: |reference - real| <= (cells + 3 roundings per term) * ulp * sum, |unit*M - real| <=
  // worst_ratio * sum
  args.relative=(negative && positive) ? 0.0 : 2.0*(worst_ratio+ulp*((double) cells+12.0));
  args.recomputed=nullptr;
  if (g_conv2d_count && (src.device >= 0) && (src.device < 64))
    {
      if (g_conv2d_recomputed[src.device] == nullptr)
        {
          MH_HIP(hipMalloc(reinterpret_cast<void **>(&g_conv2d_recomputed[src.device]),sizeof(unsigned long long)));
          MH_HIP(hipMemsetAsync(g_conv2d_recomputed[src.device],0,sizeof(unsigned long long),src.stream));
        }
      args.recomputed=g_conv2d_recomputed[src.device];
    }
  args.strips=(args.columns+kCXCols-1)/kCXCols;
  args.groups=(args.rows+step_rows-1)/step_rows;
  {
    // cuts of a strip: the schedule (one workgroup per CU) that finishes first; a cut costs the
    // staging of its first window, about a step's worth
    const int cus=compute_units(src.device);
    int best=1;
    double best_cost=1.0e300;
    for (int cuts=1; cuts <= args.groups; cuts++)
      {
        const int steps=(args.groups+cuts-1)/cuts;
        const int rounds=(args.strips*((args.groups+steps-1)/steps)+cus-1)/cus;
        const double cost=(double) rounds*((double) steps+1.0);
        if (cost < best_cost-1.0e-9)
          {
            best_cost=cost;
            best=cuts;
          }
      }
    if (const char *e=option("MAGICKHIP_CONV2D_CUTS"))          // tests: long walks on small frames
      best=(atoi(e) >= 1) && (atoi(e) <= args.groups) ? atoi(e) : best;
    args.steps_per_segment=(args.groups+best-1)/best;
    args.segments=(args.groups+args.steps_per_segment-1)/args.steps_per_segment;
  }
  args.items_per_xcd=(args.strips*args.segments+7)/8;
  *handled=true;
#define MH_LAUNCH_ROWS(MODE,ROWS) \
  return is_float ? \
    (nc == 2 ? launch_conv2d_exact_typed<float,MODE,2,ROWS>(src,args,lds) : launch_conv2d_exact_typed<float,MODE,3,ROWS>(src,args,lds)) : \
    (nc == 2 ? launch_conv2d_exact_typed<uint16_t,MODE,2,ROWS>(src,args,lds) : launch_conv2d_exact_typed<uint16_t,MODE,3,ROWS>(src,args,lds))
#define MH_LAUNCH(MODE) \
  if (step_rows == 64) \
    return is_float ? launch_conv2d_exact_typed<float,MODE,2,64>(src,args,lds) : \
      launch_conv2d_exact_typed<uint16_t,MODE,2,64>(src,args,lds); \
  MH_LAUNCH_ROWS(MODE,32)
  if (src.channels == 3)
    {
      MH_LAUNCH(MFMA_PLAIN3);
    }
  if (blend)
    MH_LAUNCH_ROWS(MFMA_BLEND4,32);
  MH_LAUNCH(MFMA_PLAIN4);
#undef MH_LAUNCH_ROWS
#undef MH_LAUNCH
}

} // namespace mh

using namespace mh;

// Host logic of the path, for tests and callers that want to know in advance: are the kernel's cells
// integer multiples (|m| <= 127) of one unit?  cells[] (width*height ints, NaN cells as 0; may be
// NULL) and *unit (may be NULL) receive the form; returns 1 or 0.
extern "C" MH_API int MhKernelIntegerCells(const MhKernelInfo *kernel,int *cells,double *unit)
{
  if ((kernel == nullptr) || (kernel->values == nullptr) || (kernel->width == 0) || (kernel->height == 0))
    return 0;
  std::vector<int> m;
  double u=0.0;
  if (!integer_cells(kernel,m,&u))
    return 0;
  if (cells != nullptr)
    std::memcpy(cells,m.data(),m.size()*sizeof(int));
  if (unit != nullptr)
    *unit=u;
  return 1;
}

// Diagnostic: samples the integer 2-D convolve recomputed in the reference's order since the
// counter was last read (enable = 1 switches the counting on and reads, 0 reads and switches off).
extern "C" MH_API unsigned long long MhConvolve2DRecomputed(int enable)
{
  unsigned long long total=0;
  int current=-1;
  (void) hipGetDevice(&current);
  for (int d=0; d < 64; d++)
    if (g_conv2d_recomputed[d] != nullptr)
      {
        unsigned long long value=0;
        if (hipSetDevice(d) == hipSuccess)
          {
            (void) hipDeviceSynchronize();
            (void) hipMemcpy(&value,g_conv2d_recomputed[d],sizeof(value),hipMemcpyDeviceToHost);
            (void) hipMemset(g_conv2d_recomputed[d],0,sizeof(value));
          }
        total+=value;
      }
  if (current >= 0)
    (void) hipSetDevice(current);
  g_conv2d_count=enable != 0;
  return total;
}
