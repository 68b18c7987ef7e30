// ResizeImage (enlargement: VerticalFilter then HorizontalFilter, MagickCore/resize.c:3846-3861)
// in ONE launch on the fp64 VECTOR pipe — the FAST (+-1 ULP / +-1 level) form of the four-channel
// frame enlarged by a whole-number horizontal factor f (2, 3, 4) and a vertical factor >= f (the
// order of the two filters, resize.c:3846).
//
// The two-pass form writes and re-reads the Quantum-typed intermediate (8.6 GB of 26.8 GB for
// 8192^2 -> 32768^2 float RGBA) and its horizontal pass reads every tap of every output out of LDS
// (7 x 32 bytes per output pixel) with per-lane weights.  Measured on MI355X (profiles/r5_notes):
// v_mfma_f64_16x16x4_f64 sustains one instruction per ~100 cycles a SIMD (48 TFLOP/s) and shares
// the fp64 unit with v_fma_f64 (72 TFLOP/s) — the banded matrix form of resize_mfma.hip spends
// 2.5x the fp64 time of plain multiply-adds.  So: plain multiply-adds, but fed from REGISTERS and
// SCALAR registers instead of LDS:
//
//   wave  = 64 adjacent source columns (lane = column) walking down `rows_per_chunk` output rows;
//           no workgroup barrier anywhere: the four waves of a workgroup are independent
//   vertical    the lane keeps the 8 source rows under the filter window as alpha-premultiplied
//               doubles in registers (a new row is loaded, converted and premultiplied once);
//               every lane works on the same output row, so the weights are wave-uniform: scalar
//               loads, scalar-register operands of v_fma_f64
//   finish      HorizontalFilter's input is the Quantum-ROUNDED intermediate (the reference stores
//               filter_image): gamma, ClampToQuantum, premultiply again
//   exchange    the lane's intermediate pixel goes to the wave's own row in LDS (32 bytes), the
//               nt neighbours come back: nt x 32 bytes per SOURCE pixel = f outputs, where the
//               two-pass kernel reads nt x 32 bytes per output
//   horizontal  with dst_columns = f * src_columns the f outputs of a source column use the same
//               f x nt weights in every column (resize_stream_plan.hpp): scalar registers.  The few
//               columns under a window clipped by the image edge take their listed weights.
//   store       f pixels a lane -> transposed through the wave's LDS so that a store instruction
//               writes 64 consecutive pixels
// Semantics are resize.c:3494-3530 / :3709-3745 with the derived gamma of the Fma64 policy
// (resize_acc.hpp): colour = sum(w*alpha*p) / sum(w*alpha), alpha = sum(w*alpha).
// What the fused sums cannot vouch for is recomputed: a second (normally empty) launch redoes the
// flagged items (a wave's strip x chunk of rows) in the reference's own operation order and
// windows (resize_stream_careful_kernel).  A wave flags its item when
//   * an intermediate value lies on a rounding boundary (the reference's last bits decide the
//     level, and one level of a small intermediate alpha is thousands of levels of the colours the
//     second filter weights with it: Triangle / Box / Catrom over small integers produce exact
//     x.5 sums in families), or an alpha sum of either filter is small enough for the quotient to
//     amplify the sums' last bits (resize_acc.hpp);
//   * a float frame's sample is not finite or huge: zero-weight padding multiplies samples outside
//     an output's window by 0, which is exact only for finite samples.
#include "mh_internal.hpp"
#include "resize_filter.hpp"
#include "resize_stream_plan.hpp"
#include "device_common.hpp"
#include "resize_acc.hpp"
#include <cstdio>
#include <vector>
#include <memory>
#include <mutex>
#include <type_traits>

namespace mh {

struct StreamResizeArgs
{
  const void *src;
  void *dst;
  int src_columns,src_rows,dst_columns,dst_rows;
  int strips,chunks,rows_per_chunk;
  int lo,edge_left,edge_right;
  const int *strip_first,*strip_count;       // [strips]: the source columns a wave finishes
  const double *strip_hw;                    // [strips][kMaxDense]: their [f][nt] weights
  const int *vbase;
  const double *vdense;
  const int *vstart,*vcount,*hstart,*hcount;
  const double *vweight,*hweight;            // [tap][out]
  // [strips*chunks][2]: bit b of the first word = the item's rows b << mark_shift .. are redone; the
  // second word = the first and the last lane (source column) that asked for it, in bytes 0 and 1
  unsigned *wild_items;
  int mark_shift;
  int nt;                                    // neighbours a lane reads (the careful launch: how far a lane's value reaches)
  const double *listed;                      // [2*kListed][kMaxDense]: the listed columns' dense weights
};

typedef double double2_t __attribute__((ext_vector_type(2)));
// wave-uniform tables are read through the constant address space: scalar loads (no vector-memory
// counter, which the pixel stores keep busy), scalar-register operands
typedef const double __attribute__((address_space(4))) *scalar_doubles;
typedef const int __attribute__((address_space(4))) *scalar_ints;

// Inf, NaN or a magnitude above 2^20 (see resize_mfma.hip): with every sample of the window at
// most 2^20 the intermediate stays finite and a finite value times a zero weight is an exact zero
static __device__ __forceinline__ bool wild_f32(float v)
{
  return !(__builtin_fabsf(v) <= 1048576.0f);  // one v_cmp_nle_f32 |v|: true for a NaN too
}

template<typename Q,bool BLEND>
static __device__ __forceinline__ void finish_sums(const double (&s)[4],Q (&q)[4])
{
  ResizeAcc<Q,4,BLEND,Fma64> f;
  f.s[0]=s[0]; f.s[1]=s[1]; f.s[2]=s[2]; f.s[3]=s[3];
  f.g=0.0;
  Q copy[4]={(Q) 0,(Q) 0,(Q) 0,(Q) 0};
  f.finish(copy,0u,q);
}

// The same with the reciprocal refined NEWTON times (v_rcp_f64 is good to ~23 bits: one step gives
// 46, enough for a result that is rounded to a float or a Q16 level once; the intermediate, whose
// rounding the horizontal sums amplify, keeps the two steps of resize_acc.hpp) and the rare case —
// PerceptibleReciprocal's clamp acts, or the alpha sum is not a number — sent down resize_acc.hpp's
// own arithmetic in a branch instead of through selects in every lane.  `doubt` collects the lanes
// whose value the fused sums cannot vouch for (a lane mask in scalar registers: no vector register
// across the walk): TIES (the intermediate) a value too close to a rounding boundary (resize_acc.hpp,
// TieWatch); the outputs: an alpha sum so small that the quotient's error is no longer negligible.
//
// Where PerceptibleReciprocal's clamp acts in the LAST filter of a Q16 frame the report can be withdrawn when the
// sums' own terms say that no order of summation matters (`magnitudes(a)`: a[c] = sum |weight * sample| of the
// window, computed only inside this rare branch): the result is +-s_c * QuantumScale / MagickEpsilon, so two orders
// differ by 1.5e7 * 6e-15 * a_c = 9e-8 * a_c level — nothing against the +-1 of the last filter while a_c < 1e4 —
// and the SIGN is the alpha sum's, certain while |s_a| is far above its own rounding (6e-15 * a_a).  The case this
// is for: a 3x enlargement has one output in three whose window is (~0 .. ~0, 1, ~0 .. ~0); over an intermediate
// pixel whose alpha the first filter's negative lobes clamped to 0 — one in 77 on a frame of random alpha — the
// alpha sum is 1e-8 out of terms of 1e-8 and the report sent three quarters of the frame down the careful launch
// (12 ms against the walk's 3.2 on 8192^2).  The window it is NOT for: a transparent pixel between two equal
// opaque ones, where the terms w and -w cancel to 0 or 1e-29 and the sign is the order's: |s_a| << a_a, reported.
struct NoMagnitudes { __device__ __forceinline__ bool operator()(double (&)[4]) const { return false; } };

template<typename Q,bool BLEND,int NEWTON,bool TIES,class Magnitudes=NoMagnitudes>
static __device__ __forceinline__ void finish_fast(const double (&s)[4],Q (&q)[4],unsigned long long &doubt,
  const Magnitudes &magnitudes=Magnitudes())
{
  // (ONE ballot, behind the branch, where every lane of the wave is present again: a ballot inside a
  // divergent branch reaches only the lanes that took it, and the wave's mask is read from one lane)
  bool flagged=false;
  bool clamped=false;
  if constexpr (BLEND)
    clamped=!((__builtin_fabs(s[3])*kQS) >= kEps);
  if (__builtin_expect(clamped,0))
    {
      finish_sums<Q,BLEND>(s,q);
      flagged=clamped_sums_count(s);
      if constexpr (!TIES && !QuantumOps<Q>::is_float)
        if (flagged)
          {
            double a[4];
            if (magnitudes(a))
              flagged=!((a[0] < 1.0e4) && (a[1] < 1.0e4) && (a[2] < 1.0e4) && (__builtin_fabs(s[3]) >= 1.0e-9*a[3]));
          }
    }
  else
    {
      double r=1.0;
      TieWatch<Q> plain,colour;
      plain.plain();
      colour.plain();
      if constexpr (BLEND)
        {
          const double sa=s[3];
          r=__builtin_amdgcn_rcp(sa);
#pragma unroll
          for (int i=0; i < NEWTON; i++)
            {
              const double e=__builtin_fma(-sa,r,1.0);
              r=__builtin_fma(r,e,r);
            }
          if constexpr (TIES)
            colour.quotient(r);
          else
            flagged=__builtin_fabs(sa) < OutputAlphaLimit<Q>::value;
        }
      // (a value, its level, its verdict: one channel after the other — the kernels sit at their register limit)
#pragma unroll
      for (int c=0; c < 4; c++)
        {
          const double v=BLEND && (c < 3) ? s[c]*r : s[c];
          q[c]=QuantumOps<Q>::clamp(v);
          if constexpr (TIES)
            flagged=flagged || (BLEND && (c < 3) ? colour.near(v) : plain.near(v));
        }
    }
  doubt|=__builtin_amdgcn_ballot_w64(flagged);
}

// a pixel as the filters sum it: (alpha*p .., alpha), or the four plain channels
template<typename Q,bool BLEND>
static __device__ __forceinline__ void premultiplied(const Q (&q)[4],double (&v)[4])
{
  if constexpr (BLEND)
    {
      v[3]=(double) q[3];
      v[0]=v[3]*(double) q[0];               // exact: two 24-bit significands (two 16-bit levels)
      v[1]=v[3]*(double) q[1];
      v[2]=v[3]*(double) q[2];
    }
  else
    {
      v[0]=(double) q[0]; v[1]=(double) q[1]; v[2]=(double) q[2]; v[3]=(double) q[3];
    }
}

// The items (strip x chunk of rows) resize_stream_kernel could not vouch for — a sample that is not
// tame (float frames), an intermediate value on a rounding boundary, a small alpha sum — again, in
// the reference's own operation order and windows (resize_acc.hpp: resize_redo_rect).  One
// workgroup per item; it leaves at once when the item's flag is clear.
template<typename Q,bool BLEND>
__global__ __launch_bounds__(256,3)
void resize_stream_careful_kernel(StreamResizeArgs a,int f)
{
  __shared__ __attribute__((aligned(16))) unsigned char scratch[12288];   // (a few rows of an item's intermediate: many workgroups a CU)
  const int item=(int) blockIdx.x;
  const unsigned marks=a.wild_items[2*item];
  if (marks == 0u)
    return;
  const int chunk=item/a.strips,strip=item-chunk*a.strips;
  int x0=f*a.strip_first[strip],x1=x0+f*a.strip_count[strip];
  {
    // the intermediate pixel of lane l (source column strip_first+lo+l) is read by the lanes
    // l-lo-(nt-1) .. l-lo: only their outputs can differ
    const unsigned lanes=a.wild_items[2*item+1];
    const int column0=a.strip_first[strip]+a.lo;
    const int from=column0+(int) (lanes & 0xffu)-a.lo-(a.nt-1),to=column0+(int) ((lanes >> 8) & 0xffu)-a.lo+1;
    x0=f*from > x0 ? f*from : x0;
    x1=f*to < x1 ? f*to : x1;
    if (x0 >= x1)
      return;
  }
  const int y0=chunk*a.rows_per_chunk;
  int y1=y0+a.rows_per_chunk;
  y1=y1 < a.dst_rows ? y1 : a.dst_rows;
  RedoTables t;
  t.vstart=a.vstart; t.vcount=a.vcount; t.hstart=a.hstart; t.hcount=a.hcount;
  t.vweight=a.vweight; t.hweight=a.hweight;
  t.src_columns=a.src_columns; t.dst_columns=a.dst_columns; t.dst_rows=a.dst_rows;
  for (int b=0; b < 32; b++)
    if (((marks >> b) & 1u) != 0u)
      {
        const int ya=y0+(b << a.mark_shift);
        int yb=ya+(1 << a.mark_shift);
        yb=yb < y1 ? yb : y1;
        if (ya < yb)
          resize_redo_rect<Q,BLEND,StreamResizePlan::kRows>(t,static_cast<const Q *>(a.src),static_cast<Q *>(a.dst),x0,x1,ya,yb,scratch,(int) sizeof(scratch));
      }
}

template<typename Q,bool BLEND,int F,int NT,int ROWS>
// three waves a SIMD (at most 168 registers); two where that spills into the walk (a reload waits for
// every store in flight): the eight-row window, and the seven-neighbour 3x enlargement of a float frame
#ifdef MH_STREAM_WAVES2
__global__ __launch_bounds__(256,2)
#else
__global__ __launch_bounds__(256,((ROWS == 8) || ((F == 3) && (NT == 7) && (sizeof(Q) == 4))) ? 2 : 3)
#endif
void resize_stream_kernel(StreamResizeArgs a)
{
  constexpr bool kFloat=QuantumOps<Q>::is_float;
  static_assert(ROWS <= StreamResizePlan::kRows,"the dense weights of a row");
  // output p of a column reads NT-1 of the NT neighbours: the first NT-1 in the left half of the
  // column, the last NT-1 in the right half (checked by the plan: the weights outside are zeros)
  auto phase_first=[](int p) constexpr -> int { return StreamResizePlan::phase_first(F,p); };
  constexpr int PAD=8;                         // slots either side of the wave's 64 (edge lanes read them)
  constexpr int SLOTS=64+2*PAD;
  constexpr int PX=(int) sizeof(Q)*4;           // bytes of a pixel
  // a phase plane of the transposition buffer, padded by 64 bytes: the 16 lanes of a read group
  // (four phases x four lanes) then fall into 16 different 16-byte (float) / 8-byte (Q16) bank slots
  constexpr int XPLANE=64*PX+64;
  // the listed columns' own weights, dense like hw[]: [2*kListed][F*NT]
  constexpr int LISTED_BYTES=2*StreamResizePlan::kListed*F*NT*8;
  constexpr int WAVE_BYTES=2*SLOTS*16+F*XPLANE+LISTED_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4*WAVE_BYTES];

  const int lane=(int) threadIdx.x & 63;
  const int wave=__builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
  const int item=(int) blockIdx.x*4+wave;
  if (item >= a.strips*a.chunks)
    return;
  const int chunk=item/a.strips,strip=item-chunk*a.strips;
  const int W=a.src_columns,H=a.src_rows,OW=a.dst_columns;
  const int lo=a.lo;
  const int first=((scalar_ints) a.strip_first)[strip],count=((scalar_ints) a.strip_count)[strip];
  const int c0=first+lo;                       // source column of lane 0
  const scalar_doubles hw=(scalar_doubles) (a.strip_hw+(size_t) strip*StreamResizePlan::kMaxDense);
  const int c=c0+lane;
  const int cc=c < 0 ? 0 : (c > W-1 ? W-1 : c);
  const Q *src=static_cast<const Q *>(a.src)+(size_t) cc*4;
  unsigned char *mine=smem+wave*WAVE_BYTES;
  double2_t *row01=reinterpret_cast<double2_t *>(mine)+PAD+lane;          // (c0, c1) of the intermediate
  double2_t *row23=row01+SLOTS;                                           // (c2, alpha)
  unsigned char *xpose=mine+2*SLOTS*16;
  // the columns near the image edges whose windows are clipped: listed weights (wave-uniform test)
  const bool edge_wave=(first < a.edge_left) || (first+count > a.edge_right);
  const bool listed=(c >= 0) && (c < W) && ((c < a.edge_left) || (c >= a.edge_right));
  const double *mine_listed=reinterpret_cast<const double *>(xpose+F*XPLANE);
  if (edge_wave)
    {
      if (listed)
        {
          const int entry=c < a.edge_left ? c : StreamResizePlan::kListed+(c-a.edge_right);
          mine_listed+=entry*(F*NT);
          double *to=const_cast<double *>(mine_listed);
          const double *from=a.listed+(size_t) entry*StreamResizePlan::kMaxDense;
#pragma unroll
          for (int i=0; i < F*NT; i++)
            to[i]=from[i];
        }
      asm volatile("" ::: "memory");
    }
  // where the wave's transposed pixels go: store j writes pixels F*c0+64*j .. +63 of the row; the
  // lanes outside the wave's columns (and the image) get an offset beyond the row's buffer
  // descriptor — the store is dropped by the range check, and no branch surrounds it (the compiler
  // can count the stores between a load and its use: vmcnt(F), not vmcnt(0))
  // (F = 2, 4: store j's lane reads plane lane%F of lane lane/F+64*j/F — one address and one
  // offset with immediate steps; F = 3: one of each per store)
  constexpr bool kSteps=(64 % F) == 0;
  constexpr int NKEPT=kSteps ? 1 : F;
  unsigned store_offset[NKEPT];
  const unsigned char *xfrom[NKEPT];
#pragma unroll
  for (int j=0; j < NKEPT; j++)
    {
      const int idx=64*j+lane;                 // the wave's pixel F*c0+idx sits in plane idx%F of lane idx/F
      const int from=idx/F,phase=idx-from*F;
      store_offset[j]=(unsigned) (F*c0+idx)*(unsigned) PX;
      xfrom[j]=xpose+phase*XPLANE+from*PX;
    }

  double win[ROWS][4];
  unsigned wild=0u;                            // bit j: window slot j holds a sample that is not tame
  unsigned seen=0u;
  unsigned marks=0u;                           // blocks of rows in which a lane met a value the fused sums cannot vouch for (finish_fast)
  unsigned lane_first=64u,lane_last=0u;        // ... and the first and the last such lane (scalar registers)
  auto fetch=[&](int row,Q (&q)[4])
  {
    row=row < H-1 ? row : H-1;
    load_pixel<Q,4>(src+(size_t) row*(size_t) W*4,q);
  };
  auto is_wild=[&](const Q (&q)[4]) -> unsigned
  {
    if constexpr (kFloat)
      {
        const bool w=wild_f32(q[0]) || wild_f32(q[1]) || wild_f32(q[2]) || wild_f32(q[3]);
        return __builtin_amdgcn_ballot_w64(w) != 0ull ? 1u : 0u;
      }
    else
      return 0u;
  };

  const int y0=chunk*a.rows_per_chunk;
  int y1=y0+a.rows_per_chunk;
  y1=y1 < a.dst_rows ? y1 : a.dst_rows;
  auto note=[&](unsigned long long lanes,int y)
  {
    // (the mask is wave-uniform, but it was merged behind finish_fast's rare branch: made scalar
    // again here, the rest is scalar arithmetic and selects — no branch in the walk)
    const unsigned long long doubt=((unsigned long long) __builtin_amdgcn_readfirstlane((unsigned) (lanes >> 32)) << 32) |
      (unsigned long long) __builtin_amdgcn_readfirstlane((unsigned) lanes);
    marks|=doubt != 0ull ? 1u << ((y-y0) >> a.mark_shift) : 0u;
    const unsigned first=doubt != 0ull ? (unsigned) __builtin_ctzll(doubt) : 64u;
    const unsigned last=doubt != 0ull ? 63u-(unsigned) __builtin_clzll(doubt) : 0u;
    lane_first=first < lane_first ? first : lane_first;
    lane_last=last > lane_last ? last : lane_last;
  };
  const scalar_ints vbase=(scalar_ints) a.vbase;
  int base=vbase[y0];
  // Source row r lives in window slot r % ROWS for as long as it is under the window (the dense
  // weights of a row are stored by slot, resize_stream_plan.hpp): a new row replaces the one that
  // left, nothing moves.  (A slot is a set of registers: the slot number selects a copy of the code.)
  auto enter=[&](int slot,const Q (&q)[4])
  {
    const unsigned bad=is_wild(q);
#pragma unroll
    for (int j=0; j < ROWS; j++)
      if (slot == j)
        {
          premultiplied<Q,BLEND>(q,win[j]);
          // (the optimiser otherwise sinks the copies' stores into one, win[slot]: a dynamically
          // indexed array in scratch memory)
#pragma unroll
          for (int k=0; k < 4; k++)
            asm volatile("" : "+v"(win[j][k]));
          wild=(wild & ~(1u << j)) | (bad << j);
        }
  };
  int slot=base % ROWS;                        // of source row `base`, and of the row that follows the window
#pragma unroll
  for (int j=0; j < ROWS; j++)
    {
      Q q[4];
      fetch(base+j,q);
      int at=slot+j;
      at=at >= ROWS ? at-ROWS : at;
      enter(at,q);
    }
  Q ahead[4];                                  // the next source row, on its way
  fetch(base+ROWS,ahead);
  {
    // a use in front of the walk: the walk is then entered with no load in flight, and the only
    // pending state at its head is the latch's — one load, F stores behind it (see below)
    unsigned any=0u;
#pragma unroll
    for (int k=0; k < 4; k++)
      {
        if constexpr (kFloat)
          any|=__builtin_bit_cast(unsigned,ahead[k]);
        else
          any|=(unsigned) ahead[k];
      }
    asm volatile("" :: "v"(any));
  }

  // VerticalFilter of row y out of the window (wave-uniform weights), the Quantum-rounded
  // intermediate pixel, premultiplied again: what the lane contributes to HorizontalFilter
  double iv[4];
  auto vertical=[&](int y)
  {
    const scalar_doubles wv=(scalar_doubles) (a.vdense+(size_t) y*StreamResizePlan::kRows);
    double s[4]={0.0,0.0,0.0,0.0};
#pragma unroll
    for (int j=0; j < ROWS; j++)
      {
        const double w=wv[j];
#pragma unroll
        for (int k=0; k < 4; k++)
          s[k]=__builtin_fma(w,win[j][k],s[k]);
      }
    Q q[4];
    unsigned long long doubt=0ull;
    finish_fast<Q,BLEND,2,true>(s,q,doubt);
    note(doubt,y);
    premultiplied<Q,BLEND>(q,iv);
  };
  // the lanes whose source column is one of the strip's own: their F outputs are the ones stored
  const unsigned long long own=__builtin_amdgcn_ballot_w64((lane >= -lo) && (lane < count-lo));
  // which lanes of store j hold a pixel of the strip: lane masks in scalar registers
  unsigned long long keep_mask[F];
#pragma unroll
  for (int j=0; j < F; j++)
    {
      const int from=(64*j+lane)/F;
      keep_mask[j]=__builtin_amdgcn_ballot_w64((from >= -lo) && (from < count-lo));
    }
  // the transposed pixels of row y: store j writes pixels F*c0+64*j .. +63
  auto store_row=[&](int y,bool really)
  {
    const __amdgpu_buffer_rsrc_t drow=__builtin_amdgcn_make_buffer_rsrc(
      static_cast<unsigned char *>(a.dst)+(size_t) y*(size_t) OW*PX,0,OW*PX,0x00020000);
#pragma unroll
    for (int j=0; j < F; j++)
      {
        typedef unsigned words4 __attribute__((ext_vector_type(4)));
        typedef unsigned words2 __attribute__((ext_vector_type(2)));
        const bool keep=(((really ? keep_mask[j] : 0ull) >> lane) & 1ull) != 0ull;
        const unsigned char *at=kSteps ? xfrom[0]+(64/F)*j*PX : xfrom[kSteps ? 0 : j];
        const unsigned offset=keep ? (kSteps ? store_offset[0]+(unsigned) (64*j*PX) : store_offset[kSteps ? 0 : j]) : 0xffffffffu;
        // (A/B builds: -DMH_STREAM_NT stores non-temporally)
#ifdef MH_STREAM_NT
        constexpr int kAux=2;
#else
        constexpr int kAux=0;
#endif
        if constexpr (PX == 16)
          __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const words4 *>(at),drow,offset,0,kAux);
        else
          __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const words2 *>(at),drow,offset,0,kAux);
      }
  };

  // The walk, software-pipelined so that each of the two LDS round trips of a row has independent
  // work behind it:
  //   write the intermediate pixel of row y | read back and store row y-1's transposed pixels
  //   read the neighbours, HorizontalFilter, finish, write row y's pixels to the transposition
  //   buffer | move the window, request the next source row, VerticalFilter of row y+1
  // The next source row is requested in EVERY iteration, used or not (the window moves down by at
  // most one source row per output row — an enlargement): between a request and its use lie
  // exactly the F stores of one output row, whether dropped or not, so the wait in front of the
  // use is vmcnt(F), not a wait for every store in flight.
  seen|=wild;
  vertical(y0);
  for (int y=y0; ; y++)
    {
      // ---- the wave's row of the intermediate (LDS operations of one wave execute in order)
      *row01=double2_t{iv[0],iv[1]};
      *row23=double2_t{iv[2],iv[3]};
      asm volatile("" ::: "memory");
      store_row(y > y0 ? y-1 : y0,y > y0);
      asm volatile("" ::: "memory");
      // ---- HorizontalFilter: f outputs out of nt neighbours, scalar weights
      double h[F][4];
#pragma unroll
      for (int p=0; p < F; p++)
#pragma unroll
        for (int k=0; k < 4; k++)
          h[p][k]=0.0;
#pragma unroll
      for (int j=0; j < NT; j++)
        {
          const double2_t n01=row01[lo+j],n23=row23[lo+j];
#pragma unroll
          for (int p=0; p < F; p++)
            if ((j >= phase_first(p)) && (j < phase_first(p)+NT-1))
            {
              const double w=hw[p*NT+j];
              h[p][0]=__builtin_fma(w,n01[0],h[p][0]);
              h[p][1]=__builtin_fma(w,n01[1],h[p][1]);
              h[p][2]=__builtin_fma(w,n23[0],h[p][2]);
              h[p][3]=__builtin_fma(w,n23[1],h[p][3]);
            }
        }
      if (edge_wave)
        {
          if (listed)
            {
              // the same sums with the column's own (clipped, renormalised) weights
#pragma unroll
              for (int p=0; p < F; p++)
#pragma unroll
                for (int k=0; k < 4; k++)
                  h[p][k]=0.0;
#pragma unroll
              for (int j=0; j < NT; j++)
                {
                  const double2_t n01=row01[lo+j],n23=row23[lo+j];
#pragma unroll
                  for (int p=0; p < F; p++)
                    if ((j >= phase_first(p)) && (j < phase_first(p)+NT-1))
                    {
                      const double w=mine_listed[p*NT+j];
                      h[p][0]=__builtin_fma(w,n01[0],h[p][0]);
                      h[p][1]=__builtin_fma(w,n01[1],h[p][1]);
                      h[p][2]=__builtin_fma(w,n23[0],h[p][2]);
                      h[p][3]=__builtin_fma(w,n23[1],h[p][3]);
                    }
                }
            }
        }
      // ---- finish; the pixels go to the transposition buffer (read back in the next iteration)
      unsigned long long doubt=0ull;
#pragma unroll
      for (int p=0; p < F; p++)
        {
          Q out[4];
          // (sum |weight * sample| of this output's window: only where the clamp acts, see finish_fast)
          auto magnitudes=[&](double (&a)[4]) -> bool
          {
            a[0]=a[1]=a[2]=a[3]=0.0;
            const bool own_weights=edge_wave && listed;
#pragma unroll
            for (int j=0; j < NT; j++)
              if ((j >= phase_first(p)) && (j < phase_first(p)+NT-1))
                {
                  const double2_t n01=row01[lo+j],n23=row23[lo+j];
                  const double w=__builtin_fabs(own_weights ? mine_listed[p*NT+j] : hw[p*NT+j]);
                  a[0]=__builtin_fma(w,__builtin_fabs(n01[0]),a[0]);
                  a[1]=__builtin_fma(w,__builtin_fabs(n01[1]),a[1]);
                  a[2]=__builtin_fma(w,__builtin_fabs(n23[0]),a[2]);
                  a[3]=__builtin_fma(w,__builtin_fabs(n23[1]),a[3]);
                }
            return true;
          };
          // (an odd factor only: no other has an output whose window is one sample's)
          if constexpr ((F % 2) == 1)
            finish_fast<Q,BLEND,1,false>(h[p],out,doubt,magnitudes);
          else
            finish_fast<Q,BLEND,1,false>(h[p],out,doubt);
          store_pixel<Q,4>(reinterpret_cast<Q *>(xpose+p*XPLANE+lane*PX),out);
        }
      // (the lanes beside the strip's own columns summed neighbours they do not have: their pixels are never stored)
      note(doubt & own,y);
      asm volatile("" ::: "memory");
      if (y+1 >= y1)
        break;
      // ---- the window of row y+1, and its VerticalFilter
      if (base < vbase[y+1])
        {
          enter(slot,ahead);                   // row base+ROWS takes the slot of row base
          base++;
          slot=slot+1 == ROWS ? 0 : slot+1;
        }
      fetch(base+ROWS,ahead);
      seen|=wild;
      vertical(y+1);
    }
  store_row(y1-1,true);
  // (the lanes beside the strip's own columns computed neighbours' intermediate pixels: they count)
  if (seen != 0u)
    {
      marks=0xffffffffu;
      lane_first=0u;
      lane_last=63u;
    }
  if ((marks != 0u) && (lane == 0))
    {
      a.wild_items[2*item]=marks;              // the careful launch rewrites these rows of the item
      a.wild_items[2*item+1]=lane_first | (lane_last << 8);
    }
}

// ------------------------------------------------------------------ host side
struct StreamPlanDevice
{
  StreamResizePlan plan;
  TableBundle tables;
  size_t i_first=0,i_count=0,i_strip_hw=0,i_listed=0,i_vbase=0,i_vdense=0,i_vstart=0,i_vcount=0,i_hstart=0,i_hcount=0,i_vw=0,i_hw=0;
  hipEvent_t ready=nullptr;
  int device=-1,strips=0;
  bool ok=false;
  ~StreamPlanDevice()
  {
    if (device >= 0)
      {
        DeviceGuard guard;
        if (guard.enter(device) == hipSuccess)
          (void) hipDeviceSynchronize();     // shared across streams, as PassTables (resize.hip)
      }
    if (ready != nullptr)
      (void) hipEventDestroy(ready);
  }
};

struct StreamPlanEntry
{
  unsigned long long vserial,hserial; int device; std::shared_ptr<StreamPlanDevice> plan;
};
static std::mutex &stream_plans_lock() { static std::mutex &m=*new std::mutex; return m; }
static std::vector<StreamPlanEntry> &stream_plans() { static std::vector<StreamPlanEntry> &v=*new std::vector<StreamPlanEntry>; return v; }

void release_resize_stream_plans()
{
  std::lock_guard<std::mutex> guard(stream_plans_lock());
  stream_plans().clear();
}

static MhStatus build_stream_plan_device(StreamPlanDevice &d,const TapTable &vt,const TapTable &ht,int src_columns,
  int src_rows,int device,hipStream_t stream)
{
  d.ok=build_stream_resize_plan(d.plan,vt,ht,src_columns,src_rows);
  if (!d.ok)
    return MH_OK;
  const StreamResizePlan &p=d.plan;
#define MH_ADD(vec) d.tables.add((vec).data(),(vec).size()*sizeof((vec)[0]))
  d.i_vbase=MH_ADD(p.vbase); d.i_vdense=MH_ADD(p.vdense); d.i_listed=MH_ADD(p.listed);
  d.i_first=MH_ADD(p.strip_first); d.i_count=MH_ADD(p.strip_count); d.i_strip_hw=MH_ADD(p.strip_hw);
  d.i_vstart=MH_ADD(vt.start); d.i_vcount=MH_ADD(vt.count); d.i_hstart=MH_ADD(ht.start); d.i_hcount=MH_ADD(ht.count);
  d.i_vw=MH_ADD(vt.weight); d.i_hw=MH_ADD(ht.weight);
#undef MH_ADD
  MH_TRY(d.tables.upload(device,stream));
  d.device=device;
  MH_HIP(hipEventCreateWithFlags(&d.ready,hipEventDisableTiming));
  MH_HIP(hipEventRecord(d.ready,stream));
  d.strips=(int) p.strip_first.size();
  d.plan.vdense.clear(); d.plan.vdense.shrink_to_fit();
  d.plan.vbase.clear(); d.plan.vbase.shrink_to_fit();
  return MH_OK;
}

static MhStatus acquire_stream_plan(std::shared_ptr<StreamPlanDevice> *out,const TapTable &vt,const TapTable &ht,
  int src_columns,int src_rows,int device,hipStream_t stream)
{
  const bool shared=(vt.serial != 0) && (ht.serial != 0);
  constexpr size_t kEntries=16;
  if (shared)
    {
      std::lock_guard<std::mutex> guard(stream_plans_lock());
      std::vector<StreamPlanEntry> &entries=stream_plans();
      for (size_t i=0; i < entries.size(); i++)
        if ((entries[i].vserial == vt.serial) && (entries[i].hserial == ht.serial) && (entries[i].device == device))
          {
            StreamPlanEntry hit=entries[i];
            entries.erase(entries.begin()+(ptrdiff_t) i);
            entries.insert(entries.begin(),hit);
            *out=hit.plan;
            if (hit.plan->ok)
              MH_HIP(hipStreamWaitEvent(stream,hit.plan->ready,0));
            return MH_OK;
          }
    }
  auto built=std::make_shared<StreamPlanDevice>();
  MH_TRY(build_stream_plan_device(*built,vt,ht,src_columns,src_rows,device,stream));
  *out=built;
  // (the evicted plan is released after the lock: its destructor drains the device)
  std::shared_ptr<StreamPlanDevice> evicted;
  if (shared)
    {
      std::lock_guard<std::mutex> guard(stream_plans_lock());
      std::vector<StreamPlanEntry> &entries=stream_plans();
      entries.insert(entries.begin(),StreamPlanEntry{vt.serial,ht.serial,device,built});
      if (entries.size() > kEntries)
        {
          evicted=std::move(entries.back().plan);
          entries.pop_back();
        }
    }
  return MH_OK;
}

template<typename Q,bool BLEND,int F,int NT,int ROWS>
static MhStatus launch_stream_typed(const View &src,const View &dst,const StreamPlanDevice &d)
{
  const StreamResizePlan &p=d.plan;
  const TableBundle &t=d.tables;
  StreamResizeArgs a;
  a.src=src.pixels; a.dst=dst.pixels;
  a.src_columns=(int) src.columns; a.src_rows=(int) src.rows;
  a.dst_columns=(int) dst.columns; a.dst_rows=(int) dst.rows;
  a.lo=p.lo; a.edge_left=p.edge_left; a.edge_right=p.edge_right;
  a.strips=d.strips;
  a.strip_first=t.at<int>(d.i_first); a.strip_count=t.at<int>(d.i_count); a.strip_hw=t.at<double>(d.i_strip_hw);
  // a wave walks `rows` output rows (it re-reads the kRows source rows above its first one): enough
  // waves for a few rounds over the chip's 12 a CU
  // (... about eight: 128 rows where the frame has them, 64 or 32 on a smaller one — 4096^2 x4 1.47 -> 1.29 ms,
  // 8192^2 x2 1.64 -> 1.60; C3's 8192^2 x4 keeps 128.  MAGICKHIP_RESIZE_STREAM_ROWS fixes it.)
  int rows=(int) option_long("MAGICKHIP_RESIZE_STREAM_ROWS",0);
  if (rows <= 0)
    {
      const long long wanted=8ll*12ll*(long long) compute_units(src.device);
      rows=128;
      while ((rows > 32) && ((long long) d.strips*(((long long) dst.rows+rows-1)/rows) < wanted))
        rows/=2;
    }
  rows=rows < 16 ? 16 : rows;
  a.rows_per_chunk=rows;
  a.mark_shift=2;                              // a chunk's rows in at most 32 blocks of 4 .. rows
  while (((rows-1) >> a.mark_shift) > 31)
    a.mark_shift++;
  a.chunks=((int) dst.rows+rows-1)/rows;
  a.vbase=t.at<int>(d.i_vbase); a.vdense=t.at<double>(d.i_vdense); a.listed=t.at<double>(d.i_listed);
  a.vstart=t.at<int>(d.i_vstart); a.vcount=t.at<int>(d.i_vcount);
  a.hstart=t.at<int>(d.i_hstart); a.hcount=t.at<int>(d.i_hcount);
  a.vweight=t.at<double>(d.i_vw); a.hweight=t.at<double>(d.i_hw);
  const long long items=(long long) a.strips*(long long) a.chunks;
  if (items >= (1ll << 30))
    return fail(MH_BAD_ARGUMENT,"resize: frame too large");
  dim3 grid((unsigned) ((items+3)/4));
  Temp flags;
  MH_TRY(flags.alloc(src.device,(size_t) items*2u*sizeof(unsigned),src.stream));
  MH_HIP(hipMemsetAsync(flags.ptr,0,(size_t) items*2u*sizeof(unsigned),src.stream));
  a.wild_items=flags.as<unsigned>();
  a.nt=p.nt;
  {
    ProfileScope prof("resize_stream",src.stream);
    hipLaunchKernelGGL((resize_stream_kernel<Q,BLEND,F,NT,ROWS>),grid,dim3(256),0,src.stream,a);
    MH_HIP(hipGetLastError());
  }
  if (option("MAGICKHIP_RESIZE_STREAM_REPORT") != nullptr)
    {
      // diagnostics: how many blocks of rows the walk handed to the careful launch
      std::vector<unsigned> host((size_t) items*2u);
      MH_HIP(hipMemcpyAsync(host.data(),flags.ptr,(size_t) items*2u*sizeof(unsigned),hipMemcpyDeviceToHost,src.stream));
      MH_HIP(hipStreamSynchronize(src.stream));
      long long marked=0,touched=0,lanes=0;
      for (long long i=0; i < items; i++)
        {
          const unsigned word=host[(size_t) (2*i)],range=host[(size_t) (2*i+1)];
          marked+=__builtin_popcount(word);
          touched+=word != 0u ? 1 : 0;
          lanes+=word != 0u ? (long long) ((range >> 8) & 0xffu)-(long long) (range & 0xffu)+1 : 0;
        }
      fprintf(stderr,"resize_stream: %lld of %lld items hold %lld marked blocks of %d rows, %.1f lanes wide on average\n",touched,
        items,marked,1 << a.mark_shift,touched > 0 ? (double) lanes/(double) touched : 0.0);
    }
  {
      ProfileScope prof("resize_stream_careful",src.stream);
      hipLaunchKernelGGL((resize_stream_careful_kernel<Q,BLEND>),dim3((unsigned) items),dim3(256),0,src.stream,a,F);
      MH_HIP(hipGetLastError());
    }
  return MH_OK;
}

template<int F,int NT,int ROWS>
static MhStatus launch_stream_layout(const View &src,const View &dst,const StreamPlanDevice &d,bool blend)
{
  if (src.quantum == MH_QUANTUM_U16)
    return blend ? launch_stream_typed<uint16_t,true,F,NT,ROWS>(src,dst,d) : launch_stream_typed<uint16_t,false,F,NT,ROWS>(src,dst,d);
  return blend ? launch_stream_typed<float,true,F,NT,ROWS>(src,dst,d) : launch_stream_typed<float,false,F,NT,ROWS>(src,dst,d);
}

template<int F>
static MhStatus launch_stream_factor(const View &src,const View &dst,const StreamPlanDevice &d,bool blend)
{
  // (five neighbours = a support of two source pixels: at most five rows under the window too)
  if (d.plan.nt == 5)
    return launch_stream_layout<F,5,6>(src,dst,d,blend);
  if (d.plan.window_rows() == 6)
    return launch_stream_layout<F,7,6>(src,dst,d,blend);
  return launch_stream_layout<F,7,8>(src,dst,d,blend);
}

// *handled = false (nothing launched) when the frame or the geometry is not this kernel's.
MhStatus launch_resize_stream(const View &src,const View &dst,const TapTable &vertical,const TapTable &horizontal,
  const Roles &roles,bool *handled)
{
  *handled=false;
  if ((src.channels != 4) || (dst.channels != 4) || (src.quantum != dst.quantum) || (roles.copy_mask != 0))
    return MH_OK;
  if (roles.blend && (roles.alpha != 3))
    return MH_OK;
  if (((int) dst.rows != vertical.out_size) || ((int) dst.columns != horizontal.out_size))
    return fail(MH_BAD_ARGUMENT,"resize: geometry mismatch");
  if ((dst.rows < src.rows) || (dst.columns < 2*src.columns) || ((dst.columns % src.columns) != 0) ||
      (dst.columns/src.columns > 4))
    return MH_OK;
  // (32-bit byte offsets inside a row — the stores' buffer descriptor —, 64-bit row offsets)
  if ((dst.columns >= (1u << 27)) || (dst.rows >= (1u << 30)))
    return MH_OK;
  std::shared_ptr<StreamPlanDevice> plan;
  MH_TRY(acquire_stream_plan(&plan,vertical,horizontal,(int) src.columns,(int) src.rows,src.device,src.stream));
  if (!plan->ok)
    return MH_OK;
  // a first filter whose sums land exactly on rounding boundaries once in a few values (StreamResizePlan::
  // weights_denominator): the two passes then — unless the suites' switch is set
  {
    const int D=plan->plan.weights_denominator;
    if ((D > 1) && ((src.quantum != MH_QUANTUM_U16) || ((D & 1) == 0)) &&
        (option_long("MAGICKHIP_RESIZE_ONE_LAUNCH_MIN_PIXELS",-1) < 0))
      return MH_OK;
  }
  *handled=true;
  switch (plan->plan.f)
  {
    case 2: return launch_stream_factor<2>(src,dst,*plan,roles.blend);
    case 3: return launch_stream_factor<3>(src,dst,*plan,roles.blend);
    default: return launch_stream_factor<4>(src,dst,*plan,roles.blend);
  }
}

} // namespace mh
