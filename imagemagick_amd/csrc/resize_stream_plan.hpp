// Host-side geometry of the one-launch resize on the fp64 VECTOR pipe (resize_stream.hip): FAST
// enlargements of four-channel frames by a whole-number horizontal factor.
//
// ResizeImage for an enlargement runs VerticalFilter, then HorizontalFilter
// (MagickCore/resize.c:3846-3861, :3549-3759, :3333-3547).  In the kernel a lane owns ONE source
// column and a wave walks down the output rows:
//   vertical    all 64 lanes work on the same output row, so VerticalFilter's weights are
//               wave-uniform (scalar registers).  The lane keeps the `kRows` source rows under the
//               filter window in registers; this header turns a row's contribution list into
//               `base` (the source row in window slot 0) and `kRows` dense weights.
//   horizontal  with dst_columns = f * src_columns the outputs f*c+p (p = 0..f-1) of source column
//               c read the same neighbours c+lo .. c+lo+nt-1 with the same weights for every c of
//               a strip away from the image edges: f*nt scalar registers again.  The contribution
//               lists are checked for exactly that (StreamResizePlan::strip_hw), and the source
//               columns near the two edges whose windows are clipped (and renormalised) are
//               listed: their outputs take the table's own weights in the kernel.
// Pure host C++ (no HIP): tests/cpu/resize_stream_plan_test.cpp emulates the kernel's walk with
// the same tables.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

namespace mh {

struct StreamResizePlan
{
  static constexpr int kRows=8;        // source rows a lane keeps (VerticalFilter's window, zero padded)
  static constexpr int kMaxDense=32;   // f*nt
  static constexpr int kListed=8;      // listed columns per image edge, at most
  // output q of a source column (q = 0..f-1) reads nt-1 of the nt neighbours, from this one on
  static constexpr int phase_first(int f,int q) { return 2*q+1 >= f ? 1 : 0; }
  int f=0;                             // dst_columns / src_columns
  int nt=0;                            // neighbours a lane reads (5 or 7; the window, zero padded)
  int lo=0;                            // first neighbour relative to the lane's own column
  int nvl=0;                           // source columns a wave finishes at most: 64-(nt-1)
  int edge_left=0,edge_right=0;        // columns c < edge_left or c >= edge_right: listed weights
  int vmax=0;                          // most source rows under VerticalFilter's window
  int window_rows() const { return vmax <= 6 ? 6 : kRows; }   // the kernel's two window sizes
  // A strip = the source columns one wave finishes.  bisect = (x+0.5)/factor+MagickEpsilon
  // (resize.c:3404-3410) rounds MagickEpsilon to the grid of its binade, so the weights of the f
  // outputs of a column are bit-identical from column to column INSIDE a binade [2^k, 2^(k+1)) and
  // differ by ~1e-12 across one: strips are cut at the powers of two and carry their own weights.
  std::vector<int> strip_first,strip_count;     // [strips]
  std::vector<double> strip_hw;        // [strips][kMaxDense]: [f][nt] dense
  std::vector<double> listed;          // [2*kListed][kMaxDense]: the listed columns' weights, dense like hw
  // The first filter's weights are multiples of 1/D for a small D (Triangle at x2: quarters; Hermite at x2: 32nds;
  // the cubic B-spline at x3: 162nds; Mitchell at x2: 1152nds; always minus MagickEpsilon's 1e-12): its sums over
  // integer samples (D even) or over float samples of like magnitude (any D: a half of the RESULT's last place is a
  // whole number of the samples' last places) then sit exactly on rounding boundaries once in about D values, most
  // blocks of rows are reported, and the careful launch does the frame at several times the two passes' cost
  // (tools/sweep_cliffs.py: Triangle x4 on 2048^2 1.78 ms against 0.33; float Mitchell x3 on 4096^2 2.8 against 0.9).
  // The smallest such D up to 4096, 0 if there is none (the windowed sincs, Gaussian, Robidoux).
  int weights_denominator=0;
  std::vector<int> vbase;              // [dst_rows]
  std::vector<double> vdense;          // [dst_rows][kRows]
};

// Table: {int out_size; std::vector<int> start,count; std::vector<double> weight /* [tap][out] */;}
// false: the geometry is not this kernel's (the caller takes another path).
template<class Table>
static bool build_stream_resize_plan(StreamResizePlan &p,const Table &vt,const Table &ht,int src_columns,
  int src_rows)
{
  constexpr int D=StreamResizePlan::kMaxDense;
  const int W=src_columns,H=src_rows,OW=ht.out_size,OH=vt.out_size;
  if ((W < 1) || (H < 1) || (OW < 2*W) || (OH < H) || ((OW % W) != 0))
    return false;
  p.f=OW/W;
  if (p.f > 4)
    return false;
  const int f=p.f;
  // ---- horizontal: offsets and counts of the middle column are every interior column's
  const int cm=W/2;
  int s[4]={0,0,0,0},n[4]={0,0,0,0};
  int lo=0x7fffffff,hi=-0x7fffffff;
  for (int q=0; q < f; q++)
    {
      const size_t x=(size_t) f*(size_t) cm+(size_t) q;
      if (ht.count[x] <= 0)
        return false;
      s[q]=ht.start[x]-cm;
      n[q]=ht.count[x];
      lo=std::min(lo,s[q]);
      hi=std::max(hi,s[q]+n[q]);
    }
  const int span=hi-lo;
  p.nt=span <= 5 ? 5 : 7;
  if ((span > 7) || (lo > 0) || (lo < -8) || (lo+p.nt-1 > 8) || (f*p.nt > D))
    return false;
  p.lo=lo;
  p.nvl=64-(p.nt-1);
  const int nt=p.nt;
  auto same_shape=[&](int c) -> bool
  {
    for (int q=0; q < f; q++)
      {
        const size_t x=(size_t) f*(size_t) c+(size_t) q;
        if ((ht.start[x] != c+s[q]) || (ht.count[x] != n[q]))
          return false;
      }
    return true;
  };
  auto dense_of=[&](int c,double *to)
  {
    for (int i=0; i < D; i++)
      to[i]=0.0;
    for (int q=0; q < f; q++)
      {
        const size_t x=(size_t) f*(size_t) c+(size_t) q;
        for (int k=0; k < ht.count[x]; k++)
          to[q*nt+(ht.start[x]-(c+lo))+k]=ht.weight[(size_t) k*(size_t) OW+x];
      }
  };
  int left=cm,right=cm+1;
  while ((left > 0) && same_shape(left-1))
    left--;
  while ((right < W) && same_shape(right))
    right++;
  p.edge_left=left;
  p.edge_right=right;
  // the listed columns are the few under a clipped window
  if ((left > StreamResizePlan::kListed) || (W-right > StreamResizePlan::kListed))
    return false;
  p.listed.assign((size_t) 2*StreamResizePlan::kListed*D,0.0);
  // a listed output's window lies inside the interior one of its column: its neighbours are in the
  // wave's row of the intermediate
  for (int c=0; c < W; c++)
    {
      if ((c >= left) && (c < right))
        continue;
      for (int q=0; q < f; q++)
        {
          const size_t x=(size_t) f*(size_t) c+(size_t) q;
          if ((ht.count[x] <= 0) || (ht.start[x] < c+lo) || (ht.start[x]+ht.count[x] > c+lo+nt) ||
              (ht.start[x] < 0) || (ht.start[x]+ht.count[x] > W))
            return false;
        }
      const int entry=c < left ? c : StreamResizePlan::kListed+(c-right);
      dense_of(c,&p.listed[(size_t) entry*D]);
    }
  // strips: at most nvl columns, cut at the powers of two; every interior column of a strip has
  // the strip's weights bit for bit — anything else is not this kernel's (weights that differ by
  // 1e-12 move a near-cancelling sum by more than a float ULP, and a weight born of cancellation
  // — Triangle at distance 1-MagickEpsilon: 1e-12 +- 1e-15 — decides a result on its own where
  // everything else under the window is transparent)
  p.strip_first.clear(); p.strip_count.clear(); p.strip_hw.clear();
  for (int c=0; c < W; )
    {
      int end=std::min(W,c+p.nvl);
      for (int power=1; (power > 0) && (power < end); power<<=1)
        if (power > c)
          {
            end=power;
            break;
          }
      p.strip_first.push_back(c);
      p.strip_count.push_back(end-c);
      double rep[D],other[D];
      bool have=false;
      for (int i=0; i < D; i++)
        rep[i]=0.0;
      for (int k=c; k < end; k++)
        {
          if ((k < left) || (k >= right))
            continue;
          if (!have)
            {
              dense_of(k,rep);
              have=true;
              continue;
            }
          dense_of(k,other);
          for (int i=0; i < D; i++)
            if (!(other[i] == rep[i]))
              return false;
        }
      p.strip_hw.insert(p.strip_hw.end(),rep,rep+D);
      c=end;
    }
  // the kernel skips the neighbour a phase does not read (StreamResizePlan::phase_first)
  auto trimmed=[&](const double *dense) -> bool
  {
    for (int q=0; q < f; q++)
      for (int j=0; j < nt; j++)
        if (((j < StreamResizePlan::phase_first(f,q)) || (j >= StreamResizePlan::phase_first(f,q)+nt-1)) &&
            (dense[q*nt+j] != 0.0))
          return false;
    return true;
  };
  for (size_t i=0; i < p.strip_hw.size(); i+=D)
    if (!trimmed(&p.strip_hw[i]))
      return false;
  for (size_t i=0; i < p.listed.size(); i+=D)
    if (!trimmed(&p.listed[i]))
      return false;
  // ---- vertical: window base and dense weights per output row.  The kernel keeps source row r in
  // window slot r % rows (rows = window_rows(): 6 or 8): the weights are stored by slot.
  p.vmax=0;
  p.vbase.assign((size_t) OH,0);
  p.vdense.assign((size_t) OH*StreamResizePlan::kRows,0.0);
  for (int y=0; y < OH; y++)
    {
      const int count=vt.count[(size_t) y],start=vt.start[(size_t) y];
      if ((count <= 0) || (count > StreamResizePlan::kRows) || (start < 0) || (start+count > H))
        return false;
      // the window moves down by at most one source row per output row (the kernel's walk)
      if ((y > 0) && ((start < p.vbase[(size_t) y-1]) || (start > p.vbase[(size_t) y-1]+1)))
        return false;
      p.vbase[(size_t) y]=start;
      p.vmax=std::max(p.vmax,count);
    }
  const int rows=p.window_rows();
  for (int y=0; y < OH; y++)
    for (int k=0; k < vt.count[(size_t) y]; k++)
      p.vdense[(size_t) y*StreamResizePlan::kRows+(size_t) ((vt.start[(size_t) y]+k) % rows)]=
        vt.weight[(size_t) k*(size_t) OH+(size_t) y];
  if ((p.nt == 5) && (p.vmax > 6))
    return false;                       // (a support of two source pixels has at most five rows: cannot happen)
  // the weights of a whole period of interior rows (the clipped windows at the edges are renormalised: other fractions)
  p.weights_denominator=0;
  {
    const int period=(OH % H) == 0 ? OH/H : 0;
    if ((period >= 1) && (period <= 64))
      {
        const int y0=(OH/2/period)*period;
        for (int D=1; (D <= 4096) && (p.weights_denominator == 0); D++)
          {
            bool all=true;
            for (int y=y0; all && (y < y0+period) && (y < OH); y++)
              for (int k=0; all && (k < vt.count[(size_t) y]); k++)
                {
                  const double scaled=vt.weight[(size_t) k*(size_t) OH+(size_t) y]*(double) D;
                  all=std::fabs(scaled-std::nearbyint(scaled)) <= 1.0e-9*(double) D;
                }
            if (all)
              p.weights_denominator=D;
          }
      }
  }
  return true;
}

} // namespace mh
