// The arguments of convolve_separable.hip's finish step, shared with the folded passes of convolve.hip.
#pragma once
#include "mh_internal.hpp"

namespace mh {

struct SeparableArgs
{
  const void *src;
  void *dst;
  double *sums;               // [rows][columns][4]
  double *bound;              // [4]: largest |P_c| of the frame (float Quantum)
  int columns,rows;
  const double *values;       // the kernel's cells (device)
  int kw,kh,shiftx,shifty;
  double error_unit;          // |difference of the two evaluations| <= error_unit * max|P_c|
  double fixed_bound[4];      // Q16: max|P_c| is known (65535^2 for alpha-weighted colour, 65535)
  // kernel = column x row + delta at one cell (SharpenImage, EdgeImage: a negated Gaussian / a
  // box whose centre carries the normalisation): delta times the sample that cell sees, which
  // is (x+delta_dx, y+delta_dy)
  double delta;
  int delta_dx,delta_dy;
  int mixed_signs;            // cells of both signs: an alpha sum of exactly zero is not "all transparent"
  unsigned long long *recomputed;
  // folded passes: the column pass queues its undecided samples ((pixel index << 4) | channel mask)
  // for separable_settle_kernel instead of settling them between its own stores
  unsigned long long *queue;
  unsigned *queue_count;
  unsigned queue_capacity;
};

// convolve.hip: the row pass (premultiply inside) and the column pass (finish inside) as two launches
MhStatus launch_separable_folded(const View &src,const SeparableArgs &sep,const Conv1DParams &horizontal,
  const Conv1DParams &vertical,bool blend);

} // namespace mh
