// Resize filter object and the per-output-index contribution ("tap") table the
// resize kernels consume.
#pragma once

#include "mh_internal.hpp"

#include <memory>

struct MhResizeFilter
{
  int filter_fn=0;          // index into the weighting-function switch
  int window_fn=0;
  double support=0.0;
  double window_support=0.0;
  double scale=1.0;
  double blur=1.0;
  double coefficient[7]={0,0,0,0,0,0,0};
  // MhAcquireResizeFilterFromCallback: weights and support come from the caller
  // (the MagickCore shim passes the reference's own GetResizeFilterWeight)
  double (*callback)(void *,double)=nullptr;
  void *callback_user=nullptr;
  double callback_support=0.0;
};

namespace mh {

// ContributionInfo lists for every output index along one axis
// (resize.c:3282-3289, :3418-3443), flattened.
struct TapTable
{
  int out_size=0;
  int max_taps=0;
  std::vector<int> start;        // first source index per output
  std::vector<int> count;        // number of taps per output (0 => output untouched)
  std::vector<int> nearest;      // source index used by Copy-trait channels
  std::vector<double> weight;    // [tap][out] : normalised weights (transposed for coalescing)
  // nonzero for a table the cache of acquire_tap_table shares between calls: the key under which
  // the resize launchers keep its device-side copies
  unsigned long long serial=0;
};

void build_tap_table(TapTable &table,const MhResizeFilter *filter,size_t in_size,
  size_t out_size,double factor);

// Shared, possibly cached table (built-in filters are cached by their parameters).
std::shared_ptr<const TapTable> acquire_tap_table(const MhResizeFilter *filter,size_t in_size,
  size_t out_size,double factor);

} // namespace mh
