// Host-side resize filters: the product's restatement of AcquireResizeFilter
// (MagickCore/resize.c:803-1298, no expert `filter:*` artifacts, orthogonal
// use only), the weighting functions (resize.c:140-620), GetResizeFilterWeight
// (resize.c:1690-1714) and the contribution set-up of HorizontalFilter /
// VerticalFilter (resize.c:3364-3443, :3580-3660).  All double precision, in
// the reference's operation order: the weights these functions return are
// compared bit-for-bit with the compiled reference in tests/.
#include "resize_filter.hpp"
#include <cstring>
#include <memory>
#include <atomic>
#include <mutex>
#include <thread>

#include <cmath>

namespace {

constexpr double kPi=3.14159265358979323846264338327950288419716939937510;
constexpr double kPi2=1.57079632679489661923132169163975144209858469968755;
constexpr double kEpsilon=mh::kMagickEpsilon;

double perceptible_reciprocal(double x)
{
  double sign=x < 0.0 ? -1.0 : 1.0;
  if ((sign*x) >= kEpsilon)
    return 1.0/x;
  return sign/kEpsilon;
}

enum Fn
{
  FN_BOX,FN_TRIANGLE,FN_CUBICBC,FN_HANN,FN_HAMMING,FN_BLACKMAN,FN_GAUSSIAN,
  FN_QUADRATIC,FN_JINC,FN_SINC,FN_SINCFAST,FN_KAISER,FN_WELCH,FN_BOHMAN,
  FN_LAGRANGE,FN_COSINE,FN_CUBICSPLINE,FN_MKS2013,FN_MKS2021
};

// I0, resize.c:1385-1408
double bessel_i0(double x)
{
  double sum=1.0,y=x*x/4.0,t=y;
  for (long i=2; t > kEpsilon; i++)
    {
      sum+=t;
      t*=y/((double) i*i);
    }
  return sum;
}

// SincFast for Quantum depth 16, resize.c:493-587 (coefficients :551-563)
double sinc_fast(double x)
{
  if (x > 4.0)
    {
      const double alpha=(double) (kPi*x);
      return sin((double) alpha)/alpha;
    }
  const double xx=x*x;
  const double c0=0.173611107357320220183368594093166520811e-2L;
  const double c1=-0.384240921114946632192116762889211361285e-3L;
  const double c2=0.394201182359318128221229891724947048771e-4L;
  const double c3=-0.250963301609117217660068889165550534856e-5L;
  const double c4=0.111902032818095784414237782071368805120e-6L;
  const double c5=-0.372895101408779549368465614321137048875e-8L;
  const double c6=0.957694196677572570319816780188718518330e-10L;
  const double c7=-0.187208577776590710853865174371617338991e-11L;
  const double c8=0.253524321426864752676094495396308636823e-13L;
  const double c9=-0.177084805010701112639035485248501049364e-15L;
  const double p=c0+xx*(c1+xx*(c2+xx*(c3+xx*(c4+xx*(c5+xx*(c6+xx*(c7+xx*(c8+xx*c9))))))));
  return (xx-1.0)*(xx-4.0)*(xx-9.0)*(xx-16.0)*p;
}

double evaluate(int fn,double x,const MhResizeFilter *f)
{
  switch (fn)
  {
    case FN_BOX:
      return 1.0;
    case FN_TRIANGLE:
      return x < 1.0 ? 1.0-x : 0.0;
    case FN_CUBICBC:
      if (x < 1.0)
        return f->coefficient[0]+x*(x*(f->coefficient[1]+x*f->coefficient[2]));
      if (x < 2.0)
        return f->coefficient[3]+x*(f->coefficient[4]+x*(f->coefficient[5]+x*f->coefficient[6]));
      return 0.0;
    case FN_HANN:
    {
      const double cosine=cos((double) (kPi*x));
      return 0.5+0.5*cosine;
    }
    case FN_HAMMING:
    {
      const double cosine=cos((double) (kPi*x));
      return 0.54+0.46*cosine;
    }
    case FN_BLACKMAN:
    {
      const double cosine=cos((double) (kPi*x));
      return 0.34+cosine*(0.5+cosine*0.16);
    }
    case FN_GAUSSIAN:
      return exp((double) (-f->coefficient[1]*x*x));
    case FN_QUADRATIC:
      if (x < 0.5)
        return 0.75-x*x;
      if (x < 1.5)
        return 0.5*(x-1.5)*(x-1.5);
      return 0.0;
    case FN_SINC:
      if (x != 0.0)
        {
          const double alpha=(double) (kPi*x);
          return sin((double) alpha)/alpha;
        }
      return 1.0;
    case FN_SINCFAST:
      return sinc_fast(x);
    case FN_KAISER:
      return f->coefficient[1]*bessel_i0(f->coefficient[0]*sqrt((double) (1.0-x*x)));
    case FN_WELCH:
      return x < 1.0 ? 1.0-x*x : 0.0;
    case FN_BOHMAN:
    {
      const double cosine=cos((double) (kPi*x));
      const double sine=sqrt(1.0-cosine*cosine);
      return (1.0-x)*cosine+(1.0/kPi)*sine;
    }
    case FN_LAGRANGE:
    {
      if (x > f->support)
        return 0.0;
      ptrdiff_t order=(ptrdiff_t) (2.0*f->window_support);
      ptrdiff_t n=(ptrdiff_t) (f->window_support+x);
      double value=1.0f;
      for (ptrdiff_t i=0; i < order; i++)
        if (i != n)
          value*=((double) (n-i)-x)/(double) (n-i);
      return value;
    }
    case FN_COSINE:
      return cos((double) (kPi2*x));
    case FN_CUBICSPLINE:
      if (f->support <= 2.0)
        {
          if (x < 1.0)
            return ((x-9.0/5.0)*x-1.0/5.0)*x+1.0;
          if (x < 2.0)
            return ((-1.0/3.0*(x-1.0)+4.0/5.0)*(x-1.0)-7.0/15.0)*(x-1.0);
          return 0.0;
        }
      if (f->support <= 3.0)
        {
          if (x < 1.0)
            return ((13.0/11.0*x-453.0/209.0)*x-3.0/209.0)*x+1.0;
          if (x < 2.0)
            return ((-6.0/11.0*(x-1.0)+270.0/209.0)*(x-1.0)-156.0/209.0)*(x-1.0);
          if (x < 3.0)
            return ((1.0/11.0*(x-2.0)-45.0/209.0)*(x-2.0)+26.0/209.0)*(x-2.0);
          return 0.0;
        }
      if (x < 1.0)
        return ((49.0/41.0*x-6387.0/2911.0)*x-3.0/2911.0)*x+1.0;
      if (x < 2.0)
        return ((-24.0/41.0*(x-1.0)+4032.0/2911.0)*(x-1.0)-2328.0/2911.0)*(x-1.0);
      if (x < 3.0)
        return ((6.0/41.0*(x-2.0)-1008.0/2911.0)*(x-2.0)+582.0/2911.0)*(x-2.0);
      if (x < 4.0)
        return ((-1.0/41.0*(x-3.0)+168.0/2911.0)*(x-3.0)-97.0/2911.0)*(x-3.0);
      return 0.0;
    case FN_MKS2013:
      if (x < 0.5)
        return 0.625+1.75*(0.5-x)*(0.5+x);
      if (x < 1.5)
        return (1.0-x)*(1.75-x);
      if (x < 2.5)
        return -0.125*(2.5-x)*(2.5-x);
      return 0.0;
    case FN_MKS2021:
      if (x < 0.5)
        return 577.0/576.0-239.0/144.0*x*x;
      if (x < 1.5)
        return 35.0/36.0*(x-1.0)*(x-239.0/140.0);
      if (x < 2.5)
        return 1.0/6.0*(x-2.0)*(65.0/24.0-x);
      if (x < 3.5)
        return 1.0/36.0*(x-3.0)*(x-3.75);
      if (x < 4.5)
        return -1.0/288.0*(x-4.5)*(x-4.5);
      return 0.0;
    default:
      break;
  }
  return 0.0;
}

struct FilterRow { int fn; double support,scale,B,C; };

// the function/support/scale/B/C table, resize.c:889-944, indexed by FilterType
const FilterRow kFilters[MH_FILTER_SENTINEL]=
{
  {FN_BOX,0.5,0.5,0.0,0.0},          // Undefined
  {FN_BOX,0.0,0.5,0.0,0.0},          // Point
  {FN_BOX,0.5,0.5,0.0,0.0},          // Box
  {FN_TRIANGLE,1.0,1.0,0.0,0.0},     // Triangle
  {FN_CUBICBC,1.0,1.0,0.0,0.0},      // Hermite
  {FN_HANN,1.0,1.0,0.0,0.0},         // Hann
  {FN_HAMMING,1.0,1.0,0.0,0.0},      // Hamming
  {FN_BLACKMAN,1.0,1.0,0.0,0.0},     // Blackman
  {FN_GAUSSIAN,2.0,1.5,0.0,0.0},     // Gaussian
  {FN_QUADRATIC,1.5,1.5,0.0,0.0},    // Quadratic
  {FN_CUBICBC,2.0,2.0,1.0,0.0},      // Cubic
  {FN_CUBICBC,2.0,1.0,0.0,0.5},      // Catrom
  {FN_CUBICBC,2.0,8.0/7.0,1./3.,1./3.}, // Mitchell
  {FN_JINC,3.0,1.2196698912665045,0.0,0.0}, // Jinc
  {FN_SINC,4.0,1.0,0.0,0.0},         // Sinc
  {FN_SINCFAST,4.0,1.0,0.0,0.0},     // SincFast
  {FN_KAISER,1.0,1.0,0.0,0.0},       // Kaiser
  {FN_WELCH,1.0,1.0,0.0,0.0},        // Welch
  {FN_CUBICBC,2.0,2.0,1.0,0.0},      // Parzen
  {FN_BOHMAN,1.0,1.0,0.0,0.0},       // Bohman
  {FN_TRIANGLE,1.0,1.0,0.0,0.0},     // Bartlett
  {FN_LAGRANGE,2.0,1.0,0.0,0.0},     // Lagrange
  {FN_SINCFAST,3.0,1.0,0.0,0.0},     // Lanczos
  {FN_SINCFAST,3.0,1.0,0.0,0.0},     // LanczosSharp
  {FN_SINCFAST,2.0,1.0,0.0,0.0},     // Lanczos2
  {FN_SINCFAST,2.0,1.0,0.0,0.0},     // Lanczos2Sharp
  {FN_CUBICBC,2.0,1.1685777620836932,0.37821575509399867,0.31089212245300067}, // Robidoux
  {FN_CUBICBC,2.0,1.105822933719019,0.2620145123990142,0.3689927438004929},    // RobidouxSharp
  {FN_COSINE,1.0,1.0,0.0,0.0},       // Cosine
  {FN_CUBICBC,2.0,2.0,1.0,0.0},      // Spline
  {FN_SINCFAST,3.0,1.0,0.0,0.0},     // LanczosRadius
  {FN_CUBICSPLINE,2.0,0.5,0.0,0.0},  // CubicSpline
  {FN_MKS2013,2.5,1.0,0.0,0.0},      // MagicKernelSharp2013
  {FN_MKS2021,4.5,1.0,0.0,0.0}       // MagicKernelSharp2021
};

struct MapRow { MhFilterType filter,window; };

// filter -> (weighting, windowing) mapping, resize.c:841-877
const MapRow kMapping[MH_FILTER_SENTINEL]=
{
  {MH_FILTER_UNDEFINED,MH_FILTER_BOX},{MH_FILTER_POINT,MH_FILTER_BOX},
  {MH_FILTER_BOX,MH_FILTER_BOX},{MH_FILTER_TRIANGLE,MH_FILTER_BOX},
  {MH_FILTER_HERMITE,MH_FILTER_BOX},{MH_FILTER_SINCFAST,MH_FILTER_HANN},
  {MH_FILTER_SINCFAST,MH_FILTER_HAMMING},{MH_FILTER_SINCFAST,MH_FILTER_BLACKMAN},
  {MH_FILTER_GAUSSIAN,MH_FILTER_BOX},{MH_FILTER_QUADRATIC,MH_FILTER_BOX},
  {MH_FILTER_CUBIC,MH_FILTER_BOX},{MH_FILTER_CATROM,MH_FILTER_BOX},
  {MH_FILTER_MITCHELL,MH_FILTER_BOX},{MH_FILTER_JINC,MH_FILTER_BOX},
  {MH_FILTER_SINC,MH_FILTER_BOX},{MH_FILTER_SINCFAST,MH_FILTER_BOX},
  {MH_FILTER_SINCFAST,MH_FILTER_KAISER},{MH_FILTER_LANCZOS,MH_FILTER_WELCH},
  {MH_FILTER_SINCFAST,MH_FILTER_CUBIC},{MH_FILTER_SINCFAST,MH_FILTER_BOHMAN},
  {MH_FILTER_SINCFAST,MH_FILTER_TRIANGLE},{MH_FILTER_LAGRANGE,MH_FILTER_BOX},
  {MH_FILTER_LANCZOS,MH_FILTER_LANCZOS},{MH_FILTER_LANCZOSSHARP,MH_FILTER_LANCZOSSHARP},
  {MH_FILTER_LANCZOS2,MH_FILTER_LANCZOS2},{MH_FILTER_LANCZOS2SHARP,MH_FILTER_LANCZOS2SHARP},
  {MH_FILTER_ROBIDOUX,MH_FILTER_BOX},{MH_FILTER_ROBIDOUXSHARP,MH_FILTER_BOX},
  {MH_FILTER_LANCZOS,MH_FILTER_COSINE},{MH_FILTER_SPLINE,MH_FILTER_BOX},
  {MH_FILTER_LANCZOSRADIUS,MH_FILTER_LANCZOS},{MH_FILTER_CUBICSPLINE,MH_FILTER_BOX},
  {MH_FILTER_MAGICKERNELSHARP2013,MH_FILTER_BOX},{MH_FILTER_MAGICKERNELSHARP2021,MH_FILTER_BOX}
};

} // namespace

namespace mh {

void build_tap_table(TapTable &table,const MhResizeFilter *filter,size_t in_size,
  size_t out_size,double factor)
{
  // resize.c:3364-3377 (horizontal) / :3580-3593 (vertical)
  double scale=1.0/factor+kEpsilon;
  if (scale < 1.0)
    scale=1.0;
  double support=scale*MhGetResizeFilterSupport(filter);
  if (support < 0.5)
    {
      support=0.5;
      scale=1.0;
    }
  const int capacity=(int) (2.0*support+3.0);
  scale=perceptible_reciprocal(scale);
  table.out_size=(int) out_size;
  table.start.assign(out_size,0);
  table.count.assign(out_size,0);
  table.nearest.assign(out_size,0);
  // One output index is independent of the next and the weighting function is pure (the
  // reference evaluates it from its OpenMP threads, resize.c:3398-3400), so long tables are
  // built by a few host threads, straight into the transposed [tap][out] layout: a
  // 32768-entry Lanczos table costs ~3 ms on one core, which would otherwise exceed the GPU
  // time of the pass it feeds.
  const size_t stride=(size_t) capacity+4;
  table.weight.assign(stride*out_size,0.0);
  double *weights=table.weight.data();
  auto build_range=[&](size_t x0,size_t x1,int *range_max)
  {
    std::vector<double> w(stride);
    int local_max=0;
    for (size_t x=x0; x < x1; x++)
      {
        // resize.c:3418-3443
        double bisect=(double) ((double) x+0.5)/factor+kEpsilon;
        double lo=bisect-support+0.5;
        if (lo < 0.0)
          lo=0.0;
        double hi=bisect+support+0.5;
        if (hi > (double) in_size)
          hi=(double) in_size;
        ptrdiff_t start=(ptrdiff_t) lo,stop=(ptrdiff_t) hi;
        ptrdiff_t n=stop-start;
        if (n < 0)
          n=0;
        if ((size_t) n > stride)
          n=(ptrdiff_t) stride;          // cannot happen: `capacity` bounds the span
        double density=0.0;
        for (ptrdiff_t i=0; i < n; i++)
          {
            w[(size_t) i]=MhGetResizeFilterWeight(filter,scale*((double) (start+i)-bisect+0.5));
            density+=w[(size_t) i];
          }
        if ((n > 0) && (density != 0.0) && (density != 1.0))
          {
            density=perceptible_reciprocal(density);
            for (ptrdiff_t i=0; i < n; i++)
              w[(size_t) i]*=density;
          }
        for (ptrdiff_t i=0; i < n; i++)
          weights[(size_t) i*out_size+x]=w[(size_t) i];
        table.start[x]=(int) start;
        table.count[x]=(int) n;
        if (n > 0)
          {
            // Copy-trait source index, resize.c:3484-3485
            double j=bisect;
            if (j < (double) start)
              j=(double) start;
            if (j > (double) stop-1.0)
              j=(double) stop-1.0;
            table.nearest[x]=(int) (ptrdiff_t) (j+0.5);
          }
        if ((int) n > local_max)
          local_max=(int) n;
      }
    *range_max=local_max;
  };
  // ~12 ns per weight on one core; a thread is worth starting for ~16k weights
  size_t workers=out_size*(size_t) capacity/16384;
  const size_t hw=std::thread::hardware_concurrency();
  if (workers > 8)
    workers=8;
  if ((hw != 0) && (workers > hw))
    workers=hw;
  if (workers < 1)
    workers=1;
  std::vector<int> range_max(workers,0);
  if (workers == 1)
    build_range(0,out_size,&range_max[0]);
  else
    {
      std::vector<std::thread> pool;
      for (size_t t=1; t < workers; t++)
        pool.emplace_back(build_range,out_size*t/workers,out_size*(t+1)/workers,&range_max[t]);
      build_range(0,out_size/workers,&range_max[0]);
      for (std::thread &t : pool)
        t.join();
    }
  int max_taps=0;
  for (int m : range_max)
    max_taps=m > max_taps ? m : max_taps;
  table.max_taps=max_taps;
  table.weight.resize((size_t) max_taps*out_size);     // drops the all-zero tail rows
}

// Contribution tables depend only on the filter's parameters and the two sizes; callers
// that resize many equally sized images (a thumbnail batch, the benchmark) would rebuild the
// same table for every image.  A small most-recently-used cache keeps the last few; filters
// that weigh through a caller's callback are never cached (their identity is opaque).
namespace {
struct TapKey
{
  int filter_fn,window_fn;
  double support,window_support,scale,blur,coefficient[7],factor;
  size_t in_size,out_size;
  bool operator==(const TapKey &o) const { return memcmp(this,&o,sizeof(TapKey)) == 0; }
};
struct TapCache
{
  std::mutex lock;
  std::vector<std::pair<TapKey,std::shared_ptr<const TapTable>>> entries;   // front = most recent
};
TapCache &tap_cache() { static TapCache *c=new TapCache(); return *c; }
constexpr size_t kTapCacheEntries=8;
}

std::shared_ptr<const TapTable> acquire_tap_table(const MhResizeFilter *filter,size_t in_size,
  size_t out_size,double factor)
{
  if (filter->callback != nullptr)
    {
      auto table=std::make_shared<TapTable>();
      build_tap_table(*table,filter,in_size,out_size,factor);
      return table;
    }
  TapKey key;
  memset(&key,0,sizeof(key));               // padding bytes take part in the comparison
  key.filter_fn=filter->filter_fn;
  key.window_fn=filter->window_fn;
  key.support=filter->support;
  key.window_support=filter->window_support;
  key.scale=filter->scale;
  key.blur=filter->blur;
  for (int i=0; i < 7; i++)
    key.coefficient[i]=filter->coefficient[i];
  key.factor=factor;
  key.in_size=in_size;
  key.out_size=out_size;
  TapCache &cache=tap_cache();
  {
    std::lock_guard<std::mutex> guard(cache.lock);
    for (size_t i=0; i < cache.entries.size(); i++)
      if (cache.entries[i].first == key)
        {
          auto hit=cache.entries[i];
          cache.entries.erase(cache.entries.begin()+(ptrdiff_t) i);
          cache.entries.insert(cache.entries.begin(),hit);
          return hit.second;
        }
  }
  auto table=std::make_shared<TapTable>();
  build_tap_table(*table,filter,in_size,out_size,factor);
  static std::atomic<unsigned long long> next_serial{1};
  table->serial=next_serial.fetch_add(1);
  std::lock_guard<std::mutex> guard(cache.lock);
  cache.entries.insert(cache.entries.begin(),{key,table});
  if (cache.entries.size() > kTapCacheEntries)
    cache.entries.pop_back();
  return table;
}

} // namespace mh

extern "C" {

MH_API MhResizeFilter *MhAcquireResizeFilter(MhFilterType filter,int /*hint*/)
{
  if ((filter <= MH_FILTER_UNDEFINED) || (filter >= MH_FILTER_SENTINEL))
    return nullptr;
  MhFilterType filter_type=kMapping[filter].filter;
  MhFilterType window_type=kMapping[filter].window;
  if ((kFilters[filter_type].fn == FN_JINC) || (kFilters[window_type].fn == FN_JINC))
    {
      mh::set_error("Jinc filters are cylindrical-only and not built here");
      return nullptr;
    }
  MhResizeFilter *f=new MhResizeFilter();
  f->blur=1.0;
  f->filter_fn=kFilters[filter_type].fn;
  f->support=kFilters[filter_type].support;
  f->window_fn=kFilters[window_type].fn;
  f->scale=kFilters[window_type].scale;
  switch (filter_type)
  {
    case MH_FILTER_LANCZOSSHARP: f->blur*=0.9812505644269356; break;
    case MH_FILTER_LANCZOS2SHARP: f->blur*=0.9549963639785485; break;
    default: break;
  }
  if ((f->filter_fn == FN_GAUSSIAN) || (f->window_fn == FN_GAUSSIAN))
    {
      const double value=0.5;
      f->coefficient[0]=value;
      f->coefficient[1]=perceptible_reciprocal(2.0*value*value);
      f->coefficient[2]=perceptible_reciprocal(6.28318530717958647692528676655900576839433879875020*value*value);
    }
  if ((f->filter_fn == FN_KAISER) || (f->window_fn == FN_KAISER))
    {
      const double value=6.5;
      f->coefficient[0]=value;
      f->coefficient[1]=perceptible_reciprocal(bessel_i0(value));
    }
  if (f->blur < kEpsilon)
    f->blur=kEpsilon;
  f->window_support=f->support;
  f->scale*=perceptible_reciprocal(f->window_support);
  if ((f->filter_fn == FN_CUBICBC) || (f->window_fn == FN_CUBICBC))
    {
      double B=kFilters[filter_type].B,C=kFilters[filter_type].C;
      if (kFilters[window_type].fn == FN_CUBICBC)
        {
          B=kFilters[window_type].B;
          C=kFilters[window_type].C;
        }
      const double twoB=B+B;
      f->coefficient[0]=1.0-(1.0/3.0)*B;
      f->coefficient[1]=-3.0+twoB+C;
      f->coefficient[2]=2.0-1.5*B-C;
      f->coefficient[3]=(4.0/3.0)*B+4.0*C;
      f->coefficient[4]=-8.0*C-twoB;
      f->coefficient[5]=B+5.0*C;
      f->coefficient[6]=(-1.0/6.0)*B-C;
    }
  return f;
}

MH_API MhResizeFilter *MhDestroyResizeFilter(MhResizeFilter *filter)
{
  delete filter;
  return nullptr;
}

MH_API MhResizeFilter *MhAcquireResizeFilterFromCallback(MhResizeWeightFunction weight,
  void *user,double support)
{
  if (weight == nullptr)
    return nullptr;
  MhResizeFilter *f=new MhResizeFilter();
  f->callback=weight;
  f->callback_user=user;
  f->callback_support=support;
  return f;
}

MH_API double MhGetResizeFilterSupport(const MhResizeFilter *filter)
{
  if (filter->callback != nullptr)
    return filter->callback_support;
  return filter->support*filter->blur;
}

MH_API double MhGetResizeFilterWeight(const MhResizeFilter *filter,double x)
{
  if (filter->callback != nullptr)
    return filter->callback(filter->callback_user,x);
  double x_blur=fabs((double) x)*perceptible_reciprocal(filter->blur);
  double scale;
  if ((filter->window_support < kEpsilon) || (filter->window_fn == FN_BOX))
    scale=1.0;
  else
    {
      scale=filter->scale;
      scale=evaluate(filter->window_fn,x_blur*scale,filter);
    }
  return scale*evaluate(filter->filter_fn,x_blur,filter);
}

} // extern "C"
