// Operator-level entry points: the host-side mirror of the reference's
// accelerate interface (MagickCore/accelerate-private.h:36-63) plus the hooks
// the reference lacks (convolve/morphology, colourspace).  Host control flow
// (kernel lists, compound stages, iteration, pass ordering, LUT construction)
// stays on the CPU exactly as in the reference; every per-pixel loop runs in a
// HIP kernel.
#include "mh_internal.hpp"

#include <cstdio>
#include "resize_filter.hpp"

#include <atomic>
#include <cmath>
#include <cstring>
#include <memory>

namespace mh {

// the accelerate gate, checkAccelerateCondition accelerate.c:110-170, applied
// to what the descriptor can express
static MhStatus gate(const MhImage *image,const char *what)
{
  MH_TRY(runtime_ready());
  MH_TRY(validate_image(image,what));
  set_call_precision(image);       // MhImage::precision of this call, else the library default
  return MH_OK;
}

static MhStatus gate_pair(const MhImage *image,const MhImage *out,const char *what,
  bool same_geometry)
{
  MH_TRY(gate(image,what));
  MH_TRY(validate_image(out,what));
  if ((out->number_channels != image->number_channels) ||
      (out->quantum != image->quantum))
    return fail(MH_BAD_ARGUMENT,"%s: destination layout differs from source",what);
  if (same_geometry && ((out->columns != image->columns) || (out->rows != image->rows)))
    return fail(MH_BAD_ARGUMENT,"%s: destination geometry differs from source",what);
  if (out->pixels == image->pixels)
    return fail(MH_BAD_ARGUMENT,"%s: destination aliases source",what);
  return MH_OK;
}

// Both images of an operator run on one device and one stream.
struct Pair
{
  DeviceGuard guard;         // declared first: the device is restored after src/dst are gone
  Resident src,dst;
  MhStatus open(const MhImage *image,MhImage *out)
  {
    int device=resolve_device(image->memory == MH_MEMORY_DEVICE ? image : out);
    hipStream_t stream;
    if (image->memory == MH_MEMORY_DEVICE)
      stream=(hipStream_t) image->stream;
    else if (out->memory == MH_MEMORY_DEVICE)
      stream=(hipStream_t) out->stream;
    else
      stream=library_stream(device);
    MH_TRY(src.open(image,0,stream,device));
    MH_TRY(dst.open(out,1,stream,device));
    src.view.stream=stream;
    dst.view.stream=stream;
    src.view.device=device;
    dst.view.device=device;
    MH_HIP(guard.enter(device));
    return MH_OK;
  }
  MhStatus commit()
  {
    MH_TRY(dst.commit());
    MH_TRY(src.commit());
    return MH_OK;
  }
};

static bool kernel_has_nan(const MhKernelInfo *k)
{
  for (size_t i=0; i < k->width*k->height; i++)
    if (std::isnan(k->values[i]))
      return true;
  return false;
}

// k[y][x] = column[y]*row[x] to within rounding?  (Gaussian:RxS, Square/Rectangle/Unity scaled
// kernels, outer products in general.)  The pivot is the largest cell; row = its kernel row,
// column = its kernel column divided by the pivot.
static bool rank_one_factors(const MhKernelInfo *k,std::vector<double> &row,std::vector<double> &column)
{
  const size_t w=k->width,h=k->height;
  size_t pivot=0;
  double largest=0.0;
  for (size_t i=0; i < w*h; i++)
    if (std::fabs(k->values[i]) > largest)
      {
        largest=std::fabs(k->values[i]);
        pivot=i;
      }
  if (!(largest > 0.0) || !std::isfinite(largest))
    return false;
  const size_t py=pivot/w,px=pivot % w;
  row.assign(k->values+py*w,k->values+(py+1)*w);
  column.resize(h);
  for (size_t y=0; y < h; y++)
    column[y]=k->values[y*w+px]/k->values[pivot];
  for (size_t y=0; y < h; y++)
    for (size_t x=0; x < w; x++)
      if (std::fabs(k->values[y*w+x]-column[y]*row[x]) > 1.0e-13*largest)
        return false;
  return true;
}

// ... or an outer product everywhere but at its origin cell: SharpenImage's negated Gaussian and
// EdgeImage's box of -1, whose centre carries the normalisation (effect.c:3640-3660, :1530-1545).
// The pivot is the largest cell off the origin's row and column, so that neither factor is read
// from the odd cell; delta = what the origin cell holds beyond column[y]*row[x].
static bool rank_one_plus_delta(const MhKernelInfo *k,std::vector<double> &row,std::vector<double> &column,
  double *delta)
{
  const size_t w=k->width,h=k->height;
  const size_t ox=(size_t) k->x,oy=(size_t) k->y;
  if ((w < 3) || (h < 3) || (k->x < 0) || (k->y < 0) || (ox >= w) || (oy >= h))
    return false;
  size_t pivot=0;
  double largest=0.0;
  for (size_t y=0; y < h; y++)
    for (size_t x=0; x < w; x++)
      if ((y != oy) && (x != ox) && (std::fabs(k->values[y*w+x]) > largest))
        {
          largest=std::fabs(k->values[y*w+x]);
          pivot=y*w+x;
        }
  if (!(largest > 0.0) || !std::isfinite(largest) || !std::isfinite(k->values[oy*w+ox]))
    return false;
  const size_t py=pivot/w,px=pivot % w;
  row.assign(k->values+py*w,k->values+(py+1)*w);
  column.resize(h);
  for (size_t y=0; y < h; y++)
    column[y]=k->values[y*w+px]/k->values[pivot];
  double scale=largest;
  for (size_t i=0; i < w*h; i++)
    if ((i != oy*w+ox) && (std::fabs(k->values[i]) > scale))
      scale=std::fabs(k->values[i]);
  for (size_t y=0; y < h; y++)
    for (size_t x=0; x < w; x++)
      if (((y != oy) || (x != ox)) && !(std::fabs(k->values[y*w+x]-column[y]*row[x]) <= 1.0e-13*scale))
        return false;
  *delta=k->values[oy*w+ox]-column[oy]*row[ox];
  return true;
}

extern "C" MH_API int MhKernelOuterProductFactors(const MhKernelInfo *kernel,double *row,double *column)
{
  if ((kernel == nullptr) || (kernel->values == nullptr) || (kernel->width == 0) || (kernel->height == 0))
    return 0;
  for (size_t i=0; i < kernel->width*kernel->height; i++)
    if (std::isnan(kernel->values[i]))
      return 0;
  std::vector<double> r,c;
  if (!rank_one_factors(kernel,r,c))
    return 0;
  if (row != nullptr)
    std::memcpy(row,r.data(),r.size()*sizeof(double));
  if (column != nullptr)
    std::memcpy(column,c.data(),c.size()*sizeof(double));
  return 1;
}

// ... and the form with one odd cell at the origin: returns 1 (outer product), 2 (outer product +
// *delta at the origin cell) or 0; what convolve_separable.hip takes
extern "C" MH_API int MhKernelOuterProductPlusDelta(const MhKernelInfo *kernel,double *row,double *column,
  double *delta)
{
  if ((kernel == nullptr) || (kernel->values == nullptr) || (kernel->width == 0) || (kernel->height == 0))
    return 0;
  for (size_t i=0; i < kernel->width*kernel->height; i++)
    if (std::isnan(kernel->values[i]))
      return 0;
  std::vector<double> r,c;
  double d=0.0;
  int kind=0;
  if (rank_one_factors(kernel,r,c))
    kind=1;
  else if (rank_one_plus_delta(kernel,r,c,&d))
    kind=2;
  if (kind == 0)
    return 0;
  if (row != nullptr)
    std::memcpy(row,r.data(),r.size()*sizeof(double));
  if (column != nullptr)
    std::memcpy(column,c.data(),c.size()*sizeof(double));
  if (delta != nullptr)
    *delta=d;
  return kind;
}

// FAST precision, Q16, a 2-D Convolve kernel that is an outer product: two 1-D passes over
// float sums instead of width*height taps per pixel (GaussianBlurImage 0x10: 79+79 instead of
// 6241).  Same window, same edge clamp (clamping is per axis) and one division at the end as in
// morphology.c:2892-2979; the float intermediate keeps the result within +-1 level.
// *handled = false: not this case, nothing launched.
static MhStatus separable_convolve(const View &src,const View &dst,const MhKernelInfo *kernel,
  const Roles &roles,bool *handled)
{
  *handled=false;
  if ((precision() != MH_PRECISION_FAST) || (src.quantum != MH_QUANTUM_U16) ||
      (kernel->width < 2) || (kernel->height < 2) || (roles.copy_mask != 0) ||
      (src.channels < 1) || (src.channels > 4) || (option("MAGICKHIP_NO_SEPARABLE") != nullptr))
    return MH_OK;
  const bool blend=roles.blend && (roles.alpha == src.channels-1) &&
    ((src.channels == 2) || (src.channels == 4));
  if (roles.blend && !blend)
    return MH_OK;
  std::vector<double> row,column;
  if (!rank_one_factors(kernel,row,column))
    return MH_OK;
  {
    // float / f16 intermediates are good to 2^-21 of sum|k|*65535: within a level for kernels
    // that average (sum|k| about 1), not for a derivative kernel with a gain of 96 — and the f16
    // planes of the matrix-core passes are scaled for sums below 2*65535
    double magnitude=0.0;
    for (size_t i=0; i < kernel->width*kernel->height; i++)
      magnitude+=std::fabs(kernel->values[i]);
    if (!(magnitude <= 8.0))
      return MH_OK;
  }
  if (blend)
    {
      // alpha-weighted sums with cells of both signs (Sobel ...): sum(k*alpha) may vanish, the
      // reference's reciprocal clamp then decides the pixel — the generic fp64 kernel's job
      bool positive=false,negative=false;
      for (size_t i=0; i < kernel->width*kernel->height; i++)
        {
          positive=positive || (kernel->values[i] > 0.0);
          negative=negative || (kernel->values[i] < 0.0);
        }
      if (positive && negative)
        return MH_OK;
    }
  // Small kernels: the two separated passes cost what they cost (0.21 ms per 4096^2 frame, three frame transfers)
  // whatever the kernel; the w x h sum in one launch is cheaper up to about 13 x 13 (convolve2d_mfma.hip: 5 x 5 0.10
  // ms, 7 x 7 0.12, 9 x 9 0.13 on the same frame, +-1 level like this route), and below 5 x 5 so is the generic
  // kernel, bit-identical (3 x 3: 0.11-0.20 ms against 0.21; tools/probe_small_separable.py,
  // profiles/r6_notes/small_separable_kernels.txt).
  if (kernel->width*kernel->height < 25)
    return MH_OK;
  if ((kernel->width <= 13) && (kernel->height <= 13) && ((src.channels == 4) || ((src.channels == 3) && !blend)) &&
      (!blend || (roles.alpha == 3)) && (option("MAGICKHIP_NO_MFMA") == nullptr) &&
      (option("MAGICKHIP_NO_MFMA_2D") == nullptr) && (option("MAGICKHIP_SEPARABLE_SMALL") == nullptr))
    {
      MH_TRY(launch_conv2d_mfma(src,dst,kernel,blend,handled));
      if (*handled)
        return MH_OK;
    }
  if ((src.channels == 4) || ((src.channels == 3) && !blend))
    {
      // both passes on the matrix cores: Quantum pixels -> float sums -> Quantum pixels (four
      // floats per pixel; the fourth is zero for RGB)
      View sums=src;
      sums.quantum=MH_QUANTUM_F32;
      sums.channels=4;
      Temp memory;
      MH_TRY(memory.alloc(src.device,sums.bytes(),src.stream));
      sums.pixels=memory.ptr;
      Conv1DParams horizontal,vertical;
      horizontal.taps=row.data();
      horizontal.ntaps=(int) kernel->width;
      horizontal.origin=(int) kernel->x;
      vertical.taps=column.data();
      vertical.ntaps=(int) kernel->height;
      vertical.origin=(int) kernel->y;
      bool first=false,second=false;
      MH_TRY(launch_conv1d_sums(src,sums,false,horizontal,blend,&first));
      if (first)
        {
          MH_TRY(launch_conv1d_sums(sums,dst,true,vertical,blend,&second));
          if (second)
            {
              *handled=true;
              return MH_OK;
            }
        }
    }
  View sums=src,work=src;
  sums.quantum=MH_QUANTUM_F32;
  work.quantum=MH_QUANTUM_F32;
  Temp sums_memory,work_memory;
  MH_TRY(sums_memory.alloc(src.device,sums.bytes(),src.stream));
  MH_TRY(work_memory.alloc(src.device,work.bytes(),src.stream));
  sums.pixels=sums_memory.ptr;
  work.pixels=work_memory.ptr;
  Roles plain;
  plain.update_mask=(1u << src.channels)-1u;
  Conv1DParams horizontal,vertical;
  horizontal.taps=row.data();
  horizontal.ntaps=(int) kernel->width;
  horizontal.origin=(int) kernel->x;
  vertical.taps=column.data();
  vertical.ntaps=(int) kernel->height;
  vertical.origin=(int) kernel->y;
  MH_TRY(launch_premultiply(src,sums,blend));
  MH_TRY(launch_conv1d(sums,work,false,horizontal,plain,MH_PRECISION_FAST,nullptr));
  MH_TRY(launch_conv1d(work,sums,true,vertical,plain,MH_PRECISION_FAST,nullptr));
  MH_TRY(launch_separable_finish(sums,dst,blend));
  *handled=true;
  return MH_OK;
}

// One MorphologyPrimitive(curr -> work), morphology.c:2566.  `changed`
// (optional, device counter, must be zero on entry) accumulates the number of
// changed channel values.
// mode: the precision of THIS primitive.  MH_PRECISION_FAST is the caller's choice only for a
// primitive whose result leaves the operator; one that feeds another primitive (the row pass of
// BlurImage, the first kernels of a list, all but the last iteration) runs EXACT, so that the
// Quantum-rounded intermediate is the reference's own and the FAST result is within +-1 level of
// the reference by construction.
static MhStatus primitive(const View &src,const View &dst,MhMorphologyMethod method,
  const MhKernelInfo *kernel,double bias,const Roles &roles,const MhImage *desc,
  unsigned long long *changed,MhPrecision mode)
{
  if ((method == MH_MORPHOLOGY_CONVOLVE) && !kernel_has_nan(kernel) &&
      ((kernel->width == 1) || (kernel->height == 1)))
    {
      Conv1DParams p;
      p.taps=kernel->values;
      p.bias=bias;
      if (kernel->width == 1)
        {
          p.ntaps=(int) kernel->height;
          p.origin=(int) kernel->y;
          return launch_conv1d(src,dst,true,p,roles,mode,changed);
        }
      p.ntaps=(int) kernel->width;
      p.origin=(int) kernel->x;
      return launch_conv1d(src,dst,false,p,roles,mode,changed);
    }
  // A 2-D Convolve of a one-channel (gray) Q16 frame: the frame's rows as four bands = the four independent
  // channels of a frame a quarter as tall (gray_bands_pack_kernel, pointwise.hip; the kernel's reach in extra rows
  // between bands), so that the wide-pixel kernels below take it — the matrix-core forms, the separated passes.
  // Every channel of a frame without alpha weighting is summed on its own (morphology.c:2892-2979): the result is
  // the frame's own under the same contract.
  if ((method == MH_MORPHOLOGY_CONVOLVE) && (src.quantum == MH_QUANTUM_U16) && (src.channels == 1) &&
      !roles.blend && (roles.copy_mask == 0) && (kernel->width >= 2) && (kernel->height >= 2) &&
      (option("MAGICKHIP_NO_GRAY_BANDS") == nullptr))
    {
      const int above=(int) kernel->y,below=(int) kernel->height-1-(int) kernel->y;
      const int halo=above > below ? above : below;
      if ((kernel->y >= 0) && ((size_t) kernel->y < kernel->height) && GrayBands::fits(src,halo,1l << 23))
        {
          GrayBands bands;
          MH_TRY(bands.pack(src,halo));
          MH_TRY(primitive(bands.packed,bands.result,method,kernel,bias,GrayBands::plain_roles(),desc,nullptr,mode));
          return bands.unpack(dst,src.pixels,changed);
        }
    }
  if ((method == MH_MORPHOLOGY_CONVOLVE) && (changed == nullptr) && (bias == 0.0) &&
      (mode == MH_PRECISION_FAST) && !kernel_has_nan(kernel))
    {
      bool handled=false;
      MH_TRY(separable_convolve(src,dst,kernel,roles,&handled));
      if (handled)
        return MH_OK;
    }
  // Q16, RGBA (alpha-weighted, alpha last), four plain channels or RGB, cells that are integer
  // multiples of a unit (flat shapes, integer kernels; NaN cells are fine), 5 x 5 and more: exact
  // sums on the i8 matrix cores, bit-identical in either mode (convolve2d_exact.hip)
  // float Quantum, the same layouts and kernels: a frame whose samples are integers of 0..65535 (what
  // an 8- or 16-bit file decodes to) has the same exact sums; the kernel finds out while it stages
  // the frame, and the generic kernel behind it runs only if it was not
  if ((method == MH_MORPHOLOGY_CONVOLVE) && (changed == nullptr) && (bias == 0.0) &&
      (src.quantum == MH_QUANTUM_F32) && ((src.channels == 4) || ((src.channels == 3) && !roles.blend)) &&
      (roles.copy_mask == 0) && (!roles.blend || (roles.alpha == 3)) &&
      (option("MAGICKHIP_NO_MFMA") == nullptr) && (option("MAGICKHIP_NO_MFMA_2D") == nullptr) &&
      (option("MAGICKHIP_NO_EXACT_2D") == nullptr) && (option("MAGICKHIP_NO_EXACT_2D_FLOAT") == nullptr))
    {
      // (outer products — boxes — have the separable path below, which takes any float frame)
      std::vector<double> row,column;
      double delta=0.0;
      bool handled=false;
      Temp flag;
      if (kernel_has_nan(kernel) || (!rank_one_factors(kernel,row,column) && !rank_one_plus_delta(kernel,row,column,&delta)))
        MH_TRY(launch_conv2d_exact(src,dst,kernel,roles.blend,&handled,&flag));
      if (handled)
        {
          bool fused=false;
          MH_TRY(launch_conv2d_tie(src,dst,kernel,roles,&fused,flag.as<unsigned>()));
          if (fused)
            return MH_OK;
          Morph2DParams fallback;
          fallback.method=method;
          fallback.kernel=kernel;
          fallback.bias=bias;
          fallback.intensity=desc->intensity != 0 ? (MhIntensityMethod) desc->intensity : MH_INTENSITY_REC709LUMA;
          fallback.colorspace=(MhColorspace) desc->colorspace;
          fallback.only_if=flag.as<unsigned>();
          return launch_morph2d(src,dst,fallback,roles,nullptr);
        }
    }
  const bool matrix_2d=(method == MH_MORPHOLOGY_CONVOLVE) && (changed == nullptr) && (bias == 0.0) &&
    (src.quantum == MH_QUANTUM_U16) && ((src.channels == 4) || ((src.channels == 3) && !roles.blend)) &&
    (roles.copy_mask == 0) && (!roles.blend || (roles.alpha == 3)) &&
    (option("MAGICKHIP_NO_MFMA") == nullptr) && (option("MAGICKHIP_NO_MFMA_2D") == nullptr);
  // (FAST, narrow kernels: the f16 kernel's band is one 32-slot chunk up to 17 cells where the
  // integer one always multiplies two, on four byte planes of alpha-weighted samples — Octagon:8 on
  // 16384^2 RGBA 3.0 against 4.0 ms; with the two planes of plain samples the integer kernel is the
  // faster one from 11 cells on (four channels: Octagon:5 2.05 against 2.13 ms, Octagon:3 1.87
  // against 1.68) and never the slower one on RGB.  Wider: Disk:15 RGBA 6.4 against 8.3 ms.)
  if (matrix_2d && (mode == MH_PRECISION_FAST) &&
      (roles.blend ? (kernel->width <= 17) : ((src.channels == 4) && (kernel->width <= 9))))
    {
      bool handled=false;
      MH_TRY(launch_conv2d_mfma(src,dst,kernel,roles.blend,&handled));
      if (handled)
        return MH_OK;
    }
  if (matrix_2d && (option("MAGICKHIP_NO_EXACT_2D") == nullptr))
    {
      bool handled=false;
      MH_TRY(launch_conv2d_exact(src,dst,kernel,roles.blend,&handled));
      if (handled)
        return MH_OK;
    }
  // an outer-product kernel in EXACT mode, or on float Quantum in either mode: two fp64 passes and
  // a tie check, bit-identical to the w x h walk (convolve_separable.hip)
  if ((method == MH_MORPHOLOGY_CONVOLVE) && (changed == nullptr) && (bias == 0.0) &&
      !kernel_has_nan(kernel) && (kernel->width >= 2) && (kernel->height >= 2) &&
      (kernel->width*kernel->height >= 25) && (option("MAGICKHIP_NO_SEPARABLE") == nullptr))
    {
      std::vector<double> row,column;
      double delta=0.0;
      if (rank_one_factors(kernel,row,column))
        {
          bool handled=false;
          MH_TRY(launch_separable_exact(src,dst,kernel,row.data(),column.data(),roles,&handled));
          if (handled)
            return MH_OK;
        }
      else if (rank_one_plus_delta(kernel,row,column,&delta))
        {
          bool handled=false;
          MH_TRY(launch_separable_exact(src,dst,kernel,row.data(),column.data(),roles,&handled,
            (int) kernel->x,(int) kernel->y,delta));
          if (handled)
            return MH_OK;
        }
    }
  // FAST, Q16, RGBA (alpha-weighted, alpha last), four plain channels or RGB, kernels of 5 x 5 and
  // more: the w x h sum as h banded products on the matrix cores (convolve2d_mfma.hip)
  if (matrix_2d && (mode == MH_PRECISION_FAST))
    {
      bool handled=false;
      MH_TRY(launch_conv2d_mfma(src,dst,kernel,roles.blend,&handled));
      if (handled)
        return MH_OK;
    }
  // what is left of Convolve — cells that are no outer product and no integer multiples, float frames
  // that are not made of integers, one- and two-channel layouts, EXACT mode: fused multiply-adds over
  // premultiplied doubles and a tie check, bit-identical (convolve2d_tie.hip)
  if ((method == MH_MORPHOLOGY_CONVOLVE) && (changed == nullptr) && (bias == 0.0))
    {
      bool handled=false;
      MH_TRY(launch_conv2d_tie(src,dst,kernel,roles,&handled));
      if (handled)
        return MH_OK;
    }
  Morph2DParams p;
  p.method=method;
  p.kernel=kernel;
  p.bias=bias;
  p.intensity=desc->intensity != 0 ? (MhIntensityMethod) desc->intensity :
    MH_INTENSITY_REC709LUMA;
  p.colorspace=(MhColorspace) desc->colorspace;
  return launch_morph2d(src,dst,p,roles,changed);
}

// Device images MorphologyApply juggles besides the (read-only) input: the caller's
// destination plus up to three pooled scratch images, handed out in a preference order so
// that simple chains end in the destination without a final copy.
struct MorphologyWorkspace
{
  static constexpr int kSlots=4;
  View slot[kSlots];
  Temp scratch[kSlots];
  bool busy[kSlots]={false,false,false,false};
  int order[kSlots]={0,1,2,3};
  MorphologyWorkspace(const View &dst,bool destination_first)
  {
    for (int i=0; i < kSlots; i++)
      {
        slot[i]=dst;
        if (i != 0)
          slot[i].pixels=nullptr;
      }
    if (!destination_first)
      {
        order[0]=1;
        order[1]=0;
      }
  }
  MhStatus acquire(const View **out)
  {
    for (int k=0; k < kSlots; k++)
      {
        const int i=order[k];
        if (busy[i])
          continue;
        if (slot[i].pixels == nullptr)
          {
            MH_TRY(scratch[i].alloc(slot[0].device,slot[0].bytes(),slot[0].stream));
            slot[i].pixels=scratch[i].ptr;
          }
        busy[i]=true;
        *out=&slot[i];
        return MH_OK;
      }
    return fail(MH_OUT_OF_MEMORY,"morphology: more than %d images in flight",kSlots);
  }
  void release(const View *v)
  {
    for (int i=0; i < kSlots; i++)
      if (v == &slot[i])
        busy[i]=false;
  }
};

static MhStatus fused_blur(const View &src,const View &dst,const MhKernelInfo *kernel,
  const Roles &roles,double bias,bool *handled,bool unsharp=false,double gain=0.0,double threshold=0.0);

// A one-channel (gray) Q16 frame through the same one-launch kernels: its rows as four bands = the four
// independent channels of a frame a quarter as tall (launch_gray_bands_pack, pointwise.hip), K-1 extra rows
// between bands.  Each channel of the one-launch kernels is computed on its own — the result is what the frame's
// own two passes give, +-1 in FAST and bit-identical in EXACT, like a four-channel frame's.  Frames from
// MAGICKHIP_GRAY_BANDS_MIN_PIXELS on (the pack and unpack launches are two more dispatches), whose bands are at
// least as tall as the rows added to them.
static MhStatus fused_blur_gray_bands(const View &src,const View &dst,const MhKernelInfo *kernel,
  double bias,bool *handled,bool unsharp,double gain,double threshold)
{
  *handled=false;
  const MhKernelInfo *column=kernel->next;
  if ((column == nullptr) || (column->width != 1) || (column->height < 2) || (column->y < 0) ||
      ((size_t) column->y >= column->height) ||
      (column->height > ((precision() == MH_PRECISION_FAST) && !unsharp ? 113u : 81u)))
    return MH_OK;
  const int K=(int) column->height;
  const int above=K-1-(int) column->y,below=(int) column->y;
  const int halo=above > below ? above : below;
  if (!GrayBands::fits(src,halo,1l << 23))
    return MH_OK;
  GrayBands bands;
  MH_TRY(bands.pack(src,halo));
  bool inner=false;
  MH_TRY(fused_blur(bands.packed,bands.result,kernel,GrayBands::plain_roles(),bias,&inner,unsharp,gain,threshold));
  if (!inner)
    return MH_OK;                                // (a kernel the one-launch forms decline: the frame's own passes)
  MH_TRY(bands.unpack(dst));
  *handled=true;
  return MH_OK;
}

// BlurImage's kernel list — a 1 x K row kernel followed by the same taps as a K x 1 column
// kernel (effect.c:765-796, "blur:RxS;blur:RxS+90") — on Q16 RGBA / RGB / four plain channels: both
// passes in one launch, the Quantum-rounded intermediate stays in LDS.  EXACT and UnsharpMaskImage:
// exact-integer sums in both passes, bit-identical (convolve_fused_exact.hip); FAST BlurImage: f16
// colour sums + exact alpha sums, +-1 by construction (convolve_fused_hybrid.hip).
// *handled = false: not this case, nothing launched.
static MhStatus fused_blur(const View &src,const View &dst,const MhKernelInfo *kernel,
  const Roles &roles,double bias,bool *handled,bool unsharp,double gain,double threshold)
{
  *handled=false;
  // (switches for tests and A/B runs, from the option table: the environment at start-up + MhSetOption)
  const bool no_mfma=option("MAGICKHIP_NO_MFMA") != nullptr;
  const bool no_fused=option("MAGICKHIP_NO_FUSED_BLUR") != nullptr;
  const bool exact=precision() == MH_PRECISION_EXACT;
  if ((src.quantum == MH_QUANTUM_U16) && (src.channels == 1) && !roles.blend && (roles.copy_mask == 0) &&
      (bias == 0.0) && !no_mfma && !no_fused && (option("MAGICKHIP_NO_GRAY_BANDS") == nullptr))
    return fused_blur_gray_bands(src,dst,kernel,bias,handled,unsharp,gain,threshold);
  if ((src.quantum != MH_QUANTUM_U16) || ((src.channels != 4) && (src.channels != 3)) ||
      (roles.copy_mask != 0) || (bias != 0.0) || no_mfma || no_fused)
    return MH_OK;
  if (roles.blend && ((roles.alpha != 3) || (src.channels != 4)))
    return MH_OK;
  const MhKernelInfo *row=kernel,*column=kernel->next;
  if ((column == nullptr) || (column->next != nullptr) || (row->height != 1) || (column->width != 1) ||
      (row->width != column->height) || (row->width < 2) || (row->x != column->y) ||
      (row->x < 0) || ((size_t) row->x >= row->width))
    return MH_OK;
  const int K=(int) row->width;
  for (int v=0; v < K; v++)
    {
      if (std::isnan(row->values[v]) || (row->values[v] != column->values[v]))
        return MH_OK;
      // alpha-weighted sums with taps of both signs stay on the fp64 kernels (launch_conv1d)
      if (roles.blend && (row->values[v] < 0.0))
        return MH_OK;
    }
  // Taps the reference zeroed (|t| < 1e-12, morphology.c:2494-2495: BlurImage with a radius far beyond
  // its sigma, "-blur 30x2") add +0.0 to sums of non-negative terms (:2746-2764, :2950-2971) and are
  // counted like any other tap (`count`, :2775): dropping the leading and trailing ones — a shorter
  // kernel with its origin moved — leaves every result bit for bit what it was.
  int lead=0,trail=0;
  while ((lead < K-1) && (row->values[K-1-lead] == 0.0))
    lead++;                                       // (the reversed walk: taps[v] = values[K-1-v])
  while ((trail < K-1-lead) && (row->values[trail] == 0.0))
    trail++;
  const int kept=K-lead-trail;
  const int shift=K-1-(int) row->x-lead;          // taps[v] multiplies the input at o-shift+v
  if ((kept < 2) || (shift < 0) || (shift >= kept))
    return MH_OK;
  std::vector<double> reversed((size_t) kept);
  for (int v=0; v < kept; v++)
    reversed[(size_t) v]=row->values[K-1-lead-v];
  // FAST BlurImage: f16 colour sums + exact alpha sums (convolve_fused_hybrid.hip), +-1 level
  // by construction at 0.6 of the all-exact row pass's matrix instructions.
  // FAST UnsharpMaskImage takes the all-exact kernel: a blurred sample one level off moves the
  // sharpened one by `gain` levels and flips the threshold test next to it — only the
  // reference's own blur keeps effect.c:4364-4369 within the +-1 contract (it is then
  // bit-identical).
  if (!exact && !unsharp)
    {
      MH_TRY(launch_blur_fused_hybrid(src,dst,reversed.data(),kept,shift,roles.blend,handled));
      if (*handled)
        return MH_OK;
      // Beyond the one-launch kernel's 81 taps (sigma > 10.3), frames WITHOUT alpha weighting: both passes on the f16
      // matrix cores with undivided f32 sums between them (convolve_mfma.hip MFMA_TO_SUMS / MFMA_FROM_SUMS, up to 113
      // taps) — the form FAST GaussianBlurImage takes.  The reference rounds the intermediate to a level
      // (morphology.c:2772-2776), which moves the value its column pass rounds by at most half a level (normalised
      // non-negative taps); the f16 terms add < 0.1: within one level, as the one-launch kernel's colour sums are.
      // (The general route runs the row pass on the fp64 kernels so that the intermediate is the reference's own:
      // 0.089 + 0.025 ms per 2048^2 frame at sigma 12 against 2 x 0.03 here.)  An alpha-weighted frame needs the
      // reference's intermediate ALPHA bit for bit and stays there.
      if (!roles.blend && (kept > 81) && (option("MAGICKHIP_NO_LONG_BLUR_SUMS") == nullptr))
        {
          bool normalised=true;
          double total=0.0;
          for (int v=0; v < kept; v++)
            {
              normalised=normalised && (reversed[(size_t) v] >= 0.0);
              total+=reversed[(size_t) v];
            }
          if (normalised && (total <= 1.0+1.0e-9))
            {
              View sums=src;
              sums.quantum=MH_QUANTUM_F32;
              sums.channels=4;
              Temp memory;
              MH_TRY(memory.alloc(src.device,sums.bytes(),src.stream));
              sums.pixels=memory.ptr;
              Conv1DParams horizontal,vertical;
              horizontal.taps=row->values;
              horizontal.ntaps=K;
              horizontal.origin=(int) row->x;
              vertical.taps=column->values;
              vertical.ntaps=K;
              vertical.origin=(int) column->y;
              bool first=false,second=false;
              MH_TRY(launch_conv1d_sums(src,sums,false,horizontal,false,&first));
              if (first)
                {
                  MH_TRY(launch_conv1d_sums(sums,dst,true,vertical,false,&second));
                  if (second)
                    {
                      *handled=true;
                      return MH_OK;
                    }
                }
            }
        }
    }
  // EXACT BlurImage of an alpha-weighted frame: the kernel may give the frame up (alpha of a few
  // levels everywhere: BlurExactArgs::give_up); the two fp64 passes queued behind it — bit-identical
  // too — then compute it, and leave at once otherwise.  MAGICKHIP_NO_GIVE_UP=1: never.  Without the
  // memory for those passes' intermediate the kernel runs unguarded (it is exact either way, only slow
  // on such frames).
  Temp give_up,rows_memory;
  bool guarded=exact && !unsharp && roles.blend && (option("MAGICKHIP_NO_GIVE_UP") == nullptr);
  if (guarded)
    {
      if ((give_up.alloc(src.device,sizeof(unsigned),src.stream) != MH_OK) ||
          (rows_memory.alloc(src.device,src.bytes(),src.stream) != MH_OK))
        guarded=false;
    }
  // "given up" = the word holds a value of this call's own, so that nothing has to clear it in front of the kernel
  // (a dispatch less a call).  Whatever the word held before: should it be this very value (one in 2^32), kernel
  // and passes agree that the frame was given up, and the passes compute it.
  static std::atomic<unsigned> tokens{1u};
  unsigned token=tokens.fetch_add(1u,std::memory_order_relaxed);
  if (token == 0u)
    token=tokens.fetch_add(1u,std::memory_order_relaxed);
  MH_TRY(launch_blur_fused_exact(src,dst,reversed.data(),kept,shift,roles.blend,handled,
    unsharp,gain,threshold,nullptr,guarded ? give_up.as<unsigned>() : nullptr,token));
  if (*handled && guarded)
    {
      View rows=src;
      rows.pixels=rows_memory.ptr;
      Conv1DParams first,second;
      first.taps=row->values;
      first.ntaps=K;
      first.origin=(int) row->x;
      first.only_if=give_up.as<unsigned>();
      first.only_if_token=second.only_if_token=token;
      second.taps=column->values;
      second.ntaps=K;
      second.origin=(int) column->y;
      second.only_if=give_up.as<unsigned>();
      MH_TRY(launch_conv1d(src,rows,false,first,roles,MH_PRECISION_EXACT,nullptr));
      MH_TRY(launch_conv1d(rows,dst,true,second,roles,MH_PRECISION_EXACT,nullptr));
    }
  // Neither form took it (taps of both signs on plain channels, taps below the certificate's floor,
  // more than 81 of them): the two passes of MorphologyApply (launch_conv1d).
  return MH_OK;
}

// MorphologyApply, morphology.c:3634-4077: the loops over method iterations, the kernel
// list, the stages of a compound method and the kernel iterations, with the
// CompositeImage post-steps (:3986-4013 Difference; :4016-4052 multi-kernel union) as
// device kernels.  Distance / Voronoi (MorphologyPrimitiveDirect) are not parallel.
static MhStatus morphology_apply(const View &src,const View &dst,const MhImage *desc,
  const Roles &roles,MhMorphologyMethod method,ptrdiff_t iterations,
  const MhKernelInfo *kernel,double bias,ptrdiff_t *changed_out,
  MhMorphologyCompose compose_override=MH_MORPHOLOGY_COMPOSE_DEFAULT)
{
  if (iterations == 0)
    return fail(MH_UNSUPPORTED,"morphology: zero iterations is a null operation");
  // (a kernel list whose results are COMPOSED instead of re-iterated is not BlurImage's list)
  const bool reiterate=(compose_override == MH_MORPHOLOGY_COMPOSE_DEFAULT) ||
    (compose_override == MH_MORPHOLOGY_COMPOSE_NONE);
  if ((method == MH_MORPHOLOGY_CONVOLVE) && (iterations == 1) && (changed_out == nullptr) &&
      (kernel->next != nullptr) && reiterate)
    {
      bool handled=false;
      MH_TRY(fused_blur(src,dst,kernel,roles,bias,&handled));
      if (handled)
        return MH_OK;
    }
  size_t kernel_limit=iterations < 0 ? (src.columns > src.rows ? src.columns : src.rows) :
    (size_t) iterations;
  size_t method_limit=1,stage_limit=1;
  int compose=-1;                                   // NoCompositeOp: re-iterate over the list
  bool need_reflected=false;
  switch (method)
  {
    case MH_MORPHOLOGY_CONVOLVE: case MH_MORPHOLOGY_ERODE: case MH_MORPHOLOGY_DILATE:
    case MH_MORPHOLOGY_ERODE_INTENSITY: case MH_MORPHOLOGY_DILATE_INTENSITY:
    case MH_MORPHOLOGY_ITERATIVE_DISTANCE: case MH_MORPHOLOGY_EDGE_IN: case MH_MORPHOLOGY_EDGE_OUT:
      break;
    case MH_MORPHOLOGY_CORRELATE:
      need_reflected=true;
      break;
    case MH_MORPHOLOGY_SMOOTH:
      stage_limit=4;
      need_reflected=true;
      break;
    case MH_MORPHOLOGY_OPEN: case MH_MORPHOLOGY_OPEN_INTENSITY: case MH_MORPHOLOGY_TOP_HAT:
    case MH_MORPHOLOGY_EDGE:
      stage_limit=2;
      break;
    case MH_MORPHOLOGY_CLOSE: case MH_MORPHOLOGY_CLOSE_INTENSITY: case MH_MORPHOLOGY_BOTTOM_HAT:
      stage_limit=2;
      need_reflected=true;
      break;
    case MH_MORPHOLOGY_HIT_AND_MISS:
      compose=MH_COMPOSITE_LIGHTEN;                 // union of the multi-kernel results
      method_limit=kernel_limit;
      kernel_limit=1;
      break;
    case MH_MORPHOLOGY_THINNING: case MH_MORPHOLOGY_THICKEN:
      method_limit=kernel_limit;                    // iterate the whole method
      kernel_limit=1;
      break;
    default:
      // Distance/Voronoi are sequential (MorphologyPrimitiveDirect)
      return fail(MH_UNSUPPORTED,"morphology method %d is not accelerated",(int) method);
  }
  // the user's morphology:compose (morphology.c:3779-3782, :4206-4215): how the results of the
  // kernels of a list are merged — overrides the method's default, nothing else
  switch (compose_override)
  {
    case MH_MORPHOLOGY_COMPOSE_DEFAULT: break;
    case MH_MORPHOLOGY_COMPOSE_NONE: compose=-1; break;
    case MH_MORPHOLOGY_COMPOSE_LIGHTEN: compose=MH_COMPOSITE_LIGHTEN; break;
    case MH_MORPHOLOGY_COMPOSE_DIFFERENCE: compose=MH_COMPOSITE_DIFFERENCE; break;
    case MH_MORPHOLOGY_COMPOSE_DARKEN: compose=MH_COMPOSITE_DARKEN; break;
    case MH_MORPHOLOGY_COMPOSE_PLUS: compose=MH_COMPOSITE_PLUS; break;
    case MH_MORPHOLOGY_COMPOSE_MULTIPLY: compose=MH_COMPOSITE_MULTIPLY; break;
    case MH_MORPHOLOGY_COMPOSE_SCREEN: compose=MH_COMPOSITE_SCREEN; break;
    case MH_MORPHOLOGY_COMPOSE_EXCLUSION: compose=MH_COMPOSITE_EXCLUSION; break;
    case MH_MORPHOLOGY_COMPOSE_MINUS_SRC: compose=MH_COMPOSITE_MINUS_SRC; break;
    case MH_MORPHOLOGY_COMPOSE_MINUS_DST: compose=MH_COMPOSITE_MINUS_DST; break;
    case MH_MORPHOLOGY_COMPOSE_LINEAR_DODGE: compose=MH_COMPOSITE_LINEAR_DODGE; break;
    case MH_MORPHOLOGY_COMPOSE_OVER: compose=MH_COMPOSITE_OVER; break;
    case MH_MORPHOLOGY_COMPOSE_DST_OVER: compose=MH_COMPOSITE_DST_OVER; break;
    default:
      return fail(MH_UNSUPPORTED,"morphology:compose operator %d is not accelerated",(int) compose_override);
  }
  std::unique_ptr<MhKernelInfo,MhKernelInfo *(*)(MhKernelInfo *)> reflected(nullptr,
    MhDestroyKernelInfo);
  if (need_reflected)
    {
      reflected.reset(MhCloneKernelInfo(kernel));
      if (!reflected)
        return fail(MH_OUT_OF_MEMORY,"cannot clone kernel");
      // RotateKernelInfo(reflected_kernel,180), morphology.c:3790: a plain
      // reversal of every kernel in the list
      for (MhKernelInfo *k=reflected.get(); k != nullptr; k=k->next)
        {
          size_t n=k->width*k->height;
          for (size_t i=0,j=n-1; i < j; i++,j--)
            {
              double t=k->values[i]; k->values[i]=k->values[j]; k->values[j]=t;
            }
          k->x=(ptrdiff_t) k->width-k->x-1;
          k->y=(ptrdiff_t) k->height-k->y-1;
        }
    }
  size_t nkernels=0;
  for (const MhKernelInfo *k=kernel; k != nullptr; k=k->next)
    nkernels++;
  const bool fixed_count=(kernel_limit == 1) && (method_limit == 1);
  // a plain chain of n primitives ping-pongs between two images: start in the destination
  // when n is odd.  Edge parks its first result and ends in the second image.
  bool destination_first=true;
  if (fixed_count && ((nkernels == 1) || (compose < 0)))
    destination_first=method == MH_MORPHOLOGY_EDGE ? false : (((nkernels*stage_limit) & 1u) != 0);
  MorphologyWorkspace ws(dst,destination_first);

  Temp counter;
  MH_TRY(counter.alloc(src.device,sizeof(unsigned long long),src.stream));
  unsigned long long *changed_dev=counter.as<unsigned long long>();
  const bool want_counts=!fixed_count || (changed_out != nullptr);
  const View *curr=&src,*work=nullptr,*save=nullptr,*rslt=nullptr;
  size_t total_changed=0;

  size_t method_loop=0,method_changed=1;
  while ((method_loop < method_limit) && (method_changed > 0))
    {
      method_loop++;
      method_changed=0;
      const MhKernelInfo *norm=kernel,*rflt=reflected.get();
      while (norm != nullptr)
        {
          for (size_t stage_loop=1; stage_loop <= stage_limit; stage_loop++)
            {
              // primitive of this stage, morphology.c:3829-3907
              const MhKernelInfo *this_kernel=norm;
              MhMorphologyMethod prim=method;
              switch (method)
              {
                case MH_MORPHOLOGY_ERODE: case MH_MORPHOLOGY_EDGE_IN:
                  prim=MH_MORPHOLOGY_ERODE; break;
                case MH_MORPHOLOGY_DILATE: case MH_MORPHOLOGY_EDGE_OUT:
                  prim=MH_MORPHOLOGY_DILATE; break;
                case MH_MORPHOLOGY_OPEN: case MH_MORPHOLOGY_TOP_HAT:
                  prim=stage_loop == 2 ? MH_MORPHOLOGY_DILATE : MH_MORPHOLOGY_ERODE; break;
                case MH_MORPHOLOGY_OPEN_INTENSITY:
                  prim=stage_loop == 2 ? MH_MORPHOLOGY_DILATE_INTENSITY : MH_MORPHOLOGY_ERODE_INTENSITY;
                  break;
                case MH_MORPHOLOGY_CLOSE: case MH_MORPHOLOGY_BOTTOM_HAT:
                  this_kernel=rflt;
                  prim=stage_loop == 2 ? MH_MORPHOLOGY_ERODE : MH_MORPHOLOGY_DILATE; break;
                case MH_MORPHOLOGY_CLOSE_INTENSITY:
                  this_kernel=rflt;
                  prim=stage_loop == 2 ? MH_MORPHOLOGY_ERODE_INTENSITY : MH_MORPHOLOGY_DILATE_INTENSITY;
                  break;
                case MH_MORPHOLOGY_SMOOTH:
                  if (stage_loop >= 3)
                    this_kernel=rflt;
                  prim=((stage_loop == 1) || (stage_loop == 4)) ? MH_MORPHOLOGY_ERODE :
                    MH_MORPHOLOGY_DILATE;
                  break;
                case MH_MORPHOLOGY_EDGE:
                  prim=MH_MORPHOLOGY_DILATE;
                  if (stage_loop == 2)
                    {
                      save=curr;                    // the dilated image, for the difference
                      curr=&src;
                      prim=MH_MORPHOLOGY_ERODE;
                    }
                  break;
                case MH_MORPHOLOGY_CORRELATE:
                  this_kernel=rflt;
                  prim=MH_MORPHOLOGY_CONVOLVE;
                  break;
                default:
                  break;
              }
              size_t kernel_loop=0;
              ptrdiff_t changed=1;
              while ((kernel_loop < kernel_limit) && (changed > 0))
                {
                  kernel_loop++;
                  if (work == nullptr)
                    MH_TRY(ws.acquire(&work));
                  if (want_counts)
                    MH_HIP(hipMemsetAsync(changed_dev,0,sizeof(unsigned long long),src.stream));
                  MhKernelInfo single=*this_kernel;
                  single.next=nullptr;
                  // FAST belongs to the primitive whose result leaves the operator
                  const bool feeds_another=(compose >= 0) || (iterations < 0) || (norm->next != nullptr) ||
                    (stage_loop != stage_limit) || (kernel_loop != kernel_limit) || (method_limit != 1);
                  MH_TRY(primitive(*curr,*work,prim,&single,bias,roles,desc,
                    want_counts ? changed_dev : nullptr,feeds_another ? MH_PRECISION_EXACT : precision()));
                  if (want_counts)
                    {
                      unsigned long long host=0;
                      MH_HIP(hipMemcpyAsync(&host,changed_dev,sizeof(host),
                        hipMemcpyDeviceToHost,src.stream));
                      MH_HIP(hipStreamSynchronize(src.stream));
                      // changed/GetImageChannels(image), morphology.c:2806, :3225
                      changed=(ptrdiff_t) (host/(unsigned long long) src.channels);
                    }
                  else
                    changed=1;
                  total_changed+=(size_t) changed;
                  method_changed+=(size_t) changed;
                  const View *t=work;               // swap, morphology.c:3952-3957
                  work=curr;
                  curr=t;
                  if (work == &src)
                    work=nullptr;
                }
            }
          // post-processing of the compound methods, morphology.c:3986-4013
          switch (method)
          {
            case MH_MORPHOLOGY_EDGE_IN: case MH_MORPHOLOGY_EDGE_OUT:
            case MH_MORPHOLOGY_TOP_HAT: case MH_MORPHOLOGY_BOTTOM_HAT:
              MH_TRY(launch_composite(*curr,src,MH_COMPOSITE_DIFFERENCE,roles));
              break;
            case MH_MORPHOLOGY_EDGE:
              MH_TRY(launch_composite(*curr,*save,MH_COMPOSITE_DIFFERENCE,roles));
              ws.release(save);
              save=nullptr;
              break;
            default:
              break;
          }
          // multi-kernel handling: re-iterate, or compose the results, :4016-4052
          if ((kernel->next == nullptr) || (compose < 0))
            rslt=curr;
          else if (rslt == nullptr)
            {
              rslt=curr;
              curr=&src;
            }
          else
            {
              MH_TRY(launch_composite(*rslt,*curr,compose,roles));
              ws.release(curr);
              curr=&src;
            }
          norm=norm->next;
          if (rflt != nullptr)
            rflt=rflt->next;
        }
    }
  if (rslt == nullptr)
    return fail(MH_BAD_ARGUMENT,"morphology: empty kernel list");
  if (rslt != &ws.slot[0])
    MH_TRY(launch_copy(*rslt,dst));
  if (changed_out != nullptr)
    *changed_out=(ptrdiff_t) total_changed;
  return MH_OK;
}

} // namespace mh

using namespace mh;

extern "C" {

MH_API MhStatus MagickHipMorphologyImage(const MhImage *image,MhImage *morphology_image,
  MhMorphologyMethod method,ptrdiff_t iterations,const MhKernelInfo *kernel,double bias)
{
  return MagickHipMorphologyImageCompose(image,morphology_image,method,iterations,kernel,bias,
    MH_MORPHOLOGY_COMPOSE_DEFAULT);
}

MH_API MhStatus MagickHipMorphologyImageCompose(const MhImage *image,MhImage *morphology_image,
  MhMorphologyMethod method,ptrdiff_t iterations,const MhKernelInfo *kernel,double bias,
  MhMorphologyCompose compose)
{
  MH_TRY(gate_pair(image,morphology_image,"MorphologyImage",true));
  if ((kernel == nullptr) || (kernel->values == nullptr))
    return fail(MH_BAD_ARGUMENT,"MorphologyImage: null kernel");
  if (compose == MH_MORPHOLOGY_COMPOSE_DEFAULT)
  {
    // host memory: row bands through a pipeline of uploads, kernels and downloads (batch.cpp)
    MhOperator op;
    op.kind=MH_OP_MORPHOLOGY;
    op.args[0]=(double) method;
    op.args[1]=(double) iterations;
    op.args[2]=bias;
    op.args[3]=0.0;
    op.text=nullptr;
    bool handled=false;
    MH_TRY(host_banded_operator(op,kernel,image,morphology_image,&handled));
    if (handled)
      return MH_OK;
  }
  Pair pair;
  MH_TRY(pair.open(image,morphology_image));
  Roles roles=channel_roles(image,morphology_image);
  MH_TRY(morphology_apply(pair.src.view,pair.dst.view,image,roles,method,iterations,kernel,
    bias,nullptr,compose));
  return pair.commit();
}

MH_API MhStatus MagickHipMorphologyPrimitive(const MhImage *image,MhImage *morphology_image,
  MhMorphologyMethod method,const MhKernelInfo *kernel,double bias,ptrdiff_t *changed)
{
  MH_TRY(gate_pair(image,morphology_image,"MorphologyPrimitive",true));
  if ((kernel == nullptr) || (kernel->values == nullptr))
    return fail(MH_BAD_ARGUMENT,"MorphologyPrimitive: null kernel");
  switch (method)
  {
    case MH_MORPHOLOGY_CONVOLVE: case MH_MORPHOLOGY_ERODE: case MH_MORPHOLOGY_DILATE:
    case MH_MORPHOLOGY_ERODE_INTENSITY: case MH_MORPHOLOGY_DILATE_INTENSITY:
    case MH_MORPHOLOGY_ITERATIVE_DISTANCE: case MH_MORPHOLOGY_HIT_AND_MISS:
    case MH_MORPHOLOGY_THINNING: case MH_MORPHOLOGY_THICKEN:
      break;
    default:
      return fail(MH_BAD_ARGUMENT,"not a primitive morphology method");
  }
  Pair pair;
  MH_TRY(pair.open(image,morphology_image));
  Roles roles=channel_roles(image,morphology_image);
  Temp counter;
  MH_TRY(counter.alloc(pair.src.view.device,sizeof(unsigned long long),pair.src.view.stream));
  MH_HIP(hipMemsetAsync(counter.ptr,0,sizeof(unsigned long long),pair.src.view.stream));
  MhKernelInfo single=*kernel;
  single.next=nullptr;
  MH_TRY(primitive(pair.src.view,pair.dst.view,method,&single,bias,roles,image,
    counter.as<unsigned long long>(),precision()));
  unsigned long long host=0;
  MH_HIP(hipMemcpyAsync(&host,counter.ptr,sizeof(host),hipMemcpyDeviceToHost,
    pair.src.view.stream));
  MH_HIP(hipStreamSynchronize(pair.src.view.stream));
  if (changed != nullptr)
    *changed=(ptrdiff_t) (host/(unsigned long long) image->number_channels);
  return pair.commit();
}

MH_API MhStatus MagickHipConvolveImage(const MhImage *image,MhImage *convolve_image,
  const MhKernelInfo *kernel)
{
  // ConvolveImage, effect.c:1170-1178
  return MagickHipMorphologyImage(image,convolve_image,MH_MORPHOLOGY_CONVOLVE,1,kernel,0.0);
}

MH_API MhStatus MagickHipBlurImage(const MhImage *image,MhImage *blur_image,
  double radius,double sigma)
{
  // BlurImage, effect.c:765-796: kernel list "blur:RxS;blur:RxS+90", then ConvolveImage
  MH_TRY(gate_pair(image,blur_image,"BlurImage",true));
  MhKernelInfo *kernel=acquire_blur_kernels(radius,sigma);
  if (kernel == nullptr)
    return fail(MH_BAD_ARGUMENT,"BlurImage: cannot build the blur kernels");
  MhStatus status=MagickHipConvolveImage(image,blur_image,kernel);
  MhDestroyKernelInfo(kernel);
  return status;
}

MH_API MhStatus MagickHipWaveletDenoiseImage(const MhImage *image,MhImage *noise_image,
  double threshold,double softness)
{
  MH_TRY(gate_pair(image,noise_image,"WaveletDenoiseImage",true));
  Pair pair;
  MH_TRY(pair.open(image,noise_image));
  Roles roles=channel_roles(image,noise_image);
  MH_TRY(launch_wavelet_denoise(pair.src.view,pair.dst.view,threshold,softness,roles));
  return pair.commit();
}

MH_API MhStatus MagickHipDespeckleImage(const MhImage *image,MhImage *despeckle_image)
{
  MH_TRY(gate_pair(image,despeckle_image,"DespeckleImage",true));
  Pair pair;
  MH_TRY(pair.open(image,despeckle_image));
  Roles roles=channel_roles(image,despeckle_image);
  MH_TRY(launch_despeckle(pair.src.view,pair.dst.view,roles));
  return pair.commit();
}

MH_API MhStatus MagickHipLocalContrastImage(const MhImage *image,MhImage *contrast_image,
  double radius,double strength)
{
  MH_TRY(gate_pair(image,contrast_image,"LocalContrastImage",true));
  Pair pair;
  MH_TRY(pair.open(image,contrast_image));
  Roles roles=channel_roles(image,contrast_image);
  MH_TRY(launch_local_contrast(pair.src.view,pair.dst.view,radius,strength,roles));
  return pair.commit();
}

MH_API MhStatus MagickHipRotationalBlurImage(const MhImage *image,MhImage *blur_image,double angle)
{
  MH_TRY(gate_pair(image,blur_image,"RotationalBlurImage",true));
  // effect.c:3256-3276: sample count and the rotation tables (host libm, as the reference)
  const double pi=3.1415926535897932384626433832795028841971693993751058209749445923078164062;
  const double radians=(double) (pi*angle/180.0);
  const double center_x=(double) (image->columns-1)/2.0,center_y=(double) (image->rows-1)/2.0;
  const double blur_radius=hypot(center_x,center_y);
  const size_t n=(size_t) fabs(4.0*radians*sqrt((double) blur_radius)+2UL);
  if ((n < 2) || (n > (1u << 24)))
    return fail(MH_UNSUPPORTED,"RotationalBlurImage: %zu samples per pixel",n);
  const double theta=radians/(double) (n-1);
  const double offset=theta*(double) (n-1)/2.0;
  std::vector<double> cos_theta(n),sin_theta(n);
  for (size_t w=0; w < n; w++)
    {
      cos_theta[w]=cos((double) (theta*(double) w-offset));
      sin_theta[w]=sin((double) (theta*(double) w-offset));
    }
  Pair pair;
  MH_TRY(pair.open(image,blur_image));
  Roles roles=channel_roles(image,blur_image);
  MH_TRY(launch_rotational_blur(pair.src.view,pair.dst.view,cos_theta.data(),sin_theta.data(),n,
    blur_radius,roles));
  return pair.commit();
}

MH_API MhStatus MagickHipMotionBlurImageWithKernel(const MhImage *image,MhImage *blur_image,
  const double *kernel,size_t width,const ptrdiff_t *offsets_xy)
{
  MH_TRY(gate_pair(image,blur_image,"MotionBlurImage",true));
  if ((kernel == nullptr) || (offsets_xy == nullptr) || (width == 0))
    return fail(MH_BAD_ARGUMENT,"MotionBlurImage: null kernel or offsets");
  Pair pair;
  MH_TRY(pair.open(image,blur_image));
  Roles roles=channel_roles(image,blur_image);
  MH_TRY(launch_motion_blur(pair.src.view,pair.dst.view,kernel,width,offsets_xy,roles));
  return pair.commit();
}

MH_API MhStatus MagickHipMotionBlurImage(const MhImage *image,MhImage *blur_image,double radius,
  double sigma,double angle)
{
  // effect.c:2378-2393: width, kernel and the offsets along the motion direction
  const size_t width=MhGetOptimalKernelWidth1D(radius,sigma);
  const double s=fabs(sigma) < 1.0e-12 ? 1.0e-12 : sigma;          // MagickSigma
  const double sq2pi=2.50662827463100024161235523934010416269302368164062;
  std::vector<double> kernel(width);
  double normalize=0.0;
  for (size_t i=0; i < width; i++)
    {
      kernel[i]=(double) (exp((-((double) i*i)/(double) (2.0*s*s)))/(sq2pi*s));
      normalize+=kernel[i];
    }
  for (size_t i=0; i < width; i++)
    kernel[i]/=normalize;
  const double pi=3.1415926535897932384626433832795028841971693993751058209749445923078164062;
  const double radians=(double) (pi*angle/180.0);                   // DegreesToRadians
  const double px=(double) width*sin(radians),py=(double) width*cos(radians);
  std::vector<ptrdiff_t> offsets(2*width);
  for (size_t w=0; w < width; w++)
    {
      offsets[2*w]=(ptrdiff_t) ceil((double) ((double) w*py)/hypot(px,py)-0.5);
      offsets[2*w+1]=(ptrdiff_t) ceil((double) ((double) w*px)/hypot(px,py)-0.5);
    }
  return MagickHipMotionBlurImageWithKernel(image,blur_image,kernel.data(),width,offsets.data());
}

// One square kernel through ConvolveImage; `equalize` adds EmbossImage's EqualizeImage of
// the result (effect.c:1675-1676), on the device before the result is handed back.
static MhStatus convolve_with(const MhImage *image,MhImage *out,MhKernelInfo *kernel,
  const char *what,bool equalize)
{
  if (kernel == nullptr)
    return fail(MH_BAD_ARGUMENT,"%s: cannot build the kernel",what);
  Pair pair;
  MhStatus status=pair.open(image,out);
  if (status == MH_OK)
    {
      Roles roles=channel_roles(image,out);
      status=morphology_apply(pair.src.view,pair.dst.view,image,roles,MH_MORPHOLOGY_CONVOLVE,1,
        kernel,0.0,nullptr);
      if ((status == MH_OK) && equalize)
        status=equalize_view(pair.dst.view,out);
    }
  MhDestroyKernelInfo(kernel);
  if (status != MH_OK)
    return status;
  return pair.commit();
}

MH_API MhStatus MagickHipGaussianBlurImage(const MhImage *image,MhImage *blur_image,
  double radius,double sigma)
{
  // GaussianBlurImage, effect.c:1709-1735: AcquireKernelInfo("gaussian:RxS")
  MH_TRY(gate_pair(image,blur_image,"GaussianBlurImage",true));
  return convolve_with(image,blur_image,acquire_gaussian_kernel(radius,sigma),"GaussianBlurImage",
    false);
}

MH_API MhStatus MagickHipSharpenImage(const MhImage *image,MhImage *sharp_image,
  double radius,double sigma)
{
  MH_TRY(gate_pair(image,sharp_image,"SharpenImage",true));
  return convolve_with(image,sharp_image,acquire_sharpen_kernel(radius,sigma),"SharpenImage",false);
}

MH_API MhStatus MagickHipEdgeImage(const MhImage *image,MhImage *edge_image,double radius)
{
  MH_TRY(gate_pair(image,edge_image,"EdgeImage",true));
  return convolve_with(image,edge_image,acquire_edge_kernel(radius),"EdgeImage",false);
}

MH_API MhStatus MagickHipEmbossImage(const MhImage *image,MhImage *emboss_image,
  double radius,double sigma)
{
  MH_TRY(gate_pair(image,emboss_image,"EmbossImage",true));
  return convolve_with(image,emboss_image,acquire_emboss_kernel(radius,sigma),"EmbossImage",true);
}

// FAST, Q16, four channels: the row pass as in BlurImage, then the column pass with the
// threshold/gain epilogue applied while its results are copied out — the blurred frame is never
// written or re-read (2 of the 5 frame transfers of the three-kernel form).
static MhStatus unsharp_fused(const View &src,const View &dst,const MhKernelInfo *kernels,
  const Roles &roles,double gain,double threshold,bool *fused)
{
  *fused=false;
  const MhKernelInfo *horizontal=kernels,*vertical=kernels != nullptr ? kernels->next : nullptr;
  if ((src.quantum != MH_QUANTUM_U16) ||
      ((src.channels != 4) && (src.channels != 3) && (src.channels != 1)) ||
      (src.columns < 2) || (option("MAGICKHIP_NO_FUSED_UNSHARP") != nullptr) ||
      (option("MAGICKHIP_NO_MFMA") != nullptr) ||
      (roles.copy_mask != 0) || (horizontal == nullptr) || (vertical == nullptr) ||
      (vertical->next != nullptr) || (horizontal->height != 1) || (vertical->width != 1) ||
      kernel_has_nan(horizontal) || kernel_has_nan(vertical))
    return MH_OK;
  if (roles.blend && ((roles.alpha != 3) || (src.channels != 4)))
    return MH_OK;
  // one launch: both passes and the epilogue (convolve_fused_exact.hip), kernels of up to 81 taps,
  // RGBA / four plain channels / RGB
  MH_TRY(fused_blur(src,dst,kernels,roles,0.0,fused,true,gain,threshold));
  if (*fused)
    return MH_OK;
  if (precision() != MH_PRECISION_FAST)
    return MH_OK;                                // EXACT: blur + unsharp_epilogue
  if (src.channels != 4)
    return MH_OK;                                // the two-launch form copies 8-byte pixels out
  if (roles.blend && !f16_taps_resolved(vertical->values,(int) vertical->height))
    return MH_OK;                                // tiny outer taps on an alpha-weighted frame: launch_conv1d
  View rows=src;
  Temp memory;
  MH_TRY(memory.alloc(src.device,rows.bytes(),src.stream));
  rows.pixels=memory.ptr;
  Conv1DParams first,second;
  first.taps=horizontal->values;
  first.ntaps=(int) horizontal->width;
  first.origin=(int) horizontal->x;
  second.taps=vertical->values;
  second.ntaps=(int) vertical->height;
  second.origin=(int) vertical->y;
  if ((second.ntaps < 2) || (second.ntaps > 113))
    return MH_OK;                                // outside the matrix-core kernel's reach
  // (the intermediate is the reference's own: exact row pass)
  MH_TRY(launch_conv1d(src,rows,false,first,roles,MH_PRECISION_EXACT,nullptr));
  MH_TRY(launch_conv1d_unsharp(rows,dst,src,second,roles.blend,gain,threshold,fused));
  return MH_OK;
}

// Any layout on the fp64 vector kernels (float Quantum; Q16 in EXACT mode outside the fused
// kernel's reach): the row pass, then the column pass with the epilogue applied as it stores —
// no blurred frame written and read back, no third kernel.
static MhStatus unsharp_vector_fused(const View &src,const View &dst,const MhKernelInfo *kernels,
  const Roles &roles,double gain,double threshold,bool *fused)
{
  *fused=false;
  const MhKernelInfo *horizontal=kernels,*vertical=kernels != nullptr ? kernels->next : nullptr;
  if ((horizontal == nullptr) || (vertical == nullptr) || (vertical->next != nullptr) ||
      (horizontal->height != 1) || (vertical->width != 1) || (vertical->height < 16) ||
      kernel_has_nan(horizontal) || kernel_has_nan(vertical) ||
      ((src.quantum == MH_QUANTUM_U16) && (precision() != MH_PRECISION_EXACT)))
    return MH_OK;
  View rows=src;
  Temp memory;
  MH_TRY(memory.alloc(src.device,rows.bytes(),src.stream));
  rows.pixels=memory.ptr;
  Conv1DParams first,second;
  first.taps=horizontal->values;
  first.ntaps=(int) horizontal->width;
  first.origin=(int) horizontal->x;
  second.taps=vertical->values;
  second.ntaps=(int) vertical->height;
  second.origin=(int) vertical->y;
  // (decide before the row pass is spent: the column entry point declines on its own terms)
  for (int v=0; v < second.ntaps; v++)
    if (!(second.taps[v] >= 0.0))
      return MH_OK;
  MH_TRY(launch_conv1d(src,rows,false,first,roles,MH_PRECISION_EXACT,nullptr));
  bool handled=false;
  MH_TRY(launch_conv1d_column_unsharp(rows,dst,src,second,roles,precision(),gain,threshold,&handled));
  if (!handled)
    {
      // the column pass and the epilogue kernel after all
      MH_TRY(launch_conv1d(rows,dst,true,second,roles,precision(),nullptr));
      MH_TRY(launch_unsharp_epilogue(src,dst,dst,gain,threshold,roles));
    }
  *fused=true;
  return MH_OK;
}

MH_API MhStatus MagickHipUnsharpMaskImage(const MhImage *image,MhImage *unsharp_image,
  double radius,double sigma,double gain,double threshold)
{
  // UnsharpMaskImage, effect.c:4256-4400: blur into the destination, then the
  // threshold/gain epilogue in place
  MH_TRY(gate_pair(image,unsharp_image,"UnsharpMaskImage",true));
  {
    // host memory: row bands through a pipeline of uploads, kernels and downloads (batch.cpp)
    MhOperator op;
    op.kind=MH_OP_UNSHARP_MASK;
    op.args[0]=radius;
    op.args[1]=sigma;
    op.args[2]=gain;
    op.args[3]=threshold;
    op.text=nullptr;
    bool handled=false;
    MH_TRY(host_banded_operator(op,nullptr,image,unsharp_image,&handled));
    if (handled)
      return MH_OK;
  }
  MhKernelInfo *kernel=acquire_blur_kernels(radius,sigma);
  if (kernel == nullptr)
    return fail(MH_BAD_ARGUMENT,"UnsharpMaskImage: cannot build the blur kernels");
  Pair pair;
  MhStatus status=pair.open(image,unsharp_image);
  if (status == MH_OK)
    {
      Roles roles=channel_roles(image,unsharp_image);
      bool fused=false;
      status=unsharp_fused(pair.src.view,pair.dst.view,kernel,roles,gain,threshold,&fused);
      if ((status == MH_OK) && !fused)
        status=unsharp_vector_fused(pair.src.view,pair.dst.view,kernel,roles,gain,threshold,&fused);
      if ((status == MH_OK) && !fused)
        {
          status=morphology_apply(pair.src.view,pair.dst.view,image,roles,MH_MORPHOLOGY_CONVOLVE,1,
            kernel,0.0,nullptr);
          if (status == MH_OK)
            status=launch_unsharp_epilogue(pair.src.view,pair.dst.view,pair.dst.view,gain,threshold,
              roles);
        }
    }
  MhDestroyKernelInfo(kernel);
  if (status != MH_OK)
    return status;
  return pair.commit();
}

// ResizeImage, resize.c:3761-3874
MH_API MhStatus MagickHipResizeImageWithFilter(const MhImage *image,MhImage *resize_image,
  const MhResizeFilter *filter)
{
  MH_TRY(gate_pair(image,resize_image,"ResizeImage",false));
  if (filter == nullptr)
    return fail(MH_BAD_ARGUMENT,"ResizeImage: null filter");
  const size_t columns=resize_image->columns,rows=resize_image->rows;
  // resize.c:3804-3805: a multiply by the reciprocal, not a division
  const double x_factor=(double) ((double) columns*(1.0/(double) image->columns));
  const double y_factor=(double) ((double) rows*(1.0/(double) image->rows));
  Pair pair;
  MH_TRY(pair.open(image,resize_image));
  // destination traits decide Blend/Copy (resize.c:3494); alpha comes from the source
  Roles roles=channel_roles(image,resize_image);
  View filter_view=pair.src.view;
  Temp scratch;
  // The first filter's result is ROUNDED to a Quantum and feeds the second: it runs in the
  // reference's own operation order in either mode.  (FAST's fused sums differ by ~1e-10 level —
  // enough to flip a value that sits on a rounding boundary, and polynomial filters over small
  // integers put whole families of sums exactly there; one level of a small intermediate alpha is
  // thousands of levels of the colours the second filter weights with it.  tests/stress_parity.py
  // found it.)  FAST belongs to the filter whose result leaves the operator; the one-launch forms
  // below find such values themselves and recompute their rows (resize_acc.hpp).
  const MhPrecision first_pass=MH_PRECISION_EXACT;
  if (x_factor > y_factor)
    {
      // HorizontalFilter then VerticalFilter, resize.c:3846-3853
      filter_view.columns=columns;
      filter_view.rows=image->rows;
      MH_TRY(scratch.alloc(pair.src.view.device,filter_view.bytes(),pair.src.view.stream));
      filter_view.pixels=scratch.ptr;
      auto horizontal=acquire_tap_table(filter,image->columns,columns,x_factor);
      auto vertical=acquire_tap_table(filter,image->rows,rows,y_factor);
      MH_TRY(launch_resize_pass(pair.src.view,filter_view,false,*horizontal,roles,first_pass));
      MH_TRY(launch_resize_pass(filter_view,pair.dst.view,true,*vertical,roles,precision()));
    }
  else
    {
      auto vertical=acquire_tap_table(filter,image->rows,rows,y_factor);
      auto horizontal=acquire_tap_table(filter,image->columns,columns,x_factor);
      bool fused=false;
      MH_TRY(launch_resize_fused(pair.src.view,pair.dst.view,*vertical,*horizontal,roles,precision(),&fused));
      if (!fused)
        {
          filter_view.columns=image->columns;
          filter_view.rows=rows;
          MH_TRY(scratch.alloc(pair.src.view.device,filter_view.bytes(),pair.src.view.stream));
          filter_view.pixels=scratch.ptr;
          MH_TRY(launch_resize_pass(pair.src.view,filter_view,true,*vertical,roles,first_pass));
          MH_TRY(launch_resize_pass(filter_view,pair.dst.view,false,*horizontal,roles,precision()));
        }
    }
  return pair.commit();
}

MH_API MhStatus MagickHipResizeImage(const MhImage *image,MhImage *resize_image,
  MhFilterType filter)
{
  MH_TRY(gate_pair(image,resize_image,"ResizeImage",false));
  // default filter choice, resize.c:3806-3816
  MhFilterType type=filter;
  if (type == MH_FILTER_UNDEFINED)
    {
      const double x_factor=(double) ((double) resize_image->columns*(1.0/(double) image->columns));
      const double y_factor=(double) ((double) resize_image->rows*(1.0/(double) image->rows));
      type=MH_FILTER_LANCZOS;
      if ((x_factor == 1.0) && (y_factor == 1.0))
        type=MH_FILTER_POINT;
      else if ((image->alpha_trait != MH_TRAIT_UNDEFINED) || ((x_factor*y_factor) > 1.0))
        type=MH_FILTER_MITCHELL;
    }
  MhResizeFilter *f=MhAcquireResizeFilter(type,0);
  if (f == nullptr)
    return fail(MH_UNSUPPORTED,"ResizeImage: filter %d is not available",(int) type);
  MhStatus status=MagickHipResizeImageWithFilter(image,resize_image,f);
  MhDestroyResizeFilter(f);
  return status;
}

} // extern "C"
