// Internal (non-ABI) declarations shared by the host runtime and the HIP
// translation units of libmagickhip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include <cstdarg>
#include <cmath>
#include <vector>
#include <memory>

#include "magickhip.h"

namespace mh {

// MagickEpsilon / QuantumRange / QuantumScale: MagickCore/magick-type.h:114-119
constexpr double kMagickEpsilon = 1.0e-12;
constexpr double kQuantumRange = 65535.0;
constexpr double kQuantumScale = 1.0/65535.0;

void set_error(const char *fmt,...) __attribute__((format(printf,1,2)));
MhStatus fail(MhStatus status,const char *fmt,...) __attribute__((format(printf,2,3)));

#define MH_HIP(expr)                                                         \
  do {                                                                       \
    hipError_t mh_err_=(expr);                                               \
    if (mh_err_ != hipSuccess)                                               \
      return ::mh::fail(MH_DEVICE_ERROR,"%s failed: %s (%s:%d)",#expr,       \
        hipGetErrorString(mh_err_),__FILE__,__LINE__);                       \
  } while (0)

#define MH_TRY(expr)                                                         \
  do { MhStatus mh_st_=(expr); if (mh_st_ != MH_OK) return mh_st_; } while (0)

// ----------------------------------------------------------------- runtime
MhStatus runtime_ready();                 // lazy MhInitialize + enabled check
int default_device();
int device_count();
int compute_units(int device);          // cached multiProcessorCount
int lds_bytes_per_workgroup(int device);   // cached MaxSharedMemoryPerBlock
bool host_block_is_pinned(const void *block,size_t bytes);   // inside a block of MhHostAlloc's
int logical_device_count();               // device_count(), or MAGICKHIP_LOGICAL_DEVICES when larger
// the precision of the operator call this thread is in (MhImage::precision of the image the entry
// point was gated with), else the library default
MhPrecision precision();
void set_call_precision(const MhImage *image);
// MAGICKHIP_* / MAGICK_HIP_* switches: the environment as it was when the runtime initialised,
// plus what MhSetOption changed since.  NULL = unset.  Never reads the environment again.
const char *option(const char *name);
long option_long(const char *name,long fallback);
hipStream_t library_stream(int device);   // non-blocking stream owned by the library

// stream-tagged caching device allocator (workspace / staging)
MhStatus pool_alloc(int device,size_t bytes,hipStream_t stream,void **ptr);
void pool_free(int device,void *ptr,hipStream_t stream);
void pool_trim();

// Makes `device` HIP's current device for the caller's thread and puts the previous one back
// when it goes out of scope: an operator on a cuda:1 image must not leave the host thread (a
// MagickCore caller's, or torch's) on device 1.
struct DeviceGuard
{
  int previous=-1;
  DeviceGuard() = default;
  DeviceGuard(const DeviceGuard &) = delete; DeviceGuard &operator=(const DeviceGuard &) = delete;
  ~DeviceGuard() { leave(); }
  hipError_t enter(int device)
  {
    leave();
    int current=-1;
    if ((hipGetDevice(&current) == hipSuccess) && (current != device))
      {
        hipError_t err=hipSetDevice(device);
        if (err != hipSuccess)
          return err;
        previous=current;
      }
    return hipSuccess;
  }
  void leave()
  {
    if (previous >= 0)
      (void) hipSetDevice(previous);
    previous=-1;
  }
};

// RAII temp device buffer
struct Temp
{
  int device=-1; hipStream_t stream=nullptr; void *ptr=nullptr;
  Temp() = default;
  Temp(const Temp &) = delete; Temp &operator=(const Temp &) = delete;
  ~Temp() { reset(); }
  MhStatus alloc(int dev,size_t bytes,hipStream_t s)
  { reset(); device=dev; stream=s; return pool_alloc(dev,bytes,s,&ptr); }
  void reset() { if (ptr != nullptr) pool_free(device,ptr,stream); ptr=nullptr; }
  template<typename T> T *as() const { return static_cast<T *>(ptr); }
};

// upload a small host table (taps, LUT, tap tables) into a Temp, async on stream.
// The host bytes are copied into a pinned bounce buffer first so the caller's
// memory may go away immediately.
MhStatus upload_table(Temp &dst,int device,hipStream_t stream,const void *host,size_t bytes);

// A small read-only table (filter taps) that many calls share: looked up by content, uploaded on
// the first use and kept on the device (a BlurImage call otherwise pays a copy and two queue
// gaps, ~10 us, in front of a 0.43 ms kernel).  *device_ptr stays valid for work enqueued on
// `stream` by this call as long as the caller holds *keep (the cache may evict the entry — a
// thread with 32 newer tables in between — but the device block is freed only when the last
// holder lets go, and hipFree then waits for the kernels already enqueued on it).
MhStatus shared_table(int device,hipStream_t stream,const void *host,size_t bytes,const void **device_ptr,
  std::shared_ptr<void> *keep=nullptr);
void release_shared_tables();       // MhTerminus
void release_resize_tables();       // the resize launchers' device-side tables (resize.hip)
void release_batch_streams();       // the worker streams of batch.cpp (after the pools were trimmed)
void release_rccl_communicators();  // batch.cpp's cached communicators

// Several host tables as ONE device block and ONE host-to-device copy (a resize pass has ten
// tables: ten stream-ordered copies cost 150-300 us of idle GPU between two kernels of a few
// milliseconds).  add() the parts (the host memory must stay valid until upload()), upload(),
// then at<T>(index).
class TableBundle
{
public:
  size_t add(const void *host,size_t bytes)
  {
    parts_.push_back({host,bytes,total_});
    total_+=(bytes+255u) & ~(size_t) 255u;
    return parts_.size()-1;
  }
  MhStatus upload(int device,hipStream_t stream);
  template<typename T> T *at(size_t index) const
  { return reinterpret_cast<T *>(static_cast<char *>(block_.ptr)+parts_[index].offset); }
private:
  struct Part { const void *host; size_t bytes,offset; };
  std::vector<Part> parts_;
  size_t total_=0;
  Temp block_;
};

// Resolved, device-resident view of an MhImage for the kernel launchers.
struct View
{
  void *pixels=nullptr;
  size_t columns=0,rows=0;
  int channels=0;
  MhQuantumKind quantum=MH_QUANTUM_U16;
  int device=0;
  hipStream_t stream=nullptr;
  size_t bytes() const
  { return columns*rows*(size_t) channels*(quantum == MH_QUANTUM_U16 ? 2u : 4u); }
};

// Brings an MhImage onto the device (HOST memory: pinned-staged
// hipMemcpyAsync) and writes results back.
class Resident
{
public:
  Resident() = default;
  ~Resident();
  Resident(const Resident &) = delete; Resident &operator=(const Resident &) = delete;
  // mode: 0 = input (upload), 1 = output (no upload, download on commit),
  //       2 = in-place (upload and download on commit)
  MhStatus open(const MhImage *image,int mode,hipStream_t stream_hint,int device_hint);
  MhStatus commit();          // output/in-place: copy back to host memory and wait
  View view;
private:
  const MhImage *image_=nullptr;
  int mode_=0;
  bool staged_=false;
  bool registered_=false;
  bool pipelined_=false;
  Temp temp_;
};

MhStatus validate_image(const MhImage *image,const char *what);
int resolve_device(const MhImage *image);
hipStream_t resolve_stream(const MhImage *image,int device);

// per-channel role masks derived from MhImage traits
struct Roles
{
  uint32_t update_mask=0;   // channels computed
  uint32_t copy_mask=0;     // channels copied from the source (Copy or Undefined trait)
  bool blend=false;         // alpha-weighted colour channels
  int alpha=-1;
};
Roles channel_roles(const MhImage *src,const MhImage *dst);

MhKernelInfo *acquire_blur_kernels(double radius,double sigma);
MhKernelInfo *acquire_gaussian_kernel(double radius,double sigma);
MhKernelInfo *acquire_sharpen_kernel(double radius,double sigma);
MhKernelInfo *acquire_edge_kernel(double radius);
MhKernelInfo *acquire_emboss_kernel(double radius,double sigma);
// EqualizeImage on a device view (operators_enhance.cpp)
MhStatus equalize_view(const View &view,const MhImage *image);

// ------------------------------------------------------------ profiling
struct ProfileScope
{
  ProfileScope(const char *name,hipStream_t stream);
  ~ProfileScope();
  const char *name; hipStream_t stream; hipEvent_t start=nullptr,stop=nullptr; bool on=false;
  int device=0;
};

// ------------------------------------------------------------ launchers
// (defined in the .hip translation units)

struct Conv1DParams
{
  const double *taps=nullptr;  // kernel values in storage order (as KernelInfo::values)
  int ntaps=0;
  int origin=0;                // kernel->x (row kernel) or kernel->y (column kernel)
  double bias=0.0;
  // launch_conv1d_column_unsharp only: the unblurred frame and UnsharpMaskImage's gain /
  // ceil-free threshold (QuantumRange*threshold), applied as the column pass stores its results
  const void *unsharp_source=nullptr;
  double unsharp_gain=0.0,unsharp_threshold=0.0;
  // a device word: the pass's kernels leave at once unless it is set (the fp64 passes queued behind
  // the exact-integer BlurImage kernel, which sets the word when it gives a frame up).  only_if_token 0: set = not
  // zero; otherwise set = equal to the token — a value of the call's own, so that nothing has to clear the word
  // in front of the kernel (one dispatch less a call)
  const unsigned *only_if=nullptr;
  unsigned only_if_token=0u;
};
// The column pass of a blur with UnsharpMaskImage's epilogue applied on the way out (the fp64
// triangular kernels: float Quantum, and Q16 in EXACT mode).  *handled = false: not this case,
// nothing launched (the caller runs the pass and the epilogue kernel).
MhStatus launch_conv1d_column_unsharp(const View &rows,const View &dst,const View &original,
  const Conv1DParams &params,const Roles &roles,MhPrecision precision,double gain,double threshold,
  bool *handled);
// One MorphologyPrimitive(Convolve) pass with a 1-D kernel without NaN cells:
// horizontal (1 x ntaps, morphology.c:2811-2979) or vertical (ntaps x 1,
// column fast path morphology.c:2654-2807).
MhStatus launch_conv1d(const View &src,const View &dst,bool vertical,
  const Conv1DParams &params,const Roles &roles,MhPrecision precision,
  unsigned long long *changed_device);

// What a matrix-core pass reads and writes
//   MFMA_Q16       Quantum pixels in, Quantum pixels out (BlurImage's passes)
//   MFMA_TO_SUMS   row pass of a separated 2-D kernel: Quantum pixels in, the four undivided
//                  f32 sums of a pixel out (16 bytes)
//   MFMA_FROM_SUMS column pass of a separated 2-D kernel: those sums in, Quantum pixels out —
//                  one division for the whole 2-D window, as morphology.c:2892-2979
//   MFMA_UNSHARP   column pass of UnsharpMaskImage: the blurred sample never reaches memory,
//                  the copy-out applies effect.c:4364-4369 against the unblurred frame
//                  (unsharp_original, gain, threshold)
enum MfmaIo { MFMA_Q16=0,MFMA_TO_SUMS=1,MFMA_FROM_SUMS=2,MFMA_UNSHARP=3 };
// The f16 tap operands are scale*tap, hi + lo (split_f16, mfma_common.hpp).  A fixed factor of 256 left the small taps of a
// kernel in f16's denormal range — absolute precision 2^-25, i.e. 2^-17 of BlurImage's outermost default
// tap and 2^-12 of a tap of 4e-7 (-blur 25x5) — which is the whole result where such a tap is the only one
// that meets a sample (a sprite on a transparent ground: up to 13 levels off, found by the stress run's
// radius draws in round 6).  scale = the power of two that puts the largest tap in [2^14, 2^15): every tap
// down to 2^-18 of the largest keeps its 22 bits, and the launchers decline kernels whose smallest tap lies
// below 2^-19 of the largest.  Returns 0 when no tap is positive.
static inline float f16_tap_scale(const double *taps,int K)
{
  double largest=0.0;
  for (int v=0; v < K; v++)
    largest=std::fabs(taps[v]) > largest ? std::fabs(taps[v]) : largest;
  if (!(largest > 0.0) || !std::isfinite(largest))
    return 0.0f;
  int exponent=0;
  (void) std::frexp(largest,&exponent);          // largest = m * 2^exponent, 0.5 <= m < 1
  return (float) std::ldexp(1.0,15-exponent);
}

// ... and whether every tap is large enough beside the largest for its hi + lo terms to carry 19 bits
static inline bool f16_taps_resolved(const double *taps,int K)
{
  double largest=0.0,smallest=INFINITY;
  for (int v=0; v < K; v++)
    {
      const double t=std::fabs(taps[v]);
      if (t == 0.0)
        continue;
      largest=t > largest ? t : largest;
      smallest=t < smallest ? t : smallest;
    }
  return (largest > 0.0) && (smallest >= std::ldexp(largest,-19));
}

// FAST Q16 pass on the f16 matrix cores (convolve_mfma.hip); *handled=false when the shape is
// outside its reach and nothing was launched.  io: an MfmaIo.  tap_scale: f16_tap_scale of the taps
// (above), MFMA_Q16 and MFMA_UNSHARP.
MhStatus launch_conv1d_mfma(const View &src,const View &dst,bool vertical,const float *taps_device,
  int ntaps,int shift,bool blend,int io,bool *handled,const View *unsharp_original=nullptr,
  double gain=0.0,double threshold=0.0,const double *taps64_device=nullptr,float tap_scale=256.0f);
// ... with cells that are integer multiples of a unit: exact sums on the i8 matrix cores,
// bit-identical in both precision modes (convolve2d_exact.hip)
MhStatus launch_conv2d_exact(const View &src,const View &dst,const MhKernelInfo *kernel,bool blend,
  bool *handled,Temp *flag=nullptr);
// ... any cells, any layout, Q16 or float: fused multiply-adds over premultiplied doubles + tie check,
// bit-identical (convolve2d_tie.hip); only_if: device word, the kernel leaves at once when it is zero
MhStatus launch_conv2d_tie(const View &src,const View &dst,const MhKernelInfo *kernel,const Roles &roles,
  bool *handled,const unsigned *only_if=nullptr);
MhStatus launch_conv2d_mfma(const View &src,const View &dst,const MhKernelInfo *kernel,bool blend,
  bool *handled);
// BlurImage's two passes in one launch: the row pass's Quantum-rounded result stays in an LDS ring and
// never reaches HBM.  Exact-integer sums (convolve_fused_exact.hip): taps = HOST doubles in the reversed
// walk of morphology.c:2746 (taps[v] multiplies the input at o-shift+v), all positive.  Both passes exact: the result
// is bit-identical to the reference (MH_PRECISION_EXACT; UnsharpMaskImage in both modes).  recomputed_device
// (optional): a device counter that receives the number of samples recomputed in the reference's operation order.
constexpr int kExactDigits=5;            // balanced signed 8-bit digits of a fixed-point tap
constexpr int kExactDigitPitch=96;       // digits of one weight, padded (K <= 81)
MhStatus launch_blur_fused_exact(const View &src,const View &dst,const double *taps,int ntaps,int shift,
  bool blend,bool *handled,bool unsharp=false,double gain=0.0,double threshold=0.0,
  unsigned long long *recomputed_device=nullptr,unsigned *give_up=nullptr,unsigned give_up_token=1u);
// FAST BlurImage in one launch: f16 colour sums in both passes, the row pass's alpha as exact integer
// sums (convolve_fused_hybrid.hip); within +-1 level by construction.  taps as above.
MhStatus launch_blur_fused_hybrid(const View &src,const View &dst,const double *taps,int ntaps,int shift,
  bool blend,bool *handled);
// UnsharpMaskImage's column pass + epilogue in one launch: rows = the row pass's result,
// original = the unblurred frame (effect.c:4343-4372)
MhStatus launch_conv1d_unsharp(const View &rows,const View &dst,const View &original,
  const Conv1DParams &params,bool blend,double gain,double threshold,bool *handled);
// One pass of a separated 2-D kernel through launch_conv1d_mfma (taps uploaded here):
// vertical = false: src Quantum RGBA -> dst float sums; vertical = true: the reverse
MhStatus launch_conv1d_sums64(const View &src,const View &dst,bool vertical,const Conv1DParams &params);
// EXACT (and float-Quantum FAST) 2-D Convolve with an outer-product kernel: two fp64 passes + tie check
// (delta: the kernel is column x row + delta at cell (delta_y, delta_x) — SharpenImage, EdgeImage)
MhStatus launch_separable_exact(const View &src,const View &dst,const MhKernelInfo *kernel,
  const double *row,const double *column,const Roles &roles,bool *handled,int delta_x=0,int delta_y=0,
  double delta=0.0);
MhStatus launch_conv1d_sums(const View &src,const View &dst,bool vertical,const Conv1DParams &params,
  bool blend,bool *handled);

struct Morph2DParams
{
  MhMorphologyMethod method=MH_MORPHOLOGY_UNDEFINED;
  const MhKernelInfo *kernel=nullptr;
  double bias=0.0;
  MhIntensityMethod intensity=MH_INTENSITY_REC709LUMA;
  MhColorspace colorspace=MH_COLORSPACE_SRGB;
  // device word: the generic kernel does the frame only if it is non-zero (the fallback behind an
  // optimistic kernel that found, on the device, that the frame is not for it)
  const unsigned *only_if=nullptr;
};
MhStatus launch_morph2d(const View &src,const View &dst,const Morph2DParams &params,
  const Roles &roles,unsigned long long *changed_device);

struct TapTable;   // resize contribution table (host), see resize_filter.cpp
MhStatus launch_resize_pass(const View &src,const View &dst,bool vertical,
  const TapTable &table,const Roles &roles,MhPrecision precision);

// VerticalFilter + HorizontalFilter in one launch (intermediate kept in LDS);
// *handled = false when the shape does not fit and nothing was launched.
MhStatus launch_resize_fused(const View &src,const View &dst,const TapTable &vertical,
  const TapTable &horizontal,const Roles &roles,MhPrecision precision,bool *handled);

// the FAST one-launch enlargement on the fp64 matrix pipe (resize_mfma.hip)
MhStatus launch_resize_mfma(const View &src,const View &dst,const TapTable &vertical,
  const TapTable &horizontal,const Roles &roles,bool *handled);
void release_resize_mfma_plans();
// the FAST one-launch enlargement by a whole-number horizontal factor on the fp64 vector pipe
// (resize_stream.hip)
MhStatus launch_resize_stream(const View &src,const View &dst,const TapTable &vertical,
  const TapTable &horizontal,const Roles &roles,bool *handled);
void release_resize_stream_plans();

MhStatus launch_unsharp_epilogue(const View &src,const View &blur,const View &dst,
  double gain,double threshold,const Roles &roles);

MhStatus launch_histogram(const View &src,int intensity_mode,const MhImage *desc,
  unsigned long long *hist_device);
// shared_column >= 0: every channel selected by apply_mask maps through that one LUT column
// device_mask (optional): a device word and-ed into apply_mask inside the kernel
MhStatus launch_apply_lut(const View &img,const void *lut_device,uint32_t apply_mask,
  const Roles &roles,int shared_column,const uint32_t *device_mask=nullptr);
// histogram [65536][channels] -> Quantum-typed LUT + per-channel apply mask, on the device
// cdf_device (optional, equalize): the running counts of channel cdf_column as 65536 uint32 (all of
// them 0xffffffff when the frame has 2^32 pixels or more)
MhStatus launch_build_lut(const View &img,const unsigned long long *hist_device,bool equalize,
  double black_point,double white_limit,void *lut_device,uint32_t *mask_device,
  const unsigned int *colour_flag_device,uint32_t *cdf_device=nullptr,int cdf_column=0);
// EqualizeImage's map evaluated per sample from the running counts (float Quantum, one histogram
// for every channel); lut_device: the tabulated map, the fallback
MhStatus launch_equalize_cdf_apply(const View &img,const uint32_t *cdf_device,const void *lut_device,
  uint32_t apply_mask,const Roles &roles,int shared_column,const uint32_t *device_mask);
// (operators_enhance.cpp) device histogram -> LUT -> apply: the second half of
// ContrastStretchImage / EqualizeImage
MhStatus apply_histogram_lut(const View &view,const MhImage *image,const unsigned long long *hist,
  int mode,bool equalize,double black_point,double white_limit,const unsigned int *colour_flag=nullptr);
// dst[i] += src[i] (histogram tables of the bands of a row-sharded image)
MhStatus launch_table_add(unsigned long long *dst,const unsigned long long *src,size_t count,
  int device,hipStream_t stream);
// (batch.cpp) a new-image stencil operator on host memory as a pipeline of row bands: uploads,
// kernels and downloads of different bands overlap.  kernel: the list of an MH_OP_MORPHOLOGY
// operator (op.text unused then); args[2] = bias.  *handled = false: run the whole-frame path.
MhStatus host_banded_operator(const MhOperator &op,const MhKernelInfo *kernel,const MhImage *image,
  MhImage *result,bool *handled);
// CompositeImage(canvas,source,Difference|Lighten,clip_to_self,0,0) in place on the canvas
enum { MH_COMPOSITE_DIFFERENCE=0,MH_COMPOSITE_LIGHTEN=1,MH_COMPOSITE_DARKEN=2,MH_COMPOSITE_PLUS=3,MH_COMPOSITE_MULTIPLY=4,
  MH_COMPOSITE_SCREEN=5,MH_COMPOSITE_EXCLUSION=6,MH_COMPOSITE_MINUS_SRC=7,MH_COMPOSITE_MINUS_DST=8,MH_COMPOSITE_LINEAR_DODGE=9,
  MH_COMPOSITE_OVER=10,MH_COMPOSITE_DST_OVER=11 };
MhStatus launch_composite(const View &canvas,const View &source,int kind,const Roles &roles);
MhStatus launch_contrast(const View &img,bool sharpen);
MhStatus launch_modulate(const View &img,bool hsb,double hue_shift,double saturation_scale,
  double brightness_scale);
// ModulateImage's other colour models (HCL, HCLp, HSI, HSV, HWB, LCH/LCHab, LCHuv)
MhStatus launch_modulate_generic(const View &img,MhColorspace colorspace,double hue_shift,
  double saturation_scale,double brightness_scale);
// sRGB, linear RGB, Lab, XYZ and the pointwise colourspaces of ConvertRGBToGeneric
bool colorspace_is_accelerated(MhColorspace c);
size_t storage_size(MhStorageType type,MhQuantumKind quantum);
MhStatus launch_pixel_io(bool import,const View &img,const MhImage *desc,int x,int y,int width,
  int height,const char *map,MhStorageType type,void *buffer_device);
// MotionBlurImage's pixel loop: `width` taps at integer (x,y) offsets
MhStatus launch_motion_blur(const View &src,const View &dst,const double *kernel,size_t width,
  const ptrdiff_t *offsets_xy,const Roles &roles);
MhStatus launch_rotational_blur(const View &src,const View &dst,const double *cos_theta,
  const double *sin_theta,size_t n,double blur_radius,const Roles &roles);
MhStatus launch_local_contrast(const View &src,const View &dst,double radius,double strength,
  const Roles &roles);
MhStatus launch_despeckle(const View &src,const View &dst,const Roles &roles);
MhStatus launch_wavelet_denoise(const View &src,const View &dst,double threshold,double softness,
  const Roles &roles);
void release_color_tables();          // frees the per-device transfer-function tables
MhStatus launch_gray_check(const View &img,const MhImage *desc,unsigned int *flag_device);
MhStatus launch_colorspace(const View &img,MhColorspace from,MhColorspace to,const MhImage *desc);
MhStatus launch_stretch_levels_apply(const View &img,const unsigned long long *hist,double black_point,
  double white_limit,uint32_t update_mask,const unsigned int *colour_flag);
MhStatus launch_lab_fast_contrast_stretch(const View &img,const MhImage *lab_desc,double black_point,
  double white_limit,uint32_t update_mask,bool *fused);
MhStatus launch_lab_fast_with_histogram(const View &img,const MhImage *lab_desc,unsigned long long *hist,
  bool *fused);
MhStatus launch_copy(const View &src,const View &dst);
// FAST separable 2-D convolution (pointwise.hip): Q16 -> float sums, float sums -> Q16
MhStatus launch_premultiply(const View &src,const View &sums,bool blend);
MhStatus launch_separable_finish(const View &sums,const View &dst,bool blend);
// a three-channel frame padded to four (pointwise.hip)
MhStatus launch_rgb_pad(const View &src,const View &padded);
MhStatus launch_rgb_unpad(const View &padded,const View &dst,const void *original=nullptr,
  unsigned long long *changed=nullptr);
// a one-channel Q16 frame as four row bands = four channels (pointwise.hip)
MhStatus launch_gray_bands_pack(const View &src,const View &packed,int band,int halo);
MhStatus launch_gray_bands_unpack(const View &packed,const View &dst,int band,int halo,const void *original=nullptr,
  unsigned long long *changed=nullptr);
// ... the two frames and the geometry of one call: band = ceil(rows / 4), `halo` extra rows either side
// of every band.  fits(): the frame has `min_pixels` (MAGICKHIP_GRAY_BANDS_MIN_PIXELS overrides every caller's
// default: the two extra dispatches must pay) and its bands are at least as tall as the rows added to them.
struct GrayBands
{
  View packed,result;
  int band=0,halo=0;
  Temp packed_memory,result_memory;
  static bool fits(const View &src,int halo,long min_pixels)
  {
    const size_t band=(src.rows+3)/4;
    return (halo >= 0) && (src.columns*src.rows >= (size_t) option_long("MAGICKHIP_GRAY_BANDS_MIN_PIXELS",min_pixels)) &&
      (band >= (size_t) 2*(size_t) halo) && (band+2*(size_t) halo <= 65535u) && (src.rows <= 0x7fffffffu/4u) &&
      (src.columns <= 0x7fffffffu);
  }
  MhStatus pack(const View &src,int rows_beyond)
  {
    halo=rows_beyond;
    band=(int) ((src.rows+3)/4);
    packed=src;
    packed.channels=4;
    packed.rows=(size_t) band+2*(size_t) halo;
    result=packed;
    MH_TRY(packed_memory.alloc(src.device,packed.bytes(),src.stream));
    MH_TRY(result_memory.alloc(src.device,result.bytes(),src.stream));
    packed.pixels=packed_memory.ptr;
    result.pixels=result_memory.ptr;
    return launch_gray_bands_pack(src,packed,band,halo);
  }
  MhStatus unpack(const View &dst,const void *original=nullptr,unsigned long long *changed=nullptr) const
  { return launch_gray_bands_unpack(result,dst,band,halo,original,changed); }
  static Roles plain_roles()
  {
    Roles plain;
    plain.update_mask=0xfu;
    return plain;
  }
};
MhStatus launch_grayscale(const View &img,int method,const MhImage *desc);
MhStatus launch_function(const View &img,int function,size_t count,const double *parameters,uint32_t update_mask);

} // namespace mh
