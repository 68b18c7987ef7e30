/*
  magickhip.h — C ABI of libmagickhip.so, an MI355X (gfx950 / CDNA4) native
  accelerate backend for MagickCore.

  This is the drop-in boundary: plain C, plain pointers and sizes, no C++ or
  torch types.  Every operator entry point below replaces (or adds, where the
  reference has no hook) one `Accelerate*Image()` function of the reference:

    reference interface                           file:line
    --------------------------------------------  -----------------------------------
    AccelerateBlurImage                           MagickCore/accelerate-private.h:36-37
    AccelerateResizeImage                         MagickCore/accelerate-private.h:43-44
    AccelerateUnsharpMaskImage                    MagickCore/accelerate-private.h:46-47
    AccelerateContrastStretchImage                MagickCore/accelerate-private.h:52-53
    AccelerateEqualizeImage                       MagickCore/accelerate-private.h:54
    AccelerateFunctionImage                       MagickCore/accelerate-private.h:56-57
    AccelerateGrayscaleImage                      MagickCore/accelerate-private.h:58-59
    (new) convolve / morphology hook              MagickCore/morphology.c:3937, :4219
    (new) colourspace hook                        MagickCore/colorspace.c:1751
    checkAccelerateCondition (the gate)           MagickCore/accelerate.c:110-170
    SetOpenCLEnabled / GetOpenCLEnabled           MagickCore/opencl.h, opencl.c:3192
    GetOpenCLKernelProfileRecords                 MagickCore/opencl.c:2081
    KernelInfo                                    MagickCore/morphology.h:100-127
    MorphologyMethod                              MagickCore/morphology.h:72-98
    PixelTrait                                    MagickCore/pixel.h:146-152
    FilterType                                    MagickCore/resample.h:31-69

  Conventions (same as the reference's accelerate layer):
    * every operator returns MH_OK (0) when it produced the result, and a
      non-zero MhStatus when it did not — the caller then runs its CPU path,
      exactly as it does when Accelerate*Image() returns NULL / MagickFalse
      (MagickCore/effect.c:783-787).  Nothing here throws or aborts.
    * the caller owns every pixel buffer; the library owns device memory,
      streams and staging buffers.
    * pixel layout is the pixel cache's: row-major, channel-interleaved
      Quantum[rows][columns][number_channels] (MagickCore/cache-private.h:130-226,
      MagickCore/pixel.c:6132-6205).  Quantum is `unsigned short` (Q16) or
      `float` (Q16-HDRI) — MagickCore/magick-type.h:80-88.
    * results are those of the reference CPU path: see DESIGN.md "Parity".
*/
#ifndef MAGICKHIP_H
#define MAGICKHIP_H

#include <stddef.h>
#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

#if defined(MAGICKHIP_BUILD)
#  define MH_API __attribute__((visibility("default")))
#else
#  define MH_API
#endif

#define MH_MAX_CHANNELS 4   /* the accelerate gate admits R[,G,B][,A] only: accelerate.c:142-167 */

typedef enum
{
  MH_OK = 0,
  MH_UNSUPPORTED = 1,     /* gate failed: caller should use the CPU path */
  MH_NO_DEVICE = 2,
  MH_BAD_ARGUMENT = 3,
  MH_OUT_OF_MEMORY = 4,
  MH_DEVICE_ERROR = 5,
  MH_DISABLED = 6         /* MhSetEnabled(0) or MAGICK_HIP_DEVICE=off */
} MhStatus;

typedef enum
{
  MH_QUANTUM_U16 = 0,     /* Q16:      Quantum = unsigned short */
  MH_QUANTUM_F32 = 1      /* Q16-HDRI: Quantum = float, nominal range 0..65535 */
} MhQuantumKind;

typedef enum
{
  MH_MEMORY_HOST = 0,     /* pixels is a host pointer (e.g. CacheInfo::pixels); staged with hipMemcpyAsync */
  MH_MEMORY_DEVICE = 1    /* pixels is a device pointer on `device`; used in place */
} MhMemoryKind;

/* PixelTrait bits, MagickCore/pixel.h:146-152 */
enum
{
  MH_TRAIT_UNDEFINED = 0x0,
  MH_TRAIT_COPY = 0x1,
  MH_TRAIT_UPDATE = 0x2,
  MH_TRAIT_BLEND = 0x4
};

/* ColorspaceType, MagickCore/colorspace.h:25-67 (same values).  TransformImageColorspace is
   accelerated between sRGB and every pointwise colourspace of ConvertRGBToGeneric /
   ConvertGenericToRGB (colorspace.c:411-595, :122-305) plus linear RGB; GRAY / LinearGRAY only
   describe single-channel images; CMYK, Log, OHTA, Rec601/709YCbCr, scRGB, YCC, Transparent
   stay on the CPU path. */
typedef enum
{
  MH_COLORSPACE_UNDEFINED = 0,
  MH_COLORSPACE_CMY = 1,
  MH_COLORSPACE_GRAY = 3,
  MH_COLORSPACE_HCL = 4,
  MH_COLORSPACE_HCLP = 5,
  MH_COLORSPACE_HSB = 6,
  MH_COLORSPACE_HSI = 7,
  MH_COLORSPACE_HSL = 8,
  MH_COLORSPACE_HSV = 9,
  MH_COLORSPACE_HWB = 10,
  MH_COLORSPACE_LAB = 11,
  MH_COLORSPACE_LCH = 12,      /* alias of LCHab in the pixel loops (colorspace.c:488-489) */
  MH_COLORSPACE_LCHAB = 13,
  MH_COLORSPACE_LCHUV = 14,
  MH_COLORSPACE_LOG = 15,          /* default film parameters only (colorspace.c:1057-1060) */
  MH_COLORSPACE_LMS = 16,
  MH_COLORSPACE_LUV = 17,
  MH_COLORSPACE_OHTA = 18,         /* the table-driven transforms, colorspace.c:1254-1420, :2591-2790 */
  MH_COLORSPACE_REC601YCBCR = 19,
  MH_COLORSPACE_REC709YCBCR = 20,
  MH_COLORSPACE_RGB = 21,      /* linear RGB */
  MH_COLORSPACE_SCRGB = 22,        /* the same pixel loops as RGB (colorspace.c:1164-1165, :2502-2503) */
  MH_COLORSPACE_SRGB = 23,
  MH_COLORSPACE_XYY = 25,
  MH_COLORSPACE_XYZ = 26,
  MH_COLORSPACE_YCBCR = 27,
  MH_COLORSPACE_YCC = 28,          /* as a target only: YCC -> sRGB (YCCMap) is declined */
  MH_COLORSPACE_YDBDR = 29,
  MH_COLORSPACE_YIQ = 30,
  MH_COLORSPACE_YPBPR = 31,
  MH_COLORSPACE_YUV = 32,
  MH_COLORSPACE_LINEARGRAY = 33,
  MH_COLORSPACE_JZAZBZ = 34,
  MH_COLORSPACE_DISPLAYP3 = 35,
  MH_COLORSPACE_ADOBE98 = 36,
  MH_COLORSPACE_PROPHOTO = 37,
  MH_COLORSPACE_OKLAB = 38,
  MH_COLORSPACE_OKLCH = 39,
  MH_COLORSPACE_CAT02LMS = 40
} MhColorspace;

/* PixelIntensityMethod, MagickCore/pixel.h */
typedef enum
{
  MH_INTENSITY_UNDEFINED = 0,
  MH_INTENSITY_AVERAGE = 1,
  MH_INTENSITY_BRIGHTNESS = 2,
  MH_INTENSITY_LIGHTNESS = 3,
  MH_INTENSITY_MS = 4,
  MH_INTENSITY_REC601LUMA = 5,
  MH_INTENSITY_REC601LUMINANCE = 6,
  MH_INTENSITY_REC709LUMA = 7,
  MH_INTENSITY_REC709LUMINANCE = 8,
  MH_INTENSITY_RMS = 9
} MhIntensityMethod;

/* ChannelType bits that change operator behaviour, MagickCore/pixel.h:60-75 */
#define MH_ALL_CHANNELS 0x7FFFFFFu
#define MH_SYNC_CHANNELS 0x20000u

/*
  One image as the operators see it: what the shim extracts from `Image` /
  `CacheInfo` (SURVEY §8b "Data access").
*/
typedef struct MhImage
{
  void *pixels;
  size_t columns;
  size_t rows;
  uint32_t number_channels;                 /* 1..MH_MAX_CHANNELS */
  uint32_t quantum;                         /* MhQuantumKind */
  uint32_t memory;                          /* MhMemoryKind */
  int32_t device;                           /* HIP device ordinal; -1 = library default; MH_DEVICE_ALL (host memory,
                                               new-image stencil operators on frames of 64 MB and more): the row
                                               bands of the frame go round every (logical) device of the node */
  uint32_t channel_traits[MH_MAX_CHANNELS]; /* PixelTrait of the channel stored at each offset */
  int32_t alpha_offset;                     /* offset of the alpha channel, -1 if none (image->alpha_trait undefined) */
  uint32_t alpha_trait;                     /* image->alpha_trait: MH_TRAIT_BLEND when alpha is active */
  uint32_t colorspace;                      /* MhColorspace */
  uint32_t intensity;                       /* MhIntensityMethod; 0 = default (Rec709Luma) */
  uint32_t channel_mask;                    /* image->channel_mask (ChannelType bits); MH_ALL_CHANNELS by default */
  void *stream;                             /* hipStream_t to run on; NULL = library stream.  DEVICE images only */
  uint32_t precision;                       /* 0 = the library default (MhSetPrecision); else MhPrecision + 1:
                                               the precision of THIS call, whatever other threads run with */
} MhImage;

/* Fill the traits the way InitializePixelChannelMap does for an image with
   `number_channels` channels and optional alpha (pixel.c:6132-6205,
   SetPixelChannelMask pixel.c:6338-6393 with the default channel mask). */
MH_API void MhInitImage(MhImage *image,void *pixels,size_t columns,size_t rows,
  uint32_t number_channels,int has_alpha,MhQuantumKind quantum,
  MhMemoryKind memory);

/* ------------------------------------------------------------------ runtime */

MH_API MhStatus MhInitialize(void);               /* InitializeOpenCL analogue, opencl.c:2430 */
MH_API void MhTerminus(void);                     /* OpenCLTerminus, opencl.c:2576 */
MH_API int MhDeviceCount(void);
#define MH_DEVICE_ALL (-2)
MH_API MhStatus MhSetDevice(int device);          /* default device for device=-1 images */
/* bands the host-image pipeline (MH_DEVICE_ALL included) has finished on a logical device: tests, bench */
MH_API unsigned long long MhBandedBands(int logical_device);
MH_API int MhGetEnabled(void);                    /* GetOpenCLEnabled */
MH_API int MhSetEnabled(int enabled);             /* SetOpenCLEnabled, opencl.c:3192; returns new state */
MH_API const char *MhGetLastError(void);          /* thread-local description of the last non-OK status */
MH_API const char *MhGetVersion(void);

typedef enum
{
  MH_PRECISION_EXACT = 0,  /* FP64, the CPU's operation order, no FMA contraction: bit-identical results */
  MH_PRECISION_FAST = 1    /* FP32 FMA accumulation where the result is still within +-1 Quantum level (Q16 only) */
} MhPrecision;
/* The library-wide DEFAULT: MH_PRECISION_FAST — what an unchanged MagickCore caller gets through the
   shim, within the +-1 level / +-1 ULP the drop-in promises (the reference's own OpenCL path computes
   in float and is not bit-identical to its CPU path either); MAGICK_HIP_PRECISION=exact at start-up
   or MhSetPrecision(MH_PRECISION_EXACT) selects the bit-identical mode.  A call whose first image
   carries a non-zero MhImage::precision runs with that precision instead: two threads can run
   EXACT and FAST operators at the same time. */
MH_API MhPrecision MhGetPrecision(void);
MH_API MhPrecision MhSetPrecision(MhPrecision precision);
#define MH_IMAGE_PRECISION(p) ((uint32_t) (p)+1u)      /* value for MhImage::precision */

/* Options.  Every MAGICKHIP_* / MAGICK_HIP_* environment variable is read ONCE, when the runtime
   initialises (MhInitialize or the first call); later changes of the environment are not seen.
   MhSetOption changes the value the library holds (value == NULL: as if the variable were unset)
   — for tests and A/B measurements; MhGetOption returns it (NULL when unset; the pointer stays
   valid).  Names as documented in DESIGN.md, e.g. "MAGICKHIP_NO_EXACT_MFMA". */
MH_API MhStatus MhSetOption(const char *name,const char *value);
MH_API const char *MhGetOption(const char *name);

/* What the device arbitration of the MagickCore binding needs (the reference: RequestOpenCLDevice,
   opencl.c:3056-3102; AcquireOpenCLCommandQueue, opencl.c:656; GetOpenCLDevices and the device
   getters, opencl.c:1823-2130).  `device` is a HIP ordinal below MhDeviceCount().
   MhLogicalDeviceCount: the number of devices a caller should arbitrate over — MhDeviceCount(),
   or MAGICKHIP_LOGICAL_DEVICES when that is larger (logical device d runs on physical device
   d mod MhDeviceCount(): how a one-GPU box rehearses the multi-GPU paths). */
typedef struct MhDeviceInfo
{
  char name[128];            /* e.g. "AMD Instinct MI355X" */
  char architecture[64];     /* e.g. "gfx950:sramecc+:xnack-" */
  int32_t compute_units;
  int32_t clock_mhz;
  uint64_t global_memory;    /* bytes */
  uint64_t local_memory;     /* LDS bytes per workgroup */
} MhDeviceInfo;
MH_API int MhLogicalDeviceCount(void);
MH_API MhStatus MhGetDeviceInfo(int device,MhDeviceInfo *info);
/* A non-blocking stream on `device` (the caller passes it as MhImage::stream and to the transfer
   helpers); destroyed by MhStreamDestroy or MhTerminus. */
MH_API MhStatus MhStreamCreate(int device,void **stream);
MH_API MhStatus MhStreamDestroy(int device,void *stream);
/* Image-sized device blocks from the library's stream-ordered caching pool: no hipMalloc /
   hipFree (a device-wide synchronisation) per operator result.  `stream` = the stream the block
   was last used on; MhDeviceFreeAsync returns at once, and the block is handed out again only
   behind the work enqueued on that stream. */
MH_API MhStatus MhDeviceAllocAsync(int device,size_t bytes,void *stream,void **ptr);
MH_API MhStatus MhDeviceFreeAsync(int device,void *ptr,void *stream);

/* Device memory helpers for callers that keep images resident.  MhUpload returns once the
   transfer is enqueued on `stream` and `src_host` may be reused; MhDownload returns once
   `dst_host` holds the data.  Pageable host blocks of 8 MiB and more move through page-locked
   staging buffers on a few host threads (replaces opencl.c's clEnqueueMapBuffer round trips,
   MagickCore/cache.c:5341-5353). */
MH_API MhStatus MhDeviceAlloc(int device,size_t bytes,void **ptr);
MH_API MhStatus MhDeviceFree(int device,void *ptr);
MH_API MhStatus MhUpload(int device,void *dst_device,const void *src_host,size_t bytes,void *stream);
MH_API MhStatus MhDownload(int device,void *dst_host,const void *src_device,size_t bytes,void *stream);
MH_API MhStatus MhSynchronize(int device,void *stream);

/* Page-locked host memory for pixel caches (what SetMagickAlignedMemoryMethods,
   MagickCore/memory.c:1541, lets an application install behind AcquireAlignedMemory,
   cache.c:3754-3758): a host image whose `pixels` block came from MhHostAlloc moves over the
   link with one DMA transfer per direction, no staging threads.  MhHostFree returns 1 when the
   block was one of MhHostAlloc's (and is now released), 0 when it is not known (the caller then
   releases it its own way).  MhHostAllocatedBytes: the bytes currently handed out. */
MH_API void *MhHostAlloc(size_t bytes);
MH_API int MhHostFree(void *block);
MH_API size_t MhHostAllocatedBytes(void);
/* ... plus the released blocks the library keeps page-locked for reuse (MAGICKHIP_PINNED_SPARE_BYTES,
   1 GiB by default): what a budget on page-locked memory has to count */
MH_API size_t MhHostPinnedBytes(void);

/* Kernel profile records (GetOpenCLKernelProfileRecords analogue). */
typedef struct MhKernelProfileRecord
{
  const char *kernel_name;
  unsigned long count;
  double min_ms,max_ms,total_ms;   /* hipEvent-timed, only while profiling is enabled */
} MhKernelProfileRecord;
MH_API int MhSetProfileEnabled(int enabled);      /* SetOpenCLKernelProfileEnabled, opencl.c:3162 */
MH_API size_t MhGetProfileRecords(MhKernelProfileRecord *records,size_t capacity);
/* ... of one device only (GetOpenCLKernelProfileRecords(device, &length), opencl.c:2081) */
MH_API size_t MhGetDeviceProfileRecords(int device,MhKernelProfileRecord *records,size_t capacity);
MH_API void MhResetProfileRecords(void);

/* Diagnostics of the exact-integer blur (convolve_fused_exact.hip): the number of samples whose
   level the integer sums' error bound could not decide and that were recomputed in the
   reference's own operation order (morphology.c:2746-2764).  enable != 0 starts counting on the
   current device (the counter is read and reset by every call); returns the count so far. */
MH_API unsigned long long MhExactBlurRecomputed(int enable);
/* The same for the separable EXACT 2-D Convolve (GaussianBlurImage in EXACT mode and on float
   Quantum): samples recomputed in the reference's w x h order. */
MH_API unsigned long long MhSeparableRecomputed(int enable);
/* Diagnostic of the exact-integer 2-D convolve (cells that are integer multiples of a unit: every
   flat shape kernel; morphology.c:2919-2979 on the i8 matrix cores, bit-identical): samples that
   lay within the reference's own rounding error of a Quantum boundary and were recomputed in the
   reference's order since the last call (enable as above). */
MH_API unsigned long long MhConvolve2DRecomputed(int enable);
/* ... and of the fused fp64 2-D convolve that takes every other kernel and frame (one fused
   multiply-add per cell and channel over alpha-premultiplied doubles + tie check, bit-identical). */
MH_API unsigned long long MhConvolve2DTieRecomputed(int enable);


/* ------------------------------------------------------ kernels and filters */

/* KernelInfoType, MagickCore/morphology.h:30-70 (same order) */
typedef enum
{
  MH_KERNEL_UNDEFINED = 0,
  MH_KERNEL_UNITY, MH_KERNEL_GAUSSIAN, MH_KERNEL_DOG, MH_KERNEL_LOG, MH_KERNEL_BLUR,
  MH_KERNEL_COMET, MH_KERNEL_BINOMIAL, MH_KERNEL_LAPLACIAN, MH_KERNEL_SOBEL,
  MH_KERNEL_FREICHEN, MH_KERNEL_ROBERTS, MH_KERNEL_PREWITT, MH_KERNEL_COMPASS,
  MH_KERNEL_KIRSCH, MH_KERNEL_DIAMOND, MH_KERNEL_SQUARE, MH_KERNEL_RECTANGLE,
  MH_KERNEL_OCTAGON, MH_KERNEL_DISK, MH_KERNEL_PLUS, MH_KERNEL_CROSS, MH_KERNEL_RING,
  MH_KERNEL_PEAKS, MH_KERNEL_EDGES, MH_KERNEL_CORNERS, MH_KERNEL_DIAGONALS,
  MH_KERNEL_LINEENDS, MH_KERNEL_LINEJUNCTIONS, MH_KERNEL_RIDGES, MH_KERNEL_CONVEXHULL,
  MH_KERNEL_THINSE, MH_KERNEL_SKELETON, MH_KERNEL_CHEBYSHEV, MH_KERNEL_MANHATTAN,
  MH_KERNEL_OCTAGONAL, MH_KERNEL_EUCLIDEAN, MH_KERNEL_USERDEFINED
} MhKernelInfoType;

/* Mirror of KernelInfo, MagickCore/morphology.h:100-127.  values is row-major
   height x width; NaN marks a cell that is not part of the neighbourhood. */
typedef struct MhKernelInfo
{
  MhKernelInfoType type;
  size_t width,height;
  ptrdiff_t x,y;
  double *values;
  double minimum,maximum,negative_range,positive_range,angle;
  struct MhKernelInfo *next;
} MhKernelInfo;

/* AcquireKernelInfo, morphology.c:485: "name:args" or "WxH+X+Y: v,v,..."
   kernel strings, ';'-separated lists.  NULL on parse failure / unsupported
   kernel name. */
MH_API MhKernelInfo *MhAcquireKernelInfo(const char *kernel_string);
MH_API MhKernelInfo *MhDestroyKernelInfo(MhKernelInfo *kernel);
MH_API MhKernelInfo *MhCloneKernelInfo(const MhKernelInfo *kernel);
/* ScaleKernelInfo, morphology.c:4571.  flags: 1 = Normalize, 2 = CorrelateNormalize */
MH_API void MhScaleKernelInfo(MhKernelInfo *kernel,double scaling_factor,unsigned flags);
/* GetOptimalKernelWidth1D / 2D, gem.c:262,302 */
MH_API size_t MhGetOptimalKernelWidth1D(double radius,double sigma);
MH_API size_t MhGetOptimalKernelWidth2D(double radius,double sigma);
/* Is the 2-D kernel an outer product column[y]*row[x] (to 1e-13 of its largest cell, no NaN
   cells)?  Returns 1 and fills row[width] / column[height] (either may be NULL), else 0.  FAST
   ConvolveImage runs such kernels (Gaussian:RxS, i.e. GaussianBlurImage) as two 1-D passes
   with one division at the end instead of morphology.c:2892-2979's width*height taps. */
MH_API int MhKernelOuterProductFactors(const MhKernelInfo *kernel,double *row,double *column);
/* ... or an outer product everywhere but at its origin cell (SharpenImage's negated Gaussian,
   EdgeImage's box of -1: effect.c:3640-3660, :1530-1545)?  Returns 1 (outer product, *delta = 0),
   2 (values[y][x] = column[y]*row[x] + *delta at the origin cell) or 0.  EXACT mode and float
   Quantum run both forms as two fp64 passes with a tie check, bit-identical to the w x h walk. */
MH_API int MhKernelOuterProductPlusDelta(const MhKernelInfo *kernel,double *row,double *column,
  double *delta);
/* Host test of the exact-integer 2-D convolve (MhConvolve2DRecomputed above): are the kernel's cells integer multiples (|m| <= 127) of one unit — every
   flat shape kernel after `convolve:scale`, binomial and hand-written integer kernels?  cells[]
   (width*height ints in the kernel's own order, NaN cells as 0; may be NULL) and *unit (may be NULL)
   receive the form values[i] = cells[i] * unit (to 1e-9 of a cell); returns 1 or 0. */
MH_API int MhKernelIntegerCells(const MhKernelInfo *kernel,int *cells,double *unit);

/* MorphologyMethod, MagickCore/morphology.h:72-98 (same values) */
typedef enum
{
  MH_MORPHOLOGY_UNDEFINED = 0,
  MH_MORPHOLOGY_CONVOLVE, MH_MORPHOLOGY_CORRELATE,
  MH_MORPHOLOGY_ERODE, MH_MORPHOLOGY_DILATE,
  MH_MORPHOLOGY_ERODE_INTENSITY, MH_MORPHOLOGY_DILATE_INTENSITY,
  MH_MORPHOLOGY_ITERATIVE_DISTANCE,
  MH_MORPHOLOGY_OPEN, MH_MORPHOLOGY_CLOSE,
  MH_MORPHOLOGY_OPEN_INTENSITY, MH_MORPHOLOGY_CLOSE_INTENSITY,
  MH_MORPHOLOGY_SMOOTH,
  MH_MORPHOLOGY_EDGE_IN, MH_MORPHOLOGY_EDGE_OUT, MH_MORPHOLOGY_EDGE,
  MH_MORPHOLOGY_TOP_HAT, MH_MORPHOLOGY_BOTTOM_HAT,
  MH_MORPHOLOGY_HIT_AND_MISS, MH_MORPHOLOGY_THINNING, MH_MORPHOLOGY_THICKEN,
  MH_MORPHOLOGY_DISTANCE, MH_MORPHOLOGY_VORONOI
} MhMorphologyMethod;

/* FilterType, MagickCore/resample.h:31-69 (same values) */
typedef enum
{
  MH_FILTER_UNDEFINED = 0,
  MH_FILTER_POINT, MH_FILTER_BOX, MH_FILTER_TRIANGLE, MH_FILTER_HERMITE,
  MH_FILTER_HANN, MH_FILTER_HAMMING, MH_FILTER_BLACKMAN, MH_FILTER_GAUSSIAN,
  MH_FILTER_QUADRATIC, MH_FILTER_CUBIC, MH_FILTER_CATROM, MH_FILTER_MITCHELL,
  MH_FILTER_JINC, MH_FILTER_SINC, MH_FILTER_SINCFAST, MH_FILTER_KAISER,
  MH_FILTER_WELCH, MH_FILTER_PARZEN, MH_FILTER_BOHMAN, MH_FILTER_BARTLETT,
  MH_FILTER_LAGRANGE, MH_FILTER_LANCZOS, MH_FILTER_LANCZOSSHARP,
  MH_FILTER_LANCZOS2, MH_FILTER_LANCZOS2SHARP, MH_FILTER_ROBIDOUX,
  MH_FILTER_ROBIDOUXSHARP, MH_FILTER_COSINE, MH_FILTER_SPLINE,
  MH_FILTER_LANCZOSRADIUS, MH_FILTER_CUBICSPLINE, MH_FILTER_MAGICKERNELSHARP2013,
  MH_FILTER_MAGICKERNELSHARP2021, MH_FILTER_SENTINEL
} MhFilterType;

/* ResizeFilter (opaque), resize.c:91-108; AcquireResizeFilter resize.c:803
   (cylindrical = False, no expert `filter:*` artifacts). */
typedef struct MhResizeFilter MhResizeFilter;
MH_API MhResizeFilter *MhAcquireResizeFilter(MhFilterType filter,int image_has_alpha_or_enlarging_hint);
/* A filter whose weights are evaluated by the caller: the MagickCore shim passes the
   reference's own GetResizeFilterWeight / GetResizeFilterSupport (resize.c:1690, :1668) for
   the ResizeFilter AccelerateResizeImage receives (accelerate-private.h:43-44), so expert
   `filter:*` artifacts are honoured.  The callback runs on the host while the tap tables are
   built (2*(columns+rows)*taps calls), never on the device. */
typedef double (*MhResizeWeightFunction)(void *user,double x);
MH_API MhResizeFilter *MhAcquireResizeFilterFromCallback(MhResizeWeightFunction weight,
  void *user,double support);
MH_API MhResizeFilter *MhDestroyResizeFilter(MhResizeFilter *filter);
MH_API double MhGetResizeFilterWeight(const MhResizeFilter *filter,double x);   /* resize.c:1690 */
MH_API double MhGetResizeFilterSupport(const MhResizeFilter *filter);           /* resize.c:1668 */

/* --------------------------------------------------------------- operators */
/*
  New-image operators take a caller-allocated destination whose geometry,
  quantum kind and channel layout the caller has set (CloneImage in the shim,
  accelerate.c:239-256); in-place operators mutate `image`.
*/

/* AccelerateBlurImage: BlurImage(image,radius,sigma), effect.c:765-796. */
MH_API MhStatus MagickHipBlurImage(const MhImage *image,MhImage *blur_image,
  double radius,double sigma);

/* ConvolveImage(image,kernel), effect.c:1170 -> MorphologyImage(Convolve,1). */
MH_API MhStatus MagickHipConvolveImage(const MhImage *image,MhImage *convolve_image,
  const MhKernelInfo *kernel);

/* The other callers of ConvolveImage: each builds its kernel on the host
   exactly as the reference does and runs MorphologyImage(Convolve,1).
     GaussianBlurImage  effect.c:1709-1735  kernel "gaussian:RxS"
     SharpenImage       effect.c:3991-4062  negated 2-D gaussian, centre -2*sum, normalised
     EdgeImage          effect.c:1523-1566  all -1, centre width*width-1 (width from sigma 0.5)
     EmbossImage        effect.c:1600-1678  anti-diagonal signed gaussian, then EqualizeImage */
MH_API MhStatus MagickHipGaussianBlurImage(const MhImage *image,MhImage *blur_image,
  double radius,double sigma);
MH_API MhStatus MagickHipSharpenImage(const MhImage *image,MhImage *sharp_image,
  double radius,double sigma);
MH_API MhStatus MagickHipEdgeImage(const MhImage *image,MhImage *edge_image,double radius);
MH_API MhStatus MagickHipEmbossImage(const MhImage *image,MhImage *emboss_image,
  double radius,double sigma);

/* AccelerateWaveletDenoiseImage: WaveletDenoiseImage(image,threshold,softness),
   visual-effects.c:3520-3760.  (The reference's hook drops `softness`; the shim's build-time
   patch passes it.)  Images below 33 pixels on a side are left to the CPU. */
MH_API MhStatus MagickHipWaveletDenoiseImage(const MhImage *image,MhImage *noise_image,
  double threshold,double softness);

/* AccelerateDespeckleImage: DespeckleImage(image), effect.c:1308-1490 (16 Hull sweeps). */
MH_API MhStatus MagickHipDespeckleImage(const MhImage *image,MhImage *despeckle_image);

/* AccelerateLocalContrastImage: LocalContrastImage(image,radius,strength), effect.c:1760-2010.
   MH_UNSUPPORTED (CPU path) when the blur width 0.002*max(columns,rows)*|radius| is 0 or does
   not leave room for the mirrored padding. */
MH_API MhStatus MagickHipLocalContrastImage(const MhImage *image,MhImage *contrast_image,
  double radius,double strength);

/* AccelerateRotationalBlurImage: RotationalBlurImage(image,angle), effect.c:3209-3430. */
MH_API MhStatus MagickHipRotationalBlurImage(const MhImage *image,MhImage *blur_image,double angle);

/* AccelerateMotionBlurImage: MotionBlurImage, effect.c:2347-2560.  The first form builds the
   kernel (GetMotionBlurKernel, :2316-2345) and the offsets along `angle` (:2385-2393) itself;
   the second takes them from the caller, as the reference's accelerate hook does
   (`offsets_xy` = width pairs x,y). */
MH_API MhStatus MagickHipMotionBlurImage(const MhImage *image,MhImage *blur_image,double radius,
  double sigma,double angle);
MH_API MhStatus MagickHipMotionBlurImageWithKernel(const MhImage *image,MhImage *blur_image,
  const double *kernel,size_t width,const ptrdiff_t *offsets_xy);

/* MorphologyImage / MorphologyApply, morphology.c:4129 / :3634, with the per-method default
   handling of multi-kernel results (re-iterate; HitAndMiss: Lighten union); `bias` is the
   convolve:bias artifact (0 by default). */
MH_API MhStatus MagickHipMorphologyImage(const MhImage *image,MhImage *morphology_image,
  MhMorphologyMethod method,ptrdiff_t iterations,const MhKernelInfo *kernel,
  double bias);

/* The same with the user's `morphology:compose` (morphology.c:4206-4215, :3779-3782): how the
   results of the kernels of a list are merged.  DEFAULT = UndefinedCompositeOp, NONE =
   NoCompositeOp (re-iterate the previous result), LIGHTEN / DIFFERENCE / DARKEN / PLUS / MULTIPLY /
   SCREEN / EXCLUSION / MINUS_SRC / MINUS_DST / LINEAR_DODGE / OVER / DST_OVER = CompositeImage with
   that operator (composite.c:2396-3124, synchronised channels).
   Other operators: MH_UNSUPPORTED (the CPU path runs). */
typedef enum
{
  MH_MORPHOLOGY_COMPOSE_DEFAULT = 0,
  MH_MORPHOLOGY_COMPOSE_NONE = 1,
  MH_MORPHOLOGY_COMPOSE_LIGHTEN = 2,
  MH_MORPHOLOGY_COMPOSE_DIFFERENCE = 3,
  MH_MORPHOLOGY_COMPOSE_OTHER = 4,
  MH_MORPHOLOGY_COMPOSE_DARKEN = 5,
  MH_MORPHOLOGY_COMPOSE_PLUS = 6,      /* `-define morphology:compose=Plus`, morphology.c:772 */
  MH_MORPHOLOGY_COMPOSE_MULTIPLY = 7,
  MH_MORPHOLOGY_COMPOSE_SCREEN = 8,
  MH_MORPHOLOGY_COMPOSE_EXCLUSION = 9,
  MH_MORPHOLOGY_COMPOSE_MINUS_SRC = 10,
  MH_MORPHOLOGY_COMPOSE_MINUS_DST = 11,
  MH_MORPHOLOGY_COMPOSE_LINEAR_DODGE = 12,
  MH_MORPHOLOGY_COMPOSE_OVER = 13,     /* OverCompositeOp and SrcOverCompositeOp: CompositeOverImage, composite.c:917 */
  MH_MORPHOLOGY_COMPOSE_DST_OVER = 14
} MhMorphologyCompose;
MH_API MhStatus MagickHipMorphologyImageCompose(const MhImage *image,MhImage *morphology_image,
  MhMorphologyMethod method,ptrdiff_t iterations,const MhKernelInfo *kernel,
  double bias,MhMorphologyCompose compose);

/* One MorphologyPrimitive pass (morphology.c:2566) with a single kernel;
   *changed receives the reference's return value. */
MH_API MhStatus MagickHipMorphologyPrimitive(const MhImage *image,MhImage *morphology_image,
  MhMorphologyMethod method,const MhKernelInfo *kernel,double bias,
  ptrdiff_t *changed);

/* AccelerateUnsharpMaskImage: UnsharpMaskImage, effect.c:4256-4400. */
MH_API MhStatus MagickHipUnsharpMaskImage(const MhImage *image,MhImage *unsharp_image,
  double radius,double sigma,double gain,double threshold);

/* AccelerateResizeImage: ResizeImage, resize.c:3761-3874.  resize_image
   carries the target columns/rows. */
MH_API MhStatus MagickHipResizeImage(const MhImage *image,MhImage *resize_image,
  MhFilterType filter);
/* As above with a filter object (what the shim passes after AcquireResizeFilter). */
MH_API MhStatus MagickHipResizeImageWithFilter(const MhImage *image,MhImage *resize_image,
  const MhResizeFilter *filter);

/* AccelerateContrastStretchImage: ContrastStretchImage, enhance.c:1544-1818.
   black_point / white_point are pixel counts as in the MagickCore API.
   *became_gray (optional) reports the IdentifyImageType side effect
   (enhance.c:1586-1588): when set, the caller must SetImageColorspace(GRAY). */
MH_API MhStatus MagickHipContrastStretchImage(MhImage *image,double black_point,
  double white_point,int *became_gray);

/* AccelerateEqualizeImage: EqualizeImage, enhance.c:2040-2280. */
MH_API MhStatus MagickHipEqualizeImage(MhImage *image);

/* TransformImageColorspace, colorspace.c:1751 — sRGB <-> linear RGB / Lab / XYZ.
   On success image->colorspace is updated. */
MH_API MhStatus MagickHipTransformImageColorspace(MhImage *image,MhColorspace colorspace);

/* AccelerateGrayscaleImage (accelerate-private.h:58-59): GrayscaleImage, enhance.c:2476-2660.
   Writes the intensity to the first channel; the caller then sets the image's colourspace to
   GRAY (LinearGRAY for the two Luminance methods) as enhance.c:2502-2510 does. */
MH_API MhStatus MagickHipGrayscaleImage(MhImage *image,MhIntensityMethod method);

/* MagickFunction, MagickCore/statistic.h:129-136 (same values) */
typedef enum
{
  MH_FUNCTION_UNDEFINED = 0,
  MH_FUNCTION_ARCSIN,
  MH_FUNCTION_ARCTAN,
  MH_FUNCTION_POLYNOMIAL,
  MH_FUNCTION_SINUSOID
} MhFunction;

/* StorageType, MagickCore/pixel.h:146-156 (same values) */
typedef enum
{
  MH_STORAGE_UNDEFINED = 0,
  MH_STORAGE_CHAR, MH_STORAGE_DOUBLE, MH_STORAGE_FLOAT, MH_STORAGE_LONG, MH_STORAGE_LONGLONG,
  MH_STORAGE_QUANTUM, MH_STORAGE_SHORT
} MhStorageType;

/* ImportImagePixels / ExportImagePixels, pixel.c:4164 / :1962: converts between
   the caller's interleaved component buffer and the Quantum pixels of `image`, for the region
   x,y,width,height (inside the image).  `map` lists the buffer's components, any order of
   R,G,B,A,O (= alpha),I (gray / intensity),P (pad); C,M,Y,K are not accelerated.  `pixels` is
   width*height*strlen(map) tightly packed elements of `type`, in host or device memory
   (`pixels_memory`).  The image must already have the layout the map implies (alpha channel
   present for A/O; the reference's side effects on alpha_trait / colourspace are the caller's). */
MH_API MhStatus MagickHipImportImagePixels(MhImage *image,ptrdiff_t x,ptrdiff_t y,size_t width,
  size_t height,const char *map,MhStorageType type,const void *pixels,MhMemoryKind pixels_memory);
MH_API MhStatus MagickHipExportImagePixels(const MhImage *image,ptrdiff_t x,ptrdiff_t y,
  size_t width,size_t height,const char *map,MhStorageType type,void *pixels,
  MhMemoryKind pixels_memory);

/* AccelerateContrastImage: ContrastImage(image,sharpen), enhance.c:1392-1480 — brightness
   pushed along a sine in HSB (Contrast(), :1370-1390).  R,G,B[,A] layouts. */
MH_API MhStatus MagickHipContrastImage(MhImage *image,int sharpen);

/* AccelerateModulateImage: ModulateImage's pixel loop, enhance.c:3776-3860, for the HSL
   (default, colorspace = MH_COLORSPACE_UNDEFINED or MH_COLORSPACE_HSL) and HSB models.  The
   three percentages are the parsed "brightness,saturation,hue" geometry (100 = unchanged). */
MH_API MhStatus MagickHipModulateImage(MhImage *image,double percent_brightness,
  double percent_saturation,double percent_hue,int colorspace);

/* AccelerateFunctionImage (accelerate-private.h:56-57): FunctionImage / ApplyFunction,
   statistic.c:975-1160, on every channel whose trait carries Update. */
MH_API MhStatus MagickHipFunctionImage(MhImage *image,MhFunction function,
  size_t number_parameters,const double *parameters);

/* ---------------------------------------------------------- building blocks */
/* Exposed so a row-sharded image (one band per GPU / per process) can run the
   global-histogram operators with one all-reduce between the phases
   (SURVEY §8e): histogram -> [all-reduce] -> LUT -> apply. */

#define MH_MAXMAP 65535u
#define MH_HISTOGRAM_BINS (MH_MAXMAP+1u)

/* histogram[bin*number_channels + c], uint64 counts.
   mode 0: each channel bins its own value; mode 1: every channel bins the
   pixel intensity (enhance.c:1637-1643, :2125-2129).  Accumulates into
   `histogram` (caller zeroes it).  histogram follows image->memory. */
MH_API MhStatus MagickHipHistogram(const MhImage *image,int intensity_mode,uint64_t *histogram);

/* Host-side LUT builders (pure C, no device).  lut[bin*number_channels + c],
   in Quantum units as double; apply_mask bit c is cleared when channel c must
   be left untouched (black == white). */
MH_API MhStatus MhContrastStretchLUT(const uint64_t *histogram,uint32_t number_channels,
  size_t columns,size_t rows,double black_point,double white_point,
  MhQuantumKind quantum,double *lut,uint32_t *apply_mask);
MH_API MhStatus MhEqualizeLUT(const uint64_t *histogram,uint32_t number_channels,
  MhQuantumKind quantum,double *lut,uint32_t *apply_mask);

/* q[c] = ClampToQuantum(lut[ScaleQuantumToMap(q[c])*number_channels + c]) for
   Update channels selected by apply_mask (enhance.c:1778-1788, :2252-2262).
   `lut` is a host pointer. */
MH_API MhStatus MagickHipApplyLUT(MhImage *image,const double *lut,uint32_t apply_mask);

/* The second half of EqualizeImage (equalize != 0) / ContrastStretchImage for a caller that
   already holds the histogram — e.g. the table all-reduced over the row bands of a sharded
   image: LUT construction (enhance.c:1652-1706, :2138-2169) and application on the
   device.  `histogram` follows image->memory; image_rows = rows of the WHOLE image (0: this
   image's), whose pixel count the white point refers to. */
MH_API MhStatus MagickHipApplyHistogram(MhImage *image,const uint64_t *histogram,int intensity_mode,
  int equalize,double black_point,double white_point,size_t image_rows);

/* TransformImageColorspace(image, colorspace) followed by ContrastStretchImage(image, black_point,
   white_point) — colorspace.c:1751-1783 then enhance.c:1544-1818 — as one call with the two
   calls' results.  What the pair can share is the pass over the pixels: a FAST sRGB -> Lab of a
   device-resident RGBA Q16 frame converts and bins the intensity of what it stores in one kernel
   (BASELINE configs[3]: one frame read + one frame write before the map is applied).  Any other
   combination runs the two operators one after the other.  MagickHipBatchImages uses it for
   adjacent MH_OP_COLORSPACE / MH_OP_CONTRAST_STRETCH operators. */
MH_API MhStatus MagickHipTransformColorspaceContrastStretchImage(MhImage *image,MhColorspace colorspace,
  double black_point,double white_point);

/* IdentifyImageGray scan (attribute.c:1564-1626): *is_gray = 1 when every
   pixel has |R-G| and |G-B| below MagickEpsilon. */
MH_API MhStatus MagickHipIsImageGray(const MhImage *image,int *is_gray);

/* ------------------------------------------------- batches and several GPUs */
/*
  SURVEY section 8e.  The reference arbitrates devices and in-order queues per call
  (RequestOpenCLDevice, MagickCore/opencl.c:3056-3102; AcquireOpenCLCommandQueue, :656; 16
  queues per device, opencl-private.h:70) but hands every operator exactly one device.  These
  two entry points take a CHAIN of operators and spread the work over the GPUs of the node
  from plain C:

    MagickHipBatchImages    independent images (BASELINE config C4): a work queue over
                            devices x streams_per_device host threads, each with its own
                            stream, so the upload of one image, the kernels of another and the
                            download of a third overlap on every device.  No collective.
    MagickHipShardedImage   ONE large image cut into row bands, one per device (config C5):
                            stencil operators exchange their halo rows between neighbouring
                            bands with hipMemcpyPeerAsync before each pass; ContrastStretch and
                            Equalize bin their band, all-reduce the 65536 x channels table
                            (RCCL ncclAllReduce when the bands sit on different GPUs and
                            librccl loads, peer copies + an add kernel otherwise) and build and
                            apply the identical LUT on every device.

  number_devices <= 0 means MhDeviceCount().  More logical devices than physical ones are
  mapped round-robin (logical d runs on physical d mod MhDeviceCount()): that is how the test
  suite rehearses both entry points on a single GPU.
*/
typedef enum
{
  MH_OP_BLUR = 1,              /* args: radius, sigma */
  MH_OP_GAUSSIAN_BLUR = 2,     /* args: radius, sigma */
  MH_OP_UNSHARP_MASK = 3,      /* args: radius, sigma, gain, threshold */
  MH_OP_RESIZE = 4,            /* args: columns, rows, MhFilterType        (batch only) */
  MH_OP_MORPHOLOGY = 5,        /* args: MhMorphologyMethod, iterations, bias; text: kernel string */
  MH_OP_COLORSPACE = 6,        /* args: MhColorspace */
  MH_OP_CONTRAST_STRETCH = 7,  /* args: black_point, white_point (pixel counts, enhance.c:1544) */
  MH_OP_EQUALIZE = 8
} MhOperatorKind;

typedef struct MhOperator
{
  uint32_t kind;               /* MhOperatorKind */
  double args[4];
  const char *text;
} MhOperator;

typedef struct MhBatchReport
{
  uint32_t devices;            /* logical devices used */
  uint32_t workers;            /* host threads = devices x streams per device */
  uint32_t used_rccl;          /* MagickHipShardedImage: the table went through ncclAllReduce */
  uint32_t halo_exchanges;     /* MagickHipShardedImage: hipMemcpyPeerAsync halo copies issued */
  uint64_t images_per_device[16];
  double seconds;              /* wall time of the call */
} MhBatchReport;

/* results[i] = operators(images[i]).  images / results: host or device memory; results[i]
   carries the geometry of the chain's output (equal to the input's unless the chain resizes).
   results == NULL: the chain must keep the geometry and images[i] is overwritten.
   Returns the first non-OK status of any image (the others still run). */
MH_API MhStatus MagickHipBatchImages(const MhOperator *operators,size_t number_operators,
  const MhImage *images,MhImage *results,size_t number_images,int number_devices,
  int streams_per_device,MhBatchReport *report);

/* result = operators(image) with the rows of `image` sharded over number_devices bands.
   Geometry-preserving operators only (no MH_OP_RESIZE); host or device memory. */
MH_API MhStatus MagickHipShardedImage(const MhOperator *operators,size_t number_operators,
  const MhImage *image,MhImage *result,int number_devices,MhBatchReport *report);

#if defined(__cplusplus)
}
#endif

#endif /* MAGICKHIP_H */
