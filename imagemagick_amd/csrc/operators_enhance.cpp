// In-place operators: ContrastStretchImage, EqualizeImage,
// TransformImageColorspace, and the histogram / LUT building blocks they are
// made of (exposed so a row-sharded image can all-reduce the histogram between
// the phases — SURVEY §8e).
//
// The histogram and the LUT application are HIP kernels (pointwise.hip); the
// 65536-entry scans that turn a histogram into a LUT are restated here on the
// host from MagickCore/enhance.c:1652-1706 (contrast stretch) and :2138-2169
// (equalize), in the reference's arithmetic.
#include "mh_internal.hpp"

#include <cmath>
#include <cstring>

namespace mh {

static double perceptible_reciprocal(double x)
{
  double sign=x < 0.0 ? -1.0 : 1.0;
  if ((sign*x) >= kMagickEpsilon)
    return 1.0/x;
  return sign/kMagickEpsilon;
}

// ScaleMapToQuantum, quantum-private.h:465-476, result as the Quantum it
// would be stored in (unsigned short or float), widened to double
static double scale_map_to_quantum(double value,MhQuantumKind quantum)
{
  if (value <= 0.0)
    return 0.0;
  if (value >= (double) MH_MAXMAP)
    return kQuantumRange;
  if (quantum == MH_QUANTUM_U16)
    return (double) (unsigned short) (value+0.5);
  return (double) (float) value;
}

struct InPlace
{
  DeviceGuard guard;         // declared first: the device is restored after img is gone
  Resident img;
  MhStatus open(MhImage *image)
  {
    int device=resolve_device(image);
    hipStream_t stream=image->memory == MH_MEMORY_DEVICE ? (hipStream_t) image->stream :
      library_stream(device);
    MH_TRY(img.open(image,2,stream,device));
    img.view.stream=stream;
    img.view.device=device;
    MH_HIP(guard.enter(device));
    return MH_OK;
  }
};

static MhStatus check_image(const MhImage *image,const char *what)
{
  MH_TRY(runtime_ready());
  MH_TRY(validate_image(image,what));
  set_call_precision(image);       // MhImage::precision of this call, else the library default
  return MH_OK;
}

// device histogram -> host
static MhStatus histogram_to_host(const View &view,int mode,const MhImage *desc,
  std::vector<unsigned long long> &host)
{
  const size_t n=(size_t) MH_HISTOGRAM_BINS*(size_t) view.channels;
  Temp hist;
  MH_TRY(hist.alloc(view.device,n*sizeof(unsigned long long),view.stream));
  MH_HIP(hipMemsetAsync(hist.ptr,0,n*sizeof(unsigned long long),view.stream));
  MH_TRY(launch_histogram(view,mode,desc,hist.as<unsigned long long>()));
  host.resize(n);
  MH_HIP(hipMemcpyAsync(host.data(),hist.ptr,n*sizeof(unsigned long long),
    hipMemcpyDeviceToHost,view.stream));
  MH_HIP(hipStreamSynchronize(view.stream));
  return MH_OK;
}

// LUT (double, Quantum-valued) -> device LUT in the image's Quantum type -> apply
static MhStatus apply_lut_host(const View &view,const MhImage *desc,const double *lut,
  uint32_t apply_mask)
{
  const size_t n=(size_t) MH_HISTOGRAM_BINS*(size_t) view.channels;
  Temp d_lut;
  int shared_column=-1;
  if (view.quantum == MH_QUANTUM_U16)
    {
      std::vector<unsigned short> q(n);
      for (size_t i=0; i < n; i++)
        {
          // ClampToQuantum of a value that already is a Quantum
          double v=lut[i];
          q[i]=(unsigned short) (!(v > 0.0) ? 0 : (v >= kQuantumRange ? 65535 : (unsigned short) (v+0.5)));
        }
      // do all selected channels share one column?
      const int C=view.channels;
      int first=-1;
      bool same=true;
      for (int c=0; (c < C) && same; c++)
        {
          if (((apply_mask >> c) & 1u) == 0)
            continue;
          if (first < 0)
            {
              first=c;
              continue;
            }
          for (size_t b=0; b < (size_t) MH_HISTOGRAM_BINS; b++)
            if (q[b*(size_t) C+(size_t) c] != q[b*(size_t) C+(size_t) first])
              {
                same=false;
                break;
              }
        }
      if (same && (first >= 0))
        shared_column=first;
      MH_TRY(upload_table(d_lut,view.device,view.stream,q.data(),n*sizeof(unsigned short)));
    }
  else
    {
      std::vector<float> q(n);
      for (size_t i=0; i < n; i++)
        q[i]=(float) lut[i];
      MH_TRY(upload_table(d_lut,view.device,view.stream,q.data(),n*sizeof(float)));
    }
  Roles roles=channel_roles(desc,desc);
  // enhance.c:1781, :2255: only channels whose traits carry Update
  uint32_t update=0;
  for (uint32_t c=0; c < desc->number_channels; c++)
    if ((desc->channel_traits[c] & MH_TRAIT_UPDATE) != 0)
      update|=1u<<c;
  roles.update_mask=update;
  return launch_apply_lut(view,d_lut.ptr,apply_mask,roles,shared_column);
}

// histogram [65536][channels] on the device -> LUT -> apply, all on the stream (the second
// half of ContrastStretchImage / EqualizeImage; MagickHipShardedImage calls it with a table
// that was all-reduced over the row bands of one image)
MhStatus apply_histogram_lut(const View &view,const MhImage *image,const unsigned long long *hist,
  int mode,bool equalize,double black_point,double white_limit,const unsigned int *colour_flag)
{
  if (!equalize && (view.channels <= 4) && (option("MAGICKHIP_NO_STRETCH_LEVELS") == nullptr))
    {
      // ContrastStretchImage: the levels of every channel, then the map evaluated per sample — no
      // table to build and gather from (pointwise.hip)
      uint32_t update=0;
      for (uint32_t c=0; c < image->number_channels; c++)
        if ((image->channel_traits[c] & MH_TRAIT_UPDATE) != 0)
          update|=1u<<c;
      return launch_stretch_levels_apply(view,hist,black_point,white_limit,update,colour_flag);
    }
  const size_t n=(size_t) MH_HISTOGRAM_BINS*(size_t) view.channels;
  Temp lut,mask;
  MH_TRY(lut.alloc(view.device,n*(view.quantum == MH_QUANTUM_U16 ? sizeof(unsigned short) :
    sizeof(float)),view.stream));
  MH_TRY(mask.alloc(view.device,sizeof(uint32_t),view.stream));
  Roles roles=channel_roles(image,image);
  // enhance.c:1781, :2255: only channels whose traits carry Update
  uint32_t update=0;
  for (uint32_t c=0; c < image->number_channels; c++)
    if ((image->channel_traits[c] & MH_TRAIT_UPDATE) != 0)
      update|=1u<<c;
  roles.update_mask=update;
  // intensity binning gives every channel the same histogram, hence the same LUT column
  int shared_column=-1;
  if (((mode != 0) || (view.channels == 1)) && (update != 0))
    shared_column=__builtin_ctz(update);
  // EqualizeImage on a float frame: the map evaluated per sample from the running counts in LDS
  // instead of gathered from a 65536-float table (pointwise.hip)
  if (equalize && (view.quantum != MH_QUANTUM_U16) && (shared_column >= 0) && (view.channels <= 4) &&
      (view.columns*view.rows >= ((size_t) 1 << 20)) && (option("MAGICKHIP_NO_EQUALIZE_COUNTS") == nullptr))
    {
      Temp counts;
      MH_TRY(counts.alloc(view.device,(size_t) MH_HISTOGRAM_BINS*sizeof(uint32_t),view.stream));
      MH_TRY(launch_build_lut(view,hist,equalize,black_point,white_limit,
        lut.ptr,mask.as<uint32_t>(),colour_flag,counts.as<uint32_t>(),shared_column));
      return launch_equalize_cdf_apply(view,counts.as<uint32_t>(),lut.ptr,~0u,roles,shared_column,mask.as<uint32_t>());
    }
  MH_TRY(launch_build_lut(view,hist,equalize,black_point,white_limit,
    lut.ptr,mask.as<uint32_t>(),colour_flag));
  return launch_apply_lut(view,lut.ptr,~0u,roles,shared_column,mask.as<uint32_t>());
}

} // namespace mh

using namespace mh;

extern "C" {

MH_API MhStatus MhContrastStretchLUT(const uint64_t *histogram,uint32_t number_channels,
  size_t columns,size_t rows,double black_point,double white_point,MhQuantumKind quantum,
  double *lut,uint32_t *apply_mask)
{
  if ((histogram == nullptr) || (lut == nullptr) || (number_channels == 0) ||
      (number_channels > MH_MAX_CHANNELS))
    return fail(MH_BAD_ARGUMENT,"ContrastStretchLUT: bad arguments");
  const size_t C=number_channels;
  uint32_t mask=0;
  memset(lut,0,(size_t) MH_HISTOGRAM_BINS*C*sizeof(double));
  for (size_t i=0; i < C; i++)
    {
      // locate the black/white levels, enhance.c:1652-1678.  black[] / white[]
      // are Quantum-typed in the reference; the indices fit either kind exactly.
      double intensity=0.0;
      ptrdiff_t j;
      for (j=0; j <= (ptrdiff_t) MH_MAXMAP; j++)
        {
          intensity+=(double) histogram[C*(size_t) j+i];
          if (intensity > black_point)
            break;
        }
      // black[i]=(Quantum) j, enhance.c:1668: when no bin exceeds black_point the scan ends with
      // j = MaxMap+1, which the Q16 build stores as (unsigned short) 65536 = 0
      const double black=quantum == MH_QUANTUM_U16 ? (double) (j & 0xffff) : (double) j;
      intensity=0.0;
      for (j=(ptrdiff_t) MH_MAXMAP; j != 0; j--)
        {
          intensity+=(double) histogram[C*(size_t) j+i];
          if (intensity > ((double) columns*(double) rows-white_point))
            break;
        }
      const double white=(double) j;
      // stretch map, enhance.c:1685-1706
      for (j=0; j <= (ptrdiff_t) MH_MAXMAP; j++)
        {
          double gamma=perceptible_reciprocal(white-black);
          if (j < (ptrdiff_t) black)
            lut[C*(size_t) j+i]=0.0;
          else if (j > (ptrdiff_t) white)
            lut[C*(size_t) j+i]=kQuantumRange;
          else if (black != white)
            lut[C*(size_t) j+i]=scale_map_to_quantum(
              (double) ((double) MH_MAXMAP*gamma*((double) j-black)),quantum);
        }
      if (black != white)
        mask|=1u<<i;
    }
  if (apply_mask != nullptr)
    *apply_mask=mask;
  return MH_OK;
}

MH_API MhStatus MhEqualizeLUT(const uint64_t *histogram,uint32_t number_channels,
  MhQuantumKind quantum,double *lut,uint32_t *apply_mask)
{
  if ((histogram == nullptr) || (lut == nullptr) || (number_channels == 0) ||
      (number_channels > MH_MAX_CHANNELS))
    return fail(MH_BAD_ARGUMENT,"EqualizeLUT: bad arguments");
  const size_t C=number_channels;
  uint32_t mask=0;
  std::vector<double> map((size_t) MH_HISTOGRAM_BINS);
  memset(lut,0,(size_t) MH_HISTOGRAM_BINS*C*sizeof(double));
  for (size_t i=0; i < C; i++)
    {
      // integrate, enhance.c:2138-2152
      double intensity=0.0;
      for (size_t j=0; j <= MH_MAXMAP; j++)
        {
          intensity+=(double) histogram[C*j+i];
          map[j]=intensity;
        }
      const double black=map[0],white=map[MH_MAXMAP];
      if (black != white)
        {
          // enhance.c:2162-2168
          for (size_t j=0; j <= MH_MAXMAP; j++)
            lut[C*j+i]=scale_map_to_quantum(
              (double) (((double) MH_MAXMAP*(map[j]-black))/(white-black)),quantum);
          mask|=1u<<i;
        }
    }
  if (apply_mask != nullptr)
    *apply_mask=mask;
  return MH_OK;
}

MH_API MhStatus MagickHipHistogram(const MhImage *image,int intensity_mode,uint64_t *histogram)
{
  MH_TRY(check_image(image,"Histogram"));
  if (histogram == nullptr)
    return fail(MH_BAD_ARGUMENT,"Histogram: null histogram");
  const size_t n=(size_t) MH_HISTOGRAM_BINS*(size_t) image->number_channels;
  DeviceGuard guard;
  if (image->memory == MH_MEMORY_DEVICE)
    {
      Resident img;
      MH_TRY(img.open(image,0,nullptr,-1));
      MH_HIP(guard.enter(img.view.device));
      return launch_histogram(img.view,intensity_mode,image,
        reinterpret_cast<unsigned long long *>(histogram));
    }
  Resident img;
  int device=resolve_device(image);
  MH_TRY(img.open(image,0,library_stream(device),device));
  MH_HIP(guard.enter(device));
  std::vector<unsigned long long> host;
  MH_TRY(histogram_to_host(img.view,intensity_mode,image,host));
  for (size_t i=0; i < n; i++)
    histogram[i]+=host[i];
  return img.commit();
}

MH_API MhStatus MagickHipApplyLUT(MhImage *image,const double *lut,uint32_t apply_mask)
{
  MH_TRY(check_image(image,"ApplyLUT"));
  if (lut == nullptr)
    return fail(MH_BAD_ARGUMENT,"ApplyLUT: null LUT");
  InPlace io;
  MH_TRY(io.open(image));
  MH_TRY(apply_lut_host(io.img.view,image,lut,apply_mask));
  return io.img.commit();
}

MH_API MhStatus MagickHipIsImageGray(const MhImage *image,int *is_gray)
{
  MH_TRY(check_image(image,"IsImageGray"));
  if (is_gray == nullptr)
    return fail(MH_BAD_ARGUMENT,"IsImageGray: null result");
  const uint32_t colour=image->number_channels-(image->alpha_offset >= 0 ? 1u : 0u);
  if (colour < 3)
    {
      *is_gray=1;
      return MH_OK;
    }
  DeviceGuard guard;
  Resident img;
  int device=resolve_device(image);
  MH_TRY(img.open(image,0,image->memory == MH_MEMORY_DEVICE ? (hipStream_t) image->stream :
    library_stream(device),device));
  MH_HIP(guard.enter(img.view.device));
  Temp flag;
  MH_TRY(flag.alloc(img.view.device,sizeof(unsigned int),img.view.stream));
  MH_HIP(hipMemsetAsync(flag.ptr,0,sizeof(unsigned int),img.view.stream));
  MH_TRY(launch_gray_check(img.view,image,flag.as<unsigned int>()));
  unsigned int host=0;
  MH_HIP(hipMemcpyAsync(&host,flag.ptr,sizeof(host),hipMemcpyDeviceToHost,img.view.stream));
  MH_HIP(hipStreamSynchronize(img.view.stream));
  *is_gray=host == 0 ? 1 : 0;
  return MH_OK;
}

// histogram -> LUT -> apply, all on the stream, no host round trip.  `colour_flag` (optional)
// is the gray-scan word: when it says "gray" the LUT builder leaves the mask at zero and the
// apply kernel touches nothing.
static MhStatus histogram_lut_apply(const View &view,const MhImage *image,int mode,bool equalize,
  double black_point,double white_limit,const unsigned int *colour_flag)
{
  const size_t n=(size_t) MH_HISTOGRAM_BINS*(size_t) view.channels;
  Temp hist;
  MH_TRY(hist.alloc(view.device,n*sizeof(unsigned long long),view.stream));
  MH_HIP(hipMemsetAsync(hist.ptr,0,n*sizeof(unsigned long long),view.stream));
  MH_TRY(launch_histogram(view,mode,image,hist.as<unsigned long long>()));
  return apply_histogram_lut(view,image,hist.as<unsigned long long>(),mode,equalize,black_point,
    white_limit,colour_flag);
}

extern "C++" {
namespace mh {
MhStatus equalize_view(const View &view,const MhImage *image)
{
  const int mode=(image->channel_mask & MH_SYNC_CHANNELS) != 0 ? 1 : 0;   // enhance.c:2125-2129
  return histogram_lut_apply(view,image,mode,true,0.0,0.0,nullptr);
}
}
}

// ContrastStretchImage, enhance.c:1544-1818
MH_API MhStatus MagickHipContrastStretchImage(MhImage *image,double black_point,
  double white_point,int *became_gray)
{
  MH_TRY(check_image(image,"ContrastStretchImage"));
  if (became_gray != nullptr)
    *became_gray=0;
  InPlace io;
  MH_TRY(io.open(image));
  const View &view=io.img.view;
  // IdentifyImageType side effect, enhance.c:1586-1588: a colour image whose
  // pixels are all gray is re-laid-out as a GRAY image by the reference; that
  // is the caller's job (SetImageColorspace), so hand such images back untouched.
  // The scan's verdict gates the LUT on the device; the host reads it once, at the end.
  const uint32_t colour=image->number_channels-(image->alpha_offset >= 0 ? 1u : 0u);
  const bool scan=(colour >= 3) && ((image->colorspace == MH_COLORSPACE_SRGB) ||
    (image->colorspace == MH_COLORSPACE_RGB));
  Temp flag;
  if (scan)
    {
      MH_TRY(flag.alloc(view.device,sizeof(unsigned int),view.stream));
      MH_HIP(hipMemsetAsync(flag.ptr,0,sizeof(unsigned int),view.stream));
      MH_TRY(launch_gray_check(view,image,flag.as<unsigned int>()));
    }
  // enhance.c:1637-1643: every channel bins the intensity under the default mask
  const int mode=image->channel_mask == MH_ALL_CHANNELS ? 1 : 0;
  MH_TRY(histogram_lut_apply(view,image,mode,false,black_point,
    (double) image->columns*(double) image->rows-white_point,scan ? flag.as<unsigned int>() : nullptr));
  if (scan)
    {
      unsigned int host=0;
      MH_HIP(hipMemcpyAsync(&host,flag.ptr,sizeof(host),hipMemcpyDeviceToHost,view.stream));
      MH_HIP(hipStreamSynchronize(view.stream));
      if (host == 0)
        {
          if (became_gray != nullptr)
            *became_gray=1;
          return fail(MH_UNSUPPORTED,"ContrastStretchImage: image is gray; convert it to the "
            "GRAY colourspace first (IdentifyImageType, enhance.c:1586)");
        }
    }
  return io.img.commit();
}

// The second half of EqualizeImage / ContrastStretchImage for a caller that holds the histogram
// (the all-reduced table of a row-sharded image): LUT construction and application on the
// device, nothing comes back to the host.
MH_API MhStatus MagickHipApplyHistogram(MhImage *image,const uint64_t *histogram,int intensity_mode,
  int equalize,double black_point,double white_point,size_t image_rows)
{
  MH_TRY(check_image(image,"ApplyHistogram"));
  if (histogram == nullptr)
    return fail(MH_BAD_ARGUMENT,"ApplyHistogram: null histogram");
  InPlace io;
  MH_TRY(io.open(image));
  const View &view=io.img.view;
  const size_t n=(size_t) MH_HISTOGRAM_BINS*(size_t) view.channels;
  Temp table;
  const unsigned long long *device_table=reinterpret_cast<const unsigned long long *>(histogram);
  if (image->memory == MH_MEMORY_HOST)
    {
      MH_TRY(upload_table(table,view.device,view.stream,histogram,n*sizeof(unsigned long long)));
      device_table=table.as<unsigned long long>();
    }
  const double rows=(double) (image_rows == 0 ? image->rows : image_rows);
  MH_TRY(apply_histogram_lut(view,image,device_table,intensity_mode != 0 ? 1 : 0,equalize != 0,black_point,
    (double) image->columns*rows-white_point,nullptr));
  return io.img.commit();
}

// EqualizeImage, enhance.c:2040-2280
MH_API MhStatus MagickHipEqualizeImage(MhImage *image)
{
  MH_TRY(check_image(image,"EqualizeImage"));
  InPlace io;
  MH_TRY(io.open(image));
  const View &view=io.img.view;
  MH_TRY(equalize_view(view,image));
  return io.img.commit();
}

// TransformImageColorspace, colorspace.c:1751-1783
MH_API MhStatus MagickHipTransformImageColorspace(MhImage *image,MhColorspace colorspace)
{
  MH_TRY(check_image(image,"TransformImageColorspace"));
  const MhColorspace from=(MhColorspace) image->colorspace;
  if (from == colorspace)
    return MH_OK;
  if ((colorspace == MH_COLORSPACE_GRAY) || (colorspace == MH_COLORSPACE_LINEARGRAY))
    {
      // sRGB -> GRAY / LinearGRAY (colorspace.c:843-957): gray = 0.212656 R + 0.715158 G +
      // 0.072186 B of the samples (GRAY) or of their gamma-decoded values (LinearGRAY), written
      // into the gray (= first) channel — the expression, operation by operation, of
      // GetPixelIntensity's Rec709Luma / Rec709Luminance on an sRGB image (pixel.c:2219-2235),
      // i.e. GrayscaleImage's kernel.  The channel LAYOUT stays: re-laying the pixels out as one
      // gray channel is SetImageColorspace's part (the caller's, as after GrayscaleImage,
      // enhance.c:2648-2652).
      const uint32_t colours=image->number_channels-(image->alpha_offset >= 0 ? 1u : 0u);
      if ((from != MH_COLORSPACE_SRGB) || (colours != 3))
        return fail(MH_UNSUPPORTED,"colourspace %d -> gray: only from sRGB with three colour channels",(int) from);
      InPlace io;
      MH_TRY(io.open(image));
      MH_TRY(launch_grayscale(io.img.view,colorspace == MH_COLORSPACE_GRAY ? (int) MH_INTENSITY_REC709LUMA :
        (int) MH_INTENSITY_REC709LUMINANCE,image));
      MH_TRY(io.img.commit());
      image->colorspace=(uint32_t) colorspace;
      return MH_OK;
    }
  if (!colorspace_is_accelerated(from) || !colorspace_is_accelerated(colorspace))
    return fail(MH_UNSUPPORTED,"colourspace %d -> %d is not accelerated",(int) from,(int) colorspace);
  const uint32_t colour=image->number_channels-(image->alpha_offset >= 0 ? 1u : 0u);
  if (colour != 3)
    return fail(MH_UNSUPPORTED,"colourspace transform needs three colour channels");
  InPlace io;
  MH_TRY(io.open(image));
  MH_TRY(launch_colorspace(io.img.view,from,colorspace,image));
  MH_TRY(io.img.commit());
  image->colorspace=(uint32_t) colorspace;
  return MH_OK;
}

// TransformImageColorspace followed by ContrastStretchImage, as one call.  The results are those
// of the two calls; what the pair can share is the pass over the pixels: FAST sRGB -> Lab on
// RGBA Q16 converts and bins the intensity of what it stores in the same kernel (pointwise.hip,
// lab_histogram_fast_kernel).  MagickHipBatchImages uses it for adjacent operators of a chain.
MH_API MhStatus MagickHipTransformColorspaceContrastStretchImage(MhImage *image,MhColorspace colorspace,
  double black_point,double white_point)
{
  MH_TRY(check_image(image,"TransformColorspaceContrastStretchImage"));
  const uint32_t colour=image->number_channels-(image->alpha_offset >= 0 ? 1u : 0u);
  if (((MhColorspace) image->colorspace == MH_COLORSPACE_SRGB) && (colorspace == MH_COLORSPACE_LAB) &&
      (colour == 3) && (image->channel_mask == MH_ALL_CHANNELS) && (image->memory == MH_MEMORY_DEVICE))
    {
      InPlace io;
      MH_TRY(io.open(image));
      const View &view=io.img.view;
      MhImage lab=*image;
      lab.colorspace=(uint32_t) MH_COLORSPACE_LAB;
      bool fused=false;
      {
        // three launches: convert + bin, levels, map (pointwise.hip)
        uint32_t update=0;
        for (uint32_t c=0; c < image->number_channels; c++)
          if ((image->channel_traits[c] & MH_TRAIT_UPDATE) != 0)
            update|=1u<<c;
        MH_TRY(launch_lab_fast_contrast_stretch(view,&lab,black_point,
          (double) image->columns*(double) image->rows-white_point,update,&fused));
        if (fused)
          {
            image->colorspace=(uint32_t) MH_COLORSPACE_LAB;
            return io.img.commit();
          }
      }
      const size_t n=(size_t) MH_HISTOGRAM_BINS*(size_t) view.channels;
      Temp hist;
      MH_TRY(hist.alloc(view.device,n*sizeof(unsigned long long),view.stream));
      MH_HIP(hipMemsetAsync(hist.ptr,0,n*sizeof(unsigned long long),view.stream));
      MH_TRY(launch_lab_fast_with_histogram(view,&lab,hist.as<unsigned long long>(),&fused));
      if (fused)
        {
          image->colorspace=(uint32_t) MH_COLORSPACE_LAB;
          MH_TRY(apply_histogram_lut(view,image,hist.as<unsigned long long>(),1,false,black_point,
            (double) image->columns*(double) image->rows-white_point,nullptr));
          return io.img.commit();
        }
    }
  MH_TRY(MagickHipTransformImageColorspace(image,colorspace));
  return MagickHipContrastStretchImage(image,black_point,white_point,nullptr);
}

// ContrastImage, enhance.c:1392-1480
MH_API MhStatus MagickHipContrastImage(MhImage *image,int sharpen)
{
  MH_TRY(check_image(image,"ContrastImage"));
  InPlace io;
  MH_TRY(io.open(image));
  MH_TRY(launch_contrast(io.img.view,sharpen != 0));
  return io.img.commit();
}

// ModulateImage, enhance.c:3665-3900 (HSL and HSB models)
MH_API MhStatus MagickHipModulateImage(MhImage *image,double percent_brightness,
  double percent_saturation,double percent_hue,int colorspace)
{
  MH_TRY(check_image(image,"ModulateImage"));
  bool generic=false;
  switch (colorspace)
  {
    case MH_COLORSPACE_UNDEFINED: case MH_COLORSPACE_HSL: case MH_COLORSPACE_HSB:
      break;
    case MH_COLORSPACE_HCL: case MH_COLORSPACE_HCLP: case MH_COLORSPACE_HSI: case MH_COLORSPACE_HSV:
    case MH_COLORSPACE_HWB: case MH_COLORSPACE_LCH: case MH_COLORSPACE_LCHAB: case MH_COLORSPACE_LCHUV:
      generic=true;                               // enhance.c:3826-3890
      break;
    default:
      // (any other value falls into ModulateHSL's `default:` in the reference; the shim maps
      // those to HSL before calling)
      return fail(MH_UNSUPPORTED,"ModulateImage: colour model %d is not accelerated",colorspace);
  }
  InPlace io;
  MH_TRY(io.open(image));
  // the loop invariants of the Modulate* helpers, enhance.c:3462-3632
  const double hue_shift=fmod((percent_hue-100.0),200.0)/200.0;
  const double saturation_scale=0.01*percent_saturation;
  const double brightness_scale=0.01*percent_brightness;
  if (generic)
    MH_TRY(launch_modulate_generic(io.img.view,(MhColorspace) colorspace,hue_shift,saturation_scale,
      brightness_scale));
  else
    MH_TRY(launch_modulate(io.img.view,colorspace == MH_COLORSPACE_HSB,hue_shift,saturation_scale,
      brightness_scale));
  return io.img.commit();
}

// ImportImagePixels / ExportImagePixels, pixel.c:4164 / :1962
static MhStatus pixel_io(bool import,const MhImage *image,ptrdiff_t x,ptrdiff_t y,size_t width,
  size_t height,const char *map,MhStorageType type,void *pixels,MhMemoryKind pixels_memory)
{
  const char *what=import ? "ImportImagePixels" : "ExportImagePixels";
  MH_TRY(check_image(image,what));
  if ((map == nullptr) || (pixels == nullptr))
    return fail(MH_BAD_ARGUMENT,"%s: null map or pixels",what);
  if ((width == 0) || (height == 0) || (x < 0) || (y < 0) ||
      ((size_t) x+width > image->columns) || ((size_t) y+height > image->rows))
    return fail(MH_UNSUPPORTED,"%s: the region must lie inside the image",what);
  const size_t element=storage_size(type,(MhQuantumKind) image->quantum);
  if (element == 0)
    return fail(MH_BAD_ARGUMENT,"%s: unknown storage type %d",what,(int) type);
  const size_t bytes=width*height*strlen(map)*element;
  DeviceGuard guard;
  Resident img;
  int device=resolve_device(image);
  hipStream_t stream=image->memory == MH_MEMORY_DEVICE ? (hipStream_t) image->stream :
    library_stream(device);
  // import modifies the image (upload + download unless the map covers it: keep it simple,
  // mode 2); export only reads it
  MH_TRY(img.open(image,import ? 2 : 0,stream,device));
  img.view.stream=stream;
  img.view.device=device;
  MH_HIP(guard.enter(device));
  Temp staged;
  void *buffer=pixels;
  if (pixels_memory == MH_MEMORY_HOST)
    {
      MH_TRY(staged.alloc(device,bytes,stream));
      buffer=staged.ptr;
      if (import)
        MH_HIP(hipMemcpyAsync(buffer,pixels,bytes,hipMemcpyHostToDevice,stream));
      else
        {
          // pads the reference leaves untouched must survive the round trip
          bool has_pad=false;
          for (const char *m=map; *m != 0; m++)
            has_pad|=(*m == 'P') || (*m == 'p');
          if (has_pad)
            MH_HIP(hipMemcpyAsync(buffer,pixels,bytes,hipMemcpyHostToDevice,stream));
        }
    }
  MH_TRY(launch_pixel_io(import,img.view,image,(int) x,(int) y,(int) width,(int) height,map,type,
    buffer));
  if (!import && (pixels_memory == MH_MEMORY_HOST))
    {
      MH_HIP(hipMemcpyAsync(pixels,buffer,bytes,hipMemcpyDeviceToHost,stream));
      MH_HIP(hipStreamSynchronize(stream));
    }
  return img.commit();
}

MH_API MhStatus MagickHipImportImagePixels(MhImage *image,ptrdiff_t x,ptrdiff_t y,size_t width,
  size_t height,const char *map,MhStorageType type,const void *pixels,MhMemoryKind pixels_memory)
{
  return pixel_io(true,image,x,y,width,height,map,type,const_cast<void *>(pixels),pixels_memory);
}

MH_API MhStatus MagickHipExportImagePixels(const MhImage *image,ptrdiff_t x,ptrdiff_t y,
  size_t width,size_t height,const char *map,MhStorageType type,void *pixels,
  MhMemoryKind pixels_memory)
{
  return pixel_io(false,image,x,y,width,height,map,type,pixels,pixels_memory);
}

// GrayscaleImage, enhance.c:2476-2660
MH_API MhStatus MagickHipGrayscaleImage(MhImage *image,MhIntensityMethod method)
{
  MH_TRY(check_image(image,"GrayscaleImage"));
  InPlace io;
  MH_TRY(io.open(image));
  MH_TRY(launch_grayscale(io.img.view,(int) method,image));
  return io.img.commit();
}

// FunctionImage, statistic.c:1069-1160
MH_API MhStatus MagickHipFunctionImage(MhImage *image,MhFunction function,
  size_t number_parameters,const double *parameters)
{
  MH_TRY(check_image(image,"FunctionImage"));
  if ((number_parameters != 0) && (parameters == nullptr))
    return fail(MH_BAD_ARGUMENT,"FunctionImage: null parameters");
  if ((function < MH_FUNCTION_ARCSIN) || (function > MH_FUNCTION_SINUSOID))
    return fail(MH_BAD_ARGUMENT,"FunctionImage: unknown function %d",(int) function);
  uint32_t update=0;
  for (uint32_t c=0; c < image->number_channels; c++)
    if ((image->channel_traits[c] & MH_TRAIT_UPDATE) != 0)
      update|=1u<<c;
  InPlace io;
  MH_TRY(io.open(image));
  MH_TRY(launch_function(io.img.view,(int) function,number_parameters,parameters,update));
  return io.img.commit();
}

} // extern "C"
