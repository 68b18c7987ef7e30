// ImportImagePixels / ExportImagePixels (MagickCore/pixel.c:4164 / :1962) on the device:
// conversion between a caller's interleaved component buffer (8/16/32/64-bit integers,
// float, double or Quantum; any order of R,G,B,A,O,I,P components) and the Quantum pixel
// cache ("decoded scanlines go straight to HBM").
//
// The reference's per-type loops (ImportCharPixel ... ExportShortPixel, pixel.c:492-2200,
// :2509-4160) all walk the map one component at a time; their named fast paths ("RGBA",
// "BGRA", ...) are the same arithmetic unrolled, except that export writes 0 for the pad of
// "RGBP"/"BGRP" and leaves a pad untouched otherwise.  Conversions: quantum-private.h:435-535
// (QuantumDepth 16, plain and HDRI), quantum.h:113-124 (ScaleQuantumToChar), quantum.h:86-97
// (ClampToQuantum).
// (Textually included at the end of pointwise.hip, inside namespace mh: it uses that file's
// GetPixelIntensity restatement.)

enum { kMaxComponents=8 };
// what a component of the caller's buffer is
enum Component : int8_t { COMP_SKIP=-1,COMP_RED=0,COMP_GREEN=1,COMP_BLUE=2,COMP_ALPHA=3,COMP_INTENSITY=4 };

struct PixelIoArgs
{
  void *image;                 // Quantum pixels
  void *buffer;                // caller's components, region-sized, tightly packed
  int columns,rows,channels;   // image
  int x,y,width,height;        // region
  int ncomp;
  int8_t comp[kMaxComponents];           // Component per buffer slot
  int8_t channel[kMaxComponents];        // image channel per buffer slot (-1: none)
  int pad_writes_zero;         // export: "RGBP" / "BGRP"
  IntensityParams intensity;   // export 'I'
  int alpha_channel;           // image channel of alpha or -1 (export reads OpaqueAlpha then)
};

// ---- storage <-> Quantum, one struct per StorageType
struct StoreChar
{
  typedef unsigned char T;
  static __device__ __forceinline__ uint16_t to_q16(T v) { return (uint16_t) (257u*v); }
  static __device__ __forceinline__ float to_f32(T v) { return (float) (257.0*v); }
  static __device__ __forceinline__ T from_q16(uint16_t q)
  { unsigned long v=(unsigned long) q+128ul; return (T) ((v-(v >> 8)) >> 8); }
  static __device__ __forceinline__ T from_f32(float q)
  {
    if (!(q > 0.0f)) return 0;                       // NaN or <= 0
    if ((q/257.0f) >= 255.0f) return 255;
    return (T) (q/257.0f+0.5f);
  }
};
struct StoreShort
{
  typedef unsigned short T;
  static __device__ __forceinline__ uint16_t to_q16(T v) { return v; }
  static __device__ __forceinline__ float to_f32(T v) { return (float) v; }
  static __device__ __forceinline__ T from_q16(uint16_t q) { return q; }
  static __device__ __forceinline__ T from_f32(float q)
  {
    if (!(q > 0.0f)) return 0;
    if (q >= 65535.0f) return 65535;
    return (T) (q+0.5f);
  }
};
struct StoreLong
{
  typedef unsigned int T;
  static __device__ __forceinline__ uint16_t to_q16(T v) { return (uint16_t) (v/65537ull); }
  static __device__ __forceinline__ float to_f32(T v) { return (float) (v/65537.0); }
  static __device__ __forceinline__ T from_q16(uint16_t q) { return (T) (65537ul*q); }
  static __device__ __forceinline__ T from_f32(float q)
  {
    if (!(q > 0.0f)) return 0u;
    if ((65537.0*(double) q) >= 4294967295.0) return 4294967295u;
    return (T) (65537.0*(double) q+0.5);
  }
};
struct StoreLongLong
{
  typedef unsigned long long T;
  static __device__ __forceinline__ uint16_t to_q16(T v) { return (uint16_t) (v/281479271743489ull); }
  static __device__ __forceinline__ float to_f32(T v) { return (float) (v/281479271743489.0); }
  static __device__ __forceinline__ T from_q16(uint16_t q) { return (T) (281479271743489ull*q); }
  static __device__ __forceinline__ T from_f32(float q)
  {
    if (!(q > 0.0f)) return 0ull;
    if ((281479271743489.0*(double) q) >= 18446744073709551615.0) return 18446744073709551615ull;
    return (T) (281479271743489.0*(double) q+0.5);
  }
};
struct StoreFloat
{
  typedef float T;
  static __device__ __forceinline__ uint16_t to_q16(T v) { return QuantumOps<uint16_t>::clamp(kQR*(double) v); }
  static __device__ __forceinline__ float to_f32(T v) { return QuantumOps<float>::clamp(kQR*(double) v); }
  static __device__ __forceinline__ T from_real(double quantum_value) { return (float) (kQS*quantum_value); }
};
struct StoreDouble
{
  typedef double T;
  static __device__ __forceinline__ uint16_t to_q16(T v) { return QuantumOps<uint16_t>::clamp(kQR*v); }
  static __device__ __forceinline__ float to_f32(T v) { return QuantumOps<float>::clamp(kQR*v); }
  static __device__ __forceinline__ T from_real(double quantum_value) { return kQS*quantum_value; }
};

template<typename S,typename Q> static __device__ __forceinline__ Q import_value(typename S::T v)
{
  if constexpr (sizeof(Q) == 2)
    return S::to_q16(v);
  else
    return S::to_f32(v);
}

// one thread per region pixel: the component buffer is contiguous per pixel, so a wave reads
// one contiguous run per component slot
template<typename S,typename Q,bool QUANTUM_STORAGE>
__global__ __launch_bounds__(256)
void import_kernel(PixelIoArgs a)
{
  const size_t n=(size_t) a.width*a.height;
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < n; i+=stride)
    {
      const int ry=(int) (i/(size_t) a.width),rx=(int) (i-(size_t) ry*a.width);
      Q *pixel=static_cast<Q *>(a.image)+((size_t) (a.y+ry)*a.columns+(a.x+rx))*a.channels;
      for (int k=0; k < a.ncomp; k++)
        {
          const int c=a.channel[k];
          if (c < 0)
            continue;
          if constexpr (QUANTUM_STORAGE)
            pixel[c]=static_cast<const Q *>(a.buffer)[i*a.ncomp+k];
          else
            pixel[c]=import_value<S,Q>(static_cast<const typename S::T *>(a.buffer)[i*a.ncomp+k]);
        }
    }
}

template<typename S,typename Q,bool QUANTUM_STORAGE,bool REAL>
__global__ __launch_bounds__(256)
void export_kernel(PixelIoArgs a)
{
  const size_t n=(size_t) a.width*a.height;
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < n; i+=stride)
    {
      const int ry=(int) (i/(size_t) a.width),rx=(int) (i-(size_t) ry*a.width);
      const Q *pixel=static_cast<const Q *>(a.image)+((size_t) (a.y+ry)*a.columns+(a.x+rx))*a.channels;
      double intensity=0.0;
      bool have_intensity=false;
      for (int k=0; k < a.ncomp; k++)
        {
          const int what=a.comp[k];
          double real=0.0;             // the Quantum value as a double (float/double storage)
          Q quantum=(Q) 0;
          if (what == COMP_SKIP)
            {
              if (a.pad_writes_zero == 0)
                continue;
            }
          else if (what == COMP_INTENSITY)
            {
              if (!have_intensity)
                {
                  // GetPixelIntensity, pixel.c:2356-2455, on the image's own channels
                  switch (a.channels)
                  {
                    case 1: { Q q1[1]={pixel[0]}; intensity=pixel_intensity<Q,1>(q1,a.intensity); break; }
                    case 2: { Q q2[2]={pixel[0],pixel[1]}; intensity=pixel_intensity<Q,2>(q2,a.intensity); break; }
                    case 3: { Q q3[3]={pixel[0],pixel[1],pixel[2]}; intensity=pixel_intensity<Q,3>(q3,a.intensity); break; }
                    default: { Q q4[4]={pixel[0],pixel[1],pixel[2],pixel[3]}; intensity=pixel_intensity<Q,4>(q4,a.intensity); break; }
                  }
                  have_intensity=true;
                }
              real=intensity;
              quantum=QuantumOps<Q>::clamp(intensity);         // ClampToQuantum(GetPixelIntensity)
            }
          else
            {
              const int c=a.channel[k];
              // GetPixelAlpha of an image without alpha is OpaqueAlpha
              quantum=c >= 0 ? pixel[c] : (Q) 65535;
              real=(double) quantum;
            }
          if constexpr (QUANTUM_STORAGE)
            static_cast<Q *>(a.buffer)[i*a.ncomp+k]=what == COMP_SKIP ? (Q) 0 : quantum;
          else if constexpr (REAL)
            static_cast<typename S::T *>(a.buffer)[i*a.ncomp+k]=
              what == COMP_SKIP ? (typename S::T) 0 : S::from_real(real);
          else
            {
              typename S::T v;
              if (what == COMP_SKIP)
                v=0;
              else if constexpr (sizeof(Q) == 2)
                v=S::from_q16(quantum);
              else
                v=S::from_f32(quantum);
              static_cast<typename S::T *>(a.buffer)[i*a.ncomp+k]=v;
            }
        }
    }
}

static unsigned io_grid(size_t n)
{
  size_t blocks=(n+255)/256;
  return (unsigned) (blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks));
}

template<typename Q>
static MhStatus launch_io_typed(bool import,MhStorageType type,const PixelIoArgs &a,hipStream_t stream)
{
  const dim3 grid(io_grid((size_t) a.width*a.height)),block(256);
#define MH_IO(S,REAL) \
  if (import) hipLaunchKernelGGL((import_kernel<S,Q,false>),grid,block,0,stream,a); \
  else hipLaunchKernelGGL((export_kernel<S,Q,false,REAL>),grid,block,0,stream,a); \
  break;
  switch (type)
  {
    case MH_STORAGE_CHAR: MH_IO(StoreChar,false)
    case MH_STORAGE_SHORT: MH_IO(StoreShort,false)
    case MH_STORAGE_LONG: MH_IO(StoreLong,false)
    case MH_STORAGE_LONGLONG: MH_IO(StoreLongLong,false)
    case MH_STORAGE_FLOAT: MH_IO(StoreFloat,true)
    case MH_STORAGE_DOUBLE: MH_IO(StoreDouble,true)
    case MH_STORAGE_QUANTUM:
      if (import) hipLaunchKernelGGL((import_kernel<StoreShort,Q,true>),grid,block,0,stream,a);
      else hipLaunchKernelGGL((export_kernel<StoreShort,Q,true,false>),grid,block,0,stream,a);
      break;
    default:
      return fail(MH_BAD_ARGUMENT,"unknown storage type %d",(int) type);
  }
#undef MH_IO
  MH_HIP(hipGetLastError());
  return MH_OK;
}

size_t storage_size(MhStorageType type,MhQuantumKind quantum)
{
  switch (type)
  {
    case MH_STORAGE_CHAR: return 1;
    case MH_STORAGE_SHORT: return 2;
    case MH_STORAGE_LONG: return 4;
    case MH_STORAGE_LONGLONG: return 8;
    case MH_STORAGE_FLOAT: return 4;
    case MH_STORAGE_DOUBLE: return 8;
    case MH_STORAGE_QUANTUM: return quantum == MH_QUANTUM_U16 ? 2 : 4;
    default: return 0;
  }
}

// map: the caller's component string; fills comp[] / channel[] for an image laid out
// R[,G,B][,A] or gray[,A]
MhStatus launch_pixel_io(bool import,const View &img,const MhImage *desc,int x,int y,int width,
  int height,const char *map,MhStorageType type,void *buffer_device)
{
  PixelIoArgs a;
  memset(&a,0,sizeof(a));
  const size_t length=strlen(map);
  if ((length == 0) || (length > kMaxComponents))
    return fail(MH_UNSUPPORTED,"pixel map `%s': 1..%d components",map,(int) kMaxComponents);
  const int colour=img.channels-(desc->alpha_offset >= 0 ? 1 : 0);
  for (size_t i=0; i < length; i++)
    {
      int what,channel=-1;
      switch (map[i])
      {
        case 'R': case 'r': what=COMP_RED; channel=0; break;
        case 'G': case 'g': what=COMP_GREEN; channel=colour >= 3 ? 1 : 0; break;
        case 'B': case 'b': what=COMP_BLUE; channel=colour >= 3 ? 2 : 0; break;
        case 'A': case 'a': case 'O': case 'o': what=COMP_ALPHA; channel=desc->alpha_offset; break;
        case 'I': case 'i': what=COMP_INTENSITY; channel=0; break;      // SetPixelGray
        case 'P': case 'p': what=COMP_SKIP; break;
        default:
          // C, M, Y, K need a CMYK image, which the accelerate gate does not admit
          return fail(MH_UNSUPPORTED,"pixel map `%s': component `%c' is not accelerated",map,map[i]);
      }
      if (import && (what == COMP_ALPHA) && (channel < 0))
        return fail(MH_UNSUPPORTED,"pixel map `%s' carries alpha but the image has no alpha channel",map);
      a.comp[i]=(int8_t) what;
      a.channel[i]=(int8_t) channel;
    }
  a.image=img.pixels;
  a.buffer=buffer_device;
  a.columns=(int) img.columns;
  a.rows=(int) img.rows;
  a.channels=img.channels;
  a.x=x; a.y=y; a.width=width; a.height=height;
  a.ncomp=(int) length;
  a.pad_writes_zero=(!import && ((strcasecmp(map,"RGBP") == 0) || (strcasecmp(map,"BGRP") == 0))) ? 1 : 0;
  a.intensity=intensity_params(desc);
  a.alpha_channel=desc->alpha_offset;
  ProfileScope prof(import ? "import_pixels" : "export_pixels",img.stream);
  if (img.quantum == MH_QUANTUM_U16)
    return launch_io_typed<uint16_t>(import,type,a,img.stream);
  return launch_io_typed<float>(import,type,a,img.stream);
}

