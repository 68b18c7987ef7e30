// Pointwise and reduction kernels: colourspace transforms, LUT apply,
// histogram, gray scan, the unsharp-mask epilogue and a plain copy.
//
// All of these stream every pixel once (HBM-bound): lanes take consecutive
// pixels with the widest load the pixel size allows, grid-stride over the
// image, >= 2048 workgroups.
//
// Reference semantics restated from:
//   DecodePixelGamma / EncodePixelGamma   MagickCore/pixel.c:260-324, :380-451
//   ConvertRGBToXYZ / XYZToLab / LabToXYZ / XYZToRGB
//                                          MagickCore/colorspace-private.h:759-779, :1066-1089, :531-557, :72-94
//   sRGBTransformImage / TransformsRGBImage MagickCore/colorspace.c:1031-1049, :1203-1214, :2373-2381, :2541-2552
//   GetPixelIntensity                      MagickCore/pixel.c:2356-2455
//   histogram / LUT apply                  MagickCore/enhance.c:1616-1647, :1778-1788, :2102-2133, :2252-2262
//   IdentifyImageGray                      MagickCore/attribute.c:1564-1626
//   UnsharpMaskImage epilogue              MagickCore/effect.c:4343-4372
#include "mh_internal.hpp"
#include "device_common.hpp"

#include <map>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <strings.h>

namespace mh {

// --------------------------------------------------------------- sRGB gamma
// x^2.4 via frexp + 9-term Chebyshev + power-of-two table, pixel.c:260-316
static __device__ double decode_gamma(double x)
{
  const double c0=1.7917488588043277509,c1=0.82045614371976854984,
    c2=0.027694100686325412819,c3=-0.00094244335181762134018,
    c4=0.000064355540911469709545,c5=-5.7224404636060757485e-06,
    c6=5.8767669437311184313e-07,c7=-6.6139920053589721168e-08,
    c8=7.9323242696227458163e-09;
  int exponent;
  double t0=1.0;
  double t1=4.0*frexp(x,&exponent)-3.0;
  double t2=2.0*t1*t1-t0;
  double t3=2.0*t1*t2-t1;
  double t4=2.0*t1*t3-t2;
  double t5=2.0*t1*t4-t3;
  double t6=2.0*t1*t5-t4;
  double t7=2.0*t1*t6-t5;
  double t8=2.0*t1*t7-t6;
  double p=c0*t0+c1*t1+c2*t2+c3*t3+c4*t4+c5*t5+c6*t6+c7*t7+c8*t8;
  int e=exponent-1;
  int quot=e/5,rem=e%5;            // div(): truncation toward zero
  if (rem < 0)
    {
      quot-=1;
      rem+=5;
    }
  double pw;
  switch (rem)
  {
    case 0: pw=1.0; break;
    case 1: pw=2.6390158215457883983; break;
    case 2: pw=6.9644045063689921093; break;
    case 3: pw=1.8379173679952558018e+01; break;
    default: pw=4.8502930128332728543e+01; break;
  }
  return x*ldexp(pw*p,7*quot);
}

// x^(5/12), pixel.c:380-443
static __device__ double encode_gamma(double x)
{
  const double c0=1.1758200232996901923,c1=0.16665763094889061230,
    c2=-0.0083154894939042125035,c3=0.00075187976780420279038,
    c4=-0.000083240178519391795367,c5=0.000010229209410070008679,
    c6=-1.3400466409860246e-06,c7=1.8333422241635376682e-07,
    c8=-2.5878596761348859722e-08;
  int exponent;
  double t0=1.0;
  double t1=4.0*frexp(x,&exponent)-3.0;
  double t2=2.0*t1*t1-t0;
  double t3=2.0*t1*t2-t1;
  double t4=2.0*t1*t3-t2;
  double t5=2.0*t1*t4-t3;
  double t6=2.0*t1*t5-t4;
  double t7=2.0*t1*t6-t5;
  double t8=2.0*t1*t7-t6;
  double p=c0*t0+c1*t1+c2*t2+c3*t3+c4*t4+c5*t5+c6*t6+c7*t7+c8*t8;
  int e=exponent-1;
  int quot=e/12,rem=e%12;
  if (rem < 0)
    {
      quot-=1;
      rem+=12;
    }
  double pw;
  switch (rem)
  {
    case 0: pw=1.0; break;
    case 1: pw=1.3348398541700343678; break;
    case 2: pw=1.7817974362806785482; break;
    case 3: pw=2.3784142300054420538; break;
    case 4: pw=3.1748021039363991669; break;
    case 5: pw=4.2378523774371812394; break;
    case 6: pw=5.6568542494923805819; break;
    case 7: pw=7.5509945014535482244; break;
    case 8: pw=1.0079368399158985525e1; break;
    case 9: pw=1.3454342644059433809e1; break;
    case 10: pw=1.7959392772949968275e1; break;
    default: pw=2.3972913230026907883e1; break;
  }
  return ldexp(pw*p,5*quot);
}

// DecodePixelGamma, pixel.c:318-324
static __device__ double decode_pixel_gamma(double pixel)
{
  if (pixel <= (0.0404482362771076*kQR))
    return pixel/12.92;
  return kQR*decode_gamma((double) (kQS*pixel+0.055)/1.055);
}

// EncodePixelGamma, pixel.c:445-451
static __device__ double encode_pixel_gamma(double pixel)
{
  if (pixel <= (0.0031306684425005883*kQR))
    return 12.92*pixel;
  return kQR*(1.055*encode_gamma(kQS*pixel)-0.055);
}

// D65, colorspace-private.h:25-44
#define MH_ILL_X 0.95047
#define MH_ILL_Y 1.00000
#define MH_ILL_Z 1.08883
#define MH_CIE_EPSILON (216.0/24389.0)
#define MH_CIE_K (24389.0/27.0)

// ConvertRGBToXYZ, colorspace-private.h:759-779
static __device__ void rgb_to_xyz(double red,double green,double blue,double &X,double &Y,double &Z)
{
  double r=kQS*decode_pixel_gamma(red);
  double g=kQS*decode_pixel_gamma(green);
  double b=kQS*decode_pixel_gamma(blue);
  X=(0.4123955889674142161*r)+(0.3575834307637148171*g)+(0.1804926473817015735*b);
  Y=(0.2125862307855955516*r)+(0.7151703037034108499*g)+(0.07220049864333622685*b);
  Z=(0.01929721549174694484*r)+(0.1191838645808485318*g)+(0.9504971251315797660*b);
}

// ConvertXYZToRGB, colorspace-private.h:72-94
static __device__ void xyz_to_rgb(double X,double Y,double Z,double &red,double &green,double &blue)
{
  double r=(3.240969941904521*X)+(-1.537383177570093*Y)+(-0.498610760293*Z);
  double g=(-0.96924363628087*X)+(1.87596750150772*Y)+(0.041555057407175*Z);
  double b=(0.055630079696993*X)+(-0.20397695888897*Y)+(1.056971514242878*Z);
  double gb=g < b ? g : b;
  double mn=r < gb ? r : gb;
  if (mn < 0.0)
    {
      r-=mn;
      g-=mn;
      b-=mn;
    }
  red=encode_pixel_gamma(kQR*r);
  green=encode_pixel_gamma(kQR*g);
  blue=encode_pixel_gamma(kQR*b);
}

// pow(t,1.0/3.0) for t in (CIEEpsilon, ~1.2].  The device's generic fp64 pow costs
// ~150 instructions; this is a float cbrt seed refined by two Newton steps whose
// residual t-y^3 is formed exactly (product split with an FMA) and whose 1/(3y^2)
// only needs float accuracy.  Error < 1 ulp of the true cube root; libm's
// pow(t,1.0/3.0) (exponent 1/3-1.85e-17) lies within 0.6 ulp of it as well, so the
// two agree to ~2 double ulps: identical after rounding to Q16, within the 1 float
// ULP the Lab tests allow for float Quantum.
static __device__ __forceinline__ double cube_root(double t)
{
  // seed: 2^(log2(t)/3) on the transcendental unit (~1e-6 relative), not libm's cbrtf
  const float seed=__builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float) t)*0.333333343f);
  double y=(double) seed;
  const double inv=(double) __builtin_amdgcn_rcpf(3.0f*seed*seed);
#pragma unroll
  for (int it=0; it < 2; it++)
    {
      double yy=y*y;
      double yy_lo=__builtin_fma(y,y,-yy);            // y*y = yy + yy_lo exactly
      double res=__builtin_fma(-yy,y,t);               // t - yy*y (one rounding)
      res=__builtin_fma(-yy_lo,y,res);
      y=__builtin_fma(res,inv,y);
    }
  return y;
}

// x/d for a compile-time constant d without the division sequence: q=RN(x*RN(1/d)) refined
// twice with exact FMA residuals.  After the first refinement q is a faithful quotient, and
// a faithful quotient corrected once more by r*RN(1/d) is the correctly rounded one
// (Markstein's theorem; needs a significand of d that is not all ones — true of every
// constant used here), i.e. the same double the reference's `/` produces.  Non-finite
// quotients pass through unrefined (inf-inf would turn them into NaN).
static __device__ __forceinline__ double div_const(double x,const double d,const double y)
{
  double q=x*y;
  double r=__builtin_fma(-d,q,x);
  double q1=__builtin_fma(r,y,q);
  r=__builtin_fma(-d,q1,x);
  q1=__builtin_fma(r,y,q1);
  return __builtin_isfinite(q) ? q1 : q;
}
#define MH_DIV(x,d) div_const((x),(d),1.0/(d))

// ConvertXYZToLab, colorspace-private.h:1066-1089
static __device__ void xyz_to_lab(double X,double Y,double Z,double &L,double &a,double &b)
{
  double x,y,z;
  const double xr=MH_DIV(X,MH_ILL_X),yr=Y,zr=MH_DIV(Z,MH_ILL_Z);     // Y/1.0 is Y
  if (xr > MH_CIE_EPSILON)
    x=cube_root(xr);
  else
    x=MH_DIV(MH_DIV(MH_CIE_K*X,MH_ILL_X)+16.0,116.0);
  if (yr > MH_CIE_EPSILON)
    y=cube_root(yr);
  else
    y=MH_DIV(MH_CIE_K*Y+16.0,116.0);
  if (zr > MH_CIE_EPSILON)
    z=cube_root(zr);
  else
    z=MH_DIV(MH_DIV(MH_CIE_K*Z,MH_ILL_Z)+16.0,116.0);
  L=MH_DIV((116.0*y)-16.0,100.0);
  a=MH_DIV(500.0*(x-y),255.0)+0.5;
  b=MH_DIV(200.0*(y-z),255.0)+0.5;
}

// ConvertLabToXYZ, colorspace-private.h:531-557
static __device__ void lab_to_xyz(double L,double a,double b,double &X,double &Y,double &Z)
{
  double y=MH_DIV(L+16.0,116.0);
  double x=y+MH_DIV(a,500.0);
  double z=y-MH_DIV(b,200.0);
  if ((x*x*x) > MH_CIE_EPSILON)
    x=(x*x*x);
  else
    x=MH_DIV(116.0*x-16.0,MH_CIE_K);
  if (L > (MH_CIE_K*MH_CIE_EPSILON))
    y=(y*y*y);
  else
    y=MH_DIV(L,MH_CIE_K);
  if ((z*z*z) > MH_CIE_EPSILON)
    z=(z*z*z);
  else
    z=MH_DIV(116.0*z-16.0,MH_CIE_K);
  X=MH_ILL_X*x;
  Y=MH_ILL_Y*y;
  Z=MH_ILL_Z*z;
}

enum ColorOp
{
  OP_SRGB_TO_RGB,OP_RGB_TO_SRGB,OP_SRGB_TO_LAB,OP_LAB_TO_SRGB,OP_SRGB_TO_XYZ,OP_XYZ_TO_SRGB
};

// Q16 pixels have only 65536 possible sRGB samples, so the transfer functions are tabulated
// once per device by the same device code that evaluates them per pixel (bit-identical):
// QuantumScale*DecodePixelGamma(j) as doubles (512 KB, L2-resident) for the XYZ/Lab kernels,
// and the Quantum-rounded decode / encode columns for sRGB <-> linear RGB, which then run as
// LUT applications.
__global__ __launch_bounds__(256)
void colorspace_table_kernel(double *decode_scaled,uint16_t *decode_q16,uint16_t *encode_q16)
{
  const unsigned j=blockIdx.x*blockDim.x+threadIdx.x;
  if (j > 65535u)
    return;
  const double d=decode_pixel_gamma((double) j);
  decode_scaled[j]=kQS*d;
  decode_q16[j]=QuantumOps<uint16_t>::clamp(d);
  encode_q16[j]=QuantumOps<uint16_t>::clamp(encode_pixel_gamma((double) j));
}

// Pointwise kernels keep kPointBatch pixels per lane in flight: with one 8-byte load per lane
// the latency of the load -> (gather ->) compute -> store chain, not HBM, sets the rate
// (measured 1.2 TB/s for the Lab kernel before batching).
constexpr int kPointBatch=4;

template<int C,int OP>
__global__ __launch_bounds__(256)
void colorspace_q16_table_kernel(uint16_t *__restrict__ pixels,size_t npixels,
  const double *__restrict__ decode_scaled)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x*kPointBatch;
  for (size_t i0=(size_t) blockIdx.x*blockDim.x*kPointBatch+threadIdx.x; i0 < npixels; i0+=stride)
    {
      uint16_t q[kPointBatch][C];
#pragma unroll
      for (int k=0; k < kPointBatch; k++)
        {
          const size_t i=i0+(size_t) k*blockDim.x;
          load_pixel<uint16_t,C>(pixels+(i < npixels ? i : npixels-1)*C,q[k]);
        }
      // ConvertRGBToXYZ, colorspace-private.h:759-779, with the three decodes looked up
      double r[kPointBatch],g[kPointBatch],b[kPointBatch];
#pragma unroll
      for (int k=0; k < kPointBatch; k++)
        {
          r[k]=decode_scaled[q[k][0]];
          g[k]=decode_scaled[q[k][1]];
          b[k]=decode_scaled[q[k][2]];
        }
#pragma unroll
      for (int k=0; k < kPointBatch; k++)
        {
          const double X=(0.4123955889674142161*r[k])+(0.3575834307637148171*g[k])+(0.1804926473817015735*b[k]);
          const double Y=(0.2125862307855955516*r[k])+(0.7151703037034108499*g[k])+(0.07220049864333622685*b[k]);
          const double Z=(0.01929721549174694484*r[k])+(0.1191838645808485318*g[k])+(0.9504971251315797660*b[k]);
          double o0,o1,o2;
          if constexpr (OP == OP_SRGB_TO_LAB)
            {
              double L,a,bb;
              xyz_to_lab(X,Y,Z,L,a,bb);
              o0=kQR*L; o1=kQR*a; o2=kQR*bb;
            }
          else
            {
              o0=kQR*X; o1=kQR*Y; o2=kQR*Z;
            }
          q[k][0]=QuantumOps<uint16_t>::clamp(o0);
          q[k][1]=QuantumOps<uint16_t>::clamp(o1);
          q[k][2]=QuantumOps<uint16_t>::clamp(o2);
          const size_t i=i0+(size_t) k*blockDim.x;
          if (i < npixels)
            store_pixel<uint16_t,C>(pixels+i*C,q[k]);
        }
    }
}

// FAST sRGB -> Lab on RGBA Q16 (config C4): the same expressions in f32 with the hardware
// log2 / exp2 for the 2.4 power and the cube roots.  Error budget in Quantum levels: the decode is
// good to 1e-6 relative, the cube roots to 3e-7, so L (x 65535 * 1.16) is within 0.03 and a, b
// (x 65535 * 500/255 of a difference of two roots) within 0.1 of the fp64 value: the rounded
// level differs from the reference's by at most one (MH_PRECISION_FAST's contract).  ~80 VALU
// slots per pixel against ~400 fp64-rate slots of the table kernel: the kernel becomes a stream.
static __device__ __forceinline__ float srgb_decode_fast(float x)
{
  // DecodePixelGamma, pixel.c:318-324, on [0,1]
  const float curve=__builtin_amdgcn_exp2f(2.4f*__builtin_amdgcn_logf((x+0.055f)*(1.0f/1.055f)));
  return x <= 0.0404482362771076f ? x*(1.0f/12.92f) : curve;
}

static __device__ __forceinline__ float lab_f_fast(float t)
{
  // ConvertXYZToLab, colorspace-private.h:1066-1089: cube root above epsilon, linear below.  Both sides are
  // computed and one is selected: as a branch, the dark lanes of a wave — there almost always are some — made the
  // wave run both sides behind six exec-mask switches per pixel pair.  (log2 of a t <= 0 is NaN / -inf: not selected.)
  float root=__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(t)*(1.0f/3.0f));
  asm volatile("" : "+v"(root));               // (evaluated here, for every lane: not sunk behind the comparison)
  const float line=__builtin_fmaf((float) (MH_CIE_K/116.0),t,(float) (16.0/116.0));
  return t > (float) MH_CIE_EPSILON ? root : line;
}

// The decode as a table in LDS: 2048 linear pieces of the curve, (value, rise) per piece.  The
// curve's second derivative is at most 3.1, so a piece is within (1/2048)^2/8*3.1 = 9e-8 of it —
// a tenth of the hardware log2/exp2 route's error — for a conversion, a multiply, a fraction, one
// ds_read_b64 and one FMA per sample instead of two quarter-rate transcendentals and six other
// operations.  Both FAST kernels decode through it (the one-call Lab + ContrastStretch leaves the
// frame TransformImageColorspace leaves, bit for bit).
constexpr int kDecodePieces=2048;
static __device__ __forceinline__ void build_decode_table(float2 *table)
{
  for (int i=(int) threadIdx.x; i < kDecodePieces; i+=(int) blockDim.x)
    {
      const float here=srgb_decode_fast((float) i*(1.0f/kDecodePieces));
      const float next=srgb_decode_fast((float) (i+1)*(1.0f/kDecodePieces));
      table[i]=make_float2(here,next-here);
    }
}

static __device__ __forceinline__ float srgb_decode_table(const float2 *table,unsigned quantum)
{
  // quantum/65535 in pieces; 65535 is the END of the last piece (fraction 1)
  const float at=(float) quantum*((float) kDecodePieces/65535.0f);
  const float whole=__builtin_fminf(__builtin_floorf(at),(float) (kDecodePieces-1));
  const float2 piece=table[(int) whole];
  return __builtin_fmaf(at-whole,piece.y,piece.x);
}

static __device__ __forceinline__ uint2 srgb_to_lab_fast_pixel(uint2 px,const float2 *decode)
{
  const float r=srgb_decode_table(decode,px.x & 0xffffu);
  const float g=srgb_decode_table(decode,px.x >> 16);
  const float b=srgb_decode_table(decode,px.y & 0xffffu);
  // (fused multiply-adds written out: the library is compiled with -ffp-contract=off for the EXACT kernels' sake,
  // and this f32 form has its own error budget; the illuminants folded into the X and Z rows)
  constexpr float ix=(float) (1.0/MH_ILL_X),iz=(float) (1.0/MH_ILL_Z);
  const float X=__builtin_fmaf(0.4123955889674142161f*ix,r,__builtin_fmaf(0.3575834307637148171f*ix,g,(0.1804926473817015735f*ix)*b));
  const float Y=__builtin_fmaf(0.2125862307855955516f,r,__builtin_fmaf(0.7151703037034108499f,g,0.07220049864333622685f*b));
  const float Z=__builtin_fmaf(0.01929721549174694484f*iz,r,__builtin_fmaf(0.1191838645808485318f*iz,g,(0.9504971251315797660f*iz)*b));
  const float fx=lab_f_fast(X),fy=lab_f_fast(Y),fz=lab_f_fast(Z);
  const float L=__builtin_fmaf(1.16f,fy,-0.16f);
  const float a=__builtin_fmaf(500.0f/255.0f,fx-fy,0.5f);
  const float bb=__builtin_fmaf(200.0f/255.0f,fy-fz,0.5f);
  // ClampToQuantum: v_cvt_pknorm_u16 clamps to [0,1] and rounds to nearest
  typedef unsigned short pk2 __attribute__((ext_vector_type(2)));
  const pk2 la=__builtin_amdgcn_cvt_pknorm_u16(L,a);
  const pk2 b0=__builtin_amdgcn_cvt_pknorm_u16(bb,0.0f);
  return make_uint2((unsigned) la[0] | ((unsigned) la[1] << 16),(unsigned) b0[0] | (px.y & 0xffff0000u));
}

__global__ __launch_bounds__(1024)
void colorspace_lab_fast_kernel(uint4 *__restrict__ pairs,size_t npairs,uint2 *__restrict__ last)
{
  __shared__ float2 decode[kDecodePieces];
  build_decode_table(decode);
  __syncthreads();
  // two 8-byte pixels per lane and load; `last` is the odd pixel of the frame, if any
  constexpr int BATCH=4;
  const size_t stride=(size_t) gridDim.x*blockDim.x*BATCH;
  for (size_t i0=(size_t) blockIdx.x*blockDim.x*BATCH+threadIdx.x; i0 < npairs; i0+=stride)
    {
      uint4 v[BATCH];
#pragma unroll
      for (int k=0; k < BATCH; k++)
        {
          const size_t i=i0+(size_t) k*blockDim.x;
          v[k]=pairs[i < npairs ? i : npairs-1];
        }
#pragma unroll
      for (int k=0; k < BATCH; k++)
        {
          const uint2 first=srgb_to_lab_fast_pixel(make_uint2(v[k].x,v[k].y),decode);
          const uint2 second=srgb_to_lab_fast_pixel(make_uint2(v[k].z,v[k].w),decode);
          const size_t i=i0+(size_t) k*blockDim.x;
          if (i < npairs)
            pairs[i]=make_uint4(first.x,first.y,second.x,second.y);
        }
    }
  if ((last != nullptr) && (blockIdx.x == 0) && (threadIdx.x == 0))
    *last=srgb_to_lab_fast_pixel(*last,decode);
}

// ... on RGB frames (6-byte pixels, no alpha: what most photographs are; they took the table kernel of the bit-identical
// route at 2.8x the time of an RGBA frame): four pixels = three 8-byte words a lane; `tail` = the frame's last
// npixels mod 4 pixels.
__global__ __launch_bounds__(1024)
void colorspace_lab_fast_rgb_kernel(uint2 *__restrict__ words,size_t nquads,uint16_t *__restrict__ tail,int ntail)
{
  __shared__ float2 decode[kDecodePieces];
  build_decode_table(decode);
  __syncthreads();
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < nquads; i+=stride)
    {
      const uint2 w0=words[3*i],w1=words[3*i+1],w2=words[3*i+2];
      const uint2 p0=srgb_to_lab_fast_pixel(make_uint2(w0.x,w0.y & 0xffffu),decode);
      const uint2 p1=srgb_to_lab_fast_pixel(make_uint2((w0.y >> 16) | (w1.x << 16),w1.x >> 16),decode);
      const uint2 p2=srgb_to_lab_fast_pixel(make_uint2(w1.y,w2.x & 0xffffu),decode);
      const uint2 p3=srgb_to_lab_fast_pixel(make_uint2((w2.x >> 16) | (w2.y << 16),w2.y >> 16),decode);
      words[3*i]=make_uint2(p0.x,(p0.y & 0xffffu) | (p1.x << 16));
      words[3*i+1]=make_uint2((p1.x >> 16) | (p1.y << 16),p2.x);
      words[3*i+2]=make_uint2((p2.y & 0xffffu) | (p3.x << 16),(p3.x >> 16) | (p3.y << 16));
    }
  if ((blockIdx.x == 0) && ((int) threadIdx.x < ntail))
    {
      uint16_t *q=tail+3*threadIdx.x;
      const uint2 p=srgb_to_lab_fast_pixel(make_uint2((unsigned) q[0] | ((unsigned) q[1] << 16),(unsigned) q[2]),decode);
      q[0]=(uint16_t) (p.x & 0xffffu);
      q[1]=(uint16_t) (p.x >> 16);
      q[2]=(uint16_t) (p.y & 0xffffu);
    }
}

template<typename Q,int C,int OP>
__global__ __launch_bounds__(256)
void colorspace_kernel(Q *__restrict__ pixels,size_t npixels)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x*kPointBatch;
  for (size_t i0=(size_t) blockIdx.x*blockDim.x*kPointBatch+threadIdx.x; i0 < npixels; i0+=stride)
    {
      Q qb[kPointBatch][C];
#pragma unroll
      for (int k=0; k < kPointBatch; k++)
        {
          const size_t ik=i0+(size_t) k*blockDim.x;
          load_pixel<Q,C>(pixels+(ik < npixels ? ik : npixels-1)*C,qb[k]);
        }
#pragma unroll
      for (int k=0; k < kPointBatch; k++)
    {
      const size_t i=i0+(size_t) k*blockDim.x;
      Q (&q)[C]=qb[k];
      double r=(double) q[0],g=(double) q[1],b=(double) q[2];
      double o0,o1,o2;
      if constexpr (OP == OP_SRGB_TO_RGB)
        {
          o0=decode_pixel_gamma(r);
          o1=decode_pixel_gamma(g);
          o2=decode_pixel_gamma(b);
        }
      else if constexpr (OP == OP_RGB_TO_SRGB)
        {
          o0=encode_pixel_gamma(r);
          o1=encode_pixel_gamma(g);
          o2=encode_pixel_gamma(b);
        }
      else if constexpr (OP == OP_SRGB_TO_LAB)
        {
          double X,Y,Z,L,a,bb;
          rgb_to_xyz(r,g,b,X,Y,Z);
          xyz_to_lab(X,Y,Z,L,a,bb);
          o0=kQR*L; o1=kQR*a; o2=kQR*bb;          // colorspace.c:1041-1043
        }
      else if constexpr (OP == OP_SRGB_TO_XYZ)
        {
          double X,Y,Z;
          rgb_to_xyz(r,g,b,X,Y,Z);
          o0=kQR*X; o1=kQR*Y; o2=kQR*Z;
        }
      else if constexpr (OP == OP_LAB_TO_SRGB)
        {
          // ConvertGenericToRGB(QuantumScale*R,...) -> ConvertLabToRGB, colorspace.c:2373-2377
          double L=kQS*r,a=kQS*g,bb=kQS*b,X,Y,Z;
          lab_to_xyz(100.0*L,255.0*(a-0.5),255.0*(bb-0.5),X,Y,Z);
          xyz_to_rgb(X,Y,Z,o0,o1,o2);
        }
      else
        {
          xyz_to_rgb(kQS*r,kQS*g,kQS*b,o0,o1,o2);
        }
      q[0]=QuantumOps<Q>::clamp(o0);
      q[1]=QuantumOps<Q>::clamp(o1);
      q[2]=QuantumOps<Q>::clamp(o2);
      if (i < npixels)
        store_pixel<Q,C>(pixels+i*C,q);
    }
    }
}

static unsigned stream_grid(size_t npixels)
{
  size_t blocks=(npixels+255)/256;
  if (blocks > 8192)
    blocks=8192;
  if (blocks < 1)
    blocks=1;
  return (unsigned) blocks;
}

template<typename Q,int C>
static MhStatus colorspace_typed(const View &img,int op)
{
  const size_t n=img.columns*img.rows;
  Q *p=static_cast<Q *>(img.pixels);
  dim3 grid(stream_grid((n+kPointBatch-1)/kPointBatch)),block(256);
  ProfileScope prof("colorspace",img.stream);
  switch (op)
  {
    case OP_SRGB_TO_RGB: hipLaunchKernelGGL((colorspace_kernel<Q,C,OP_SRGB_TO_RGB>),grid,block,0,img.stream,p,n); break;
    case OP_RGB_TO_SRGB: hipLaunchKernelGGL((colorspace_kernel<Q,C,OP_RGB_TO_SRGB>),grid,block,0,img.stream,p,n); break;
    case OP_SRGB_TO_LAB: hipLaunchKernelGGL((colorspace_kernel<Q,C,OP_SRGB_TO_LAB>),grid,block,0,img.stream,p,n); break;
    case OP_LAB_TO_SRGB: hipLaunchKernelGGL((colorspace_kernel<Q,C,OP_LAB_TO_SRGB>),grid,block,0,img.stream,p,n); break;
    case OP_SRGB_TO_XYZ: hipLaunchKernelGGL((colorspace_kernel<Q,C,OP_SRGB_TO_XYZ>),grid,block,0,img.stream,p,n); break;
    default: hipLaunchKernelGGL((colorspace_kernel<Q,C,OP_XYZ_TO_SRGB>),grid,block,0,img.stream,p,n); break;
  }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// per-device transfer-function tables (see colorspace_table_kernel); built on first use
struct ColorTables
{
  void *block=nullptr;
  double *decode_scaled=nullptr;
  uint16_t *decode_q16=nullptr,*encode_q16=nullptr;
};
static std::mutex color_tables_lock;
static std::map<int,ColorTables> color_tables;

static MhStatus acquire_color_tables(int device,hipStream_t stream,ColorTables *out)
{
  std::lock_guard<std::mutex> guard(color_tables_lock);
  auto it=color_tables.find(device);
  if (it != color_tables.end())
    {
      *out=it->second;
      return MH_OK;
    }
  ColorTables t;
  MH_HIP(hipMalloc(&t.block,65536*(sizeof(double)+2*sizeof(uint16_t))));
  t.decode_scaled=static_cast<double *>(t.block);
  t.decode_q16=reinterpret_cast<uint16_t *>(t.decode_scaled+65536);
  t.encode_q16=t.decode_q16+65536;
  hipLaunchKernelGGL(colorspace_table_kernel,dim3(256),dim3(256),0,stream,t.decode_scaled,
    t.decode_q16,t.encode_q16);
  hipError_t err=hipGetLastError();
  if (err == hipSuccess)
    err=hipStreamSynchronize(stream);      // once per device: other streams may use it next
  if (err != hipSuccess)
    {
      (void) hipFree(t.block);
      MH_HIP(err);
    }
  color_tables[device]=t;
  *out=t;
  return MH_OK;
}

void release_color_tables()
{
  std::lock_guard<std::mutex> guard(color_tables_lock);
  for (auto &kv : color_tables)
    (void) hipFree(kv.second.block);
  color_tables.clear();
}

static MhStatus apply_q16_column(const View &img,const uint16_t *column,uint32_t mask,const char *label);

static MhStatus colorspace_step(const View &img,int op)
{
  if ((op == OP_SRGB_TO_LAB) && (img.quantum == MH_QUANTUM_U16) && (img.channels == 4) &&
      (precision() == MH_PRECISION_FAST) && (option("MAGICKHIP_NO_FAST_LAB") == nullptr) &&
      ((reinterpret_cast<uintptr_t>(img.pixels) & 15u) == 0))
    {
      const size_t n=img.columns*img.rows,npairs=n/2;
      uint2 *last=(n & 1) != 0 ? static_cast<uint2 *>(img.pixels)+(n-1) : nullptr;
      if (npairs == 0)
        {
          hipLaunchKernelGGL(colorspace_lab_fast_kernel,dim3(1),dim3(1024),0,img.stream,
            static_cast<uint4 *>(img.pixels),(size_t) 0,last);
          MH_HIP(hipGetLastError());
          return MH_OK;
        }
      ProfileScope prof("colorspace",img.stream);
      // persistent workgroups, two per CU: each builds the decode table once
      const size_t wanted=(npairs+4095)/4096,most=2*(size_t) compute_units(img.device);
      hipLaunchKernelGGL(colorspace_lab_fast_kernel,dim3((unsigned) (wanted < most ? wanted : most)),dim3(1024),0,
        img.stream,static_cast<uint4 *>(img.pixels),npairs,last);
      MH_HIP(hipGetLastError());
      return MH_OK;
    }
  if ((op == OP_SRGB_TO_LAB) && (img.quantum == MH_QUANTUM_U16) && (img.channels == 3) &&
      (precision() == MH_PRECISION_FAST) && (option("MAGICKHIP_NO_FAST_LAB") == nullptr) &&
      ((reinterpret_cast<uintptr_t>(img.pixels) & 7u) == 0))
    {
      const size_t n=img.columns*img.rows,nquads=n/4;
      ProfileScope prof("colorspace",img.stream);
      const size_t wanted=(nquads+1023)/1024,most=2*(size_t) compute_units(img.device);
      hipLaunchKernelGGL(colorspace_lab_fast_rgb_kernel,dim3((unsigned) (wanted < 1 ? 1 : (wanted < most ? wanted : most))),
        dim3(1024),0,img.stream,static_cast<uint2 *>(img.pixels),nquads,
        static_cast<uint16_t *>(img.pixels)+12*nquads,(int) (n-4*nquads));
      MH_HIP(hipGetLastError());
      return MH_OK;
    }
  if ((img.quantum == MH_QUANTUM_U16) && (option("MAGICKHIP_NO_COLOR_TABLES") == nullptr) &&
      ((op == OP_SRGB_TO_RGB) || (op == OP_RGB_TO_SRGB) || (op == OP_SRGB_TO_LAB) ||
       (op == OP_SRGB_TO_XYZ)))
    {
      ColorTables t;
      MH_TRY(acquire_color_tables(img.device,img.stream,&t));
      if ((op == OP_SRGB_TO_RGB) || (op == OP_RGB_TO_SRGB))
        {
          return apply_q16_column(img,op == OP_SRGB_TO_RGB ? t.decode_q16 : t.encode_q16,0x7u,
            "colorspace");
        }
      const size_t n=img.columns*img.rows;
      uint16_t *p=static_cast<uint16_t *>(img.pixels);
      dim3 grid(stream_grid((n+kPointBatch-1)/kPointBatch)),block(256);
      ProfileScope prof("colorspace",img.stream);
      if (img.channels == 3)
        {
          if (op == OP_SRGB_TO_LAB)
            hipLaunchKernelGGL((colorspace_q16_table_kernel<3,OP_SRGB_TO_LAB>),grid,block,0,img.stream,p,n,t.decode_scaled);
          else
            hipLaunchKernelGGL((colorspace_q16_table_kernel<3,OP_SRGB_TO_XYZ>),grid,block,0,img.stream,p,n,t.decode_scaled);
        }
      else
        {
          if (op == OP_SRGB_TO_LAB)
            hipLaunchKernelGGL((colorspace_q16_table_kernel<4,OP_SRGB_TO_LAB>),grid,block,0,img.stream,p,n,t.decode_scaled);
          else
            hipLaunchKernelGGL((colorspace_q16_table_kernel<4,OP_SRGB_TO_XYZ>),grid,block,0,img.stream,p,n,t.decode_scaled);
        }
      MH_HIP(hipGetLastError());
      return MH_OK;
    }
  if (img.quantum == MH_QUANTUM_U16)
    return img.channels == 3 ? colorspace_typed<uint16_t,3>(img,op) :
      colorspace_typed<uint16_t,4>(img,op);
  return img.channels == 3 ? colorspace_typed<float,3>(img,op) :
    colorspace_typed<float,4>(img,op);
}

// (defined with the generic colourspace kernels further down)
static bool colorspace_is_generic(MhColorspace c);
static MhStatus colorspace_generic_forward_or_inverse(const View &img,MhColorspace colorspace,bool forward);

MhStatus launch_colorspace(const View &img,MhColorspace from,MhColorspace to,const MhImage *)
{
  if ((img.channels != 3) && (img.channels != 4))
    return fail(MH_UNSUPPORTED,"colourspace transform needs R,G,B[,A] channels");
  // TransformImageColorspace, colorspace.c:1751-1783: X -> sRGB -> Y
  if (from != MH_COLORSPACE_SRGB)
    {
      int op=-1;
      switch (from)
      {
        case MH_COLORSPACE_RGB: case MH_COLORSPACE_SCRGB: op=OP_RGB_TO_SRGB; break;   // (one case: colorspace.c:2502-2503)
        case MH_COLORSPACE_LAB: op=OP_LAB_TO_SRGB; break;
        case MH_COLORSPACE_XYZ: op=OP_XYZ_TO_SRGB; break;
        default:
          if (!colorspace_is_generic(from))
            return fail(MH_UNSUPPORTED,"source colourspace %d is not accelerated",(int) from);
          break;
      }
      if (op >= 0)
        MH_TRY(colorspace_step(img,op));
      else
        MH_TRY(colorspace_generic_forward_or_inverse(img,from,false));
    }
  if (to != MH_COLORSPACE_SRGB)
    {
      int op=-1;
      switch (to)
      {
        case MH_COLORSPACE_RGB: case MH_COLORSPACE_SCRGB: op=OP_SRGB_TO_RGB; break;   // (colorspace.c:1164-1165)
        case MH_COLORSPACE_LAB: op=OP_SRGB_TO_LAB; break;
        case MH_COLORSPACE_XYZ: op=OP_SRGB_TO_XYZ; break;
        default:
          if (!colorspace_is_generic(to))
            return fail(MH_UNSUPPORTED,"target colourspace %d is not accelerated",(int) to);
          break;
      }
      if (op >= 0)
        MH_TRY(colorspace_step(img,op));
      else
        MH_TRY(colorspace_generic_forward_or_inverse(img,to,true));
    }
  return MH_OK;
}

// ---------------------------------------------------------------- intensity
struct IntensityParams
{
  int method;        // MhIntensityMethod
  int linear;        // colourspace is linear RGB / LinearGRAY
  int nonlinear;     // colourspace is sRGB / GRAY
  int gray;          // R,G,B all live at offset 0 (GRAY / LinearGRAY image)
};

// GetPixelIntensity, pixel.c:2356-2455
template<typename Q,int C>
static __device__ __forceinline__ double pixel_intensity(const Q (&q)[C],const IntensityParams &ip)
{
  double red=(double) q[0];
  if (C == 1)
    return red;
  double green=(double) q[(C >= 3) && !ip.gray ? 1 : 0];
  double blue=(double) q[(C >= 3) && !ip.gray ? 2 : 0];
  switch (ip.method)
  {
    case MH_INTENSITY_AVERAGE:
      return (red+green+blue)/3.0;
    case MH_INTENSITY_BRIGHTNESS:
    {
      double m=red > green ? red : green;
      return m > blue ? m : blue;
    }
    case MH_INTENSITY_LIGHTNESS:
    {
      double mn=red < green ? red : green;
      mn=mn < blue ? mn : blue;
      double mx=red > green ? red : green;
      mx=mx > blue ? mx : blue;
      return (mn+mx)/2.0;
    }
    case MH_INTENSITY_MS:
      return (red*red+green*green+blue*blue)/(3.0*kQR);
    case MH_INTENSITY_REC601LUMA:
      if (ip.linear)
        {
          red=encode_pixel_gamma(red);
          green=encode_pixel_gamma(green);
          blue=encode_pixel_gamma(blue);
        }
      return 0.298839*red+0.586811*green+0.114350*blue;
    case MH_INTENSITY_REC601LUMINANCE:
      if (ip.nonlinear)
        {
          red=decode_pixel_gamma(red);
          green=decode_pixel_gamma(green);
          blue=decode_pixel_gamma(blue);
        }
      return 0.298839*red+0.586811*green+0.114350*blue;
    case MH_INTENSITY_REC709LUMINANCE:
      if (ip.nonlinear)
        {
          red=decode_pixel_gamma(red);
          green=decode_pixel_gamma(green);
          blue=decode_pixel_gamma(blue);
        }
      return 0.212656*red+0.715158*green+0.072186*blue;
    case MH_INTENSITY_RMS:
      return sqrt(red*red+green*green+blue*blue)/sqrt(3.0);
    default:
      break;
  }
  if (ip.linear)
    {
      red=encode_pixel_gamma(red);
      green=encode_pixel_gamma(green);
      blue=encode_pixel_gamma(blue);
    }
  return 0.212656*red+0.715158*green+0.072186*blue;
}

static IntensityParams intensity_params(const MhImage *desc)
{
  IntensityParams ip;
  ip.method=(int) desc->intensity;
  ip.linear=(desc->colorspace == MH_COLORSPACE_RGB) || (desc->colorspace == MH_COLORSPACE_LINEARGRAY);
  ip.nonlinear=(desc->colorspace == MH_COLORSPACE_SRGB) || (desc->colorspace == MH_COLORSPACE_GRAY);
  ip.gray=(desc->colorspace == MH_COLORSPACE_GRAY) || (desc->colorspace == MH_COLORSPACE_LINEARGRAY) ||
    (desc->number_channels < 3);
  return ip;
}

// The default method on a frame that needs no gamma step (Rec709Luma of sRGB, Lab ... pixels),
// pixel.c:2446-2454 — three products and two sums, against the whole switch above inlined at
// every call site (the packed-table kernels evaluate 16 pixels per thread and step: 67 000
// instructions of ISA with the switch, and an instruction-cache-bound loop).
static bool intensity_is_plain_luma(const IntensityParams &ip,int channels)
{
  switch (ip.method)
  {
    case MH_INTENSITY_AVERAGE: case MH_INTENSITY_BRIGHTNESS: case MH_INTENSITY_LIGHTNESS: case MH_INTENSITY_MS:
    case MH_INTENSITY_REC601LUMA: case MH_INTENSITY_REC601LUMINANCE: case MH_INTENSITY_REC709LUMINANCE:
    case MH_INTENSITY_RMS:
      return false;
    default:
      break;
  }
  return (channels >= 3) && (ip.linear == 0) && (ip.gray == 0);
}

template<typename Q,int C>
static __device__ __noinline__ double pixel_intensity_call(const Q (&q)[C],const IntensityParams &ip)
{
  return pixel_intensity<Q,C>(q,ip);
}

template<bool PLAIN,typename Q,int C>
static __device__ __forceinline__ double pixel_intensity_of(const Q (&q)[C],const IntensityParams &ip)
{
  if constexpr (PLAIN && (C >= 3))
    return 0.212656*(double) q[0]+0.715158*(double) q[1]+0.072186*(double) q[2];
  else
    return pixel_intensity_call<Q,C>(q,ip);
}

// ---------------------------------------------------------------- histogram
// 65536 bins x C channels of 64-bit counts (0.5-2 MiB) do not fit LDS, so the
// counts live in global memory (L2 / Infinity-Cache resident) and are updated
// with device-scope atomics.  Two things keep the atomic traffic down:
//   * in intensity mode every channel bins the same value, so only channel 0
//     is accumulated and a 65536-element kernel replicates it afterwards;
//   * when neighbouring lanes agree on the bin (smooth images, 8-bit data) the
//     wave folds equal bins first and issues one atomic per distinct bin.
static __device__ __forceinline__ void histogram_add(unsigned long long *counts,unsigned bin,
  int stride,int c,bool valid)
{
  const int lane=(int) (threadIdx.x & 63);
  unsigned neighbour=__shfl(bin,(lane+1) & 63,64);
  unsigned long long agree=__ballot(valid && (neighbour == bin));
  if (__popcll(agree) < 16)
    {
      if (valid)
        atomicAdd(counts+(size_t) bin*stride+c,1ull);
      return;
    }
  // match-any by leader election over the still-unserved lanes
  unsigned long long remaining=__ballot(valid);
  while (remaining != 0)
    {
      int leader=__ffsll((long long) remaining)-1;
      unsigned leader_bin=__shfl(bin,leader,64);
      unsigned long long same=__ballot(valid && (bin == leader_bin)) & remaining;
      if (lane == leader)
        atomicAdd(counts+(size_t) leader_bin*stride+c,(unsigned long long) __popcll(same));
      remaining&=~same;
    }
}

template<typename Q,int C>
__global__ __launch_bounds__(256)
void histogram_kernel(const Q *pixels,size_t npixels,int intensity_mode,IntensityParams ip,
  unsigned long long *counts)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  const size_t rounds=(npixels+stride-1)/stride;
  size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x;
  for (size_t k=0; k < rounds; k++,i+=stride)
    {
      const bool valid=i < npixels;
      Q q[C];
      load_pixel<Q,C>(pixels+(valid ? i : 0)*C,q);
      if (intensity_mode)
        {
          // ScaleQuantumToMap(ClampToQuantum(intensity)): every channel bins the same value
          unsigned bin=QuantumOps<Q>::map_index(QuantumOps<Q>::clamp(pixel_intensity<Q,C>(q,ip)));
          histogram_add(counts,bin,C,0,valid);
        }
      else
        {
#pragma unroll
          for (int c=0; c < C; c++)
            {
              unsigned bin=QuantumOps<Q>::map_index(QuantumOps<Q>::clamp((double) q[c]));
              histogram_add(counts,bin,C,c,valid);
            }
        }
    }
}

// intensity mode: channel 0 holds this call's counts (added on top of what the
// caller passed in); propagate the increment to the other channels
__global__ __launch_bounds__(256)
void histogram_replicate_kernel(unsigned long long *counts,const unsigned long long *before,int channels)
{
  unsigned bin=blockIdx.x*blockDim.x+threadIdx.x;
  if (bin > 65535u)
    return;
  unsigned long long delta=counts[(size_t) bin*channels]-before[bin];
  for (int c=1; c < channels; c++)
    counts[(size_t) bin*channels+c]+=delta;
}

__global__ __launch_bounds__(256)
void histogram_snapshot_kernel(const unsigned long long *counts,unsigned long long *before,int channels)
{
  unsigned bin=blockIdx.x*blockDim.x+threadIdx.x;
  if (bin <= 65535u)
    before[bin]=counts[(size_t) bin*channels];
}

// Intensity mode, LDS-privatised.  65 536 bins do not fit LDS as 32-bit counters
// (256 KB > 160 KB), so a persistent workgroup (one per CU, 1024 threads) bins its
// slice of the image twice — bins 0..32767, then 32768..65535 — into a 128 KB LDS
// table with ds_add (no global atomics: on uniform-random Q16 data the global-atomic
// kernel above is bound by 16.7 M L2 atomics per 4096^2 image), and writes each
// half to its own slab; a small kernel sums the slabs into the caller's table
// (all channels of a bin receive the same count in this mode).  The second read
// of the slice comes from L2 / Infinity Cache.
constexpr int kHistHalf=32768;

template<typename Q,int C>
__global__ __launch_bounds__(1024)
void histogram_lds_kernel(const Q *pixels,size_t npixels,IntensityParams ip,unsigned *slabs)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned *table=reinterpret_cast<unsigned *>(smem_raw);
  const size_t per=(npixels+gridDim.x-1)/gridDim.x;
  const size_t begin=(size_t) blockIdx.x*per;
  size_t end=begin+per;
  end=end < npixels ? end : npixels;
  for (int half=0; half < 2; half++)
    {
      for (int i=(int) threadIdx.x; i < kHistHalf; i+=1024)
        table[i]=0u;
      __syncthreads();
      const unsigned base=(unsigned) half*kHistHalf;
      constexpr int BATCH=4;
      for (size_t i0=begin+threadIdx.x; i0 < end; i0+=(size_t) 1024*BATCH)
        {
          Q q[BATCH][C];
#pragma unroll
          for (int k=0; k < BATCH; k++)
            {
              size_t i=i0+(size_t) 1024*k;
              load_pixel<Q,C>(pixels+(i < end ? i : end-1)*C,q[k]);
            }
#pragma unroll
          for (int k=0; k < BATCH; k++)
            if (i0+(size_t) 1024*k < end)
              {
                unsigned bin=QuantumOps<Q>::map_index(QuantumOps<Q>::clamp(pixel_intensity<Q,C>(q[k],ip)));
                unsigned local=bin-base;
                if (local < (unsigned) kHistHalf)
                  atomicAdd(table+local,1u);
              }
        }
      __syncthreads();
      unsigned *slab=slabs+((size_t) blockIdx.x*2+half)*kHistHalf;
      for (int i=(int) threadIdx.x; i < kHistHalf; i+=1024)
        slab[i]=table[i];
      __syncthreads();
    }
}

// grid (256, kSlabGroups): block y sums its share of the per-workgroup slabs for 256 bins
// and adds the partial sum to the caller's table (one 64-bit atomic per bin and channel,
// kSlabGroups-way contention at most) — 8x the loads in flight of a single pass per bin.
constexpr int kSlabGroups=8;

__global__ __launch_bounds__(256)
void histogram_slab_reduce_kernel(const unsigned *slabs,int nblocks,unsigned long long *counts,int channels)
{
  const unsigned bin=blockIdx.x*blockDim.x+threadIdx.x;
  if (bin > 65535u)
    return;
  const unsigned half=bin/kHistHalf,local=bin%kHistHalf;
  const int per=(nblocks+kSlabGroups-1)/kSlabGroups;
  const int b0=(int) blockIdx.y*per;
  const int b1=b0+per < nblocks ? b0+per : nblocks;
  unsigned long long sum=0;
#pragma unroll 8
  for (int b=b0; b < b1; b++)
    sum+=slabs[((size_t) b*2+half)*kHistHalf+local];
  if (sum != 0)
    for (int c=0; c < channels; c++)
      atomicAdd(counts+(size_t) bin*channels+c,sum);
}

template<typename Q,int C>
static MhStatus histogram_intensity_lds(const View &src,const IntensityParams &ip,unsigned long long *hist)
{
  const size_t n=src.columns*src.rows;
  const int nblocks=compute_units(src.device);
  Temp slabs;
  MH_TRY(slabs.alloc(src.device,(size_t) nblocks*2*kHistHalf*sizeof(unsigned),src.stream));
  const size_t lds=(size_t) kHistHalf*sizeof(unsigned);
  MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&histogram_lds_kernel<Q,C>),
    hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  {
    ProfileScope prof("histogram",src.stream);
    hipLaunchKernelGGL((histogram_lds_kernel<Q,C>),dim3(nblocks),dim3(1024),lds,src.stream,
      static_cast<const Q *>(src.pixels),n,ip,slabs.as<unsigned>());
    hipLaunchKernelGGL(histogram_slab_reduce_kernel,dim3(256,kSlabGroups),dim3(256),0,src.stream,
      slabs.as<unsigned>(),nblocks,hist,C);
  }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// Intensity mode on Q16, one pass: 65 536 sixteen-bit counters DO fit (128 KB), two to a
// 32-bit LDS word, incremented with ds_add_u32 of 1 or 1<<16.  A workgroup bins at most 65 534
// pixels into its table, so no counter can wrap into its neighbour; the handful of pixels of an
// over-full share go straight to the caller's table with global atomics.  Each workgroup then
// writes its packed table (128 KB) to its slab and histogram_packed_reduce_kernel sums the slabs:
// the frame is read once, against twice by the 32-bit half-range kernel below.
constexpr unsigned kPackedCapacity=65534u;     // even: the RGBA path loads pixel pairs

// the shared tail of the packed-table kernels: LDS table -> the workgroup's slab
static __device__ __forceinline__ void packed_table_to_slab(const unsigned *table,unsigned *slabs)
{
  uint4 *slab=reinterpret_cast<uint4 *>(slabs+(size_t) blockIdx.x*32768);
  const uint4 *packed=reinterpret_cast<const uint4 *>(table);
  for (int i=(int) threadIdx.x; i < 8192; i+=1024)
    slab[i]=packed[i];
}

template<typename Q,int C,bool PLAIN>
__global__ __launch_bounds__(1024)
void histogram_packed_kernel(const Q *pixels,size_t npixels,IntensityParams ip,unsigned *slabs,
  unsigned long long *counts,int wide)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned *table=reinterpret_cast<unsigned *>(smem_raw);
  for (int i=(int) threadIdx.x; i < 32768; i+=1024)
    table[i]=0u;
  __syncthreads();
  size_t per=(npixels+gridDim.x-1)/gridDim.x;
  per+=per & 1;                                // shares start on a pixel pair
  size_t begin=(size_t) blockIdx.x*per;
  begin=begin < npixels ? begin : npixels;
  size_t end=begin+per;
  end=end < npixels ? end : npixels;
  const size_t packed_end=end-begin > kPackedCapacity ? begin+kPackedCapacity : end;
  auto count=[&](const Q (&q)[C])
  {
    const unsigned bin=QuantumOps<Q>::map_index(QuantumOps<Q>::clamp(pixel_intensity_of<PLAIN,Q,C>(q,ip)));
    atomicAdd(table+(bin >> 1),(bin & 1u) != 0u ? 0x10000u : 1u);
  };
  size_t done=begin;                           // pixels [begin, done) are in the LDS table
  if constexpr ((C == 4) && (sizeof(Q) == 2))
    if (wide != 0)
      {
        // 16-byte loads: two RGBA pixels per lane, eight loads in flight
        constexpr int BATCH=8;
        const uint4 *pairs=reinterpret_cast<const uint4 *>(pixels)+begin/2;
        const size_t npairs=(packed_end-begin)/2;
        for (size_t i0=threadIdx.x; i0 < npairs; i0+=(size_t) 1024*BATCH)
          {
            uint4 v[BATCH];
#pragma unroll
            for (int k=0; k < BATCH; k++)
              {
                const size_t i=i0+(size_t) 1024*k;
                v[k]=pairs[i < npairs ? i : npairs-1];
              }
#pragma unroll
            for (int k=0; k < BATCH; k++)
              if (i0+(size_t) 1024*k < npairs)
                {
                  const uint16_t a[4]={(uint16_t) v[k].x,(uint16_t) (v[k].x >> 16),(uint16_t) v[k].y,(uint16_t) (v[k].y >> 16)};
                  const uint16_t b[4]={(uint16_t) v[k].z,(uint16_t) (v[k].z >> 16),(uint16_t) v[k].w,(uint16_t) (v[k].w >> 16)};
                  count(a);
                  count(b);
                }
          }
        done=begin+2*npairs;
      }
  constexpr int BATCH=4;
  for (size_t i0=done+threadIdx.x; i0 < end; i0+=(size_t) 1024*BATCH)
    {
      Q q[BATCH][C];
#pragma unroll
      for (int k=0; k < BATCH; k++)
        {
          const size_t i=i0+(size_t) 1024*k;
          load_pixel<Q,C>(pixels+(i < end ? i : end-1)*C,q[k]);
        }
#pragma unroll
      for (int k=0; k < BATCH; k++)
        {
          const size_t i=i0+(size_t) 1024*k;
          if (i >= end)
            continue;
          if (i < packed_end)
            count(q[k]);
          else
            {
              // beyond what the 16-bit counters may hold: the caller's table directly
              const unsigned bin=QuantumOps<Q>::map_index(QuantumOps<Q>::clamp(pixel_intensity_of<PLAIN,Q,C>(q[k],ip)));
              for (int c=0; c < C; c++)
                atomicAdd(counts+(size_t) bin*C+c,1ull);
            }
        }
    }
  __syncthreads();
  packed_table_to_slab(table,slabs);
}

// One workgroup per 256 bins: thread (bin, quarter) sums a quarter of the packed slabs, the four
// partial sums meet in LDS and one thread per bin adds the total to every channel of the caller's
// table (no atomics: a bin belongs to one thread).
__global__ __launch_bounds__(1024)
void histogram_packed_reduce_kernel(const unsigned *slabs,int nblocks,unsigned long long *counts,int channels)
{
  __shared__ unsigned partial[4][256];
  const unsigned local=threadIdx.x & 255u,quarter=threadIdx.x >> 8;
  const unsigned bin=blockIdx.x*256u+local;
  const int per=(nblocks+3)/4;
  const int b0=(int) quarter*per;
  const int b1=b0+per < nblocks ? b0+per : nblocks;
  const unsigned shift=16u*(bin & 1u);
  unsigned sum=0;                              // at most 65 534 per slab: 65 536 slabs fit 32 bits
#pragma unroll 8
  for (int b=b0; b < b1; b++)
    sum+=(slabs[(size_t) b*32768+(bin >> 1)] >> shift) & 0xffffu;
  partial[quarter][local]=sum;
  __syncthreads();
  if (quarter == 0)
    {
      const unsigned long long total=(unsigned long long) partial[0][local]+partial[1][local]+partial[2][local]+partial[3][local];
      if (total != 0)
        for (int c=0; c < channels; c++)
          counts[(size_t) bin*channels+c]+=total;
    }
}

// workgroups of the packed-table kernels for a frame of n pixels (0: too many slabs)
static size_t packed_histogram_blocks(int device,size_t n)
{
  // one workgroup per CU when a share fits the 16-bit counters (plus a few pixels of slack that
  // go through global atomics), more workgroups otherwise
  size_t nblocks=(size_t) compute_units(device);
  if ((n+nblocks-1)/nblocks > kPackedCapacity+256u)
    nblocks=(n+kPackedCapacity-1)/kPackedCapacity;
  return nblocks > 65536 ? 0 : nblocks;
}

template<typename Q,int C>
static MhStatus histogram_intensity_packed(const View &src,const IntensityParams &ip,unsigned long long *hist)
{
  const size_t n=src.columns*src.rows;
  const size_t nblocks=packed_histogram_blocks(src.device,n);
  if (nblocks == 0)
    return histogram_intensity_lds<Q,C>(src,ip,hist);
  Temp slabs;
  MH_TRY(slabs.alloc(src.device,nblocks*32768*sizeof(unsigned),src.stream));
  const size_t lds=32768*sizeof(unsigned);
  const bool plain=intensity_is_plain_luma(ip,C);
  MH_HIP(hipFuncSetAttribute(plain ? reinterpret_cast<const void *>(&histogram_packed_kernel<Q,C,true>) :
    reinterpret_cast<const void *>(&histogram_packed_kernel<Q,C,false>),
    hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  const int wide=(C == 4) && (sizeof(Q) == 2) && ((reinterpret_cast<uintptr_t>(src.pixels) & 15u) == 0) ? 1 : 0;
  {
    ProfileScope prof("histogram",src.stream);
    if (plain)
      hipLaunchKernelGGL((histogram_packed_kernel<Q,C,true>),dim3((unsigned) nblocks),dim3(1024),lds,src.stream,
        static_cast<const Q *>(src.pixels),n,ip,slabs.as<unsigned>(),hist,wide);
    else
      hipLaunchKernelGGL((histogram_packed_kernel<Q,C,false>),dim3((unsigned) nblocks),dim3(1024),lds,src.stream,
        static_cast<const Q *>(src.pixels),n,ip,slabs.as<unsigned>(),hist,wide);
    hipLaunchKernelGGL(histogram_packed_reduce_kernel,dim3(256),dim3(1024),0,src.stream,
      slabs.as<unsigned>(),(int) nblocks,hist,C);
  }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// sRGB -> Lab (FAST, RGBA Q16) and the intensity histogram of the Lab frame in ONE pass over the
// pixels: what TransformImageColorspace followed by ContrastStretchImage / EqualizeImage needs
// (config C4) at one frame read and one frame write.  Converts as colorspace_lab_fast_kernel
// and bins the values it stores as histogram_packed_kernel does, so the table is exactly the
// histogram of the frame it leaves behind.
__device__ __forceinline__ unsigned long long lut_block_sum(unsigned long long v,unsigned long long *shared16);
__device__ __forceinline__ double lut_scale_map_to_quantum(double value,int is_u16);

// What the levels kernels below need besides the slabs: a workgroup's pixels beyond the 16-bit
// counters' capacity as a list of bins (count first).
constexpr int kExtraPitch=264;                 // 1 + the 258 pixels a share may exceed the capacity by, padded
struct StretchScratch
{
  int black,white;                             // the levels, enhance.c:1652-1678
  unsigned apply;                              // black != white: the map is applied (enhance.c:1690)
  unsigned pad;
};

template<bool PLAIN>
__global__ __launch_bounds__(1024)
void lab_histogram_fast_kernel(uint16_t *pixels,size_t npixels,IntensityParams ip,unsigned *slabs,
  unsigned long long *counts,unsigned short *extras)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ unsigned extra_count;
  unsigned *table=reinterpret_cast<unsigned *>(smem_raw);
  float2 *decode=reinterpret_cast<float2 *>(smem_raw+32768*sizeof(unsigned));
  for (int i=(int) threadIdx.x; i < 32768; i+=1024)
    table[i]=0u;
  build_decode_table(decode);
  if (threadIdx.x == 0)
    extra_count=0u;
  __syncthreads();
  size_t per=(npixels+gridDim.x-1)/gridDim.x;
  per+=per & 1;
  size_t begin=(size_t) blockIdx.x*per;
  begin=begin < npixels ? begin : npixels;
  size_t end=begin+per;
  end=end < npixels ? end : npixels;
  const size_t packed_end=end-begin > kPackedCapacity ? begin+kPackedCapacity : end;
  auto bin_of=[&](uint2 px) -> unsigned
  {
    const uint16_t q[4]={(uint16_t) px.x,(uint16_t) (px.x >> 16),(uint16_t) px.y,(uint16_t) (px.y >> 16)};
    return QuantumOps<uint16_t>::map_index(QuantumOps<uint16_t>::clamp(pixel_intensity_of<PLAIN,uint16_t,4>(q,ip)));
  };
  auto count=[&](uint2 px)
  {
    const unsigned bin=bin_of(px);
    atomicAdd(table+(bin >> 1),(bin & 1u) != 0u ? 0x10000u : 1u);
  };
  constexpr int BATCH=8;
  uint4 *pairs=reinterpret_cast<uint4 *>(pixels)+begin/2;
  const size_t npairs=(packed_end-begin)/2;
  for (size_t i0=threadIdx.x; i0 < npairs; i0+=(size_t) 1024*BATCH)
    {
      uint4 v[BATCH];
#pragma unroll
      for (int k=0; k < BATCH; k++)
        {
          const size_t i=i0+(size_t) 1024*k;
          v[k]=pairs[i < npairs ? i : npairs-1];
        }
#pragma unroll
      for (int k=0; k < BATCH; k++)
        if (i0+(size_t) 1024*k < npairs)
          {
            const uint2 first=srgb_to_lab_fast_pixel(make_uint2(v[k].x,v[k].y),decode);
            const uint2 second=srgb_to_lab_fast_pixel(make_uint2(v[k].z,v[k].w),decode);
            pairs[i0+(size_t) 1024*k]=make_uint4(first.x,first.y,second.x,second.y);
            count(first);
            count(second);
          }
    }
  // the pixels past the pairs (an odd one, or a share beyond the 16-bit counters' capacity)
  uint2 *single=reinterpret_cast<uint2 *>(pixels);
  for (size_t i=begin+2*npairs+threadIdx.x; i < end; i+=1024)
    {
      const uint2 lab=srgb_to_lab_fast_pixel(single[i],decode);
      single[i]=lab;
      if (i < packed_end)
        count(lab);
      else
        {
          const unsigned bin=bin_of(lab);
          if (extras != nullptr)
            extras[(size_t) blockIdx.x*kExtraPitch+1u+atomicAdd(&extra_count,1u)]=(unsigned short) bin;
          else
            for (int c=0; c < 4; c++)
              atomicAdd(counts+(size_t) bin*4+c,1ull);
        }
    }
  __syncthreads();
  if ((extras != nullptr) && (threadIdx.x == 0))
    extras[(size_t) blockIdx.x*kExtraPitch]=(unsigned short) extra_count;
  packed_table_to_slab(table,slabs);
}

// ContrastStretchImage's levels straight from the slabs, two small launches and no table of
// 65536 x channels 64-bit counts: stretch_bins_kernel sums the slabs (as
// histogram_packed_reduce_kernel) into 32-bit `bins` and leaves each workgroup's share (256 bins)
// in `shares`; stretch_levels_kernel, ONE workgroup, adds the listed extra pixels and finds the
// black and the white level (enhance.c:1652-1678: the first bin from below whose running count
// exceeds black_point, the last bin from above whose count exceeds white_limit) by locating the
// 1024-bin chunk from the 64 chunk sums and scanning it.  Intensity binning: all channels share
// the histogram, hence the levels.  Replaces a 2 MB memset, the slab reduction and the three LUT
// kernels; the map itself is evaluated by stretch_apply_kernel.  (One launch with a last-
// workgroup-done ticket measured 32 us: every __threadfence() writes back an L2 full of the
// frame the previous kernel stored.)
__global__ __launch_bounds__(1024)
void stretch_bins_kernel(const unsigned *slabs,int nblocks,unsigned *bins,unsigned long long *shares)
{
  // 256 bins = 32 groups of four packed words; thread (group, slice) sums every 32nd slab with
  // 16-byte loads, all in flight at once
  __shared__ unsigned partial[32][32][8];
  __shared__ unsigned long long wave_sums[16];
  {
    const int group=(int) threadIdx.x & 31,slice=(int) threadIdx.x >> 5;
    const uint4 *words=reinterpret_cast<const uint4 *>(slabs)+(size_t) blockIdx.x*32+group;
    unsigned sum[8]={0u,0u,0u,0u,0u,0u,0u,0u};     // at most 65 534 per slab: 65 536 slabs fit 32 bits
#pragma unroll 8
    for (int b=slice; b < nblocks; b+=32)
      {
        const uint4 v=words[(size_t) b*8192];
        sum[0]+=v.x & 0xffffu; sum[1]+=v.x >> 16;
        sum[2]+=v.y & 0xffffu; sum[3]+=v.y >> 16;
        sum[4]+=v.z & 0xffffu; sum[5]+=v.z >> 16;
        sum[6]+=v.w & 0xffffu; sum[7]+=v.w >> 16;
      }
#pragma unroll
    for (int k=0; k < 8; k++)
      partial[slice][group][k]=sum[k];
  }
  __syncthreads();
  unsigned long long mine=0ull;
  if (threadIdx.x < 256u)
    {
      unsigned total=0u;
#pragma unroll 8
      for (int slice=0; slice < 32; slice++)
        total+=partial[slice][threadIdx.x >> 3][threadIdx.x & 7u];
      bins[blockIdx.x*256u+threadIdx.x]=total;
      mine=total;
    }
  const unsigned long long share=lut_block_sum(mine,wave_sums);
  if (threadIdx.x == 0)
    shares[blockIdx.x]=share;
}

__global__ __launch_bounds__(1024)
void stretch_levels_kernel(const unsigned *bins,const unsigned long long *shares,const unsigned short *extras,
  int nblocks,StretchScratch *levels,double black_point,double white_limit)
{
  __shared__ unsigned long long chunk_s[64];
  __shared__ unsigned long long wave_sums[16];
  __shared__ unsigned low_s[1024],high_s[1024];    // the bins of the black and of the white chunk
  __shared__ int found_s;
  const int t=(int) threadIdx.x,lane=t & 63;
  if (t < 64)
    chunk_s[t]=shares[4*t]+shares[4*t+1]+shares[4*t+2]+shares[4*t+3];
  __syncthreads();
  // the pixels a share held beyond the packed counters' capacity (a thread per list: a couple of
  // entries each unless the shares were ragged)
  for (int w=t; w < nblocks; w+=1024)
    {
      const unsigned short *list=extras+(size_t) w*kExtraPitch;
      const int listed=(int) list[0];
      for (int i=1; i <= listed; i++)
        atomicAdd(&chunk_s[list[i] >> 10],1ull);
    }
  __syncthreads();
  // counts below each chunk (every wave computes them: 64 chunks, 64 lanes)
  const unsigned long long chunk=chunk_s[lane];
  unsigned long long inclusive=chunk;
  for (int off=1; off < 64; off<<=1)
    {
      const unsigned long long up=__shfl_up(inclusive,off,64);
      if (lane >= off)
        inclusive+=up;
    }
  const unsigned long long total=__shfl(inclusive,63,64);
  const unsigned long long below=inclusive-chunk;
  // black lies in the first chunk whose inclusive running count exceeds black_point, white in the
  // last chunk whose count from its first bin upwards exceeds white_limit
  const unsigned long long low_hit=__ballot((double) inclusive > black_point);
  const unsigned long long high_hit=__ballot((double) (total-below) > white_limit);
  const int low_chunk=low_hit != 0ull ? __builtin_ctzll(low_hit) : -1;
  const int high_chunk=high_hit != 0ull ? 63-__builtin_clzll(high_hit) : -1;
  low_s[t]=low_chunk >= 0 ? bins[low_chunk*1024+t] : 0u;
  high_s[t]=high_chunk >= 0 ? bins[high_chunk*1024+t] : 0u;
  __syncthreads();
  for (int w=t; w < nblocks; w+=1024)
    {
      const unsigned short *list=extras+(size_t) w*kExtraPitch;
      const int listed=(int) list[0];
      for (int i=1; i <= listed; i++)
        {
          const int b=(int) list[i];
          if ((b >> 10) == low_chunk)
            atomicAdd(&low_s[b & 1023],1u);
          if ((b >> 10) == high_chunk)
            atomicAdd(&high_s[b & 1023],1u);
        }
    }
  __syncthreads();
  // inclusive running count of a chunk's 1024 bins, one per thread
  auto scan=[&](unsigned long long h,unsigned long long prefix) -> unsigned long long
  {
    unsigned long long cum=h;
    for (int off=1; off < 64; off<<=1)
      {
        const unsigned long long up=__shfl_up(cum,off,64);
        if (lane >= off)
          cum+=up;
      }
    __syncthreads();
    if (lane == 63)
      wave_sums[t >> 6]=cum;
    __syncthreads();
    for (int w=0; w < (t >> 6); w++)
      cum+=wave_sums[w];
    return cum+prefix;
  };
  int black=65536;                                 // none: the reference's loop ends with j = 65536
  if (low_chunk >= 0)
    {
      const unsigned long long cum=scan(low_s[t],__shfl(below,low_chunk,64));
      if (t == 0)
        found_s=65536;
      __syncthreads();
      if ((double) cum > black_point)
        atomicMin(&found_s,low_chunk*1024+t);
      __syncthreads();
      black=found_s;
      __syncthreads();
    }
  int white=0;                                     // none
  if (high_chunk >= 0)
    {
      const unsigned long long h=high_s[t];
      const unsigned long long cum=scan(h,__shfl(below,high_chunk,64));
      if (t == 0)
        found_s=0;
      __syncthreads();
      const int j=high_chunk*1024+t;
      if ((j >= 1) && ((double) (total-(cum-h)) > white_limit))
        atomicMax(&found_s,j);
      __syncthreads();
      white=found_s;
    }
  if (t == 0)
    {
      // black[i]=(Quantum) j, enhance.c:1668: a scan that found no bin above black_point ends with
      // j = 65536, which the Q16 build stores as (unsigned short) 65536 = 0
      const int black_i=black == 65536 ? 0 : black;
      levels->black=black_i;
      levels->white=white;
      levels->apply=black_i != white ? 1u : 0u;
    }
}

// The stretch map of enhance.c:1685-1706 evaluated per sample (Q16, the levels from
// stretch_levels_kernel): no 65536-entry table to build, stage or gather from.
template<int C>
__global__ __launch_bounds__(256)
void stretch_apply_kernel(uint16_t *pixels,size_t npixels,const StretchScratch *levels,uint32_t mask)
{
  if ((levels->apply == 0u) || (mask == 0u))
    return;
  const int black_i=levels->black,white_i=levels->white;
  const double black=(double) black_i;
  const double scale=65535.0*perceptible_reciprocal((double) white_i-black);
  auto map=[&](unsigned j) -> unsigned
  {
    const double value=scale*((double) j-black);           // 65535.0*gamma*((double) j-black)
    unsigned v=(unsigned) lut_scale_map_to_quantum(value,1);
    v=(int) j > white_i ? 65535u : v;
    v=(int) j < black_i ? 0u : v;                         // (tested first in enhance.c:1694)
    return v;
  };
  constexpr int BATCH=4;
  if constexpr (C == 4)
    {
      if ((reinterpret_cast<uintptr_t>(pixels) & 15u) == 0)
        {
          // two RGBA pixels per lane and load
          uint4 *pairs=reinterpret_cast<uint4 *>(pixels);
          const size_t npairs=npixels/2;
          const size_t stride=(size_t) gridDim.x*blockDim.x*BATCH;
          for (size_t i0=(size_t) blockIdx.x*blockDim.x*BATCH+threadIdx.x; i0 < npairs; i0+=stride)
            {
              uint4 v[BATCH];
#pragma unroll
              for (int k=0; k < BATCH; k++)
                {
                  const size_t i=i0+(size_t) k*blockDim.x;
                  v[k]=pairs[i < npairs ? i : npairs-1];
                }
#pragma unroll
              for (int k=0; k < BATCH; k++)
                {
                  unsigned words[4]={v[k].x,v[k].y,v[k].z,v[k].w};
#pragma unroll
                  for (int w=0; w < 4; w++)
                    {
                      const int c0=2*(w & 1);
                      unsigned lo=words[w] & 0xffffu,hi=words[w] >> 16;
                      if ((mask >> c0) & 1u)
                        lo=map(lo);
                      if ((mask >> (c0+1)) & 1u)
                        hi=map(hi);
                      words[w]=lo | (hi << 16);
                    }
                  const size_t i=i0+(size_t) k*blockDim.x;
                  if (i < npairs)
                    pairs[i]=make_uint4(words[0],words[1],words[2],words[3]);
                }
            }
          if (((npixels & 1u) != 0) && (blockIdx.x == 0) && (threadIdx.x < 4u) && ((mask >> threadIdx.x) & 1u))
            pixels[(npixels-1)*4+threadIdx.x]=(uint16_t) map(pixels[(npixels-1)*4+threadIdx.x]);
          return;
        }
    }
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < npixels; i+=stride)
    {
      uint16_t q[C];
      load_pixel<uint16_t,C>(pixels+i*C,q);
#pragma unroll
      for (int c=0; c < C; c++)
        if ((mask >> c) & 1u)
          q[c]=(uint16_t) map(q[c]);
      store_pixel<uint16_t,C>(pixels+i*C,q);
    }
}

// *fused stays false when the frame does not qualify (the caller then runs the two operators)
MhStatus launch_lab_fast_with_histogram(const View &img,const MhImage *lab_desc,unsigned long long *hist,
  bool *fused)
{
  *fused=false;
  const size_t n=img.columns*img.rows;
  if ((img.quantum != MH_QUANTUM_U16) || (img.channels != 4) || (precision() != MH_PRECISION_FAST) ||
      (n < ((size_t) 1 << 20)) || (n >= ((size_t) 1 << 31)) ||
      ((reinterpret_cast<uintptr_t>(img.pixels) & 15u) != 0) ||
      (option("MAGICKHIP_NO_FAST_LAB") != nullptr) || (option("MAGICKHIP_NO_PACKED_HISTOGRAM") != nullptr) ||
      (option("MAGICKHIP_NO_LDS_HISTOGRAM") != nullptr) || (option("MAGICKHIP_NO_FUSED_LAB_HISTOGRAM") != nullptr))
    return MH_OK;
  const size_t nblocks=packed_histogram_blocks(img.device,n);
  if (nblocks == 0)
    return MH_OK;
  const IntensityParams ip=intensity_params(lab_desc);
  Temp slabs;
  MH_TRY(slabs.alloc(img.device,nblocks*32768*sizeof(unsigned),img.stream));
  const size_t lds=32768*sizeof(unsigned)+kDecodePieces*sizeof(float2);
  const bool plain=intensity_is_plain_luma(ip,4);
  MH_HIP(hipFuncSetAttribute(plain ? reinterpret_cast<const void *>(&lab_histogram_fast_kernel<true>) :
    reinterpret_cast<const void *>(&lab_histogram_fast_kernel<false>),
    hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  {
    ProfileScope prof("colorspace_histogram",img.stream);
    if (plain)
      hipLaunchKernelGGL(lab_histogram_fast_kernel<true>,dim3((unsigned) nblocks),dim3(1024),lds,img.stream,
        static_cast<uint16_t *>(img.pixels),n,ip,slabs.as<unsigned>(),hist,nullptr);
    else
      hipLaunchKernelGGL(lab_histogram_fast_kernel<false>,dim3((unsigned) nblocks),dim3(1024),lds,img.stream,
        static_cast<uint16_t *>(img.pixels),n,ip,slabs.as<unsigned>(),hist,nullptr);
    hipLaunchKernelGGL(histogram_packed_reduce_kernel,dim3(256),dim3(1024),0,img.stream,
      slabs.as<unsigned>(),(int) nblocks,hist,4);
  }
  MH_HIP(hipGetLastError());
  *fused=true;
  return MH_OK;
}

// FAST sRGB -> Lab followed by ContrastStretchImage on RGBA Q16 in FOUR launches: convert + bin
// (lab_histogram_fast_kernel), bins and levels (stretch_bins_kernel, stretch_levels_kernel), map
// (stretch_apply_kernel) — the general route takes seven (memset, convert + bin, slab reduction, three LUT kernels, LUT apply).
// The results are the general route's, bit for bit: the same bins, the same levels, the same map
// expression.  *fused stays false when the frame does not qualify.
MhStatus launch_lab_fast_contrast_stretch(const View &img,const MhImage *lab_desc,double black_point,
  double white_limit,uint32_t update_mask,bool *fused)
{
  *fused=false;
  const size_t n=img.columns*img.rows;
  if ((img.quantum != MH_QUANTUM_U16) || (img.channels != 4) || (precision() != MH_PRECISION_FAST) ||
      (n < ((size_t) 1 << 20)) || (n >= ((size_t) 1 << 31)) ||
      ((reinterpret_cast<uintptr_t>(img.pixels) & 15u) != 0) ||
      (option("MAGICKHIP_NO_FAST_LAB") != nullptr) || (option("MAGICKHIP_NO_PACKED_HISTOGRAM") != nullptr) ||
      (option("MAGICKHIP_NO_LDS_HISTOGRAM") != nullptr) || (option("MAGICKHIP_NO_FUSED_LAB_HISTOGRAM") != nullptr) ||
      (option("MAGICKHIP_NO_STRETCH_LEVELS") != nullptr))
    return MH_OK;
  const size_t nblocks=packed_histogram_blocks(img.device,n);
  if (nblocks == 0)
    return MH_OK;
  const IntensityParams ip=intensity_params(lab_desc);
  // one allocation: slabs, the reduced bins, the extra-pixel lists, the scratch
  const size_t slab_bytes=nblocks*32768*sizeof(unsigned);
  const size_t bins_bytes=65536*sizeof(unsigned);
  const size_t extras_bytes=((nblocks*kExtraPitch*sizeof(unsigned short))+15u) & ~(size_t) 15u;
  const size_t shares_bytes=256*sizeof(unsigned long long);
  Temp work;
  MH_TRY(work.alloc(img.device,slab_bytes+bins_bytes+extras_bytes+shares_bytes+sizeof(StretchScratch),img.stream));
  unsigned char *at=static_cast<unsigned char *>(work.ptr);
  unsigned *slabs=reinterpret_cast<unsigned *>(at);
  unsigned *bins=reinterpret_cast<unsigned *>(at+slab_bytes);
  unsigned short *extras=reinterpret_cast<unsigned short *>(at+slab_bytes+bins_bytes);
  unsigned long long *shares=reinterpret_cast<unsigned long long *>(at+slab_bytes+bins_bytes+extras_bytes);
  StretchScratch *scratch=reinterpret_cast<StretchScratch *>(at+slab_bytes+bins_bytes+extras_bytes+shares_bytes);
  const size_t lds=32768*sizeof(unsigned)+kDecodePieces*sizeof(float2);
  const bool plain=intensity_is_plain_luma(ip,4);
  static bool configured[2][64]={};
  if ((img.device < 0) || (img.device >= 64) || !configured[plain ? 1 : 0][img.device])
    {
      MH_HIP(hipFuncSetAttribute(plain ? reinterpret_cast<const void *>(&lab_histogram_fast_kernel<true>) :
        reinterpret_cast<const void *>(&lab_histogram_fast_kernel<false>),
        hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
      if ((img.device >= 0) && (img.device < 64))
        configured[plain ? 1 : 0][img.device]=true;
    }
  {
    ProfileScope prof("colorspace_histogram",img.stream);
    if (plain)
      hipLaunchKernelGGL(lab_histogram_fast_kernel<true>,dim3((unsigned) nblocks),dim3(1024),lds,img.stream,
        static_cast<uint16_t *>(img.pixels),n,ip,slabs,nullptr,extras);
    else
      hipLaunchKernelGGL(lab_histogram_fast_kernel<false>,dim3((unsigned) nblocks),dim3(1024),lds,img.stream,
        static_cast<uint16_t *>(img.pixels),n,ip,slabs,nullptr,extras);
  }
  {
    ProfileScope prof("build_lut",img.stream);
    hipLaunchKernelGGL(stretch_bins_kernel,dim3(256),dim3(1024),0,img.stream,slabs,(int) nblocks,bins,shares);
    hipLaunchKernelGGL(stretch_levels_kernel,dim3(1),dim3(1024),0,img.stream,bins,shares,extras,(int) nblocks,
      scratch,black_point,white_limit);
  }
  {
    ProfileScope prof("apply_lut",img.stream);
    hipLaunchKernelGGL(stretch_apply_kernel<4>,dim3(stream_grid((n/2+3)/4)),dim3(256),0,img.stream,
      static_cast<uint16_t *>(img.pixels),n,scratch,update_mask);
  }
  MH_HIP(hipGetLastError());
  *fused=true;
  return MH_OK;
}

template<typename Q,int C>
static MhStatus histogram_typed(const View &src,int mode,const IntensityParams &ip,
  unsigned long long *hist)
{
  const size_t n=src.columns*src.rows;
  // large frames in intensity mode: the LDS-privatised kernel (a frame below ~1 Mpixel
  // does not amortise the 2 x 256 x 128 KB slab traffic)
  if ((mode != 0) && (n >= ((size_t) 1 << 20)) && (option("MAGICKHIP_NO_LDS_HISTOGRAM") == nullptr))
    {
      // one pass with 16-bit counters, Q16 and float Quantum alike (the bin of a float sample is
      // ScaleQuantumToMap's)
      if ((option("MAGICKHIP_NO_PACKED_HISTOGRAM") == nullptr) && (n < ((size_t) 1 << 31)))
        return histogram_intensity_packed<Q,C>(src,ip,hist);
      return histogram_intensity_lds<Q,C>(src,ip,hist);
    }
  Temp before;
  if ((mode != 0) && (C > 1))
    {
      MH_TRY(before.alloc(src.device,65536*sizeof(unsigned long long),src.stream));
      hipLaunchKernelGGL(histogram_snapshot_kernel,dim3(256),dim3(256),0,src.stream,hist,
        before.as<unsigned long long>(),C);
    }
  {
    ProfileScope prof("histogram",src.stream);
    hipLaunchKernelGGL((histogram_kernel<Q,C>),dim3(stream_grid(n)),dim3(256),0,src.stream,
      static_cast<const Q *>(src.pixels),n,mode,ip,hist);
  }
  if ((mode != 0) && (C > 1))
    hipLaunchKernelGGL(histogram_replicate_kernel,dim3(256),dim3(256),0,src.stream,hist,
      before.as<unsigned long long>(),C);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

MhStatus launch_histogram(const View &src,int intensity_mode,const MhImage *desc,
  unsigned long long *hist)
{
  IntensityParams ip=intensity_params(desc);
#define MH_CASE(QT) \
  switch (src.channels) { \
    case 1: return histogram_typed<QT,1>(src,intensity_mode,ip,hist); \
    case 2: return histogram_typed<QT,2>(src,intensity_mode,ip,hist); \
    case 3: return histogram_typed<QT,3>(src,intensity_mode,ip,hist); \
    default: return histogram_typed<QT,4>(src,intensity_mode,ip,hist); }
  if (src.quantum == MH_QUANTUM_U16)
    { MH_CASE(uint16_t) }
  MH_CASE(float)
#undef MH_CASE
}

// ---------------------------------------------------------------- LUT build
// The 65536-entry scans that turn a histogram into a LUT (enhance.c:1652-1706 contrast
// stretch, :2138-2169 equalize) on the device, so the operator never waits for a histogram
// download: one workgroup per channel, 64 consecutive bins per thread, counts summed as
// 64-bit integers (the reference adds them as doubles — exact below 2^53, so the order of
// the additions does not matter) and every floating-point expression written as there.
struct LutBuildArgs
{
  const unsigned long long *hist;     // [65536][channels]
  int channels;
  int equalize;                       // 0: contrast stretch, 1: equalize
  double black_point,white_limit;     // stretch: counts; white_limit = columns*rows-white_point
  int is_u16;
  void *lut;                          // Quantum-typed [65536][channels]
  uint32_t *mask;                     // out: bit c set when channel c has black != white
  const unsigned int *colour_flag;    // optional: 0 => image is gray => leave every channel alone
  uint32_t *cdf;                      // optional (equalize): the running counts of channel cdf_column
  int cdf_column;
};

__device__ __forceinline__ double lut_scale_map_to_quantum(double value,int is_u16)
{
  // ScaleMapToQuantum, quantum-private.h:465-476, as the Quantum it is stored in
  if (value <= 0.0)
    return 0.0;
  if (value >= 65535.0)
    return 65535.0;
  if (is_u16)
    return (double) (unsigned short) (value+0.5);
  return (double) (float) value;
}

// Three small grids of (channels x 64) workgroups, one bin per thread, coalesced:
//   lut_chunk_sums_kernel   sum of each 1024-bin chunk
//   lut_scan_kernel         running count of every bin (chunk prefix + workgroup scan);
//                           equalize writes its map here, stretch records the black / white
//                           levels (first bin from below / from above whose running count
//                           passes the threshold) with one atomic per workgroup
//   lut_stretch_map_kernel  stretch map from the two levels
constexpr int kLutChunks=64;

struct LutScratch
{
  unsigned long long chunk_sum[MH_MAX_CHANNELS][kLutChunks];
  int black[MH_MAX_CHANNELS],white[MH_MAX_CHANNELS];
};

__device__ __forceinline__ bool lut_image_is_gray(const LutBuildArgs &a)
{
  return (a.colour_flag != nullptr) && (*a.colour_flag == 0);
}

__device__ __forceinline__ void lut_store(const LutBuildArgs &a,int j,int c,double v)
{
  if (a.is_u16)
    static_cast<unsigned short *>(a.lut)[(size_t) j*a.channels+c]=(unsigned short) v;
  else
    static_cast<float *>(a.lut)[(size_t) j*a.channels+c]=(float) v;
}

// workgroup-wide sum of one 64-bit value per thread (1024 threads); result in every thread
__device__ __forceinline__ unsigned long long lut_block_sum(unsigned long long v,unsigned long long *shared16)
{
  for (int off=32; off > 0; off>>=1)
    v+=__shfl_xor(v,off,64);
  if ((threadIdx.x & 63) == 0)
    shared16[threadIdx.x >> 6]=v;
  __syncthreads();
  unsigned long long total=0;
  for (int w=0; w < 16; w++)
    total+=shared16[w];
  __syncthreads();
  return total;
}

__global__ __launch_bounds__(1024)
void lut_chunk_sums_kernel(LutBuildArgs a,LutScratch *scratch)
{
  __shared__ unsigned long long wave_sums[16];
  const int c=(int) blockIdx.x,chunk=(int) blockIdx.y,j=chunk*1024+(int) threadIdx.x;
  unsigned long long total=lut_block_sum(a.hist[(size_t) j*a.channels+c],wave_sums);
  if (threadIdx.x == 0)
    {
      scratch->chunk_sum[c][chunk]=total;
      if (chunk == 0)
        {
          scratch->black[c]=65536;
          scratch->white[c]=0;
          // this channel's bit of the mask and, once, the bits above the channels (the later
          // kernels of the stream set the bit again; a memset would be one more launch)
          atomicAnd(a.mask,~(1u << c));
          if (c == 0)
            atomicAnd(a.mask,a.channels >= 32 ? 0xffffffffu : (1u << a.channels)-1u);
        }
    }
}

__global__ __launch_bounds__(1024)
void lut_scan_kernel(LutBuildArgs a,LutScratch *scratch)
{
  __shared__ unsigned long long wave_sums[16];
  __shared__ unsigned long long prefix_s,total_s;
  __shared__ int black_s,white_s;
  if (lut_image_is_gray(a))
    return;
  const int c=(int) blockIdx.x,chunk=(int) blockIdx.y,t=(int) threadIdx.x,j=chunk*1024+t;
  if (t < 64)
    {
      // wave 0: counts below this chunk, and of the whole channel
      unsigned long long v=scratch->chunk_sum[c][t];
      unsigned long long below=t < chunk ? v : 0ull;
      for (int off=32; off > 0; off>>=1)
        {
          v+=__shfl_xor(v,off,64);
          below+=__shfl_xor(below,off,64);
        }
      if (t == 0)
        {
          prefix_s=below;
          total_s=v;
          black_s=65536;
          white_s=0;
        }
    }
  const unsigned long long h=a.hist[(size_t) j*a.channels+c];
  // inclusive scan of the 1024 bins: within the wave, then across the 16 waves
  unsigned long long cum=h;
  const int lane=t & 63;
  for (int off=1; off < 64; off<<=1)
    {
      unsigned long long up=__shfl_up(cum,off,64);
      if (lane >= off)
        cum+=up;
    }
  if (lane == 63)
    wave_sums[t >> 6]=cum;
  __syncthreads();
  for (int w=0; w < (t >> 6); w++)
    cum+=wave_sums[w];
  cum+=prefix_s;
  const unsigned long long total=total_s;
  if (a.equalize != 0)
    {
      // integrate, enhance.c:2138-2152; map, :2162-2168
      const double black=(double) a.hist[c],white=(double) total;
      if ((a.cdf != nullptr) && (c == a.cdf_column))
        a.cdf[j]=(total >> 32) != 0ull ? 0xffffffffu : (uint32_t) cum;
      if (black == white)
        return;
      lut_store(a,j,c,lut_scale_map_to_quantum(
        (double) ((65535.0*((double) cum-black))/(white-black)),a.is_u16));
      if (j == 0)
        atomicOr(a.mask,1u<<c);
      return;
    }
  // black / white levels, enhance.c:1652-1678
  const unsigned long long from_top=total-(cum-h);          // bins j..65535
  if ((double) cum > a.black_point)
    atomicMin(&black_s,j);
  if ((j >= 1) && ((double) from_top > a.white_limit))
    atomicMax(&white_s,j);
  __syncthreads();
  if (t == 0)
    {
      if (black_s < 65536)
        atomicMin(&scratch->black[c],black_s);
      if (white_s > 0)
        atomicMax(&scratch->white[c],white_s);
    }
}

__global__ __launch_bounds__(1024)
void lut_stretch_map_kernel(LutBuildArgs a,const LutScratch *scratch)
{
  if (lut_image_is_gray(a))
    return;
  const int c=(int) blockIdx.x,j=(int) blockIdx.y*1024+(int) threadIdx.x;
  // black[i]=(Quantum) j, enhance.c:1668: a scan that found no bin above black_point ends with
  // j = 65536, which the Q16 build stores as (unsigned short) 65536 = 0
  const int black_i=((a.is_u16 != 0) && (scratch->black[c] == 65536)) ? 0 : scratch->black[c];
  const int white_i=scratch->white[c];
  const double black=(double) black_i,white=(double) white_i;
  // stretch map, enhance.c:1685-1706
  const double gamma=perceptible_reciprocal(white-black);
  double v=0.0;
  if (j < black_i)
    v=0.0;
  else if (j > white_i)
    v=65535.0;
  else if (black != white)
    v=lut_scale_map_to_quantum((double) (65535.0*gamma*((double) j-black)),a.is_u16);
  lut_store(a,j,c,v);
  if ((j == 0) && (black != white))
    atomicOr(a.mask,1u<<c);
}

__global__ __launch_bounds__(256)
void table_add_kernel(unsigned long long *dst,const unsigned long long *src,size_t count)
{
  for (size_t i=(size_t) blockIdx.x*256u+threadIdx.x; i < count; i+=(size_t) gridDim.x*256u)
    dst[i]+=src[i];
}

MhStatus launch_table_add(unsigned long long *dst,const unsigned long long *src,size_t count,
  int device,hipStream_t stream)
{
  DeviceGuard guard;
  MH_HIP(guard.enter(device));
  const unsigned blocks=(unsigned) ((count+255)/256 < 1024 ? (count+255)/256 : 1024);
  hipLaunchKernelGGL(table_add_kernel,dim3(blocks == 0 ? 1u : blocks),dim3(256),0,stream,dst,src,count);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

MhStatus launch_build_lut(const View &img,const unsigned long long *hist,bool equalize,
  double black_point,double white_limit,void *lut,uint32_t *mask,const unsigned int *colour_flag,
  uint32_t *cdf,int cdf_column)
{
  LutBuildArgs a;
  a.cdf=equalize ? cdf : nullptr;
  a.cdf_column=cdf_column;
  a.hist=hist;
  a.channels=img.channels;
  a.equalize=equalize ? 1 : 0;
  a.black_point=black_point;
  a.white_limit=white_limit;
  a.is_u16=img.quantum == MH_QUANTUM_U16 ? 1 : 0;
  a.lut=lut;
  a.mask=mask;
  a.colour_flag=colour_flag;
  Temp scratch;
  MH_TRY(scratch.alloc(img.device,sizeof(LutScratch),img.stream));
  const dim3 grid((unsigned) img.channels,kLutChunks);
  ProfileScope prof("build_lut",img.stream);
  hipLaunchKernelGGL(lut_chunk_sums_kernel,grid,dim3(1024),0,img.stream,a,scratch.as<LutScratch>());
  hipLaunchKernelGGL(lut_scan_kernel,grid,dim3(1024),0,img.stream,a,scratch.as<LutScratch>());
  if (!equalize)
    hipLaunchKernelGGL(lut_stretch_map_kernel,grid,dim3(1024),0,img.stream,a,scratch.as<LutScratch>());
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// ContrastStretchImage without the 65536 x channels table: the two scan kernels above leave the
// black and the white level of every channel in the scratch block, and this kernel evaluates
// enhance.c:1685-1706's map per sample (as lut_stretch_map_kernel would have tabulated it — the
// same expressions, so the same bits) for Q16 and float Quantum, any channel count, intensity or
// per-channel binning.  A float frame gathered four floats per pixel from a 1 MB table before
// (1.4 ms per 8192^2 RGBA frame); a Q16 frame staged a 128 KB column in LDS per workgroup.
template<typename Q,int C>
__global__ __launch_bounds__(256)
void stretch_apply_levels_kernel(Q *pixels,size_t npixels,const LutScratch *scratch,uint32_t mask,
  const unsigned int *colour_flag)
{
  if ((colour_flag != nullptr) && (*colour_flag == 0))
    return;                                      // the image is gray: every channel is left alone
  constexpr int is_u16=sizeof(Q) == 2 ? 1 : 0;
  int black_i[C],white_i[C];
  double black[C],scale[C];
  uint32_t apply=0;
#pragma unroll
  for (int c=0; c < C; c++)
    {
      // black[i]=(Quantum) j, enhance.c:1668: a scan that found no bin above black_point ends with
      // j = 65536, which the Q16 build stores as (unsigned short) 65536 = 0
      black_i[c]=((is_u16 != 0) && (scratch->black[c] == 65536)) ? 0 : scratch->black[c];
      white_i[c]=scratch->white[c];
      black[c]=(double) black_i[c];
      scale[c]=65535.0*perceptible_reciprocal((double) white_i[c]-black[c]);
      if ((black_i[c] != white_i[c]) && (((mask >> c) & 1u) != 0u))
        apply|=1u << c;
    }
  if (apply == 0u)
    return;
  constexpr int BATCH=4;
  const size_t stride=(size_t) gridDim.x*blockDim.x*BATCH;
  for (size_t i0=(size_t) blockIdx.x*blockDim.x*BATCH+threadIdx.x; i0 < npixels; i0+=stride)
    {
      Q q[BATCH][C];
#pragma unroll
      for (int k=0; k < BATCH; k++)
        {
          const size_t i=i0+(size_t) k*blockDim.x;
          load_pixel<Q,C>(pixels+(i < npixels ? i : npixels-1)*C,q[k]);
        }
#pragma unroll
      for (int k=0; k < BATCH; k++)
        {
          const size_t i=i0+(size_t) k*blockDim.x;
          if (i >= npixels)
            continue;
#pragma unroll
          for (int c=0; c < C; c++)
            if ((apply >> c) & 1u)
              {
                const int j=(int) QuantumOps<Q>::map_index(q[k][c]);
                double v=lut_scale_map_to_quantum(scale[c]*((double) j-black[c]),is_u16);   // 65535.0*gamma*((double) j-black)
                v=j > white_i[c] ? 65535.0 : v;
                v=j < black_i[c] ? 0.0 : v;        // (tested first in enhance.c:1694)
                q[k][c]=(Q) v;
              }
          store_pixel<Q,C>(pixels+i*C,q[k]);
        }
    }
}

// levels of every channel from the table [65536][channels], then the map: three launches
MhStatus launch_stretch_levels_apply(const View &img,const unsigned long long *hist,double black_point,
  double white_limit,uint32_t update_mask,const unsigned int *colour_flag)
{
  LutBuildArgs a;
  a.hist=hist;
  a.channels=img.channels;
  a.equalize=0;
  a.cdf=nullptr;
  a.cdf_column=0;
  a.black_point=black_point;
  a.white_limit=white_limit;
  a.is_u16=img.quantum == MH_QUANTUM_U16 ? 1 : 0;
  a.lut=nullptr;
  a.colour_flag=colour_flag;
  Temp scratch;
  MH_TRY(scratch.alloc(img.device,sizeof(LutScratch)+sizeof(uint32_t),img.stream));
  LutScratch *levels=scratch.as<LutScratch>();
  a.mask=reinterpret_cast<uint32_t *>(levels+1);   // (the scan kernels keep their mask word)
  const dim3 grid((unsigned) img.channels,kLutChunks);
  {
    ProfileScope prof("build_lut",img.stream);
    hipLaunchKernelGGL(lut_chunk_sums_kernel,grid,dim3(1024),0,img.stream,a,levels);
    hipLaunchKernelGGL(lut_scan_kernel,grid,dim3(1024),0,img.stream,a,levels);
  }
  const size_t n=img.columns*img.rows;
  {
    ProfileScope prof("apply_lut",img.stream);
    const dim3 apply_grid(stream_grid((n+3)/4)),block(256);
#define MH_CASE(QT) \
    switch (img.channels) { \
      case 1: hipLaunchKernelGGL((stretch_apply_levels_kernel<QT,1>),apply_grid,block,0,img.stream,static_cast<QT *>(img.pixels),n,levels,update_mask,colour_flag); break; \
      case 2: hipLaunchKernelGGL((stretch_apply_levels_kernel<QT,2>),apply_grid,block,0,img.stream,static_cast<QT *>(img.pixels),n,levels,update_mask,colour_flag); break; \
      case 3: hipLaunchKernelGGL((stretch_apply_levels_kernel<QT,3>),apply_grid,block,0,img.stream,static_cast<QT *>(img.pixels),n,levels,update_mask,colour_flag); break; \
      default: hipLaunchKernelGGL((stretch_apply_levels_kernel<QT,4>),apply_grid,block,0,img.stream,static_cast<QT *>(img.pixels),n,levels,update_mask,colour_flag); break; }
    if (img.quantum == MH_QUANTUM_U16)
      { MH_CASE(uint16_t) }
    else
      { MH_CASE(float) }
#undef MH_CASE
  }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// ---------------------------------------------------------------- LUT apply
template<typename Q,int C>
__global__ __launch_bounds__(256)
void apply_lut_kernel(Q *pixels,size_t npixels,const Q *lut,uint32_t mask,
  const uint32_t *device_mask,int lut_stride,int column_step)
{
  if (device_mask != nullptr)
    mask&=*device_mask;                 // uniform: written by build_lut_kernel earlier in the stream
  if (mask == 0)
    return;
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < npixels; i+=stride)
    {
      Q q[C];
      load_pixel<Q,C>(pixels+i*C,q);
#pragma unroll
      for (int c=0; c < C; c++)
        if ((mask >> c) & 1u)
          q[c]=lut[(size_t) QuantumOps<Q>::map_index(q[c])*lut_stride+c*column_step];
      store_pixel<Q,C>(pixels+i*C,q);
    }
}

template<typename Q,int C>
static MhStatus apply_lut_typed(const View &img,const void *lut,uint32_t mask,
  const uint32_t *device_mask,bool single_column=false,const char *label="apply_lut")
{
  const size_t n=img.columns*img.rows;
  ProfileScope prof(label,img.stream);
  hipLaunchKernelGGL((apply_lut_kernel<Q,C>),dim3(stream_grid(n)),dim3(256),0,img.stream,
    static_cast<Q *>(img.pixels),n,static_cast<const Q *>(lut),mask,device_mask,
    single_column ? 1 : C,single_column ? 0 : 1);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// Q16, every masked channel maps through the SAME 65536-entry column (the intensity-
// binned operators give all channels one histogram, hence one LUT: enhance.c:1637-1643):
// the 128 KB column lives in LDS of a persistent workgroup per CU and the 4 lookups per
// pixel are LDS reads instead of scattered 2-byte global gathers.
template<int C>
__global__ __launch_bounds__(1024)
void apply_lut_shared_kernel(uint16_t *pixels,size_t npixels,const uint16_t *lut,int column,uint32_t mask,
  const uint32_t *device_mask,int lut_stride)
{
  if (device_mask != nullptr)
    mask&=*device_mask;
  if (mask == 0)
    return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint16_t *table=reinterpret_cast<uint16_t *>(smem_raw);
  for (int i=(int) threadIdx.x; i < 65536; i+=1024)
    table[i]=lut[(size_t) i*lut_stride+column];
  __syncthreads();
  constexpr int BATCH=4;
  const size_t stride=(size_t) gridDim.x*1024*BATCH;
  for (size_t i0=(size_t) blockIdx.x*1024*BATCH+threadIdx.x; i0 < npixels; i0+=stride)
    {
      uint16_t q[BATCH][C];
#pragma unroll
      for (int k=0; k < BATCH; k++)
        {
          size_t i=i0+(size_t) 1024*k;
          load_pixel<uint16_t,C>(pixels+(i < npixels ? i : npixels-1)*C,q[k]);
        }
#pragma unroll
      for (int k=0; k < BATCH; k++)
        {
          size_t i=i0+(size_t) 1024*k;
          if (i < npixels)
            {
#pragma unroll
              for (int c=0; c < C; c++)
                if ((mask >> c) & 1u)
                  q[k][c]=table[q[k][c]];
              store_pixel<uint16_t,C>(pixels+i*C,q[k]);
            }
        }
    }
}

template<int C>
static MhStatus apply_lut_shared(const View &img,const void *lut,int column,uint32_t mask,
  const uint32_t *device_mask,bool single_column=false,const char *label="apply_lut")
{
  const size_t n=img.columns*img.rows;
  const int cus=compute_units(img.device);
  const size_t lds=65536*sizeof(uint16_t);
  MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&apply_lut_shared_kernel<C>),
    hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds));
  ProfileScope prof(label,img.stream);
  hipLaunchKernelGGL((apply_lut_shared_kernel<C>),dim3(cus),dim3(1024),lds,img.stream,
    static_cast<uint16_t *>(img.pixels),n,static_cast<const uint16_t *>(lut),single_column ? 0 : column,mask,
    device_mask,single_column ? 1 : C);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// EqualizeImage on a float frame whose channels share one histogram (intensity binning: the
// default, synchronised channels).  Its map is  ScaleMapToQuantum(65535*(cum[j]-black)/(white-black))
// (enhance.c:2162-2168) with INTEGER running counts cum[j]: 65536 floats of table (256 KB, a
// gather through L2 per sample: 0.32 ms per 4096^2 RGBA frame) do not fit the LDS, the counts do —
// a uint32 every 16 bins (16 KB) and a uint16 offset per bin (128 KB) — and the map is three
// fp64 operations on them, the table's own, so the same bits.  A 16-bin group that holds more than
// 65535 pixels (a frame with a flat background) or a frame of 2^32 pixels sends the workgroup to
// the tabulated map.
template<int C>
__global__ __launch_bounds__(1024)
void equalize_cdf_apply_kernel(float *pixels,size_t npixels,const uint32_t *cdf,const float *lut,int column,
  uint32_t mask,const uint32_t *device_mask)
{
  if (device_mask != nullptr)
    mask&=*device_mask;                 // (zero: gray frame, or black == white)
  if (mask == 0)
    return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint32_t *base=reinterpret_cast<uint32_t *>(smem_raw);                     // [4096]
  uint16_t *offset=reinterpret_cast<uint16_t *>(smem_raw+4096*sizeof(uint32_t));   // [65536]
  __shared__ int overflow;
  if (threadIdx.x == 0)
    overflow=cdf[65535] == 0xffffffffu ? 1 : 0;
  __syncthreads();
  for (int j=(int) threadIdx.x; j < 65536; j+=1024)
    {
      const uint32_t count=cdf[j],first=cdf[j & ~15];
      const uint32_t d=count-first;
      if (d > 65535u)
        overflow=1;                     // (every writer writes the same 1)
      offset[j]=(uint16_t) d;
      if ((j & 15) == 0)
        base[j >> 4]=count;
    }
  __syncthreads();
  const bool tabulated=overflow != 0;
  const double black=(double) cdf[0],white=(double) cdf[65535];
  constexpr int BATCH=4;
  const size_t stride=(size_t) gridDim.x*1024*BATCH;
  for (size_t i0=(size_t) blockIdx.x*1024*BATCH+threadIdx.x; i0 < npixels; i0+=stride)
    {
      float q[BATCH][C];
#pragma unroll
      for (int k=0; k < BATCH; k++)
        {
          const size_t i=i0+(size_t) 1024*k;
          load_pixel<float,C>(pixels+(i < npixels ? i : npixels-1)*C,q[k]);
        }
#pragma unroll
      for (int k=0; k < BATCH; k++)
        {
          const size_t i=i0+(size_t) 1024*k;
          if (i < npixels)
            {
#pragma unroll
              for (int c=0; c < C; c++)
                if ((mask >> c) & 1u)
                  {
                    const unsigned j=QuantumOps<float>::map_index(q[k][c]);
                    if (tabulated)
                      {
                        q[k][c]=lut[(size_t) j*C+column];
                        continue;
                      }
                    const double cum=(double) (base[j >> 4]+(uint32_t) offset[j]);
                    const double value=(65535.0*(cum-black))/(white-black);
                    q[k][c]=value <= 0.0 ? 0.0f : (value >= 65535.0 ? 65535.0f : (float) value);
                  }
              store_pixel<float,C>(pixels+i*C,q[k]);
            }
        }
    }
}

MhStatus launch_equalize_cdf_apply(const View &img,const uint32_t *cdf,const void *lut,uint32_t apply_mask,
  const Roles &roles,int shared_column,const uint32_t *device_mask)
{
  const uint32_t mask=apply_mask & roles.update_mask;
  const size_t n=img.columns*img.rows;
  const int cus=compute_units(img.device);
  const size_t lds=4096*sizeof(uint32_t)+65536*sizeof(uint16_t);
  float *pixels=static_cast<float *>(img.pixels);
  const float *table=static_cast<const float *>(lut);
#define MH_CASE(CH) \
  { \
    MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&equalize_cdf_apply_kernel<CH>), \
      hipFuncAttributeMaxDynamicSharedMemorySize,(int) lds)); \
    ProfileScope prof("apply_lut",img.stream); \
    hipLaunchKernelGGL((equalize_cdf_apply_kernel<CH>),dim3(cus),dim3(1024),lds,img.stream,pixels,n,cdf,table, \
      shared_column,mask,device_mask); \
  }
  switch (img.channels)
  {
    case 1: MH_CASE(1) break;
    case 2: MH_CASE(2) break;
    case 3: MH_CASE(3) break;
    case 4: MH_CASE(4) break;
    default: return fail(MH_UNSUPPORTED,"%d channels",img.channels);
  }
#undef MH_CASE
  MH_HIP(hipGetLastError());
  return MH_OK;
}

MhStatus launch_apply_lut(const View &img,const void *lut,uint32_t apply_mask,const Roles &roles,
  int shared_column,const uint32_t *device_mask)
{
  uint32_t mask=apply_mask & roles.update_mask;
  if ((shared_column >= 0) && (img.quantum == MH_QUANTUM_U16) &&
      (img.columns*img.rows >= ((size_t) 1 << 20)))
    switch (img.channels)
    {
      case 1: return apply_lut_shared<1>(img,lut,shared_column,mask,device_mask);
      case 2: return apply_lut_shared<2>(img,lut,shared_column,mask,device_mask);
      case 3: return apply_lut_shared<3>(img,lut,shared_column,mask,device_mask);
      default: return apply_lut_shared<4>(img,lut,shared_column,mask,device_mask);
    }
#define MH_CASE(QT) \
  switch (img.channels) { \
    case 1: return apply_lut_typed<QT,1>(img,lut,mask,device_mask); \
    case 2: return apply_lut_typed<QT,2>(img,lut,mask,device_mask); \
    case 3: return apply_lut_typed<QT,3>(img,lut,mask,device_mask); \
    default: return apply_lut_typed<QT,4>(img,lut,mask,device_mask); }
  if (img.quantum == MH_QUANTUM_U16)
    { MH_CASE(uint16_t) }
  MH_CASE(float)
#undef MH_CASE
}


// one 65536-entry Quantum column applied to the masked channels of a Q16 image
static MhStatus apply_q16_column(const View &img,const uint16_t *column,uint32_t mask,const char *label)
{
  if (img.columns*img.rows >= ((size_t) 1 << 20))
    return img.channels == 3 ? apply_lut_shared<3>(img,column,0,mask,nullptr,true,label) :
      apply_lut_shared<4>(img,column,0,mask,nullptr,true,label);
  return img.channels == 3 ? apply_lut_typed<uint16_t,3>(img,column,mask,nullptr,true,label) :
    apply_lut_typed<uint16_t,4>(img,column,mask,nullptr,true,label);
}

// ---------------------------------------------------------------- CompositeImage
// The two mathematical compositions MorphologyApply uses as post-steps (morphology.c:
// 3986-4044): Difference (EdgeIn/EdgeOut/Edge/TopHat/BottomHat) and Lighten (union of the
// HitAndMiss results of a kernel list).  Canvas and source have the same geometry, offset
// 0,0, default artifacts (compose:sync and compose:clamp on): composite.c:2396-2428 (alpha),
// :2523-2711 (alpha channel), :2716-2751 (Sca/Dca/gamma), :2922-2933 (Difference),
// :3110-3124 (Lighten), ClampPixel pixel-accessor.h:35-46.
enum CompositeKind { COMPOSITE_DIFFERENCE=0,COMPOSITE_LIGHTEN=1,COMPOSITE_DARKEN=2,COMPOSITE_PLUS=3,COMPOSITE_MULTIPLY=4,
  COMPOSITE_SCREEN=5,COMPOSITE_EXCLUSION=6,COMPOSITE_MINUS_SRC=7,COMPOSITE_MINUS_DST=8,COMPOSITE_LINEAR_DODGE=9,
  COMPOSITE_OVER=10,COMPOSITE_DST_OVER=11 };

template<typename Q> static __device__ __forceinline__ Q clamp_pixel(double pixel);
template<> __device__ __forceinline__ uint16_t clamp_pixel<uint16_t>(double pixel)
{
  if (pixel < 0.0)
    return 0;
  if (pixel >= kQR)
    return 65535;
  return (uint16_t) (pixel+0.5);
}
template<> __device__ __forceinline__ float clamp_pixel<float>(double pixel)
{
  if (pixel < 0.0)
    return 0.0f;
  if (pixel >= kQR)
    return 65535.0f;
  return (float) pixel;
}

template<typename Q,int C,int OP>
__global__ __launch_bounds__(256)
void composite_kernel(Q *__restrict__ canvas,const Q *__restrict__ source,size_t npixels,int alpha_index,
  uint32_t update_mask,uint32_t copy_mask)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < npixels; i+=stride)
    {
      Q s[C],d[C];
      load_pixel<Q,C>(source+i*C,s);
      load_pixel<Q,C>(canvas+i*C,d);
      // GetPixelAlpha gives OpaqueAlpha for an image without an alpha channel
      const double Sa=kQS*(alpha_index >= 0 ? (double) s[alpha_index < 0 ? 0 : alpha_index] : kQR);
      const double Da=kQS*(alpha_index >= 0 ? (double) d[alpha_index < 0 ? 0 : alpha_index] : kQR);
      // Plus: RoundToUnity(source_dissolve*Sa+canvas_dissolve*Da), both 1 (composite.c:2462-2467);
      // the others RoundToUnity(Sa+Da-Sa*Da) (:2398-2426)
      double alpha=OP == COMPOSITE_PLUS ? 1.0*Sa+1.0*Da : Sa+Da-Sa*Da;
      if ((OP != COMPOSITE_OVER) && (OP != COMPOSITE_DST_OVER))      // (Over / DstOver: not rounded to unity, :2443-2449)
        alpha=alpha < 0.0 ? 0.0 : (alpha > 1.0 ? 1.0 : alpha);
      // composite.c:2737-2751
      const double gamma=perceptible_reciprocal(((OP == COMPOSITE_LIGHTEN) || (OP == COMPOSITE_DARKEN)) ? 1.0-alpha : alpha);
#pragma unroll
      for (int c=0; c < C; c++)
        {
          // CompositeOverImage sets the alpha channel whatever its trait says, Update or Copy
          // (composite.c:1096-1104): `-channel RGB` does not keep the merged alpha out
          if ((c == alpha_index) && ((OP == COMPOSITE_OVER) || (((update_mask >> c) & 1u) != 0)))
            {
              // composite.c:2580-2712 (Multiply with synchronised channels: the default case)
              const double pixel=OP == COMPOSITE_DIFFERENCE ? kQR*fabs(Sa-Da) : kQR*alpha;
              d[c]=clamp_pixel<Q>(pixel);
              continue;
            }
          if (((copy_mask >> c) & 1u) != 0)
            {
              // ClampToQuantum(Dc) = Dc; Over runs CompositeOverImage (composite.c:1489-1495), whose
              // copy channels take the SOURCE (:1115-1121)
              if (OP == COMPOSITE_OVER)
                d[c]=s[c];
              continue;
            }
          const double Sc=(double) s[c],Dc=(double) d[c];
          const double Sca=kQS*Sa*Sc,Dca=kQS*Da*Dc;
          double pixel;
          if (OP == COMPOSITE_DIFFERENCE)
            {
              const double a=Sca*Da,b=Dca*Sa;
              pixel=kQR*gamma*(Sca+Dca-2.0*(a < b ? a : b));
            }
          else if (OP == COMPOSITE_PLUS)
            pixel=kQR*(Sca+Dca);                                 // composite.c:3369-3378
          else if (OP == COMPOSITE_MULTIPLY)
            pixel=kQR*gamma*(Sca*Dca+Sca*(1.0-Da)+Dca*(1.0-Sa)); // :3288-3299
          else if (OP == COMPOSITE_SCREEN)
            pixel=kQR*gamma*(Sca+Dca-Sca*Dca);                   // :3447-3461
          else if (OP == COMPOSITE_EXCLUSION)
            pixel=kQR*gamma*(Sca*Da+Dca*Sa-2.0*Sca*Dca+Sca*(1.0-Da)+Dca*(1.0-Sa));   // :3011-3016
          else if (OP == COMPOSITE_MINUS_SRC)
            pixel=gamma*(Da*Dc+Sa*Sc-2.0*Sa*Sc*Da);              // :3211-3224
          else if (OP == COMPOSITE_MINUS_DST)
            pixel=gamma*(Sa*Sc+Da*Dc-2.0*Da*Dc*Sa);              // :3201-3210
          else if (OP == COMPOSITE_LINEAR_DODGE)
            pixel=gamma*(Sa*Sc+Da*Dc);                           // :3094-3098
          else if (OP == COMPOSITE_OVER)
            pixel=kQR*gamma*(Sca+Dca*(1.0-Sa));                  // :3310-3315
          else if (OP == COMPOSITE_DST_OVER)
            pixel=kQR*gamma*(Dca+Sca*(1.0-Da));                  // :3006-3010
          else if (OP == COMPOSITE_DARKEN ? (Sca*Da) < (Dca*Sa) : (Sca*Da) > (Dca*Sa))
            pixel=kQR*(Sca+Dca*(1.0-Sa));                        // :2892-2910, :3110-3124
          else
            pixel=kQR*(Dca+Sca*(1.0-Da));
          d[c]=clamp_pixel<Q>(pixel);
        }
      store_pixel<Q,C>(canvas+i*C,d);
    }
}

template<typename Q,int C>
static MhStatus composite_typed(const View &canvas,const View &source,int kind,const Roles &roles)
{
  const size_t n=canvas.columns*canvas.rows;
  dim3 grid(stream_grid(n)),block(256);
  ProfileScope prof("composite",canvas.stream);
#define MH_COMPOSE(OP) \
  hipLaunchKernelGGL((composite_kernel<Q,C,OP>),grid,block,0,canvas.stream,static_cast<Q *>(canvas.pixels), \
    static_cast<const Q *>(source.pixels),n,roles.alpha,roles.update_mask,roles.copy_mask)
  switch (kind)
  {
    case COMPOSITE_DIFFERENCE: MH_COMPOSE(COMPOSITE_DIFFERENCE); break;
    case COMPOSITE_LIGHTEN: MH_COMPOSE(COMPOSITE_LIGHTEN); break;
    case COMPOSITE_DARKEN: MH_COMPOSE(COMPOSITE_DARKEN); break;
    case COMPOSITE_PLUS: MH_COMPOSE(COMPOSITE_PLUS); break;
    case COMPOSITE_MULTIPLY: MH_COMPOSE(COMPOSITE_MULTIPLY); break;
    case COMPOSITE_SCREEN: MH_COMPOSE(COMPOSITE_SCREEN); break;
    case COMPOSITE_EXCLUSION: MH_COMPOSE(COMPOSITE_EXCLUSION); break;
    case COMPOSITE_MINUS_SRC: MH_COMPOSE(COMPOSITE_MINUS_SRC); break;
    case COMPOSITE_MINUS_DST: MH_COMPOSE(COMPOSITE_MINUS_DST); break;
    case COMPOSITE_LINEAR_DODGE: MH_COMPOSE(COMPOSITE_LINEAR_DODGE); break;
    case COMPOSITE_OVER: MH_COMPOSE(COMPOSITE_OVER); break;
    case COMPOSITE_DST_OVER: MH_COMPOSE(COMPOSITE_DST_OVER); break;
    default: return fail(MH_UNSUPPORTED,"composite operator %d",kind);
  }
#undef MH_COMPOSE
  MH_HIP(hipGetLastError());
  return MH_OK;
}

MhStatus launch_composite(const View &canvas,const View &source,int kind,const Roles &roles)
{
  if ((canvas.columns != source.columns) || (canvas.rows != source.rows) ||
      (canvas.channels != source.channels) || (canvas.quantum != source.quantum))
    return fail(MH_BAD_ARGUMENT,"composite: canvas and source differ in geometry");
#define MH_CASE(QT) \
  switch (canvas.channels) { \
    case 1: return composite_typed<QT,1>(canvas,source,kind,roles); \
    case 2: return composite_typed<QT,2>(canvas,source,kind,roles); \
    case 3: return composite_typed<QT,3>(canvas,source,kind,roles); \
    default: return composite_typed<QT,4>(canvas,source,kind,roles); }
  if (canvas.quantum == MH_QUANTUM_U16)
    { MH_CASE(uint16_t) }
  MH_CASE(float)
#undef MH_CASE
}

// ---------------------------------------------------------------- ContrastImage / ModulateImage
// HSB and HSL round trips in fp64 exactly as the CPU path writes them (every expression in
// the reference's order; -ffp-contract=off): ConvertRGBToHSB / ConvertHSBToRGB
// colorspace-private.h:867-908 / :292-365, ConvertRGBToHSL / ConvertHSLToRGB
// colorspace.c:597-640 / :307-378.
static __device__ void rgb_to_hsb(double red,double green,double blue,double &hue,double &saturation,
  double &brightness)
{
  hue=0.0;
  saturation=0.0;
  brightness=0.0;
  double mn=red < green ? red : green;
  if (blue < mn)
    mn=blue;
  double mx=red > green ? red : green;
  if (blue > mx)
    mx=blue;
  if (fabs(mx) < kEps)
    return;
  const double delta=mx-mn;
  saturation=delta/mx;
  brightness=kQS*mx;
  if (fabs(delta) < kEps)
    return;
  if (fabs(red-mx) < kEps)
    hue=(green-blue)/delta;
  else if (fabs(green-mx) < kEps)
    hue=2.0+(blue-red)/delta;
  else
    hue=4.0+(red-green)/delta;
  hue/=6.0;
  if (hue < 0.0)
    hue+=1.0;
}

static __device__ void hsb_to_rgb(double hue,double saturation,double brightness,double &red,
  double &green,double &blue)
{
  if (fabs(saturation) < kEps)
    {
      red=kQR*brightness;
      green=red;
      blue=red;
      return;
    }
  const double h=6.0*(hue-floor(hue));
  const double f=h-floor(h);
  const double p=brightness*(1.0-saturation);
  const double q=brightness*(1.0-saturation*f);
  const double t=brightness*(1.0-(saturation*(1.0-f)));
  switch ((int) h)
  {
    case 1: red=kQR*q; green=kQR*brightness; blue=kQR*p; break;
    case 2: red=kQR*p; green=kQR*brightness; blue=kQR*t; break;
    case 3: red=kQR*p; green=kQR*q; blue=kQR*brightness; break;
    case 4: red=kQR*t; green=kQR*p; blue=kQR*brightness; break;
    case 5: red=kQR*brightness; green=kQR*p; blue=kQR*q; break;
    default: red=kQR*brightness; green=kQR*t; blue=kQR*p; break;        // 0
  }
}

static __device__ void rgb_to_hsl(double red,double green,double blue,double &hue,double &saturation,
  double &lightness)
{
  const double r=kQS*red,g=kQS*green,b=kQS*blue;
  const double gb_max=g > b ? g : b,gb_min=g < b ? g : b;
  const double mx=r > gb_max ? r : gb_max,mn=r < gb_min ? r : gb_min;
  const double c=mx-mn;
  lightness=(mx+mn)/2.0;
  if (c <= 0.0)
    {
      hue=0.0;
      saturation=0.0;
      return;
    }
  if (fabs(mx-r) < kEps)
    {
      hue=(g-b)/c;
      if (g < b)
        hue+=6.0;
    }
  else if (fabs(mx-g) < kEps)
    hue=2.0+(b-r)/c;
  else
    hue=4.0+(r-g)/c;
  hue*=60.0/360.0;
  if (lightness <= 0.5)
    saturation=c*perceptible_reciprocal(2.0*lightness);
  else
    saturation=c*perceptible_reciprocal(2.0-2.0*lightness);
}

static __device__ void hsl_to_rgb(double hue,double saturation,double lightness,double &red,
  double &green,double &blue)
{
  double h=hue*360.0,c;
  if (lightness <= 0.5)
    c=2.0*lightness*saturation;
  else
    c=(2.0-2.0*lightness)*saturation;
  const double mn=lightness-0.5*c;
  h-=360.0*floor(h/360.0);
  h/=60.0;
  const double x=c*(1.0-fabs(h-2.0*floor(h/2.0)-1.0));
  switch ((int) floor(h))
  {
    case 1: red=kQR*(mn+x); green=kQR*(mn+c); blue=kQR*mn; break;
    case 2: red=kQR*mn; green=kQR*(mn+c); blue=kQR*(mn+x); break;
    case 3: red=kQR*mn; green=kQR*(mn+x); blue=kQR*(mn+c); break;
    case 4: red=kQR*(mn+x); green=kQR*mn; blue=kQR*(mn+c); break;
    case 5: red=kQR*(mn+c); green=kQR*mn; blue=kQR*(mn+x); break;
    default: red=kQR*(mn+c); green=kQR*(mn+x); blue=kQR*mn; break;       // 0
  }
}

enum ToneOp { TONE_CONTRAST=0,TONE_MODULATE_HSL=1,TONE_MODULATE_HSB=2 };

struct ToneArgs
{
  int op;
  double sign;                 // Contrast: +1 sharpen, -1 dull
  double hue_shift,saturation_scale,brightness_scale;      // Modulate
};

// ContrastImage enhance.c:1370-1390 (per pixel), ModulateHSL / ModulateHSB :3499-3554
template<typename Q,int C>
__global__ __launch_bounds__(256)
void tone_kernel(Q *__restrict__ pixels,size_t npixels,ToneArgs a)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x*kPointBatch;
  for (size_t i0=(size_t) blockIdx.x*blockDim.x*kPointBatch+threadIdx.x; i0 < npixels; i0+=stride)
    {
      Q qb[kPointBatch][C];
#pragma unroll
      for (int k=0; k < kPointBatch; k++)
        {
          const size_t ik=i0+(size_t) k*blockDim.x;
          load_pixel<Q,C>(pixels+(ik < npixels ? ik : npixels-1)*C,qb[k]);
        }
#pragma unroll
      for (int k=0; k < kPointBatch; k++)
        {
          const size_t i=i0+(size_t) k*blockDim.x;
          double red=(double) qb[k][0],green=(double) qb[k][1],blue=(double) qb[k][2];
          double hue,saturation,third;
          if (a.op == TONE_CONTRAST)
            {
              rgb_to_hsb(red,green,blue,hue,saturation,third);
              third+=0.5*a.sign*(0.5*(sin((double) (3.14159265358979323846264338327950288*(third-0.5)))+1.0)-third);
              third=third > 1.0 ? 1.0 : (third < 0.0 ? 0.0 : third);
              hsb_to_rgb(hue,saturation,third,red,green,blue);
            }
          else if (a.op == TONE_MODULATE_HSB)
            {
              rgb_to_hsb(red,green,blue,hue,saturation,third);
              hue+=a.hue_shift;
              saturation*=a.saturation_scale;
              third*=a.brightness_scale;
              hsb_to_rgb(hue,saturation,third,red,green,blue);
            }
          else
            {
              rgb_to_hsl(red,green,blue,hue,saturation,third);
              hue+=a.hue_shift;
              saturation*=a.saturation_scale;
              third*=a.brightness_scale;
              hsl_to_rgb(hue,saturation,third,red,green,blue);
            }
          qb[k][0]=QuantumOps<Q>::clamp(red);
          qb[k][1]=QuantumOps<Q>::clamp(green);
          qb[k][2]=QuantumOps<Q>::clamp(blue);
          if (i < npixels)
            store_pixel<Q,C>(pixels+i*C,qb[k]);
        }
    }
}

static MhStatus launch_tone(const View &img,const ToneArgs &a,const char *label)
{
  if ((img.channels != 3) && (img.channels != 4))
    return fail(MH_UNSUPPORTED,"%s needs R,G,B[,A] channels",label);
  const size_t n=img.columns*img.rows;
  dim3 grid(stream_grid((n+kPointBatch-1)/kPointBatch)),block(256);
  ProfileScope prof(label,img.stream);
  if (img.quantum == MH_QUANTUM_U16)
    {
      if (img.channels == 3)
        hipLaunchKernelGGL((tone_kernel<uint16_t,3>),grid,block,0,img.stream,static_cast<uint16_t *>(img.pixels),n,a);
      else
        hipLaunchKernelGGL((tone_kernel<uint16_t,4>),grid,block,0,img.stream,static_cast<uint16_t *>(img.pixels),n,a);
    }
  else
    {
      if (img.channels == 3)
        hipLaunchKernelGGL((tone_kernel<float,3>),grid,block,0,img.stream,static_cast<float *>(img.pixels),n,a);
      else
        hipLaunchKernelGGL((tone_kernel<float,4>),grid,block,0,img.stream,static_cast<float *>(img.pixels),n,a);
    }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

MhStatus launch_contrast(const View &img,bool sharpen)
{
  ToneArgs a;
  a.op=TONE_CONTRAST;
  a.sign=sharpen ? 1.0 : -1.0;
  a.hue_shift=a.saturation_scale=a.brightness_scale=0.0;
  return launch_tone(img,a,"contrast");
}

MhStatus launch_modulate(const View &img,bool hsb,double hue_shift,double saturation_scale,
  double brightness_scale)
{
  ToneArgs a;
  a.op=hsb ? TONE_MODULATE_HSB : TONE_MODULATE_HSL;
  a.sign=0.0;
  a.hue_shift=hue_shift;
  a.saturation_scale=saturation_scale;
  a.brightness_scale=brightness_scale;
  return launch_tone(img,a,"modulate");
}

// ---------------------------------------------------------------- the other pointwise colourspaces
constexpr double kPi=3.1415926535897932384626433832795028841971693993751058209749445923078164062;   // MagickPI
#include "colorspace_generic.inc.hpp"

static bool generic_colorspace(MhColorspace c)
{
  switch (c)
  {
    case MH_COLORSPACE_CMY: case MH_COLORSPACE_HCL: case MH_COLORSPACE_HCLP: case MH_COLORSPACE_HSB:
    case MH_COLORSPACE_HSI: case MH_COLORSPACE_HSL: case MH_COLORSPACE_HSV: case MH_COLORSPACE_HWB:
    case MH_COLORSPACE_LCH: case MH_COLORSPACE_LCHAB: case MH_COLORSPACE_LCHUV: case MH_COLORSPACE_LMS:
    case MH_COLORSPACE_LUV: case MH_COLORSPACE_XYY: case MH_COLORSPACE_YCBCR: case MH_COLORSPACE_YDBDR:
    case MH_COLORSPACE_YIQ: case MH_COLORSPACE_YPBPR: case MH_COLORSPACE_YUV: case MH_COLORSPACE_JZAZBZ:
    case MH_COLORSPACE_DISPLAYP3: case MH_COLORSPACE_ADOBE98: case MH_COLORSPACE_PROPHOTO:
    case MH_COLORSPACE_OKLAB: case MH_COLORSPACE_OKLCH: case MH_COLORSPACE_CAT02LMS:
      return true;
    default:
      return false;
  }
}

bool colorspace_is_accelerated(MhColorspace c)
{
  return (c == MH_COLORSPACE_SRGB) || (c == MH_COLORSPACE_RGB) || (c == MH_COLORSPACE_SCRGB) ||
    (c == MH_COLORSPACE_LAB) || (c == MH_COLORSPACE_XYZ) || generic_colorspace(c) ||
    (c == MH_COLORSPACE_OHTA) || (c == MH_COLORSPACE_REC601YCBCR) || (c == MH_COLORSPACE_REC709YCBCR) ||
    (c == MH_COLORSPACE_YCC) || (c == MH_COLORSPACE_LOG);
}

template<typename Q,int C>
static MhStatus colorspace_generic_typed(const View &img,MhColorspace colorspace,bool forward,
  double white_luminance)
{
  const size_t n=img.columns*img.rows;
  dim3 grid(stream_grid(n)),block(256);
  ProfileScope prof("colorspace",img.stream);
  if (forward)
    hipLaunchKernelGGL((colorspace_generic_kernel<Q,C,true>),grid,block,0,img.stream,
      static_cast<Q *>(img.pixels),n,(int) colorspace,white_luminance);
  else
    hipLaunchKernelGGL((colorspace_generic_kernel<Q,C,false>),grid,block,0,img.stream,
      static_cast<Q *>(img.pixels),n,(int) colorspace,white_luminance);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

static MhStatus colorspace_generic_step(const View &img,MhColorspace colorspace,bool forward)
{
  const double white_luminance=10000.0;          // colorspace.c:993 (no "white-luminance" property)
  if (img.quantum == MH_QUANTUM_U16)
    return img.channels == 3 ? colorspace_generic_typed<uint16_t,3>(img,colorspace,forward,white_luminance) :
      colorspace_generic_typed<uint16_t,4>(img,colorspace,forward,white_luminance);
  return img.channels == 3 ? colorspace_generic_typed<float,3>(img,colorspace,forward,white_luminance) :
    colorspace_generic_typed<float,4>(img,colorspace,forward,white_luminance);
}

// ---------------------------------------------------------------- the table-driven colourspaces
// OHTA, Rec601YCbCr, Rec709YCbCr and YCC are the half of sRGBTransformImage / TransformsRGBImage
// that works through three tables of MaxMap+1 TransformPackets (colorspace.c:1226-1420,
// :2560-2790): x_map[i], y_map[i], z_map[i] hold what map index i of the red, green and blue sample
// contributes to each result, the pixel loop adds the three entries (+ the primary offsets) and
// ScaleMapToQuantum rounds.  Every entry is ONE rounded product k*f(i) — f(i) = i (forward; the
// inverse's first table), 2i - MaxMap (inverse, second and third table: the constant 0.5*k is
// folded by the compiler as it would be multiplied), 1.099i - 0.099 (YCC above 0.018 MaxMap) — so
// the kernel forms the entries in place, in the table's own operations, instead of reading them.
// YCC -> sRGB goes through a 1389-entry film curve (YCCMap) and stays on the CPU.
struct MatrixColorspaceArgs
{
  double k[3][3];             // [table: red, green, blue index][result channel]
  double low[3][3];           // YCC forward: the entries of indices up to 0.018 MaxMap
  double primary[3];
  int form;                   // 0 forward, 1 inverse, 2 YCC forward
};

template<typename Q,int C>
__global__ __launch_bounds__(256)
void colorspace_matrix_kernel(Q *pixels,size_t npixels,MatrixColorspaceArgs a)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t at=(size_t) blockIdx.x*blockDim.x+threadIdx.x; at < npixels; at+=stride)
    {
      Q q[C];
      load_pixel<Q,C>(pixels+at*C,q);
      double entry[3][3];
#pragma unroll
      for (int t=0; t < 3; t++)
        {
          // ScaleQuantumToMap(ClampToQuantum(sample)), colorspace.c:1459-1464, :2738-2740
          const unsigned index=QuantumOps<Q>::map_index(q[t]);
          const double i=(double) index;
          if (a.form == 2)
            {
              const bool low=index <= 1179u;      // (ssize_t) (0.018*MaxMap)
              const double upper=1.099*i-0.099;
#pragma unroll
              for (int c=0; c < 3; c++)
                entry[t][c]=low ? a.low[t][c]*i : a.k[t][c]*upper;
            }
          else
            {
              const double f=((a.form == 1) && (t != 0)) ? 2.0*i-65535.0 : i;
#pragma unroll
              for (int c=0; c < 3; c++)
                entry[t][c]=a.k[t][c]*f;
            }
        }
#pragma unroll
      for (int c=0; c < 3; c++)
        {
          double value=entry[0][c]+entry[1][c]+entry[2][c];
          if (a.form != 1)
            value=value+a.primary[c];
          // ScaleMapToQuantum, quantum-private.h:465-476
          if (value <= 0.0)
            q[c]=(Q) 0;
          else if (value >= 65535.0)
            q[c]=(Q) 65535;
          else
            q[c]=QuantumOps<Q>::is_float ? (Q) value : (Q) (value+0.5);
        }
      store_pixel<Q,C>(pixels+at*C,q);
    }
}

static bool matrix_colorspace(MhColorspace c)
{
  return (c == MH_COLORSPACE_OHTA) || (c == MH_COLORSPACE_REC601YCBCR) || (c == MH_COLORSPACE_REC709YCBCR) ||
    (c == MH_COLORSPACE_YCC);
}

static MhStatus colorspace_matrix_step(const View &img,MhColorspace colorspace,bool forward)
{
  MatrixColorspaceArgs a={};
  auto rows=[&](double (&to)[3][3],const double (&red)[3],const double (&green)[3],const double (&blue)[3])
  {
    for (int c=0; c < 3; c++)
      {
        to[0][c]=red[c];
        to[1][c]=green[c];
        to[2][c]=blue[c];
      }
  };
  a.form=forward ? 0 : 1;
  if (forward)
    {
      a.primary[0]=0.0;
      a.primary[1]=a.primary[2]=32768.0;          // (MaxMap+1)/2
    }
  switch (colorspace)
  {
    case MH_COLORSPACE_OHTA:
      if (forward)
        rows(a.k,{0.33333,0.50000,-0.25000},{0.33334,0.00000,0.50000},{0.33333,-0.50000,-0.25000});
      else
        rows(a.k,{1.0,1.0,1.0},{0.5*1.00000,0.5*0.00000,-0.5*1.00000},{-0.5*0.66668,0.5*1.33333,-0.5*0.66668});
      break;
    case MH_COLORSPACE_REC601YCBCR:
      if (forward)
        rows(a.k,{0.298839,-0.1687367,0.500000},{0.586811,-0.331264,-0.418688},{0.114350,0.500000,-0.081312});
      else
        rows(a.k,{0.99999999999914679361,0.99999975910502514331,1.00000124040004623180},
          {0.5*(-1.2188941887145875e-06),0.5*(-0.34413567816504303521),0.5*1.77200006607230409200},
          {0.5*1.4019995886561440468,0.5*(-0.71413649331646789076),0.5*2.1453384174593273e-06});
      break;
    case MH_COLORSPACE_REC709YCBCR:
      if (forward)
        rows(a.k,{0.212656,-0.114572,0.500000},{0.715158,-0.385428,-0.454153},{0.072186,0.500000,-0.045847});
      else
        rows(a.k,{1.0,1.0,1.0},{0.5*0.000000,0.5*(-0.187324),0.5*1.855600},{0.5*1.574800,0.5*(-0.468124),0.5*0.000000});
      break;
    case MH_COLORSPACE_YCC:
      if (!forward)
        return fail(MH_UNSUPPORTED,"YCC -> sRGB (the YCCMap film curve) is not accelerated");
      a.form=2;
      rows(a.low,{0.005382,-0.003296,0.009410},{0.010566,-0.006471,-0.007880},{0.002052,0.009768,-0.001530});
      rows(a.k,{0.298839,-0.298839,0.70100},{0.586811,-0.586811,-0.586811},{0.114350,0.88600,-0.114350});
      a.primary[1]=40092.0;                        // ScaleQuantumToMap(ScaleCharToQuantum(156))
      a.primary[2]=35209.0;                        // ... (137)
      break;
    default:
      return fail(MH_UNSUPPORTED,"colourspace %d is not table-driven",(int) colorspace);
  }
  const size_t n=img.columns*img.rows;
  dim3 grid(stream_grid(n)),block(256);
  ProfileScope prof("colorspace",img.stream);
  if (img.quantum == MH_QUANTUM_U16)
    {
      if (img.channels == 3)
        hipLaunchKernelGGL((colorspace_matrix_kernel<uint16_t,3>),grid,block,0,img.stream,static_cast<uint16_t *>(img.pixels),n,a);
      else
        hipLaunchKernelGGL((colorspace_matrix_kernel<uint16_t,4>),grid,block,0,img.stream,static_cast<uint16_t *>(img.pixels),n,a);
    }
  else
    {
      if (img.channels == 3)
        hipLaunchKernelGGL((colorspace_matrix_kernel<float,3>),grid,block,0,img.stream,static_cast<float *>(img.pixels),n,a);
      else
        hipLaunchKernelGGL((colorspace_matrix_kernel<float,4>),grid,block,0,img.stream,static_cast<float *>(img.pixels),n,a);
    }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// Log (Cineon film density, colorspace.c:1055-1163, :2391-2501): a table of MaxMap+1 Quantum values
// built from four film parameters — here with their defaults (the image properties "gamma",
// "film-gamma", "reference-black", "reference-white" change them: the caller declines then) —
// indexed by the DECODED sample on the way in, and followed by EncodePixelGamma on the way back.
// The table is built on the host in the reference's own expressions (libm's pow and log10).
template<typename Q>
static void build_log_table(bool forward,std::vector<Q> &table)
{
  const double density=1.0/1.7,gamma=1.0/1.7,film_gamma=0.6,reference_black=95.0,reference_white=685.0;
  const double max_map=65535.0;
  // PerceptibleReciprocal (pixel-accessor.h: 1/x, clamped at MagickEpsilon)
  auto reciprocal=[](double x) { const double sign=x < 0.0 ? -1.0 : 1.0; return (sign*x) >= 1.0e-12 ? 1.0/x : sign/1.0e-12; };
  auto map_to_quantum=[&](double value) -> Q    // ScaleMapToQuantum, quantum-private.h:465-476
  {
    if (value <= 0.0)
      return (Q) 0;
    if (value >= max_map)
      return (Q) 65535;
    return std::is_same<Q,float>::value ? (Q) value : (Q) (value+0.5);
  };
  table.assign(65536,(Q) 0);
  const double black=std::pow(10.0,(reference_black-reference_white)*(gamma/density)*0.002*reciprocal(film_gamma));
  if (forward)
    {
      for (long i=0; i <= 65535; i++)
        table[(size_t) i]=map_to_quantum((max_map*(reference_white+std::log10(black+(1.0*(double) i/max_map)*(1.0-black))/
          ((gamma/density)*0.002*reciprocal(film_gamma)))/1024.0));
      return;
    }
  long i=0;
  for ( ; i <= (long) (reference_black*max_map/1024.0); i++)
    table[(size_t) i]=(Q) 0;
  for ( ; i < (long) (reference_white*max_map/1024.0); i++)
    {
      const double value=65535.0/(1.0-black)*(std::pow(10.0,(1024.0*(double) i/max_map-reference_white)*(gamma/density)*0.002*
        reciprocal(film_gamma))-black);
      // ClampToQuantum (quantum.h:86-97)
      if (std::is_same<Q,float>::value)
        table[(size_t) i]=(Q) value;
      else
        table[(size_t) i]=!(value > 0.0) ? (Q) 0 : (value >= 65535.0 ? (Q) 65535 : (Q) (value+0.5));
    }
  for ( ; i <= 65535; i++)
    table[(size_t) i]=(Q) 65535;
}

template<typename Q,int C,bool FORWARD>
__global__ __launch_bounds__(256)
void colorspace_log_kernel(Q *pixels,size_t npixels,const Q *table)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t at=(size_t) blockIdx.x*blockDim.x+threadIdx.x; at < npixels; at+=stride)
    {
      Q q[C];
      load_pixel<Q,C>(pixels+at*C,q);
#pragma unroll
      for (int c=0; c < 3; c++)
        {
          if constexpr (FORWARD)
            q[c]=table[QuantumOps<Q>::map_index(QuantumOps<Q>::clamp(decode_pixel_gamma((double) q[c])))];
          else
            q[c]=QuantumOps<Q>::clamp(encode_pixel_gamma((double) table[QuantumOps<Q>::map_index(q[c])]));
        }
      store_pixel<Q,C>(pixels+at*C,q);
    }
}

template<typename Q>
static MhStatus colorspace_log_typed(const View &img,bool forward)
{
  std::vector<Q> host;
  build_log_table<Q>(forward,host);
  Temp table;
  MH_TRY(upload_table(table,img.device,img.stream,host.data(),host.size()*sizeof(Q)));
  const size_t n=img.columns*img.rows;
  dim3 grid(stream_grid(n)),block(256);
  Q *pixels=static_cast<Q *>(img.pixels);
  ProfileScope prof("colorspace",img.stream);
  if (img.channels == 3)
    {
      if (forward)
        hipLaunchKernelGGL((colorspace_log_kernel<Q,3,true>),grid,block,0,img.stream,pixels,n,table.as<Q>());
      else
        hipLaunchKernelGGL((colorspace_log_kernel<Q,3,false>),grid,block,0,img.stream,pixels,n,table.as<Q>());
    }
  else
    {
      if (forward)
        hipLaunchKernelGGL((colorspace_log_kernel<Q,4,true>),grid,block,0,img.stream,pixels,n,table.as<Q>());
      else
        hipLaunchKernelGGL((colorspace_log_kernel<Q,4,false>),grid,block,0,img.stream,pixels,n,table.as<Q>());
    }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

static bool colorspace_is_generic(MhColorspace c)
{
  return generic_colorspace(c) || matrix_colorspace(c) || (c == MH_COLORSPACE_LOG);
}
static MhStatus colorspace_generic_forward_or_inverse(const View &img,MhColorspace colorspace,bool forward)
{
  if (colorspace == MH_COLORSPACE_LOG)
    return img.quantum == MH_QUANTUM_U16 ? colorspace_log_typed<uint16_t>(img,forward) :
      colorspace_log_typed<float>(img,forward);
  if (matrix_colorspace(colorspace))
    return colorspace_matrix_step(img,colorspace,forward);
  return colorspace_generic_step(img,colorspace,forward);
}

MhStatus launch_modulate_generic(const View &img,MhColorspace colorspace,double hue_shift,
  double saturation_scale,double brightness_scale)
{
  if ((img.channels != 3) && (img.channels != 4))
    return fail(MH_UNSUPPORTED,"modulate needs R,G,B[,A] channels");
  const size_t n=img.columns*img.rows;
  dim3 grid(stream_grid(n)),block(256);
  ProfileScope prof("modulate",img.stream);
#define MH_MODULATE(QT,CH) \
  hipLaunchKernelGGL((modulate_generic_kernel<QT,CH>),grid,block,0,img.stream,static_cast<QT *>(img.pixels),n, \
    (int) colorspace,hue_shift,saturation_scale,brightness_scale)
  if (img.quantum == MH_QUANTUM_U16)
    {
      if (img.channels == 3)
        MH_MODULATE(uint16_t,3);
      else
        MH_MODULATE(uint16_t,4);
    }
  else
    {
      if (img.channels == 3)
        MH_MODULATE(float,3);
      else
        MH_MODULATE(float,4);
    }
#undef MH_MODULATE
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// ---------------------------------------------------------------- GrayscaleImage
// enhance.c:2476-2660: the intensity of (R,G,B) by `method` is written to the Gray
// (= first) channel only; the caller then switches the image to GRAY / LinearGRAY.
// Note MS here is (r^2+g^2+b^2)/3 (:2572-2577), unlike GetPixelIntensity's MS.
template<typename Q,int C>
__global__ __launch_bounds__(256)
void grayscale_kernel(Q *pixels,size_t npixels,int method,int is_rgb,int is_srgb)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < npixels; i+=stride)
    {
      Q q[C];
      load_pixel<Q,C>(pixels+i*C,q);
      double red=(double) q[0],green=(double) q[C >= 3 ? 1 : 0],blue=(double) q[C >= 3 ? 2 : 0];
      double intensity=0.0;
      switch (method)
      {
        case MH_INTENSITY_AVERAGE: intensity=(red+green+blue)/3.0; break;
        case MH_INTENSITY_BRIGHTNESS:
        {
          double m=red > green ? red : green;
          intensity=m > blue ? m : blue;
          break;
        }
        case MH_INTENSITY_LIGHTNESS:
        {
          double mn=red < green ? red : green;
          mn=mn < blue ? mn : blue;
          double mx=red > green ? red : green;
          mx=mx > blue ? mx : blue;
          intensity=(mn+mx)/2.0;
          break;
        }
        case MH_INTENSITY_MS: intensity=(red*red+green*green+blue*blue)/3.0; break;
        case MH_INTENSITY_RMS: intensity=sqrt(red*red+green*green+blue*blue)/sqrt(3.0); break;
        case MH_INTENSITY_REC601LUMA:
        case MH_INTENSITY_REC601LUMINANCE:
        {
          if ((method == MH_INTENSITY_REC601LUMA) ? is_rgb : is_srgb)
            {
              if (method == MH_INTENSITY_REC601LUMA)
                { red=encode_pixel_gamma(red); green=encode_pixel_gamma(green); blue=encode_pixel_gamma(blue); }
              else
                { red=decode_pixel_gamma(red); green=decode_pixel_gamma(green); blue=decode_pixel_gamma(blue); }
            }
          intensity=0.298839*red+0.586811*green+0.114350*blue;
          break;
        }
        case MH_INTENSITY_REC709LUMINANCE:
        {
          if (is_srgb)
            { red=decode_pixel_gamma(red); green=decode_pixel_gamma(green); blue=decode_pixel_gamma(blue); }
          intensity=0.212656*red+0.715158*green+0.072186*blue;
          break;
        }
        default:      // Rec709Luma
        {
          if (is_rgb)
            { red=encode_pixel_gamma(red); green=encode_pixel_gamma(green); blue=encode_pixel_gamma(blue); }
          intensity=0.212656*red+0.715158*green+0.072186*blue;
          break;
        }
      }
      pixels[i*C]=QuantumOps<Q>::clamp(intensity);       // SetPixelGray: the first channel only
    }
}

MhStatus launch_grayscale(const View &img,int method,const MhImage *desc)
{
  const size_t n=img.columns*img.rows;
  const int is_rgb=desc->colorspace == MH_COLORSPACE_RGB;
  const int is_srgb=desc->colorspace == MH_COLORSPACE_SRGB;
  ProfileScope prof("grayscale",img.stream);
#define MH_CASE(QT) \
  switch (img.channels) { \
    case 1: hipLaunchKernelGGL((grayscale_kernel<QT,1>),dim3(stream_grid(n)),dim3(256),0,img.stream,static_cast<QT *>(img.pixels),n,method,is_rgb,is_srgb); break; \
    case 2: hipLaunchKernelGGL((grayscale_kernel<QT,2>),dim3(stream_grid(n)),dim3(256),0,img.stream,static_cast<QT *>(img.pixels),n,method,is_rgb,is_srgb); break; \
    case 3: hipLaunchKernelGGL((grayscale_kernel<QT,3>),dim3(stream_grid(n)),dim3(256),0,img.stream,static_cast<QT *>(img.pixels),n,method,is_rgb,is_srgb); break; \
    default: hipLaunchKernelGGL((grayscale_kernel<QT,4>),dim3(stream_grid(n)),dim3(256),0,img.stream,static_cast<QT *>(img.pixels),n,method,is_rgb,is_srgb); break; }
  if (img.quantum == MH_QUANTUM_U16)
    { MH_CASE(uint16_t) }
  else
    { MH_CASE(float) }
#undef MH_CASE
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// ----------------------------------------------------------------- FunctionImage
// ApplyFunction, statistic.c:975-1067, on every channel selected by `mask`.
struct FunctionParams
{
  int function;               // MhFunction
  int count;
  double p[8];
};

template<typename Q,int C>
__global__ __launch_bounds__(256)
void function_kernel(Q *pixels,size_t npixels,FunctionParams fp,uint32_t mask)
{
  // (kPi: MagickPI, defined with the colourspace helpers)
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < npixels; i+=stride)
    {
      Q q[C];
      load_pixel<Q,C>(pixels+i*C,q);
#pragma unroll
      for (int c=0; c < C; c++)
        {
          if (((mask >> c) & 1u) == 0)
            continue;
          const double pixel=(double) q[c];
          double result=0.0;
          switch (fp.function)
          {
            case MH_FUNCTION_POLYNOMIAL:
              for (int k=0; k < fp.count; k++)
                result=result*kQS*pixel+fp.p[k];
              result*=kQR;
              break;
            case MH_FUNCTION_SINUSOID:
            {
              double frequency=fp.count >= 1 ? fp.p[0] : 1.0,phase=fp.count >= 2 ? fp.p[1] : 0.0;
              double amplitude=fp.count >= 3 ? fp.p[2] : 0.5,bias=fp.count >= 4 ? fp.p[3] : 0.5;
              result=kQR*(amplitude*sin(2.0*kPi*(frequency*kQS*pixel+phase/360.0))+bias);
              break;
            }
            case MH_FUNCTION_ARCSIN:
            {
              double width=fp.count >= 1 ? fp.p[0] : 1.0,center=fp.count >= 2 ? fp.p[1] : 0.5;
              double range=fp.count >= 3 ? fp.p[2] : 1.0,bias=fp.count >= 4 ? fp.p[3] : 0.5;
              result=2.0*perceptible_reciprocal(width)*(kQS*pixel-center);
              if (result <= -1.0)
                result=bias-range/2.0;
              else if (result >= 1.0)
                result=bias+range/2.0;
              else
                result=range/kPi*asin(result)+bias;
              result*=kQR;
              break;
            }
            case MH_FUNCTION_ARCTAN:
            {
              double slope=fp.count >= 1 ? fp.p[0] : 1.0,center=fp.count >= 2 ? fp.p[1] : 0.5;
              double range=fp.count >= 3 ? fp.p[2] : 1.0,bias=fp.count >= 4 ? fp.p[3] : 0.5;
              result=kPi*slope*(kQS*pixel-center);
              result=kQR*(range/kPi*atan(result)+bias);
              break;
            }
            default: break;
          }
          q[c]=QuantumOps<Q>::clamp(result);
        }
      store_pixel<Q,C>(pixels+i*C,q);
    }
}

MhStatus launch_function(const View &img,int function,size_t count,const double *parameters,uint32_t mask)
{
  if (count > 8)
    return fail(MH_UNSUPPORTED,"FunctionImage: more than 8 parameters");
  FunctionParams fp;
  fp.function=function;
  fp.count=(int) count;
  for (int k=0; k < 8; k++)
    fp.p[k]=k < (int) count ? parameters[k] : 0.0;
  const size_t n=img.columns*img.rows;
  ProfileScope prof("function",img.stream);
#define MH_CASE(QT) \
  switch (img.channels) { \
    case 1: hipLaunchKernelGGL((function_kernel<QT,1>),dim3(stream_grid(n)),dim3(256),0,img.stream,static_cast<QT *>(img.pixels),n,fp,mask); break; \
    case 2: hipLaunchKernelGGL((function_kernel<QT,2>),dim3(stream_grid(n)),dim3(256),0,img.stream,static_cast<QT *>(img.pixels),n,fp,mask); break; \
    case 3: hipLaunchKernelGGL((function_kernel<QT,3>),dim3(stream_grid(n)),dim3(256),0,img.stream,static_cast<QT *>(img.pixels),n,fp,mask); break; \
    default: hipLaunchKernelGGL((function_kernel<QT,4>),dim3(stream_grid(n)),dim3(256),0,img.stream,static_cast<QT *>(img.pixels),n,fp,mask); break; }
  if (img.quantum == MH_QUANTUM_U16)
    { MH_CASE(uint16_t) }
  else
    { MH_CASE(float) }
#undef MH_CASE
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// ---------------------------------------------------------------- gray scan
// SetImageGray's scan (attribute.c:1416-1450: IsPixelGray on every pixel).  A wave leaves at its
// first colour pixel and reports it with a plain store — on a colour frame the kernel used to be
// 32 768 atomicOr on one word (0.4 ms per 8192^2 RGBA call, more than the histogram) — and Q16
// pixels are compared as integers (|a-b| <
// MagickEpsilon on integer levels is a == b), two RGBA pixels per 16-byte load.
template<typename Q,int C>
__global__ __launch_bounds__(256)
void gray_check_kernel(const Q *pixels,size_t npixels,unsigned int *not_gray)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  bool bad=false;
  if constexpr ((sizeof(Q) == 2) && (C == 4))
    {
      if ((reinterpret_cast<uintptr_t>(pixels) & 15u) == 0)
        {
          const uint4 *pairs=reinterpret_cast<const uint4 *>(pixels);
          const size_t npairs=npixels/2;
          for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < npairs; i+=stride)
            {
              const uint4 v=pairs[i];
              // red == green and green == blue, both pixels
              bad=(((v.x >> 16) ^ v.x) & 0xffffu) != 0u || ((v.y ^ v.x) & 0xffffu) != 0u ||
                  (((v.z >> 16) ^ v.z) & 0xffffu) != 0u || ((v.w ^ v.z) & 0xffffu) != 0u;
              if (__any(bad))
                break;
            }
          if (((npixels & 1u) != 0) && (blockIdx.x == 0) && (threadIdx.x == 0))
            {
              const uint16_t *last=pixels+(npixels-1)*4;
              bad=bad || (last[0] != last[1]) || (last[1] != last[2]);
            }
          if (__any(bad) && ((threadIdx.x & 63) == 0))
            *not_gray=1u;                        // (a plain store: every writer writes the same 1)
          return;
        }
    }
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < npixels; i+=stride)
    {
      Q q[C];
      load_pixel<Q,C>(pixels+i*C,q);
      // IsPixelGray, pixel-accessor.h: |red-green| < eps && |green-blue| < eps
      double rg=(double) q[0]-(double) q[1],gb=(double) q[1]-(double) q[2];
      if (!((fabs(rg) < kEps) && (fabs(gb) < kEps)))
        bad=true;
      if (__any(bad))
        break;
    }
  if (__any(bad) && ((threadIdx.x & 63) == 0))
    *not_gray=1u;
}

MhStatus launch_gray_check(const View &img,const MhImage *,unsigned int *flag)
{
  const size_t n=img.columns*img.rows;
  dim3 grid(stream_grid(n)),block(256);
  ProfileScope prof("gray_check",img.stream);
  if (img.quantum == MH_QUANTUM_U16)
    {
      if (img.channels == 3)
        hipLaunchKernelGGL((gray_check_kernel<uint16_t,3>),grid,block,0,img.stream,
          static_cast<const uint16_t *>(img.pixels),n,flag);
      else
        hipLaunchKernelGGL((gray_check_kernel<uint16_t,4>),grid,block,0,img.stream,
          static_cast<const uint16_t *>(img.pixels),n,flag);
    }
  else
    {
      if (img.channels == 3)
        hipLaunchKernelGGL((gray_check_kernel<float,3>),grid,block,0,img.stream,
          static_cast<const float *>(img.pixels),n,flag);
      else
        hipLaunchKernelGGL((gray_check_kernel<float,4>),grid,block,0,img.stream,
          static_cast<const float *>(img.pixels),n,flag);
    }
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// ------------------------------------------------------- unsharp epilogue
template<typename Q,int C>
__global__ __launch_bounds__(256)
void unsharp_kernel(const Q *src,const Q *blur,Q *dst,size_t npixels,double gain,
  double quantum_threshold,uint32_t copy_mask)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < npixels; i+=stride)
    {
      Q p[C],b[C],o[C];
      load_pixel<Q,C>(src+i*C,p);
      load_pixel<Q,C>(blur+i*C,b);
#pragma unroll
      for (int c=0; c < C; c++)
        {
          if ((copy_mask >> c) & 1u)
            {
              o[c]=p[c];
              continue;
            }
          // effect.c:4364-4369
          double pixel=(double) p[c]-(double) b[c];
          if (fabs(2.0*pixel) < quantum_threshold)
            pixel=(double) p[c];
          else
            pixel=(double) p[c]+gain*pixel;
          o[c]=QuantumOps<Q>::clamp(pixel);
        }
      store_pixel<Q,C>(dst+i*C,o);
    }
}

template<typename Q,int C>
static MhStatus unsharp_typed(const View &src,const View &blur,const View &dst,double gain,
  double threshold,uint32_t copy_mask)
{
  const size_t n=src.columns*src.rows;
  ProfileScope prof("unsharp_epilogue",src.stream);
  hipLaunchKernelGGL((unsharp_kernel<Q,C>),dim3(stream_grid(n)),dim3(256),0,src.stream,
    static_cast<const Q *>(src.pixels),static_cast<const Q *>(blur.pixels),
    static_cast<Q *>(dst.pixels),n,gain,kQuantumRange*threshold,copy_mask);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

MhStatus launch_unsharp_epilogue(const View &src,const View &blur,const View &dst,double gain,
  double threshold,const Roles &roles)
{
#define MH_CASE(QT) \
  switch (src.channels) { \
    case 1: return unsharp_typed<QT,1>(src,blur,dst,gain,threshold,roles.copy_mask); \
    case 2: return unsharp_typed<QT,2>(src,blur,dst,gain,threshold,roles.copy_mask); \
    case 3: return unsharp_typed<QT,3>(src,blur,dst,gain,threshold,roles.copy_mask); \
    default: return unsharp_typed<QT,4>(src,blur,dst,gain,threshold,roles.copy_mask); }
  if (src.quantum == MH_QUANTUM_U16)
    { MH_CASE(uint16_t) }
  MH_CASE(float)
#undef MH_CASE
}

// ------------------------------------------- separable 2-D convolution, FAST
// A rank-1 2-D kernel (Gaussian:RxS ...) is run as a row and a column pass over float sums.
// The reference's 2-D loop (morphology.c:2892-2979) forms sum k*alpha*p and sum k*alpha over
// the whole window and divides once, so the passes must carry the *undivided* sums:
// premultiply_kernel writes alpha*p (alpha = QuantumScale*a) and a as floats, the two 1-D
// passes convolve those as plain channels, separable_finish_kernel divides and quantises.
template<int C,bool BLEND>
__global__ __launch_bounds__(256)
void premultiply_kernel(const uint16_t *src,float *dst,size_t npixels)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < npixels; i+=stride)
    {
      uint16_t p[C];
      float o[C];
      load_pixel<uint16_t,C>(src+i*C,p);
      const float alpha=BLEND ? (float) p[C-1]*(1.0f/65535.0f) : 1.0f;
#pragma unroll
      for (int c=0; c < C; c++)
        o[c]=BLEND && (c < C-1) ? alpha*(float) p[c] : (float) p[c];
      store_pixel<float,C>(dst+i*C,o);
    }
}

template<int C,bool BLEND>
__global__ __launch_bounds__(256)
void separable_finish_kernel(const float *sums,uint16_t *dst,size_t npixels)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < npixels; i+=stride)
    {
      float v[C];
      uint16_t o[C];
      load_pixel<float,C>(sums+i*C,v);
      // gamma = 1/sum(k*alpha) = 65535/v[alpha]; an all-transparent window has zero sums: 0
      const float gamma=BLEND ? (v[C-1] > 0.0f ? 65535.0f/v[C-1] : 0.0f) : 1.0f;
#pragma unroll
      for (int c=0; c < C; c++)
        {
          const float pixel=BLEND && (c < C-1) ? gamma*v[c] : v[c];
          const float q=floorf(pixel+0.5f);
          o[c]=(uint16_t) (q < 0.0f ? 0.0f : (q > 65535.0f ? 65535.0f : q));
        }
      store_pixel<uint16_t,C>(dst+i*C,o);
    }
}

MhStatus launch_premultiply(const View &src,const View &sums,bool blend)
{
  const size_t n=src.columns*src.rows;
  const uint16_t *in=static_cast<const uint16_t *>(src.pixels);
  float *out=static_cast<float *>(sums.pixels);
  ProfileScope prof("premultiply",src.stream);
#define MH_CASE(CV,BV) \
  hipLaunchKernelGGL((premultiply_kernel<CV,BV>),dim3(stream_grid(n)),dim3(256),0,src.stream,in,out,n)
  switch (src.channels)
  {
    case 1: MH_CASE(1,false); break;
    case 2: if (blend) MH_CASE(2,true); else MH_CASE(2,false); break;
    case 3: MH_CASE(3,false); break;
    case 4: if (blend) MH_CASE(4,true); else MH_CASE(4,false); break;
    default: return fail(MH_UNSUPPORTED,"separable convolution: %d channels",src.channels);
  }
#undef MH_CASE
  MH_HIP(hipGetLastError());
  return MH_OK;
}

MhStatus launch_separable_finish(const View &sums,const View &dst,bool blend)
{
  const size_t n=dst.columns*dst.rows;
  const float *in=static_cast<const float *>(sums.pixels);
  uint16_t *out=static_cast<uint16_t *>(dst.pixels);
  ProfileScope prof("separable_finish",dst.stream);
#define MH_CASE(CV,BV) \
  hipLaunchKernelGGL((separable_finish_kernel<CV,BV>),dim3(stream_grid(n)),dim3(256),0,dst.stream,in,out,n)
  switch (dst.channels)
  {
    case 1: MH_CASE(1,false); break;
    case 2: if (blend) MH_CASE(2,true); else MH_CASE(2,false); break;
    case 3: MH_CASE(3,false); break;
    case 4: if (blend) MH_CASE(4,true); else MH_CASE(4,false); break;
    default: return fail(MH_UNSUPPORTED,"separable convolution: %d channels",dst.channels);
  }
#undef MH_CASE
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// ------------------------------------------------------------------ one-channel frames as four row bands
// A gray Q16 frame has no wide pixel to hand the one-launch blur kernels (convolve_fused_hybrid.hip,
// convolve_fused_exact.hip: 8-byte pixels, four independent channels).  Its rows cut into four bands ARE four
// independent channels of a frame a quarter as tall: channel c of packed row r is source row c*band + r - halo,
// clamped into the frame (the virtual pixels of cache.c:2663-2679 above the first and below the last row; between
// bands, the neighbouring band's real rows).  The kernels' own edge clamp then only ever decides packed rows
// [0, halo) and [band+halo, band+2*halo), which unpacking drops.
__global__ __launch_bounds__(256)
void gray_bands_pack_kernel(const uint16_t *src,uint16_t *dst,int W,int H,int band,int halo)
{
  const int x=(int) (blockIdx.x*blockDim.x+threadIdx.x);
  const int r=(int) blockIdx.y;
  if (x >= W)
    return;
  uint16_t p[4];
#pragma unroll
  for (int c=0; c < 4; c++)
    {
      int y=c*band+r-halo;
      y=y < 0 ? 0 : (y > H-1 ? H-1 : y);
      p[c]=src[(size_t) y*(size_t) W+(size_t) x];
    }
  store_pixel<uint16_t,4>(dst+((size_t) r*(size_t) W+(size_t) x)*4,p);
}

// ... and back; `changed` (MorphologyPrimitive's count, morphology.c:3199): the samples that differ from `original`
__global__ __launch_bounds__(256)
void gray_bands_unpack_kernel(const uint16_t *src,uint16_t *dst,int W,int H,int band,int halo,
  const uint16_t *original,unsigned long long *changed)
{
  const int x=(int) (blockIdx.x*blockDim.x+threadIdx.x);
  const int r=(int) blockIdx.y;
  unsigned differ=0;
  if (x < W)
    {
      uint16_t p[4];
      load_pixel<uint16_t,4>(src+((size_t) (r+halo)*(size_t) W+(size_t) x)*4,p);
#pragma unroll
      for (int c=0; c < 4; c++)
        {
          const int y=c*band+r;
          if (y < H)
            {
              const size_t at=(size_t) y*(size_t) W+(size_t) x;
              if (changed != nullptr)
                differ+=original[at] != p[c] ? 1u : 0u;
              dst[at]=p[c];
            }
        }
    }
  if (changed != nullptr)
    {
      differ=wave_sum(differ);
      if (((threadIdx.x & 63) == 0) && (differ != 0))
        atomicAdd(changed,(unsigned long long) differ);
    }
}

MhStatus launch_gray_bands_pack(const View &src,const View &packed,int band,int halo)
{
  ProfileScope prof("gray_bands_pack",src.stream);
  hipLaunchKernelGGL(gray_bands_pack_kernel,dim3((unsigned) ((src.columns+255)/256),(unsigned) (band+2*halo)),dim3(256),0,
    src.stream,static_cast<const uint16_t *>(src.pixels),static_cast<uint16_t *>(packed.pixels),(int) src.columns,
    (int) src.rows,band,halo);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

MhStatus launch_gray_bands_unpack(const View &packed,const View &dst,int band,int halo,const void *original,
  unsigned long long *changed)
{
  ProfileScope prof("gray_bands_unpack",dst.stream);
  hipLaunchKernelGGL(gray_bands_unpack_kernel,dim3((unsigned) ((dst.columns+255)/256),(unsigned) band),dim3(256),0,
    dst.stream,static_cast<const uint16_t *>(packed.pixels),static_cast<uint16_t *>(dst.pixels),(int) dst.columns,
    (int) dst.rows,band,halo,static_cast<const uint16_t *>(original),changed);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

// ------------------------------------------------------------------ three-channel frames with a fourth, empty one
// RGB without alpha — 6-byte (Q16) or 12-byte (float) pixels — has no vector load of its own either; the
// union-of-rectangles kernel (morphology.hip) takes 8- and 16-byte pixels.  Padded to four channels (the fourth 0) it
// is an ordinary four-channel frame; minima and maxima are per channel.
template<typename Q>
__global__ __launch_bounds__(256)
void rgb_pad_kernel(const Q *src,Q *dst,size_t npixels)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < npixels; i+=stride)
    {
      Q p[3],o[4];
      load_pixel<Q,3>(src+i*3,p);
      o[0]=p[0]; o[1]=p[1]; o[2]=p[2]; o[3]=(Q) 0;
      store_pixel<Q,4>(dst+i*4,o);
    }
}

// ... and back; `changed` counts the samples with |result - original| >= MagickEpsilon (morphology.c:3195-3199; a
// NaN difference is not one)
template<typename Q>
__global__ __launch_bounds__(256)
void rgb_unpad_kernel(const Q *src,Q *dst,size_t npixels,const Q *original,unsigned long long *changed)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  unsigned differ=0;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < npixels; i+=stride)
    {
      Q p[4],o[3];
      load_pixel<Q,4>(src+i*4,p);
      o[0]=p[0]; o[1]=p[1]; o[2]=p[2];
      if (changed != nullptr)
        {
          Q was[3];
          load_pixel<Q,3>(original+i*3,was);
#pragma unroll
          for (int c=0; c < 3; c++)
            differ+=fabs((double) o[c]-(double) was[c]) >= 1.0e-12 ? 1u : 0u;
        }
      store_pixel<Q,3>(dst+i*3,o);
    }
  if (changed != nullptr)
    {
      differ=wave_sum(differ);
      if (((threadIdx.x & 63) == 0) && (differ != 0))
        atomicAdd(changed,(unsigned long long) differ);
    }
}

MhStatus launch_rgb_pad(const View &src,const View &padded)
{
  const size_t n=src.columns*src.rows;
  ProfileScope prof("rgb_pad",src.stream);
  if (src.quantum == MH_QUANTUM_U16)
    hipLaunchKernelGGL(rgb_pad_kernel<uint16_t>,dim3(stream_grid(n)),dim3(256),0,src.stream,
      static_cast<const uint16_t *>(src.pixels),static_cast<uint16_t *>(padded.pixels),n);
  else
    hipLaunchKernelGGL(rgb_pad_kernel<float>,dim3(stream_grid(n)),dim3(256),0,src.stream,
      static_cast<const float *>(src.pixels),static_cast<float *>(padded.pixels),n);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

MhStatus launch_rgb_unpad(const View &padded,const View &dst,const void *original,unsigned long long *changed)
{
  const size_t n=dst.columns*dst.rows;
  ProfileScope prof("rgb_unpad",dst.stream);
  if (dst.quantum == MH_QUANTUM_U16)
    hipLaunchKernelGGL((rgb_unpad_kernel<uint16_t>),dim3(stream_grid(n)),dim3(256),0,dst.stream,
      static_cast<const uint16_t *>(padded.pixels),static_cast<uint16_t *>(dst.pixels),n,
      static_cast<const uint16_t *>(original),changed);
  else
    hipLaunchKernelGGL((rgb_unpad_kernel<float>),dim3(stream_grid(n)),dim3(256),0,dst.stream,
      static_cast<const float *>(padded.pixels),static_cast<float *>(dst.pixels),n,
      static_cast<const float *>(original),changed);
  MH_HIP(hipGetLastError());
  return MH_OK;
}

MhStatus launch_copy(const View &src,const View &dst)
{
  MH_HIP(hipMemcpyAsync(dst.pixels,src.pixels,src.bytes(),hipMemcpyDeviceToDevice,src.stream));
  return MH_OK;
}

#include "pixel_io.inc.hpp"

} // namespace mh
