// The pointwise colourspaces of sRGBTransformImage / TransformsRGBImage beyond sRGB <-> linear
// RGB / Lab / XYZ (MagickCore/colorspace.c:958-1051, :2292-2389): ConvertRGBToGeneric (:411-595)
// and ConvertGenericToRGB (:122-305) for CMY, HCL, HCLp, HSB, HSI, HSL, HSV, HWB, LCHab (LCH),
// LCHuv, Luv, LMS, CAT02LMS, xyY, YCbCr, YPbPr, YDbDr, YIQ, YUV, Jzazbz, OkLab, OkLCH, Adobe98,
// DisplayP3 and ProPhoto, and ModulateImage's colour models built on them
// (MagickCore/enhance.c:3462-3632).  Included by pointwise.hip after its gamma / XYZ / Lab / HSB
// / HSL helpers.
//
// One kernel with a wave-uniform switch on the colourspace: these are streams of fp64
// arithmetic, the branch costs nothing next to a pow or an atan2.  Every expression keeps the
// reference's association (-ffp-contract=off); signed coefficients are stored with their sign
// (x-c*y == x+(-c)*y exactly).  pow / cbrt / atan2 / hypot / sin / cos / fmod are the device
// library's: a last-bit difference of a double survives Quantum rounding only on an exact
// tie (Q16) and as at most one float ULP (float Quantum), which is what the tests allow.
//
// Formulas: MagickCore/colorspace-private.h — the line of each is cited at its restatement.

// ---------------------------------------------------------------- small helpers
static __device__ __forceinline__ double cs_min3(double a,double b,double c)
{
  const double bc=b < c ? b : c;               // MagickMin(a,MagickMin(b,c))
  return a < bc ? a : bc;
}

static __device__ __forceinline__ double cs_max3(double a,double b,double c)
{
  const double bc=b > c ? b : c;
  return a > bc ? a : bc;
}

// a 3x3 matrix applied to (u,v,w), rows in the reference's left-to-right association
struct Matrix3
{
  double m[3][3];
};

static __device__ __forceinline__ void cs_mul(const Matrix3 &t,double u,double v,double w,double &x,
  double &y,double &z)
{
  x=t.m[0][0]*u+t.m[0][1]*v+t.m[0][2]*w;
  y=t.m[1][0]*u+t.m[1][1]*v+t.m[1][2]*w;
  z=t.m[2][0]*u+t.m[2][1]*v+t.m[2][2]*w;
}

// ---------------------------------------------------------------- luma / colour-difference spaces
// Y = QuantumScale*(0.298839 R + 0.586811 G + 0.114350 B); the two difference channels are
// QuantumScale*(row . RGB)+0.5.  colorspace-private.h:1551-1587 (forward), :1637-1701 (inverse:
// QuantumRange*(cY*Y + c1*(P-0.5) + c2*(Q-0.5)) per output channel).
struct LumaSpace
{
  double forward[2][3];        // rows of the two difference channels
  double inverse[3][3];        // per output channel: coefficient of Y, of (P-0.5), of (Q-0.5)
};

static __device__ __forceinline__ void luma_forward(const LumaSpace &s,double R,double G,double B,
  double &Y,double &P,double &Q)
{
  Y=kQS*(0.298839*R+0.586811*G+0.114350*B);
  P=kQS*(s.forward[0][0]*R+s.forward[0][1]*G+s.forward[0][2]*B)+0.5;
  Q=kQS*(s.forward[1][0]*R+s.forward[1][1]*G+s.forward[1][2]*B)+0.5;
}

static __device__ __forceinline__ void luma_inverse(const LumaSpace &s,bool scaled_luma,double Y,
  double P,double Q,double &R,double &G,double &B)
{
  // YPbPr / YCbCr multiply Y by a constant next to 1; the others add Y itself (1.0*Y == Y)
  const double p=P-0.5,q=Q-0.5;
  if (scaled_luma)
    {
      R=kQR*(s.inverse[0][0]*Y+s.inverse[0][1]*p+s.inverse[0][2]*q);
      G=kQR*(s.inverse[1][0]*Y+s.inverse[1][1]*p+s.inverse[1][2]*q);
      B=kQR*(s.inverse[2][0]*Y+s.inverse[2][1]*p+s.inverse[2][2]*q);
      return;
    }
  R=kQR*(Y+s.inverse[0][1]*p+s.inverse[0][2]*q);
  G=kQR*(Y+s.inverse[1][1]*p+s.inverse[1][2]*q);
  B=kQR*(Y+s.inverse[2][1]*p+s.inverse[2][2]*q);
}

static __device__ __forceinline__ LumaSpace luma_space(int colorspace)
{
  switch (colorspace)
  {
    case MH_COLORSPACE_YDBDR:
      return LumaSpace{{{-0.450,-0.883,1.333},{-1.333,1.116,0.217}},
        {{1.0,9.2303716147657e-05,-0.52591263066186533},{1.0,-0.12913289889050927,0.26789932820759876},
         {1.0,0.66467905997895482,-7.9202543533108e-05}}};
    case MH_COLORSPACE_YIQ:
      return LumaSpace{{{0.595716,-0.274453,-0.321263},{0.211456,-0.522591,0.311135}},
        {{1.0,0.9562957197589482261,0.6210244164652610754},{1.0,-0.2721220993185104464,-0.6473805968256950427},
         {1.0,-1.1069890167364901945,1.7046149983646481374}}};
    case MH_COLORSPACE_YUV:
      return LumaSpace{{{-0.147,-0.289,0.436},{0.615,-0.515,-0.100}},
        {{1.0,-3.945707070708279e-05,1.1398279671717170825},{1.0,-0.3946101641414141437,-0.5805003156565656797},
         {1.0,2.0319996843434342537,-4.813762626262513e-04}}};
    default:                                     // YPbPr, YCbCr
      return LumaSpace{{{-0.1687367,-0.331264,0.5},{0.5,-0.418688,-0.081312}},
        {{0.99999999999914679361,-1.2188941887145875e-06,1.4019995886561440468},
         {0.99999975910502514331,-0.34413567816504303521,-0.71413649331646789076},
         {1.00000124040004623180,1.77200006607230409200,2.1453384174593273e-06}}};
  }
}

// ---------------------------------------------------------------- hue / chroma models
// ConvertRGBToHCL / HCLp, colorspace-private.h:801-866 (the two are the same function)
static __device__ void rgb_to_hcl(double R,double G,double B,double &hue,double &chroma,double &luma)
{
  const double top=cs_max3(R,G,B);
  const double c=top-cs_min3(R,G,B);
  double h=0.0;
  if (fabs(c) < kEps)
    h=0.0;
  else if (fabs(R-top) < kEps)
    h=fmod((G-B)/c+6.0,6.0);
  else if (fabs(G-top) < kEps)
    h=((B-R)/c)+2.0;
  else if (fabs(B-top) < kEps)
    h=((R-G)/c)+4.0;
  hue=h/6.0;
  chroma=kQS*c;
  luma=kQS*(0.298839*R+0.586811*G+0.114350*B);
}

// the sextant walk shared by ConvertHCLToRGB and ConvertHCLpToRGB (:149-291)
static __device__ __forceinline__ void hcl_sextant(double hue,double c,double &r,double &g,double &b)
{
  const double h=6.0*hue;
  const double x=c*(1.0-fabs(fmod(h,2.0)-1.0));
  r=0.0;
  g=0.0;
  b=0.0;
  if ((0.0 <= h) && (h < 1.0))
    { r=c; g=x; }
  else if ((1.0 <= h) && (h < 2.0))
    { r=x; g=c; }
  else if ((2.0 <= h) && (h < 3.0))
    { g=c; b=x; }
  else if ((3.0 <= h) && (h < 4.0))
    { g=x; b=c; }
  else if ((4.0 <= h) && (h < 5.0))
    { r=x; b=c; }
  else if ((5.0 <= h) && (h < 6.0))
    { r=c; b=x; }
}

static __device__ void hcl_to_rgb(double hue,double chroma,double luma,double &R,double &G,double &B)
{
  double r,g,b;
  hcl_sextant(hue,chroma,r,g,b);
  const double m=luma-(0.298839*r+0.586811*g+0.114350*b);
  R=kQR*(r+m);
  G=kQR*(g+m);
  B=kQR*(b+m);
}

static __device__ void hclp_to_rgb(double hue,double chroma,double luma,double &R,double &G,double &B)
{
  double r,g,b;
  hcl_sextant(hue,chroma,r,g,b);
  double m=luma-(0.298839*r+0.586811*g+0.114350*b);
  double z=1.0;
  if (m < 0.0)
    {
      z=luma/(luma-m);
      m=0.0;
    }
  else if (m+chroma > 1.0)
    {
      z=(1.0-luma)/(m+chroma-luma);
      m=1.0-z*chroma;
    }
  R=kQR*(z*r+m);
  G=kQR*(z*g+m);
  B=kQR*(z*b+m);
}

// ConvertRGBToHSI :909-936, ConvertHSIToRGB :368-412
static __device__ void rgb_to_hsi(double R,double G,double B,double &hue,double &saturation,
  double &intensity)
{
  intensity=(kQS*R+kQS*G+kQS*B)/3.0;
  if (intensity <= 0.0)
    {
      hue=0.0;
      saturation=0.0;
      return;
    }
  saturation=1.0-cs_min3(kQS*R,kQS*G,kQS*B)/intensity;
  const double alpha=0.5*(2.0*kQS*R-kQS*G-kQS*B);
  const double beta=0.8660254037844385*(kQS*G-kQS*B);
  hue=atan2(beta,alpha)*(180.0/kPi)/360.0;
  if (hue < 0.0)
    hue+=1.0;
}

static __device__ void hsi_to_rgb(double hue,double saturation,double intensity,double &R,double &G,
  double &B)
{
  double h=360.0*hue;
  h-=360.0*floor(h/360.0);
  double r,g,b;
  if (h < 120.0)
    {
      b=intensity*(1.0-saturation);
      r=intensity*(1.0+saturation*cos(h*(kPi/180.0))/cos((60.0-h)*(kPi/180.0)));
      g=3.0*intensity-r-b;
    }
  else if (h < 240.0)
    {
      h-=120.0;
      r=intensity*(1.0-saturation);
      g=intensity*(1.0+saturation*cos(h*(kPi/180.0))/cos((60.0-h)*(kPi/180.0)));
      b=3.0*intensity-r-g;
    }
  else
    {
      h-=240.0;
      g=intensity*(1.0-saturation);
      b=intensity*(1.0+saturation*cos(h*(kPi/180.0))/cos((60.0-h)*(kPi/180.0)));
      r=3.0*intensity-g-b;
    }
  R=kQR*r;
  G=kQR*g;
  B=kQR*b;
}

// ConvertRGBToHSV :994-1033, ConvertHSVToRGB :414-481
static __device__ void rgb_to_hsv(double R,double G,double B,double &hue,double &saturation,double &value)
{
  const double r=kQS*R,g=kQS*G,b=kQS*B;
  const double top=cs_max3(r,g,b),bottom=cs_min3(r,g,b);
  const double c=top-bottom;
  value=top;
  if (c <= 0.0)
    {
      hue=0.0;
      saturation=0.0;
      return;
    }
  if (fabs(top-r) < kEps)
    {
      hue=(g-b)/c;
      if (g < b)
        hue+=6.0;
    }
  else if (fabs(top-g) < kEps)
    hue=2.0+(b-r)/c;
  else
    hue=4.0+(r-g)/c;
  hue*=60.0/360.0;
  saturation=c*perceptible_reciprocal(top);
}

static __device__ void hsv_to_rgb(double hue,double saturation,double value,double &R,double &G,double &B)
{
  double h=hue*360.0;
  const double c=value*saturation;
  const double bottom=value-c;
  h-=360.0*floor(h/360.0);
  h/=60.0;
  const double x=c*(1.0-fabs(h-2.0*floor(h/2.0)-1.0));
  switch ((int) floor(h))
  {
    case 1: R=kQR*(bottom+x); G=kQR*(bottom+c); B=kQR*bottom; break;
    case 2: R=kQR*bottom; G=kQR*(bottom+c); B=kQR*(bottom+x); break;
    case 3: R=kQR*bottom; G=kQR*(bottom+x); B=kQR*(bottom+c); break;
    case 4: R=kQR*(bottom+x); G=kQR*bottom; B=kQR*(bottom+c); break;
    case 5: R=kQR*(bottom+c); G=kQR*bottom; B=kQR*(bottom+x); break;
    default: R=kQR*(bottom+c); G=kQR*(bottom+x); B=kQR*bottom; break;      // 0
  }
}

// ConvertRGBToHWB :1035-1064, ConvertHWBToRGB :483-529
static __device__ void rgb_to_hwb(double R,double G,double B,double &hue,double &whiteness,double &blackness)
{
  const double w=cs_min3(R,G,B),v=cs_max3(R,G,B);
  blackness=1.0-kQS*v;
  whiteness=kQS*w;
  if (fabs(v-w) < kEps)
    {
      hue=-1.0;
      return;
    }
  const double f=(fabs(R-w) < kEps) ? G-B : ((fabs(G-w) < kEps) ? B-R : R-G);
  const double p=(fabs(R-w) < kEps) ? 3.0 : ((fabs(G-w) < kEps) ? 5.0 : 1.0);
  hue=(p-f/(v-1.0*w))/6.0;
}

static __device__ void hwb_to_rgb(double hue,double whiteness,double blackness,double &R,double &G,double &B)
{
  const double v=1.0-blackness;
  if (fabs(hue-(-1.0)) < kEps)
    {
      R=kQR*v;
      G=kQR*v;
      B=kQR*v;
      return;
    }
  // CastDoubleToLong(floor(6*hue)): hue comes from a Quantum in [0,1] (or a modulated one)
  const double sextant=floor(6.0*hue);
  const long long i=sextant >= 9.2e18 ? 0x7fffffffffffffffLL : (sextant <= -9.2e18 ? (-0x7fffffffffffffffLL-1) :
    (long long) sextant);
  double f=6.0*hue-(double) i;
  if ((i & 0x01) != 0)
    f=1.0-f;
  const double n=whiteness+f*(v-whiteness);
  double r,g,b;
  switch (i)
  {
    case 1: r=n; g=v; b=whiteness; break;
    case 2: r=whiteness; g=v; b=n; break;
    case 3: r=whiteness; g=n; b=v; break;
    case 4: r=n; g=whiteness; b=v; break;
    case 5: r=v; g=whiteness; b=n; break;
    default: r=v; g=n; b=whiteness; break;         // 0
  }
  R=kQR*r;
  G=kQR*g;
  B=kQR*b;
}

// ---------------------------------------------------------------- Lab / Luv polar forms
// ConvertXYZToLCHab :1104-1117, ConvertLCHabToRGB :572-598
static __device__ void xyz_to_lchab(double X,double Y,double Z,double &luma,double &chroma,double &hue)
{
  double a,b;
  xyz_to_lab(X,Y,Z,luma,a,b);
  chroma=hypot(a-0.5,b-0.5)/1.0+0.5;
  hue=180.0*atan2(b-0.5,a-0.5)/kPi/360.0;
  if (hue < 0.0)
    hue+=1.0;
}

static __device__ __forceinline__ double cs_radians(double degrees)
{
  return kPi*degrees/180.0;                      // DegreesToRadians, image-private.h
}

static __device__ void lchab_to_rgb(double luma,double chroma,double hue,double &R,double &G,double &B)
{
  const double l=100.0*luma,c=255.0*(chroma-0.5),h=360.0*hue;
  double X,Y,Z;
  lab_to_xyz(l,c*cos(cs_radians(h)),c*sin(cs_radians(h)),X,Y,Z);
  xyz_to_rgb(X,Y,Z,R,G,B);
}

// ConvertXYZToLuv :1138-1161, ConvertLuvToXYZ :600-625 (D65)
static __device__ void xyz_to_luv(double X,double Y,double Z,double &L,double &u,double &v)
{
  if ((Y/MH_ILL_Y) > MH_CIE_EPSILON)
    L=116.0*pow(Y/MH_ILL_Y,1.0/3.0)-16.0;
  else
    L=MH_CIE_K*(Y/MH_ILL_Y);
  const double alpha=perceptible_reciprocal(X+15.0*Y+3.0*Z);
  u=13.0*L*((4.0*alpha*X)-(4.0*MH_ILL_X/(MH_ILL_X+15.0*MH_ILL_Y+3.0*MH_ILL_Z)));
  v=13.0*L*((9.0*alpha*Y)-(9.0*MH_ILL_Y/(MH_ILL_X+15.0*MH_ILL_Y+3.0*MH_ILL_Z)));
  L/=100.0;
  u=(u+134.0)/354.0;
  v=(v+140.0)/262.0;
}

static __device__ void luv_to_xyz(double L,double u,double v,double &X,double &Y,double &Z)
{
  if (L > (MH_CIE_K*MH_CIE_EPSILON))
    Y=pow((L+16.0)/116.0,3.0);
  else
    Y=L/MH_CIE_K;
  const double un=4.0*MH_ILL_X/(MH_ILL_X+15.0*MH_ILL_Y+3.0*MH_ILL_Z);
  const double vn=9.0*MH_ILL_Y/(MH_ILL_X+15.0*MH_ILL_Y+3.0*MH_ILL_Z);
  const double gamma=perceptible_reciprocal((((52.0*L*perceptible_reciprocal(u+13.0*L*un))-1.0)/3.0)-
    (-1.0/3.0));
  X=gamma*((Y*((39.0*L*perceptible_reciprocal(v+13.0*L*vn))-5.0))+5.0*Y);
  Z=(X*(((52.0*L*perceptible_reciprocal(u+13.0*L*un))-1.0)/3.0))-5.0*Y;
}

// ConvertXYZToLCHuv :1163-1176, ConvertLCHuvToRGB :627-653
static __device__ void xyz_to_lchuv(double X,double Y,double Z,double &luma,double &chroma,double &hue)
{
  double u,v;
  xyz_to_luv(X,Y,Z,luma,u,v);
  chroma=hypot(354.0*u-134.0,262.0*v-140.0)/255.0+0.5;
  hue=180.0*atan2(262.0*v-140.0,354.0*u-134.0)/kPi/360.0;
  if (hue < 0.0)
    hue+=1.0;
}

static __device__ void lchuv_to_rgb(double luma,double chroma,double hue,double &R,double &G,double &B)
{
  const double l=100.0*luma,c=255.0*(chroma-0.5),h=360.0*hue;
  double X,Y,Z;
  luv_to_xyz(l,c*cos(cs_radians(h)),c*sin(cs_radians(h)),X,Y,Z);
  xyz_to_rgb(X,Y,Z,R,G,B);
}

// ---------------------------------------------------------------- Jzazbz, colorspace-private.h:1274-1478
static __device__ void xyz_to_jzazbz(double X,double Y,double Z,double white_luminance,double &Jz,
  double &az,double &bz)
{
  const double b=1.15,g=0.66,c1=3424.0/4096.0,c2=2413.0/128.0,c3=2392.0/128.0,n=2610.0/16384.0,
    p=1.7*2523.0/32.0,d=-0.56,d0=1.6295499532821566e-11;
  const double WLr=perceptible_reciprocal(white_luminance);
  const double Xp=Z+b*(X-Z);
  const double Yp=X+g*(Y-X);
  double L=0.0146480*Z,M=0.0531008*Z,S=0.6684799*Z;
  L+=0.41478972*Xp;
  M+=(-0.2015100)*Xp;
  S+=(-0.0166008)*Xp;
  L+=0.579999*Yp;
  M+=1.120649*Yp;
  S+=0.264800*Yp;
  const double gL=pow(L*WLr,n),gM=pow(M*WLr,n),gS=pow(S*WLr,n);
  const double nL=c1+c2*gL,nM=c1+c2*gM,nS=c1+c2*gS;
  const double dL=1.0+c3*gL,dM=1.0+c3*gM,dS=1.0+c3*gS;
  const double Lp=pow(nL/dL,p),Mp=pow(nM/dM,p),Sp=pow(nS/dS,p);
  const double Iz=(Lp+Mp)*0.5;
  const double JdI=d*Iz;
  const double J=(JdI+Iz)/(JdI+1.0)-d0;
  double a=0.5+3.52400*Lp,bb=0.5+0.199076*Lp;
  a+=(-4.066708)*Mp;
  bb+=1.096799*Mp;
  a+=0.542708*Sp;
  bb+=(-1.295875)*Sp;
  Jz=(J != J) ? 0.0 : J;
  az=(a != a) ? 0.5 : a;
  bz=(bb != bb) ? 0.5 : bb;
}

static __device__ void jzazbz_to_xyz(double Jz,double az,double bz,double white_luminance,double &X,
  double &Y,double &Z)
{
  const double b=1.15,g0=0.66,c1=3424.0/4096.0,c2=2413.0/128.0,mc3=-2392.0/128.0,n=2610.0/16384.0,
    p=1.7*2523.0/32.0,d=-0.56,d0=1.6295499532821566e-11;
  const double g=Jz+d0;
  const double azz=az-0.5,bzz=bz-0.5;
  const double C=0.138605043271539*azz+0.0580473161561189*bzz;
  double Sp=g/(1.0+d*(1.0-g));
  const double Lp=Sp+C,Mp=Sp-C;
  Sp+=(-0.0960192420263189)*azz;
  Sp+=(-0.811891896056039)*bzz;
  const double Jpr=1.0/p;
  const double gL=pow(Lp,Jpr),gM=pow(Mp,Jpr),gS=pow(Sp,Jpr);
  const double Jnr=1.0/n;
  const double nL=gL-c1,nM=gM-c1,nS=gS-c1;
  const double dL=c2+mc3*gL,dM=c2+mc3*gM,dS=c2+mc3*gS;
  double L=pow(nL/dL,Jnr),M=pow(nM/dM,Jnr),S=pow(nS/dS,Jnr);
  L*=white_luminance;
  M*=white_luminance;
  S*=white_luminance;
  double Zp=(-0.0909828109828476)*L,Xp=1.92422643578761*L,Yp=0.350316762094999*L;
  Zp+=(-0.312728290523074)*M;
  Xp+=(-1.00479231259537)*M;
  Yp+=0.726481193931655*M;
  Zp+=1.52276656130526*S;
  Xp+=0.037651404030618*S;
  Yp+=(-0.065384422948085)*S;
  Zp=(Zp != Zp) ? 0.0 : Zp;
  Xp=Zp+(Xp-Zp)/b;
  Xp=(Xp != Xp) ? 0.0 : Xp;
  Yp=Xp+(Yp-Xp)/g0;
  Yp=(Yp != Yp) ? 0.0 : Yp;
  Z=Zp;
  X=Xp;
  Y=Yp;
}

// ---------------------------------------------------------------- OkLab / OkLCH, :1480-1549
static __device__ void rgb_to_oklab(double R,double G,double B,double &L,double &a,double &b)
{
  const double r=kQS*decode_pixel_gamma(R),g=kQS*decode_pixel_gamma(G),bl=kQS*decode_pixel_gamma(B);
  const double l=cbrt(0.4122214708*r+0.5363325363*g+0.0514459929*bl);
  const double m=cbrt(0.2119034982*r+0.6806995451*g+0.1073969566*bl);
  const double s=cbrt(0.0883024619*r+0.2817188376*g+0.6299787005*bl);
  L=0.2104542553*l+0.7936177850*m+(-0.0040720468)*s;
  a=1.9779984951*l+(-2.4285922050)*m+0.4505937099*s+0.5;
  b=0.0259040371*l+0.7827717662*m+(-0.8086757660)*s+0.5;
}

static __device__ void oklab_to_rgb(double L,double a,double b,double &R,double &G,double &B)
{
  double l=L+0.3963377774*(a-0.5)+0.2158037573*(b-0.5);
  double m=L+(-0.1055613458)*(a-0.5)+(-0.0638541728)*(b-0.5);
  double s=L+(-0.0894841775)*(a-0.5)+(-1.2914855480)*(b-0.5);
  l*=l*l;
  m*=m*m;
  s*=s*s;
  const double r=4.0767416621*l+(-3.3077115913)*m+0.2309699292*s;
  const double g=(-1.2684380046)*l+2.6097574011*m+(-0.3413193965)*s;
  const double bl=(-0.0041960863)*l+(-0.7034186147)*m+1.7076147010*s;
  R=encode_pixel_gamma(kQR*r);
  G=encode_pixel_gamma(kQR*g);
  B=encode_pixel_gamma(kQR*bl);
}

// ---------------------------------------------------------------- RGB working spaces through XYZ
// Adobe98 :53-70 / :938-964, DisplayP3 :675-704 / :966-992, ProPhoto :719-749 / :1197-1223
static __device__ __forceinline__ void working_space(int colorspace,Matrix3 &to_xyz,Matrix3 &from_xyz)
{
  switch (colorspace)
  {
    case MH_COLORSPACE_ADOBE98:
      to_xyz=Matrix3{{{0.57666904291013050,0.18555823790654630,0.18822864623499470},
        {0.29734497525053605,0.62736356625546610,0.07529145849399788},
        {0.02703136138641234,0.07068885253582723,0.99133753683763880}}};
      from_xyz=Matrix3{{{2.041587903810746500,-0.56500697427885960,-0.34473135077832956},
        {-0.969243636280879500,1.87596750150772020,0.04155505740717557},
        {0.013444280632031142,-0.11836239223101838,1.01517499439120540}}};
      break;
    case MH_COLORSPACE_DISPLAYP3:
      to_xyz=Matrix3{{{0.4865709486482162,0.26566769316909306,0.1982172852343625},
        {0.2289745640697488,0.69173852183650640,0.0792869140937450},
        {0.0000000000000000,0.04511338185890264,1.0439443689009760}}};
      from_xyz=Matrix3{{{2.49349691194142500,-0.93138361791912390,-0.402710784450716840},
        {-0.82948896956157470,1.76266406031834630,0.023624685841943577},
        {0.03584583024378447,-0.07617238926804182,0.956884524007687200}}};
      break;
    default:                                     // ProPhoto
      to_xyz=Matrix3{{{0.7977604896723027,0.13518583717574031,0.03134934958152480000},
        {0.2880711282292934,0.71184321781010140,0.00008565396060525902},
        {0.0000000000000000,0.00000000000000000,0.82510460251046010000}}};
      from_xyz=Matrix3{{{1.3457989731028281,-0.25558010007997534,-0.05110628506753401},
        {-0.5446224939028347,1.50823274131327810,0.02053603239147973},
        {0.0000000000000000,0.0000000000000000,1.21196754563894540}}};
      break;
  }
}

// ConvertXYZToLMS :1225-1231 == ConvertXYZToCAT02LMS :751-757; ConvertLMSToXYZ :655-661 ==
// ConvertCAT02LMSToXYZ :108-117
static __device__ __forceinline__ void xyz_to_lms(double X,double Y,double Z,double &L,double &M,double &S)
{
  L=0.7328*X+0.4296*Y+(-0.1624)*Z;
  M=(-0.7036)*X+1.6975*Y+0.0061*Z;
  S=0.0030*X+0.0136*Y+0.9834*Z;
}

static __device__ __forceinline__ void lms_to_xyz(double L,double M,double S,double &X,double &Y,double &Z)
{
  X=1.096123820835514*L+(-0.278869000218287)*M+0.182745179382773*S;
  Y=0.454369041975359*L+0.473533154307412*M+0.072097803717229*S;
  Z=(-0.009627608738429)*L+(-0.005698031216113)*M+1.015325639954543*S;
}

// ---------------------------------------------------------------- the two dispatchers
// ConvertRGBToGeneric, colorspace.c:411-595: R,G,B in Quantum units -> three components,
// nominally in [0,1] (the caller stores ClampToQuantum(QuantumRange*component)).
static __device__ void rgb_to_generic(int colorspace,double R,double G,double B,double white_luminance,
  double &c0,double &c1,double &c2)
{
  double X,Y,Z;
  switch (colorspace)
  {
    case MH_COLORSPACE_CMY:                      // :793-799
      c0=kQS*(kQR-R);
      c1=kQS*(kQR-G);
      c2=kQS*(kQR-B);
      return;
    case MH_COLORSPACE_HCL: case MH_COLORSPACE_HCLP:
      rgb_to_hcl(R,G,B,c0,c1,c2);
      return;
    case MH_COLORSPACE_HSB:
      rgb_to_hsb(R,G,B,c0,c1,c2);
      return;
    case MH_COLORSPACE_HSI:
      rgb_to_hsi(R,G,B,c0,c1,c2);
      return;
    case MH_COLORSPACE_HSL:
      rgb_to_hsl(R,G,B,c0,c1,c2);
      return;
    case MH_COLORSPACE_HSV:
      rgb_to_hsv(R,G,B,c0,c1,c2);
      return;
    case MH_COLORSPACE_HWB:
      rgb_to_hwb(R,G,B,c0,c1,c2);
      return;
    case MH_COLORSPACE_YCBCR: case MH_COLORSPACE_YPBPR: case MH_COLORSPACE_YDBDR:
    case MH_COLORSPACE_YIQ: case MH_COLORSPACE_YUV:
      luma_forward(luma_space(colorspace),R,G,B,c0,c1,c2);
      return;
    case MH_COLORSPACE_OKLAB:
      rgb_to_oklab(R,G,B,c0,c1,c2);
      return;
    case MH_COLORSPACE_OKLCH:                    // :1539-1549
      {
        double a,b;
        rgb_to_oklab(R,G,B,c0,a,b);
        c1=sqrt(a*a+b*b);
        c2=0.5+0.5*atan2(-b,-a)/kPi;
        return;
      }
    case MH_COLORSPACE_JZAZBZ:                   // :1365-1376: green and blue change places
      rgb_to_xyz(R,B,G,X,Y,Z);
      xyz_to_jzazbz(X,Y,Z,white_luminance,c0,c1,c2);
      return;
    default:
      break;
  }
  rgb_to_xyz(R,G,B,X,Y,Z);
  switch (colorspace)
  {
    case MH_COLORSPACE_LAB:
      xyz_to_lab(X,Y,Z,c0,c1,c2);
      return;
    case MH_COLORSPACE_LCH: case MH_COLORSPACE_LCHAB:
      xyz_to_lchab(X,Y,Z,c0,c1,c2);
      return;
    case MH_COLORSPACE_LCHUV:
      xyz_to_lchuv(X,Y,Z,c0,c1,c2);
      return;
    case MH_COLORSPACE_LUV:
      xyz_to_luv(X,Y,Z,c0,c1,c2);
      return;
    case MH_COLORSPACE_LMS:
      xyz_to_lms(X,Y,Z,c0,c1,c2);
      return;
    case MH_COLORSPACE_CAT02LMS:                 // colorspace.c:422-431: to LMS and back to XYZ
      {
        double L,M,S;
        xyz_to_lms(X,Y,Z,L,M,S);
        lms_to_xyz(L,M,S,c0,c1,c2);
        return;
      }
    case MH_COLORSPACE_XYY:                      // :1258-1272
      {
        const double gamma=perceptible_reciprocal(X+Y+Z);
        c0=gamma*X;
        c1=gamma*Y;
        c2=Y;
        return;
      }
    case MH_COLORSPACE_ADOBE98: case MH_COLORSPACE_DISPLAYP3: case MH_COLORSPACE_PROPHOTO:
      {
        Matrix3 to_xyz,from_xyz;
        working_space(colorspace,to_xyz,from_xyz);
        double r,g,b;
        cs_mul(from_xyz,X,Y,Z,r,g,b);
        c0=kQS*encode_pixel_gamma(kQR*r);
        c1=kQS*encode_pixel_gamma(kQR*g);
        c2=kQS*encode_pixel_gamma(kQR*b);
        return;
      }
    default:                                     // XYZ
      c0=X;
      c1=Y;
      c2=Z;
      return;
  }
}

// ConvertGenericToRGB, colorspace.c:122-305: components (QuantumScale*stored value) -> R,G,B in
// Quantum units.
static __device__ void generic_to_rgb(int colorspace,double c0,double c1,double c2,double white_luminance,
  double &R,double &G,double &B)
{
  double X,Y,Z;
  switch (colorspace)
  {
    case MH_COLORSPACE_CMY:                      // :141-147
      R=kQR*(1.0-c0);
      G=kQR*(1.0-c1);
      B=kQR*(1.0-c2);
      return;
    case MH_COLORSPACE_HCL:
      hcl_to_rgb(c0,c1,c2,R,G,B);
      return;
    case MH_COLORSPACE_HCLP:
      hclp_to_rgb(c0,c1,c2,R,G,B);
      return;
    case MH_COLORSPACE_HSB:
      hsb_to_rgb(c0,c1,c2,R,G,B);
      return;
    case MH_COLORSPACE_HSI:
      hsi_to_rgb(c0,c1,c2,R,G,B);
      return;
    case MH_COLORSPACE_HSL:
      hsl_to_rgb(c0,c1,c2,R,G,B);
      return;
    case MH_COLORSPACE_HSV:
      hsv_to_rgb(c0,c1,c2,R,G,B);
      return;
    case MH_COLORSPACE_HWB:
      hwb_to_rgb(c0,c1,c2,R,G,B);
      return;
    case MH_COLORSPACE_YCBCR: case MH_COLORSPACE_YPBPR:
      luma_inverse(luma_space(colorspace),true,c0,c1,c2,R,G,B);
      return;
    case MH_COLORSPACE_YDBDR: case MH_COLORSPACE_YIQ: case MH_COLORSPACE_YUV:
      luma_inverse(luma_space(colorspace),false,c0,c1,c2,R,G,B);
      return;
    case MH_COLORSPACE_OKLAB:
      oklab_to_rgb(c0,c1,c2,R,G,B);
      return;
    case MH_COLORSPACE_OKLCH:                    // :1527-1537
      oklab_to_rgb(c0,c1*cos(2.0*kPi*c2),c1*sin(2.0*kPi*c2),R,G,B);
      return;
    case MH_COLORSPACE_LCH: case MH_COLORSPACE_LCHAB:
      lchab_to_rgb(c0,c1,c2,R,G,B);
      return;
    case MH_COLORSPACE_LCHUV:
      lchuv_to_rgb(c0,c1,c2,R,G,B);
      return;
    case MH_COLORSPACE_JZAZBZ:                   // :1467-1478: green and blue change places
      jzazbz_to_xyz(c0,c1,c2,white_luminance,X,Y,Z);
      xyz_to_rgb(X,Y,Z,R,B,G);
      return;
    case MH_COLORSPACE_LAB:
      lab_to_xyz(100.0*c0,255.0*(c1-0.5),255.0*(c2-0.5),X,Y,Z);
      break;
    case MH_COLORSPACE_LUV:                      // :706-717
      luv_to_xyz(100.0*c0,354.0*c1-134.0,262.0*c2-140.0,X,Y,Z);
      break;
    case MH_COLORSPACE_LMS:
      lms_to_xyz(c0,c1,c2,X,Y,Z);
      break;
    case MH_COLORSPACE_CAT02LMS:                 // colorspace.c:133-143
      {
        double L,M,S;
        xyz_to_lms(c0,c1,c2,L,M,S);
        lms_to_xyz(L,M,S,X,Y,Z);
        break;
      }
    case MH_COLORSPACE_XYY:                      // :1676-1690
      {
        const double gamma=perceptible_reciprocal(c1);
        X=gamma*c2*c0;
        Y=c2;
        Z=gamma*c2*(1.0-c0-c1);
        break;
      }
    case MH_COLORSPACE_ADOBE98: case MH_COLORSPACE_DISPLAYP3: case MH_COLORSPACE_PROPHOTO:
      {
        Matrix3 to_xyz,from_xyz;
        working_space(colorspace,to_xyz,from_xyz);
        const double r=kQS*decode_pixel_gamma(kQR*c0),g=kQS*decode_pixel_gamma(kQR*c1),
          b=kQS*decode_pixel_gamma(kQR*c2);
        cs_mul(to_xyz,r,g,b,X,Y,Z);
        break;
      }
    default:                                     // XYZ
      X=c0;
      Y=c1;
      Z=c2;
      break;
  }
  xyz_to_rgb(X,Y,Z,R,G,B);
}

// ---------------------------------------------------------------- kernels
// forward: sRGB -> colourspace (colorspace.c:1032-1043); inverse: colourspace -> sRGB (:2373-2380)
template<typename Q,int C,bool FORWARD>
__global__ __launch_bounds__(256)
void colorspace_generic_kernel(Q *__restrict__ pixels,size_t npixels,int colorspace,double white_luminance)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < npixels; i+=stride)
    {
      Q q[C];
      load_pixel<Q,C>(pixels+i*C,q);
      double o0,o1,o2;
      if (FORWARD)
        {
          double c0,c1,c2;
          rgb_to_generic(colorspace,(double) q[0],(double) q[1],(double) q[2],white_luminance,c0,c1,c2);
          o0=kQR*c0;
          o1=kQR*c1;
          o2=kQR*c2;
        }
      else
        generic_to_rgb(colorspace,kQS*(double) q[0],kQS*(double) q[1],kQS*(double) q[2],white_luminance,
          o0,o1,o2);
      q[0]=QuantumOps<Q>::clamp(o0);
      q[1]=QuantumOps<Q>::clamp(o1);
      q[2]=QuantumOps<Q>::clamp(o2);
      store_pixel<Q,C>(pixels+i*C,q);
    }
}

// ModulateImage's per-pixel step for any of its colour models (enhance.c:3462-3632, :3826-3890):
// RGB -> model, hue += shift, the two other components scaled, model -> RGB.
// LCHab / LCHuv keep (luma, chroma, hue) in that order; HWB scales whiteness by the saturation
// percentage and blackness by the brightness percentage.
template<typename Q,int C>
__global__ __launch_bounds__(256)
void modulate_generic_kernel(Q *__restrict__ pixels,size_t npixels,int colorspace,double hue_shift,
  double saturation_scale,double brightness_scale)
{
  const size_t stride=(size_t) gridDim.x*blockDim.x;
  for (size_t i=(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < npixels; i+=stride)
    {
      Q q[C];
      load_pixel<Q,C>(pixels+i*C,q);
      double c0,c1,c2,R,G,B;
      rgb_to_generic(colorspace,(double) q[0],(double) q[1],(double) q[2],10000.0,c0,c1,c2);
      switch (colorspace)
      {
        case MH_COLORSPACE_LCH: case MH_COLORSPACE_LCHAB: case MH_COLORSPACE_LCHUV:
          c0*=brightness_scale;                  // luma
          c1*=saturation_scale;                  // chroma
          c2+=hue_shift;
          break;
        case MH_COLORSPACE_HWB:
          c0+=hue_shift;
          c2*=brightness_scale;                  // blackness
          c1*=saturation_scale;                  // whiteness
          break;
        default:                                 // HCL, HCLp, HSB, HSI, HSL, HSV: (hue, saturation|chroma, third)
          c0+=hue_shift;
          c1*=saturation_scale;
          c2*=brightness_scale;
          break;
      }
      generic_to_rgb(colorspace,c0,c1,c2,10000.0,R,G,B);
      q[0]=QuantumOps<Q>::clamp(R);
      q[1]=QuantumOps<Q>::clamp(G);
      q[2]=QuantumOps<Q>::clamp(B);
      store_pixel<Q,C>(pixels+i*C,q);
    }
}
