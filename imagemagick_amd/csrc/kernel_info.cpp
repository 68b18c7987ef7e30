// Host-side morphology kernel builder: the product's own restatement of the
// parts of AcquireKernelInfo / AcquireKernelBuiltIn the hot path needs
// (MagickCore/morphology.c:210-560 parser, :950-1800 built-ins, :2485-2504
// CalcKernelMetaData, :4571-4632 ScaleKernelInfo, :4258-4429 RotateKernelInfo,
// MagickCore/gem.c:262-345 optimal widths).  In a MagickCore integration the
// shim hands the reference's own KernelInfo values to the operators, so this
// file only serves standalone callers (bench, tests, the Python binding); it
// is validated value-for-value against the compiled reference in
// tests/test_host.py (CPU) and tests/test_gpu_parity.py (the operators that use them).
//
// Every name of morphology.c's KernelInfoType table is built, including the FreiChen set, the
// Laplacian / LoG constants and the hit-and-miss families (Edges, Corners, Diagonals, LineEnds,
// LineJunctions, Ridges, ConvexHull, ThinSE, Skeleton) with their rotation expansions; a string
// that MagickCore's own parser would reject returns NULL here too (the caller falls back).
#include "mh_internal.hpp"

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <strings.h>

namespace {

constexpr double kPi = 3.14159265358979323846264338327950288419716939937510;
constexpr double k2Pi = 6.28318530717958647692528676655900576839433879875020;
constexpr double kSq2Pi = 2.50662827463100024161235523934010416269302368164062;
constexpr double kEpsilon = mh::kMagickEpsilon;
constexpr double kQScale = mh::kQuantumScale;

double perceptible_reciprocal(double x)
{
  double sign=x < 0.0 ? -1.0 : 1.0;
  if ((sign*x) >= kEpsilon)
    return 1.0/x;
  return sign/kEpsilon;
}

// ------------------------------------------------------ geometry arguments
enum : unsigned
{
  kRho=0x0004,kSigma=0x0008,kXi=0x0001,kPsi=0x0002,
  kPercent=0x1000,kAspect=0x2000,kLess=0x4000,kGreater=0x8000,kArea=0x10000
};

struct Geometry { double rho=0,sigma=0,xi=0,psi=0; unsigned flags=0; };

// strtod that reads "0x2" as 0 followed by the separator, not as hexadecimal
// (ParseGeometry special-cases the prefix the same way, geometry.c:1083-1098)
double geometry_number(const char *p,char **end)
{
  if ((p[0] == '0') && ((p[1] == 'x') || (p[1] == 'X')))
    {
      *end=const_cast<char *>(p)+1;
      return 0.0;
    }
  return strtod(p,end);
}

// "rho[x|,sigma][{+-,}xi[{+-,}psi]]" with the % ! < > @ meta characters, the
// subset of ParseGeometry (geometry.c:922) kernel arguments use.
bool parse_geometry(const std::string &text,Geometry &g)
{
  std::string s;
  for (char ch : text)
    {
      if (isspace((unsigned char) ch))
        continue;
      switch (ch)
      {
        case '%': g.flags|=kPercent; continue;
        case '!': g.flags|=kAspect; continue;
        case '<': g.flags|=kLess; continue;
        case '>': g.flags|=kGreater; continue;
        case '@': g.flags|=kArea; continue;
        case '^': case '#': case '(': case ')': continue;
        default: break;
      }
      if (isdigit((unsigned char) ch) || ch == '.' || ch == 'x' || ch == 'X' ||
          ch == ',' || ch == '+' || ch == '-' || ch == 'e' || ch == 'E')
        s.push_back(ch);
      else
        return false;
    }
  const char *p=s.c_str();
  if (*p == '\0')
    return true;
  char *q=nullptr;
  // rho: a number followed by x , or end
  (void) geometry_number(p,&q);
  if ((q != p) && ((*q == 'x') || (*q == 'X') || (*q == ',') || (*q == '\0')))
    {
      g.rho=geometry_number(p,&q);
      g.flags|=kRho;
      p=q;
    }
  if ((*p == 'x') || (*p == 'X') || (*p == ','))
    {
      char sep=*p;
      p++;
      if ((sep == ',') || ((*p != '+') && (*p != '-')))
        {
          double v=geometry_number(p,&q);
          if (q != p)
            {
              g.sigma=v;
              g.flags|=kSigma;
              p=q;
            }
        }
    }
  for (int which=0; which < 2; which++)
    {
      if ((*p != '+') && (*p != '-') && (*p != ','))
        break;
      bool negative=false;
      if (*p == ',')
        p++;
      while ((*p == '+') || (*p == '-'))
        {
          if (*p == '-')
            negative=!negative;
          p++;
        }
      double v=geometry_number(p,&q);
      if (q == p)
        break;
      p=q;
      if (negative)
        v=(-v);
      if (which == 0) { g.xi=v; g.flags|=kXi; }
      else { g.psi=v; g.flags|=kPsi; }
    }
  return true;
}

// ------------------------------------------------------------ kernel basics
MhKernelInfo *new_kernel(MhKernelInfoType type,size_t width,size_t height)
{
  MhKernelInfo *k=(MhKernelInfo *) calloc(1,sizeof(MhKernelInfo));
  if (k == nullptr)
    return nullptr;
  k->type=type;
  k->width=width;
  k->height=height;
  k->values=(double *) calloc(width*height > 0 ? width*height : 1,sizeof(double));
  if (k->values == nullptr)
    {
      free(k);
      return nullptr;
    }
  return k;
}

void destroy_chain(MhKernelInfo *k)
{
  while (k != nullptr)
    {
      MhKernelInfo *next=k->next;
      free(k->values);
      free(k);
      k=next;
    }
}

MhKernelInfo *last_kernel(MhKernelInfo *k)
{
  while (k->next != nullptr)
    k=k->next;
  return k;
}

// CalcKernelMetaData, morphology.c:2485-2504
void calc_meta(MhKernelInfo *k)
{
  k->minimum=k->maximum=0.0;
  k->negative_range=k->positive_range=0.0;
  for (size_t i=0; i < k->width*k->height; i++)
    {
      double &v=k->values[i];
      if (fabs(v) < kEpsilon)
        v=0.0;
      if (v < 0)
        k->negative_range+=v;
      else
        k->positive_range+=v;
      if (v < k->minimum) k->minimum=v;
      if (v > k->maximum) k->maximum=v;
    }
}

// ScaleKernelInfo, morphology.c:4571-4632.  flags: 1 Normalize, 2 CorrelateNormalize
void scale_kernel(MhKernelInfo *k,double factor,unsigned flags)
{
  if (k->next != nullptr)
    scale_kernel(k->next,factor,flags);
  double pos=1.0,neg;
  if ((flags & 1u) != 0)
    {
      if (fabs(k->positive_range+k->negative_range) >= kEpsilon)
        pos=fabs(k->positive_range+k->negative_range);
      else
        pos=k->positive_range;
    }
  if ((flags & 2u) != 0)
    {
      pos=fabs(k->positive_range) >= kEpsilon ? k->positive_range : 1.0;
      neg=fabs(k->negative_range) >= kEpsilon ? -k->negative_range : 1.0;
    }
  else
    neg=pos;
  pos=factor/pos;
  neg=factor/neg;
  for (size_t i=0; i < k->width*k->height; i++)
    if (!std::isnan(k->values[i]))
      k->values[i]*=(k->values[i] >= 0) ? pos : neg;
  k->positive_range*=pos;
  k->negative_range*=neg;
  k->maximum*=(k->maximum >= 0.0) ? pos : neg;
  k->minimum*=(k->minimum >= 0.0) ? pos : neg;
  if (factor < kEpsilon)
    {
      double t=k->positive_range;
      k->positive_range=k->negative_range;
      k->negative_range=t;
      k->maximum=k->minimum;
      k->minimum=1;
    }
}

// RotateKernelInfo, morphology.c:4258-4429
void rotate_kernel(MhKernelInfo *k,double angle)
{
  if (k->next != nullptr)
    rotate_kernel(k->next,angle);
  angle=fmod(angle,360.0);
  if (angle < 0)
    angle+=360.0;
  if ((337.5 < angle) || (angle <= 22.5))
    return;
  switch (k->type)
  {
    case MH_KERNEL_GAUSSIAN: case MH_KERNEL_DOG: case MH_KERNEL_LOG:
    case MH_KERNEL_DISK: case MH_KERNEL_PEAKS: case MH_KERNEL_LAPLACIAN:
    case MH_KERNEL_CHEBYSHEV: case MH_KERNEL_MANHATTAN: case MH_KERNEL_EUCLIDEAN:
    case MH_KERNEL_SQUARE: case MH_KERNEL_DIAMOND: case MH_KERNEL_PLUS:
    case MH_KERNEL_CROSS:
      return;
    case MH_KERNEL_BLUR:
      if ((135.0 < angle) && (angle <= 225.0))
        return;
      if ((225.0 < angle) && (angle <= 315.0))
        angle-=180;
      break;
    default:
      break;
  }
  const ptrdiff_t w=(ptrdiff_t) k->width,h=(ptrdiff_t) k->height;
  double *v=k->values;
  if ((22.5 < fmod(angle,90.0)) && (fmod(angle,90.0) <= 67.5))
    {
      if ((w == 3) && (h == 3))
        {
          // cycle the outer ring one step
          static const int ring[8]={0,3,6,7,8,5,2,1};
          double t=v[ring[0]];
          for (int i=0; i < 7; i++)
            v[ring[i]]=v[ring[i+1]];
          v[ring[7]]=t;
          if ((k->x != 1) || (k->y != 1))
            {
              ptrdiff_t x=k->x-1,y=k->y-1;
              if (x == y) x=0;
              else if (x == 0) x=-y;
              else if (x == -y) y=0;
              else if (y == 0) y=x;
              k->x=x+1;
              k->y=y+1;
            }
          angle=fmod(angle+315.0,360.0);
          k->angle=fmod(k->angle+45.0,360.0);
        }
    }
  if ((45.0 < fmod(angle,180.0)) && (fmod(angle,180.0) <= 135.0))
    {
      if ((w == 1) || (h == 1))
        {
          // transpose of a 1-D kernel
          size_t t=k->width; k->width=k->height; k->height=t;
          ptrdiff_t o=k->x; k->x=k->y; k->y=o;
          if (k->width == 1)
            {
              angle=fmod(angle+270.0,360.0);
              k->angle=fmod(k->angle+90.0,360.0);
            }
          else
            {
              angle=fmod(angle+90.0,360.0);
              k->angle=fmod(k->angle+270.0,360.0);
            }
        }
      else if (w == h)
        {
          for (ptrdiff_t i=0,x=w-1; i <= x; i++,x--)
            for (ptrdiff_t j=0,y=h-1; j < y; j++,y--)
              {
                double t=v[i+j*w];
                v[i+j*w]=v[j+x*w];
                v[j+x*w]=v[x+y*w];
                v[x+y*w]=v[y+i*w];
                v[y+i*w]=t;
              }
          ptrdiff_t x=k->x*2-w+1,y=k->y*2-h+1;
          k->x=(-y+w-1)/2;
          k->y=(+x+h-1)/2;
          angle=fmod(angle+270.0,360.0);
          k->angle=fmod(k->angle+90.0,360.0);
        }
    }
  if ((135.0 < angle) && (angle <= 225.0))
    {
      for (ptrdiff_t i=0,j=(ptrdiff_t) (k->width*k->height)-1; i < j; i++,j--)
        {
          double t=v[i]; v[i]=v[j]; v[j]=t;
        }
      k->x=(ptrdiff_t) k->width-k->x-1;
      k->y=(ptrdiff_t) k->height-k->y-1;
      k->angle=fmod(k->angle+180.0,360.0);
    }
}

MhKernelInfo *clone_one(const MhKernelInfo *k)
{
  MhKernelInfo *c=new_kernel(k->type,k->width,k->height);
  if (c == nullptr)
    return nullptr;
  memcpy(c->values,k->values,k->width*k->height*sizeof(double));
  c->x=k->x; c->y=k->y;
  c->minimum=k->minimum; c->maximum=k->maximum;
  c->negative_range=k->negative_range; c->positive_range=k->positive_range;
  c->angle=k->angle;
  return c;
}

bool same_kernel(const MhKernelInfo *a,const MhKernelInfo *b)
{
  if ((a->width != b->width) || (a->height != b->height) || (a->x != b->x) ||
      (a->y != b->y))
    return false;
  for (size_t i=0; i < a->width*a->height; i++)
    {
      bool na=std::isnan(a->values[i]),nb=std::isnan(b->values[i]);
      if (na != nb)
        return false;
      if (!na && (fabs(a->values[i]-b->values[i]) >= kEpsilon))
        return false;
    }
  return true;
}

MhKernelInfo *clone_chain(const MhKernelInfo *k)
{
  MhKernelInfo *head=nullptr,*tail=nullptr;
  for (; k != nullptr; k=k->next)
    {
      MhKernelInfo *c=clone_one(k);
      if (c == nullptr)
        {
          destroy_chain(head);
          return nullptr;
        }
      if (head == nullptr) head=c; else tail->next=c;
      tail=c;
    }
  return head;
}

// ExpandRotateKernelInfo, morphology.c:2424-2450: append rotated clones (of the whole
// list from the last appended group on, as CloneKernelInfo clones a chain) until the
// rotation returns to the first kernel.
void expand_rotate(MhKernelInfo *kernel,double angle)
{
  MhKernelInfo *last=kernel;
  for (int guard=0; guard < 16; guard++)
    {
      MhKernelInfo *c=clone_chain(last);
      if (c == nullptr)
        return;
      rotate_kernel(c,angle);
      if (same_kernel(kernel,c))
        {
          destroy_chain(c);
          return;
        }
      last_kernel(last)->next=c;
      last=c;
    }
}

// ExpandMirrorKernelInfo, morphology.c:2332-2361: the list, its 180-degree rotation,
// the transpose of that, and the 180-degree rotation of the transpose.
void expand_mirror(MhKernelInfo *kernel)
{
  const double angles[3]={180.0,90.0,180.0};
  MhKernelInfo *last=kernel;
  for (double angle : angles)
    {
      MhKernelInfo *c=clone_chain(last);
      if (c == nullptr)
        return;
      rotate_kernel(c,angle);
      last_kernel(last)->next=c;
      last=c;
    }
}

// ------------------------------------------------------- user-defined arrays
// ParseKernelArray, morphology.c:210-372
MhKernelInfo *parse_array(const std::string &text)
{
  Geometry g;
  size_t colon=text.find(':');
  std::string body;
  size_t width=0,height=0;
  ptrdiff_t ox=0,oy=0;
  if (colon != std::string::npos)
    {
      if (!parse_geometry(text.substr(0,colon),g))
        return nullptr;
      if ((g.flags & kRho) == 0)
        g.rho=g.sigma;
      if (g.rho < 1.0)
        g.rho=1.0;
      if (g.sigma < 1.0)
        g.sigma=g.rho;
      width=(size_t) g.rho;
      height=(size_t) g.sigma;
      if ((g.xi < 0.0) || (g.psi < 0.0))
        return nullptr;
      ox=(g.flags & kXi) != 0 ? (ptrdiff_t) g.xi : (ptrdiff_t) (width-1)/2;
      oy=(g.flags & kPsi) != 0 ? (ptrdiff_t) g.psi : (ptrdiff_t) (height-1)/2;
      if ((ox >= (ptrdiff_t) width) || (oy >= (ptrdiff_t) height))
        return nullptr;
      body=text.substr(colon+1);
    }
  else
    body=text;
  // tokenise values: separated by spaces and/or commas; "nan" and "-" are holes
  std::vector<double> values;
  const char *p=body.c_str();
  while (*p != '\0')
    {
      while (isspace((unsigned char) *p) || (*p == ',') || (*p == '\''))
        p++;
      if (*p == '\0')
        break;
      const char *start=p;
      while ((*p != '\0') && !isspace((unsigned char) *p) && (*p != ',') && (*p != '\''))
        p++;
      std::string token(start,(size_t) (p-start));
      if ((strcasecmp(token.c_str(),"nan") == 0) || (token == "-"))
        values.push_back(std::numeric_limits<double>::quiet_NaN());
      else
        {
          char *end=nullptr;
          double v=strtod(token.c_str(),&end);
          if ((end == token.c_str()) || (*end != '\0'))
            return nullptr;
          values.push_back(v);
        }
    }
  if (colon == std::string::npos)
    {
      // old style: odd square kernel sized from the value count
      width=height=(size_t) sqrt((double) values.size()+1.0);
      ox=oy=(ptrdiff_t) (width-1)/2;
    }
  if ((width == 0) || (height == 0) || (values.size() != width*height))
    return nullptr;
  MhKernelInfo *k=new_kernel(MH_KERNEL_USERDEFINED,width,height);
  if (k == nullptr)
    return nullptr;
  k->x=ox;
  k->y=oy;
  k->minimum=std::numeric_limits<double>::max();
  k->maximum=-std::numeric_limits<double>::max();
  bool any=false;
  for (size_t i=0; i < values.size(); i++)
    {
      k->values[i]=values[i];
      if (std::isnan(values[i]))
        continue;
      any=true;
      if (values[i] < 0) k->negative_range+=values[i];
      else k->positive_range+=values[i];
      if (values[i] < k->minimum) k->minimum=values[i];
      if (values[i] > k->maximum) k->maximum=values[i];
    }
  if (!any)
    {
      destroy_chain(k);
      return nullptr;
    }
  if ((g.flags & kArea) != 0)
    expand_rotate(k,45.0);
  else if ((g.flags & kGreater) != 0)
    expand_rotate(k,90.0);
  else if ((g.flags & kLess) != 0)
    expand_mirror(k);
  return k;
}

// ----------------------------------------------------------------- built-ins
struct NameEntry { const char *name; MhKernelInfoType type; };
const NameEntry kNames[]=
{
  {"Unity",MH_KERNEL_UNITY},{"Gaussian",MH_KERNEL_GAUSSIAN},{"DoG",MH_KERNEL_DOG},
  {"LoG",MH_KERNEL_LOG},{"Blur",MH_KERNEL_BLUR},{"Comet",MH_KERNEL_COMET},
  {"Binomial",MH_KERNEL_BINOMIAL},{"Laplacian",MH_KERNEL_LAPLACIAN},
  {"Sobel",MH_KERNEL_SOBEL},{"FreiChen",MH_KERNEL_FREICHEN},
  {"Roberts",MH_KERNEL_ROBERTS},{"Prewitt",MH_KERNEL_PREWITT},
  {"Compass",MH_KERNEL_COMPASS},{"Kirsch",MH_KERNEL_KIRSCH},
  {"Diamond",MH_KERNEL_DIAMOND},{"Square",MH_KERNEL_SQUARE},
  {"Rectangle",MH_KERNEL_RECTANGLE},{"Disk",MH_KERNEL_DISK},
  {"Octagon",MH_KERNEL_OCTAGON},{"Plus",MH_KERNEL_PLUS},{"Cross",MH_KERNEL_CROSS},
  {"Ring",MH_KERNEL_RING},{"Peaks",MH_KERNEL_PEAKS},{"Edges",MH_KERNEL_EDGES},
  {"Corners",MH_KERNEL_CORNERS},{"Diagonals",MH_KERNEL_DIAGONALS},
  {"LineEnds",MH_KERNEL_LINEENDS},{"LineJunctions",MH_KERNEL_LINEJUNCTIONS},
  {"Ridges",MH_KERNEL_RIDGES},{"ConvexHull",MH_KERNEL_CONVEXHULL},
  {"ThinSe",MH_KERNEL_THINSE},{"Skeleton",MH_KERNEL_SKELETON},
  {"Chebyshev",MH_KERNEL_CHEBYSHEV},{"Manhattan",MH_KERNEL_MANHATTAN},
  {"Octagonal",MH_KERNEL_OCTAGONAL},{"Euclidean",MH_KERNEL_EUCLIDEAN}
};

size_t factorial(size_t n)
{
  size_t f=1;
  for (size_t i=2; i <= n; i++)
    f*=i;
  return f;
}

MhKernelInfo *constant_kernel(MhKernelInfoType type,const char *array,double angle)
{
  MhKernelInfo *k=parse_array(array);
  if (k == nullptr)
    return nullptr;
  k->type=type;
  rotate_kernel(k,angle);
  return k;
}

MhKernelInfo *acquire_kernel_list(const char *kernel_string);

void retype_chain(MhKernelInfo *k,MhKernelInfoType type)
{
  for (; k != nullptr; k=k->next)
    k->type=type;
}

// 3x3 hit-and-miss structuring element from 9 cells in raster order:
// '1' foreground, '0' background, '-' don't care (NaN); origin at the centre.
MhKernelInfo *structuring_element(MhKernelInfoType type,const char *cells)
{
  MhKernelInfo *k=new_kernel(type,3,3);
  if (k == nullptr)
    return nullptr;
  k->x=k->y=1;
  for (int i=0; i < 9; i++)
    k->values[i]=cells[i] == '-' ? std::numeric_limits<double>::quiet_NaN() :
      (cells[i] == '1' ? 1.0 : 0.0);
  calc_meta(k);
  return k;
}

// The thinning structuring elements of D. S. Bloomberg, "Connectivity-Preserving
// Morphological Image Transformations" (SE_4_n -> 4n, SE_8_n -> 8n, the combined ones
// 423/823/481/482), as ThinSE:<id> names them (morphology.c:1998-2087).
struct ThinElement { int id; const char *cells; };
const ThinElement kThinElements[]={
  {41,"--10-1--1"},{42,"--10-1-0-"},{43,"-0-0-1--1"},{44,"-0-0-1-0-"},{45,"-010-1-0-"},
  {46,"-0-0-1-01"},{47,"-110-1-0-"},{48,"--10-10-1"},{49,"0-10-1--1"},
  {81,"-1-0-1-1-"},{82,"-1-0-10--"},{83,"0--0-1-1-"},{84,"0--0-10--"},{85,"0-10-10--"},
  {86,"0--0-10-1"},{87,"-1-0-100-"},{88,"-1-0-101-"},{89,"01-0-1-1-"},
  {423,"--10---0-"},{823,"-1---10--"},{481,"-110-100-"},{482,"0-10-10-1"}
};

MhKernelInfo *thin_element(MhKernelInfoType type,int id,double angle)
{
  const char *cells="0-10-10-1";                  // 482, also the default
  for (const ThinElement &e : kThinElements)
    if (e.id == id)
      cells=e.cells;
  MhKernelInfo *k=structuring_element(type,cells);
  if (k != nullptr)
    rotate_kernel(k,angle);
  return k;
}

// a list of structuring elements given as cell strings, typed `type`
MhKernelInfo *element_list(MhKernelInfoType type,std::initializer_list<const char *> cells)
{
  MhKernelInfo *head=nullptr;
  for (const char *c : cells)
    {
      MhKernelInfo *k=structuring_element(type,c);
      if (k == nullptr)
        {
          destroy_chain(head);
          return nullptr;
        }
      if (head == nullptr) head=k; else last_kernel(head)->next=k;
    }
  return head;
}

MhKernelInfo *array_typed(MhKernelInfoType type,const char *array)
{
  MhKernelInfo *k=parse_array(array);
  if (k != nullptr)
    k->type=type;
  return k;
}

// FreiChen, morphology.c:1416-1538: Sobel-like edge kernels with sqrt(2) weights and
// the nine-kernel orthogonal basis (11..19)
MhKernelInfo *frei_chen(const Geometry &args)
{
  const double sq2=1.41421356237309504880168872420969807856967187537695;
  const MhKernelInfoType type=MH_KERNEL_FREICHEN;
  MhKernelInfo *k=nullptr;
  struct Patch { int index; double value; };
  auto build=[&](const char *array,std::initializer_list<Patch> patches,bool recalc,double scale)
  {
    MhKernelInfo *r=array_typed(type,array);
    if (r == nullptr)
      return r;
    for (const Patch &p : patches)
      r->values[p.index]=p.value;
    if (recalc)
      calc_meta(r);
    if (scale != 0.0)
      scale_kernel(r,scale,0u);
    return r;
  };
  const double half_sq2=(double) (1.0/2.0*sq2);
  switch ((int) args.rho)
  {
    default:
    case 0: k=build("3: 1,0,-1  2,0,-2  1,0,-1",{{3,sq2},{5,-sq2}},true,0.0); break;
    case 2: k=build("3: 1,2,0  2,0,-2  0,-2,-1",{{1,sq2},{3,sq2},{5,-sq2},{7,-sq2}},true,half_sq2); break;
    case 10:
      return acquire_kernel_list("FreiChen:11;FreiChen:12;FreiChen:13;FreiChen:14;FreiChen:15;"
        "FreiChen:16;FreiChen:17;FreiChen:18;FreiChen:19");
    case 1:
    case 11: k=build("3: 1,0,-1  2,0,-2  1,0,-1",{{3,sq2},{5,-sq2}},true,half_sq2); break;
    case 12: k=build("3: 1,2,1  0,0,0  1,2,1",{{1,sq2},{7,sq2}},true,half_sq2); break;
    case 13: k=build("3: 2,-1,0  -1,0,1  0,1,-2",{{0,sq2},{8,-sq2}},true,half_sq2); break;
    case 14: k=build("3: 0,1,-2  -1,0,1  2,-1,0",{{2,-sq2},{6,sq2}},true,half_sq2); break;
    case 15: k=build("3: 0,-1,0  1,0,1  0,-1,0",{},false,1.0/2.0); break;
    case 16: k=build("3: 1,0,-1  0,0,0  -1,0,1",{},false,1.0/2.0); break;
    case 17: k=build("3: 1,-2,1  -2,4,-2  -1,-2,1",{},false,1.0/6.0); break;
    case 18: k=build("3: -2,1,-2  1,4,1  -2,1,-2",{},false,1.0/6.0); break;
    case 19: k=build("3: 1,1,1  1,1,1  1,1,1",{},false,1.0/3.0); break;
  }
  if (k == nullptr)
    return nullptr;
  if (fabs(args.sigma) >= kEpsilon)
    rotate_kernel(k,args.sigma);                   // the angle argument
  else if ((args.rho > 30.0) || (args.rho < -30.0))
    rotate_kernel(k,args.rho);                     // an out-of-range 'type' is an angle
  return k;
}

MhKernelInfo *builtin(MhKernelInfoType type,const Geometry &args)
{
  const double nan=std::numeric_limits<double>::quiet_NaN();
  MhKernelInfo *k=nullptr;
  switch (type)
  {
    case MH_KERNEL_UNITY:
    {
      k=new_kernel(type,1,1);
      if (k == nullptr) return nullptr;
      k->maximum=k->values[0]=args.rho;
      break;
    }
    case MH_KERNEL_GAUSSIAN:
    case MH_KERNEL_DOG:
    case MH_KERNEL_LOG:
    {
      // morphology.c:1045-1138
      double sigma=fabs(args.sigma),sigma2=fabs(args.xi);
      size_t width;
      if (args.rho >= 1.0)
        width=(size_t) args.rho*2+1;
      else if ((type != MH_KERNEL_DOG) || (sigma >= sigma2))
        width=MhGetOptimalKernelWidth2D(args.rho,sigma);
      else
        width=MhGetOptimalKernelWidth2D(args.rho,sigma2);
      k=new_kernel(type,width,width);
      if (k == nullptr) return nullptr;
      k->x=k->y=(ptrdiff_t) (width-1)/2;
      const ptrdiff_t r=k->x;
      const size_t centre=(size_t) (k->x+k->y*(ptrdiff_t) width);
      if ((type == MH_KERNEL_GAUSSIAN) || (type == MH_KERNEL_DOG))
        {
          if (sigma > kEpsilon)
            {
              double A=1.0/(2.0*sigma*sigma),B=(double) (1.0/(k2Pi*sigma*sigma));
              size_t i=0;
              for (ptrdiff_t v=-r; v <= r; v++)
                for (ptrdiff_t u=-r; u <= r; u++,i++)
                  k->values[i]=exp(-((double) (u*u+v*v))*A)*B;
            }
          else
            k->values[centre]=1.0;
        }
      if (type == MH_KERNEL_DOG)
        {
          if (sigma2 > kEpsilon)
            {
              sigma=sigma2;
              double A=1.0/(2.0*sigma*sigma),B=(double) (1.0/(k2Pi*sigma*sigma));
              size_t i=0;
              for (ptrdiff_t v=-r; v <= r; v++)
                for (ptrdiff_t u=-r; u <= r; u++,i++)
                  k->values[i]-=exp(-((double) (u*u+v*v))*A)*B;
            }
          else
            k->values[centre]-=1.0;
        }
      if (type == MH_KERNEL_LOG)
        {
          if (sigma > kEpsilon)
            {
              double A=1.0/(2.0*sigma*sigma),B=(double) (1.0/(kPi*sigma*sigma*sigma*sigma));
              size_t i=0;
              for (ptrdiff_t v=-r; v <= r; v++)
                for (ptrdiff_t u=-r; u <= r; u++,i++)
                  {
                    double R=((double) (u*u+v*v))*A;
                    k->values[i]=(1-R)*exp(-R)*B;
                  }
            }
          else
            k->values[centre]=1.0;
        }
      calc_meta(k);
      scale_kernel(k,1.0,2u);
      break;
    }
    case MH_KERNEL_BLUR:
    {
      // morphology.c:1140-1227 — 3x oversampled 1-D gaussian, binned
      double sigma=fabs(args.sigma);
      size_t width=args.rho >= 1.0 ? (size_t) args.rho*2+1 :
        MhGetOptimalKernelWidth1D(args.rho,sigma);
      k=new_kernel(type,width,1);
      if (k == nullptr) return nullptr;
      k->x=(ptrdiff_t) (width-1)/2;
      k->y=0;
      const int rank=3;
      if (sigma > kEpsilon)
        {
          ptrdiff_t v=(ptrdiff_t) (width*rank-1)/2;
          sigma*=rank;
          double alpha=1.0/(2.0*sigma*sigma),beta=(double) (1.0/(kSq2Pi*sigma));
          for (ptrdiff_t u=-v; u <= v; u++)
            k->values[(u+v)/rank]+=exp(-((double) (u*u))*alpha)*beta;
        }
      else
        k->values[k->x]=1.0;
      calc_meta(k);
      scale_kernel(k,1.0,2u);
      rotate_kernel(k,args.xi);
      break;
    }
    case MH_KERNEL_COMET:
    {
      // morphology.c:1228-1292
      double sigma=fabs(args.sigma);
      size_t width=args.rho < 1.0 ? (MhGetOptimalKernelWidth1D(args.rho,sigma)-1)/2+1 :
        (size_t) args.rho;
      k=new_kernel(type,width,1);
      if (k == nullptr) return nullptr;
      k->x=k->y=0;
      if (sigma > kEpsilon)
        {
          const int rank=3;
          ptrdiff_t v=(ptrdiff_t) width*rank;
          sigma*=rank;
          double A=1.0/(2.0*sigma*sigma);
          for (ptrdiff_t u=0; u < v; u++)
            k->values[u/rank]+=exp(-((double) (u*u))*A);
          for (size_t i=0; i < width; i++)
            k->positive_range+=k->values[i];
        }
      else
        {
          k->values[0]=1.0;
          k->positive_range=1.0;
        }
      k->minimum=0.0;
      k->maximum=k->values[0];
      k->negative_range=0.0;
      scale_kernel(k,1.0,1u);
      rotate_kernel(k,args.xi);
      break;
    }
    case MH_KERNEL_BINOMIAL:
    {
      size_t width=args.rho < 1.0 ? 3 : ((size_t) args.rho)*2+1;
      k=new_kernel(type,width,width);
      if (k == nullptr) return nullptr;
      k->x=k->y=(ptrdiff_t) (width-1)/2;
      size_t order=factorial(width-1),i=0;
      for (size_t v=0; v < width; v++)
        {
          size_t alpha=order/(factorial(v)*factorial(width-v-1));
          for (size_t u=0; u < width; u++,i++)
            k->positive_range+=k->values[i]=(double)
              (alpha*order/(factorial(u)*factorial(width-u-1)));
        }
      k->minimum=1.0;
      k->maximum=k->values[k->x+k->y*(ptrdiff_t) width];
      k->negative_range=0.0;
      break;
    }
    case MH_KERNEL_LAPLACIAN:
    {
      const char *array=nullptr;
      switch ((int) args.rho)
      {
        case 1: array="3: 0,-1,0  -1,4,-1  0,-1,0"; break;
        case 2: array="3: -2,1,-2  1,4,1  -2,1,-2"; break;
        case 3: array="3: 1,-2,1  -2,4,-2  1,-2,1"; break;
        case 5: array="5: -4,-1,0,-1,-4  -1,2,3,2,-1  0,3,4,3,0  -1,2,3,2,-1  -4,-1,0,-1,-4"; break;
        case 7: array="7:-10,-5,-2,-1,-2,-5,-10 -5,0,3,4,3,0,-5 -2,3,6,7,6,3,-2 -1,4,7,8,7,4,-1 "
          "-2,3,6,7,6,3,-2 -5,0,3,4,3,0,-5 -10,-5,-2,-1,-2,-5,-10"; break;
        case 15: array="5: 0,0,-1,0,0  0,-1,-2,-1,0  -1,-2,16,-2,-1  0,-1,-2,-1,0  0,0,-1,0,0"; break;
        case 19: array="9: 0,-1,-1,-2,-2,-2,-1,-1,0  -1,-2,-4,-5,-5,-5,-4,-2,-1  -1,-4,-5,-3,-0,-3,-5,-4,-1  "
          "-2,-5,-3,12,24,12,-3,-5,-2  -2,-5,-0,24,40,24,-0,-5,-2  -2,-5,-3,12,24,12,-3,-5,-2  "
          "-1,-4,-5,-3,-0,-3,-5,-4,-1  -1,-2,-4,-5,-5,-5,-4,-2,-1  0,-1,-1,-2,-2,-2,-1,-1,0"; break;
        default: array="3: -1,-1,-1  -1,8,-1  -1,-1,-1"; break;
      }
      k=parse_array(array);
      if (k == nullptr) return nullptr;
      k->type=type;
      break;
    }
    case MH_KERNEL_SOBEL:
      return constant_kernel(type,"3: 1,0,-1  2,0,-2  1,0,-1",args.rho);
    case MH_KERNEL_ROBERTS:
      return constant_kernel(type,"3: 0,0,0  1,-1,0  0,0,0",args.rho);
    case MH_KERNEL_PREWITT:
      return constant_kernel(type,"3: 1,0,-1  1,0,-1  1,0,-1",args.rho);
    case MH_KERNEL_COMPASS:
      return constant_kernel(type,"3: 1,1,-1  1,-2,-1  1,1,-1",args.rho);
    case MH_KERNEL_KIRSCH:
      return constant_kernel(type,"3: 5,-3,-3  5,0,-3  5,-3,-3",args.rho);
    case MH_KERNEL_DIAMOND:
    case MH_KERNEL_OCTAGON:
    case MH_KERNEL_PLUS:
    case MH_KERNEL_CROSS:
    {
      const size_t def=(type == MH_KERNEL_DIAMOND) ? 3 : 5;
      size_t width=args.rho < 1.0 ? def : ((size_t) args.rho)*2+1;
      k=new_kernel(type,width,width);
      if (k == nullptr) return nullptr;
      k->x=k->y=(ptrdiff_t) (width-1)/2;
      const ptrdiff_t r=k->x;
      size_t i=0;
      for (ptrdiff_t v=-r; v <= r; v++)
        for (ptrdiff_t u=-r; u <= r; u++,i++)
          {
            bool inside;
            if (type == MH_KERNEL_DIAMOND)
              inside=(labs((long) u)+labs((long) v)) <= (long) r;
            else if (type == MH_KERNEL_OCTAGON)
              inside=(labs((long) u)+labs((long) v)) <= ((long) r+(long) (r/2));
            else if (type == MH_KERNEL_PLUS)
              inside=(u == 0) || (v == 0);
            else
              inside=(u == v) || (u == -v);
            if (inside)
              {
                k->values[i]=args.sigma;
                if ((type == MH_KERNEL_DIAMOND) || (type == MH_KERNEL_OCTAGON))
                  k->positive_range+=args.sigma;
              }
            else
              k->values[i]=nan;
          }
      k->minimum=k->maximum=args.sigma;
      if ((type == MH_KERNEL_PLUS) || (type == MH_KERNEL_CROSS))
        k->positive_range=args.sigma*((double) width*2.0-1.0);
      break;
    }
    case MH_KERNEL_SQUARE:
    case MH_KERNEL_RECTANGLE:
    {
      double scale;
      size_t width,height;
      ptrdiff_t ox,oy;
      if (type == MH_KERNEL_SQUARE)
        {
          width=height=args.rho < 1.0 ? 3 : (size_t) (2*args.rho+1);
          ox=oy=(ptrdiff_t) (width-1)/2;
          scale=args.sigma;
        }
      else
        {
          if ((args.rho < 1.0) || (args.sigma < 1.0))
            return nullptr;
          width=(size_t) args.rho;
          height=(size_t) args.sigma;
          if ((args.xi < 0.0) || (args.xi > (double) width) ||
              (args.psi < 0.0) || (args.psi > (double) height))
            return nullptr;
          ox=(ptrdiff_t) args.xi;
          oy=(ptrdiff_t) args.psi;
          scale=1.0;
        }
      k=new_kernel(type,width,height);
      if (k == nullptr) return nullptr;
      k->x=ox;
      k->y=oy;
      for (size_t i=0; i < width*height; i++)
        k->values[i]=scale;
      k->minimum=k->maximum=scale;
      k->positive_range=scale*(double) (ptrdiff_t) (width*height);
      break;
    }
    case MH_KERNEL_DISK:
    {
      // morphology.c:1625-1649
      ptrdiff_t limit=(ptrdiff_t) (args.rho*args.rho);
      size_t width;
      if (args.rho < 0.4)
        {
          width=9;
          limit=18;
        }
      else
        width=(size_t) fabs(args.rho)*2+1;
      k=new_kernel(type,width,width);
      if (k == nullptr) return nullptr;
      k->x=k->y=(ptrdiff_t) (width-1)/2;
      const ptrdiff_t r=k->x;
      size_t i=0;
      for (ptrdiff_t v=-r; v <= r; v++)
        for (ptrdiff_t u=-r; u <= r; u++,i++)
          if ((u*u+v*v) <= limit)
            k->positive_range+=k->values[i]=args.sigma;
          else
            k->values[i]=nan;
      k->minimum=k->maximum=args.sigma;
      break;
    }
    case MH_KERNEL_RING:
    case MH_KERNEL_PEAKS:
    {
      ptrdiff_t limit1,limit2;
      size_t width;
      if (args.rho < args.sigma)
        {
          width=((size_t) args.sigma)*2+1;
          limit1=(ptrdiff_t) (args.rho*args.rho);
          limit2=(ptrdiff_t) (args.sigma*args.sigma);
        }
      else
        {
          width=((size_t) args.rho)*2+1;
          limit1=(ptrdiff_t) (args.sigma*args.sigma);
          limit2=(ptrdiff_t) (args.rho*args.rho);
        }
      if (limit2 <= 0)
        {
          width=7;
          limit1=7;
          limit2=11;
        }
      k=new_kernel(type,width,width);
      if (k == nullptr) return nullptr;
      k->x=k->y=(ptrdiff_t) (width-1)/2;
      const ptrdiff_t r=k->x;
      ptrdiff_t scale=(ptrdiff_t) (type == MH_KERNEL_PEAKS ? 0.0 : args.xi);
      size_t i=0;
      for (ptrdiff_t v=-r; v <= r; v++)
        for (ptrdiff_t u=-r; u <= r; u++,i++)
          {
            ptrdiff_t radius=u*u+v*v;
            if ((limit1 < radius) && (radius <= limit2))
              k->positive_range+=k->values[i]=(double) scale;
            else
              k->values[i]=nan;
          }
      k->minimum=k->maximum=(double) scale;
      if (type == MH_KERNEL_PEAKS)
        {
          k->values[k->x+k->y*(ptrdiff_t) width]=1.0;
          k->positive_range=1.0;
          k->maximum=1.0;
        }
      break;
    }
    case MH_KERNEL_CHEBYSHEV:
    case MH_KERNEL_MANHATTAN:
    case MH_KERNEL_OCTAGONAL:
    case MH_KERNEL_EUCLIDEAN:
    {
      // distance kernels, morphology.c:2316-2412
      size_t width=args.rho < 1.0 ? 3 : ((size_t) args.rho)*2+1;
      if ((type == MH_KERNEL_OCTAGONAL) && (args.rho < 2.0))
        width=5;
      k=new_kernel(type,width,width);
      if (k == nullptr) return nullptr;
      k->x=k->y=(ptrdiff_t) (width-1)/2;
      const ptrdiff_t r=k->x;
      size_t i=0;
      for (ptrdiff_t v=-r; v <= r; v++)
        for (ptrdiff_t u=-r; u <= r; u++,i++)
          {
            double d;
            if (type == MH_KERNEL_CHEBYSHEV)
              d=args.sigma*(double) ((labs((long) u) > labs((long) v)) ? labs((long) u) :
                labs((long) v));
            else if (type == MH_KERNEL_MANHATTAN)
              d=args.sigma*(double) (labs((long) u)+labs((long) v));
            else if (type == MH_KERNEL_OCTAGONAL)
              {
                double r1=(double) ((labs((long) u) > labs((long) v)) ? labs((long) u) :
                  labs((long) v));
                double r2=floor((double) (labs((long) u)+labs((long) v)+1)/1.5);
                d=args.sigma*(r1 > r2 ? r1 : r2);
              }
            else
              d=args.sigma*sqrt((double) (u*u+v*v));
            k->positive_range+=k->values[i]=d;
          }
      k->maximum=k->values[0];
      break;
    }
    case MH_KERNEL_FREICHEN:
      return frei_chen(args);
    // ---- hit-and-miss kernel sets, morphology.c:1748-2087
    case MH_KERNEL_THINSE:
      return thin_element(type,(int) args.rho,args.sigma);
    case MH_KERNEL_EDGES:
      k=thin_element(type,482,0.0);
      if (k == nullptr) return nullptr;
      expand_mirror(k);
      return k;
    case MH_KERNEL_CORNERS:
      k=thin_element(type,87,0.0);
      if (k == nullptr) return nullptr;
      expand_rotate(k,90.0);
      return k;
    case MH_KERNEL_DIAGONALS:
      switch ((int) args.rho)
      {
        case 1: k=structuring_element(type,"0000-111-"); break;
        case 2: k=structuring_element(type,"0010-101-"); break;
        default:
          k=element_list(type,{"0000-111-","0010-101-"});
          if (k == nullptr) return nullptr;
          expand_mirror(k);
          return k;
      }
      if (k == nullptr) return nullptr;
      rotate_kernel(k,args.sigma);
      return k;
    case MH_KERNEL_LINEENDS:
      switch ((int) args.rho)
      {
        case 1: k=structuring_element(type,"00-01100-"); break;    // 4-connected line ends
        case 2: k=structuring_element(type,"000010001"); break;    // added for 8-connected lines
        case 3: k=structuring_element(type,"000011000"); break;    // orthogonal ends only
        case 4: k=structuring_element(type,"00001-00-"); break;    // traditional
        default: return acquire_kernel_list("LineEnds:1>;LineEnds:2>");
      }
      if (k == nullptr) return nullptr;
      rotate_kernel(k,args.sigma);
      return k;
    case MH_KERNEL_LINEJUNCTIONS:
      switch ((int) args.rho)
      {
        case 1: k=structuring_element(type,"1-1-1--1-"); break;    // Y
        case 2: k=structuring_element(type,"1---1-1-1"); break;    // diagonal T
        case 3: k=structuring_element(type,"---111-1-"); break;    // orthogonal T
        case 4: k=structuring_element(type,"1-1-1-1-1"); break;    // diagonal X
        case 5: k=structuring_element(type,"-1-111-1-"); break;    // orthogonal X
        default: return acquire_kernel_list("LineJunctions:1@;LineJunctions:2>");
      }
      if (k == nullptr) return nullptr;
      rotate_kernel(k,args.sigma);
      return k;
    case MH_KERNEL_RIDGES:
      if ((int) args.rho == 2)
        {
          k=array_typed(type,"4x1:0,1,1,0");
          if (k == nullptr) return nullptr;
          expand_rotate(k,90.0);
          // the stepped 'thick' lines: listed explicitly, a non-square kernel cannot
          // be rotated
          static const char *const kSteps[]={
            "4x3+1+1:0,1,1,- -,1,1,- -,1,1,0","4x3+2+1:0,1,1,- -,1,1,- -,1,1,0",
            "4x3+1+1:-,1,1,0 -,1,1,- 0,1,1,-","4x3+2+1:-,1,1,0 -,1,1,- 0,1,1,-",
            "3x4+1+1:0,-,- 1,1,1 1,1,1 -,-,0","3x4+1+2:0,-,- 1,1,1 1,1,1 -,-,0",
            "3x4+1+1:-,-,0 1,1,1 1,1,1 0,-,-","3x4+1+2:-,-,0 1,1,1 1,1,1 0,-,-"};
          for (const char *step : kSteps)
            {
              MhKernelInfo *n=array_typed(type,step);
              if (n == nullptr)
                {
                  destroy_chain(k);
                  return nullptr;
                }
              last_kernel(k)->next=n;
            }
          return k;
        }
      k=array_typed(type,"3x1:0,1,0");
      if (k == nullptr) return nullptr;
      expand_rotate(k,90.0);
      return k;
    case MH_KERNEL_CONVEXHULL:
    {
      k=structuring_element(type,"11-10-1-0");
      if (k == nullptr) return nullptr;
      expand_rotate(k,90.0);
      MhKernelInfo *mirror=structuring_element(type,"11110---0");
      if (mirror == nullptr)
        {
          destroy_chain(k);
          return nullptr;
        }
      expand_rotate(mirror,90.0);
      last_kernel(k)->next=mirror;
      return k;
    }
    case MH_KERNEL_SKELETON:
      switch ((int) args.rho)
      {
        case 2:
          k=acquire_kernel_list("ThinSE:482; ThinSE:87x90;");
          if (k == nullptr) return nullptr;
          retype_chain(k,type);
          expand_rotate(k,90.0);
          return k;
        case 3:
          k=acquire_kernel_list("ThinSE:41; ThinSE:42; ThinSE:43");
          if (k == nullptr) return nullptr;
          retype_chain(k,type);
          expand_mirror(k);
          return k;
        default:
          k=thin_element(type,482,0.0);
          if (k == nullptr) return nullptr;
          expand_rotate(k,45.0);
          return k;
      }
    default:
      return nullptr;
  }
  return k;
}

// ParseKernelName, morphology.c:374-483
MhKernelInfo *parse_named(const std::string &text)
{
  size_t n=0;
  while ((n < text.size()) && isalnum((unsigned char) text[n]))
    n++;
  std::string name=text.substr(0,n);
  MhKernelInfoType type=MH_KERNEL_UNDEFINED;
  for (const NameEntry &e : kNames)
    if (strcasecmp(e.name,name.c_str()) == 0)
      type=e.type;
  if (type == MH_KERNEL_UNDEFINED)
    return nullptr;
  while ((n < text.size()) && (isspace((unsigned char) text[n]) || (text[n] == ',') ||
         (text[n] == ':')))
    n++;
  Geometry args;
  if (!parse_geometry(text.substr(n),args))
    return nullptr;
  switch (type)
  {
    case MH_KERNEL_UNITY:
      if ((args.flags & kRho) == 0) args.rho=1.0;
      break;
    case MH_KERNEL_SQUARE: case MH_KERNEL_DIAMOND: case MH_KERNEL_OCTAGON:
    case MH_KERNEL_DISK: case MH_KERNEL_PLUS: case MH_KERNEL_CROSS:
      if ((args.flags & kSigma) == 0) args.sigma=1.0;
      break;
    case MH_KERNEL_RING:
      if ((args.flags & kXi) == 0) args.xi=1.0;
      break;
    case MH_KERNEL_RECTANGLE:
      if ((args.flags & kRho) == 0) args.rho=args.sigma;
      if (args.rho < 1.0) args.rho=3;
      if (args.sigma < 1.0) args.sigma=args.rho;
      if ((args.flags & kXi) == 0) args.xi=(double) (((ptrdiff_t) args.rho-1)/2);
      if ((args.flags & kPsi) == 0) args.psi=(double) (((ptrdiff_t) args.sigma-1)/2);
      break;
    case MH_KERNEL_CHEBYSHEV: case MH_KERNEL_MANHATTAN:
    case MH_KERNEL_OCTAGONAL: case MH_KERNEL_EUCLIDEAN:
      if ((args.flags & kSigma) == 0) args.sigma=100.0;
      else if ((args.flags & kAspect) != 0) args.sigma=mh::kQuantumRange/(args.sigma+1);
      else if ((args.flags & kPercent) != 0) args.sigma*=mh::kQuantumRange/100.0;
      break;
    default:
      break;
  }
  MhKernelInfo *k=builtin(type,args);
  if (k == nullptr)
    return nullptr;
  if (k->next == nullptr)
    {
      if ((args.flags & kArea) != 0)
        expand_rotate(k,45.0);
      else if ((args.flags & kGreater) != 0)
        expand_rotate(k,90.0);
    }
  return k;
}

// AcquireKernelInfo, morphology.c:485-560: a ';'-separated list of kernels
MhKernelInfo *acquire_kernel_list(const char *kernel_string)
{
  if (kernel_string == nullptr)
    return nullptr;
  MhKernelInfo *head=nullptr;
  std::string all(kernel_string);
  size_t pos=0;
  while (pos <= all.size())
    {
      size_t semi=all.find(';',pos);
      std::string part=all.substr(pos,semi == std::string::npos ? std::string::npos : semi-pos);
      pos=semi == std::string::npos ? all.size()+1 : semi+1;
      size_t b=0;
      while ((b < part.size()) && (isspace((unsigned char) part[b]) || (part[b] == '\'')))
        b++;
      part=part.substr(b);
      while (!part.empty() && (isspace((unsigned char) part.back()) || (part.back() == '\'')))
        part.pop_back();
      if (part.empty())
        continue;
      MhKernelInfo *k=isalpha((unsigned char) part[0]) ? parse_named(part) : parse_array(part);
      if (k == nullptr)
        {
          destroy_chain(head);
          mh::set_error("cannot build kernel `%s'",part.c_str());
          return nullptr;
        }
      if (head == nullptr)
        head=k;
      else
        last_kernel(head)->next=k;
    }
  return head;
}


} // namespace

namespace mh {

// "blur:RxS;blur:RxS+90" without the locale-dependent string round trip
// BlurImage goes through (effect.c:788-790)
MhKernelInfo *acquire_blur_kernels(double radius,double sigma)
{
  Geometry args;
  args.rho=radius;
  args.sigma=sigma;
  args.flags=kRho|kSigma;
  MhKernelInfo *first=builtin(MH_KERNEL_BLUR,args);
  if (first == nullptr)
    return nullptr;
  args.xi=90.0;
  args.flags|=kXi;
  MhKernelInfo *second=builtin(MH_KERNEL_BLUR,args);
  if (second == nullptr)
    {
      destroy_chain(first);
      return nullptr;
    }
  first->next=second;
  return first;
}

// "gaussian:RxS" (GaussianBlurImage, effect.c:1724-1726) without the string round trip
MhKernelInfo *acquire_gaussian_kernel(double radius,double sigma)
{
  Geometry args;
  args.rho=radius;
  args.sigma=sigma;
  args.flags=kRho|kSigma;
  return builtin(MH_KERNEL_GAUSSIAN,args);
}

// Square kernels of the ConvolveImage callers in effect.c; meta-data stays zero as in the
// reference (memset / fresh AcquireKernelInfo(NULL)), MorphologyApply only reads the values.
static MhKernelInfo *square_kernel(size_t width)
{
  MhKernelInfo *k=new_kernel(MH_KERNEL_USERDEFINED,width,width);
  if (k == nullptr)
    return nullptr;
  k->x=k->y=(ptrdiff_t) (width-1)/2;
  return k;
}

static double magick_sigma(double sigma)
{
  return fabs(sigma) < kEpsilon ? kEpsilon : sigma;          // MagickSigma, effect.c:96
}

// SharpenImage, effect.c:4017-4058
MhKernelInfo *acquire_sharpen_kernel(double radius,double sigma)
{
  const size_t width=MhGetOptimalKernelWidth2D(radius,sigma);
  MhKernelInfo *k=square_kernel(width);
  if (k == nullptr)
    return nullptr;
  const double s=magick_sigma(sigma);
  const ptrdiff_t j=(ptrdiff_t) (width-1)/2;
  double normalize=0.0;
  size_t i=0;
  for (ptrdiff_t v=-j; v <= j; v++)
    for (ptrdiff_t u=-j; u <= j; u++)
      {
        k->values[i]=(double) (-exp(-((double) u*u+v*v)/(2.0*s*s))/(2.0*kPi*s*s));
        normalize+=k->values[i];
        i++;
      }
  k->values[i/2]=(double) ((-2.0)*normalize);
  normalize=0.0;
  for (i=0; i < width*width; i++)
    normalize+=k->values[i];
  const double gamma=perceptible_reciprocal(normalize);
  for (i=0; i < width*width; i++)
    k->values[i]*=gamma;
  return k;
}

// EdgeImage, effect.c:1541-1563
MhKernelInfo *acquire_edge_kernel(double radius)
{
  const size_t width=MhGetOptimalKernelWidth1D(radius,0.5);
  MhKernelInfo *k=square_kernel(width);
  if (k == nullptr)
    return nullptr;
  size_t i;
  for (i=0; i < width*width; i++)
    k->values[i]=(-1.0);
  k->values[i/2]=(double) width*width-1.0;
  return k;
}

// EmbossImage, effect.c:1633-1672
MhKernelInfo *acquire_emboss_kernel(double radius,double sigma)
{
  const size_t width=MhGetOptimalKernelWidth1D(radius,sigma);
  MhKernelInfo *k=square_kernel(width);
  if (k == nullptr)
    return nullptr;
  const double s=magick_sigma(sigma);
  const ptrdiff_t j=(ptrdiff_t) (width-1)/2;
  ptrdiff_t diagonal=j;
  size_t i=0;
  for (ptrdiff_t v=-j; v <= j; v++)
    {
      for (ptrdiff_t u=-j; u <= j; u++)
        {
          k->values[i]=(double) ((((u < 0) || (v < 0)) ? -8.0 : 8.0)*
            exp(-((double) u*u+v*v)/(2.0*s*s))/(2.0*kPi*s*s));
          if (u != diagonal)
            k->values[i]=0.0;
          i++;
        }
      diagonal--;
    }
  double normalize=0.0;
  for (i=0; i < width*width; i++)
    normalize+=k->values[i];
  const double gamma=perceptible_reciprocal(normalize);
  for (i=0; i < width*width; i++)
    k->values[i]*=gamma;
  return k;
}

} // namespace mh

// ===================================================================== C ABI
extern "C" {

// GetOptimalKernelWidth1D, gem.c:262-300
MH_API size_t MhGetOptimalKernelWidth1D(double radius,double sigma)
{
  if (radius > kEpsilon)
    return (size_t) (2.0*ceil(radius)+1.0);
  double gamma=fabs(sigma);
  if (gamma <= kEpsilon)
    return 3;
  double alpha=perceptible_reciprocal(2.0*gamma*gamma);
  double beta=perceptible_reciprocal(kSq2Pi*gamma);
  size_t width=5;
  for ( ; ; )
    {
      double normalize=0.0;
      ptrdiff_t j=(ptrdiff_t) (width-1)/2;
      for (ptrdiff_t i=-j; i <= j; i++)
        normalize+=exp(-((double) (i*i))*alpha)*beta;
      double value=exp(-((double) (j*j))*alpha)*beta/normalize;
      if ((value < kQScale) || (value < kEpsilon))
        break;
      width+=2;
    }
  return width-2;
}

// GetOptimalKernelWidth2D, gem.c:302-345
MH_API size_t MhGetOptimalKernelWidth2D(double radius,double sigma)
{
  if (radius > kEpsilon)
    return (size_t) (2.0*ceil(radius)+1.0);
  double gamma=fabs(sigma);
  if (gamma <= kEpsilon)
    return 3;
  double alpha=perceptible_reciprocal(2.0*gamma*gamma);
  double beta=perceptible_reciprocal(k2Pi*gamma*gamma);
  size_t width=5;
  for ( ; ; )
    {
      double normalize=0.0;
      ptrdiff_t j=(ptrdiff_t) (width-1)/2;
      for (ptrdiff_t v=-j; v <= j; v++)
        for (ptrdiff_t u=-j; u <= j; u++)
          normalize+=exp(-((double) (u*u+v*v))*alpha)*beta;
      double value=exp(-((double) (j*j))*alpha)*beta/normalize;
      if ((value < kQScale) || (value < kEpsilon))
        break;
      width+=2;
    }
  return width-2;
}

MH_API MhKernelInfo *MhAcquireKernelInfo(const char *kernel_string)
{
  return acquire_kernel_list(kernel_string);
}

MH_API MhKernelInfo *MhDestroyKernelInfo(MhKernelInfo *kernel)
{
  destroy_chain(kernel);
  return nullptr;
}

MH_API MhKernelInfo *MhCloneKernelInfo(const MhKernelInfo *kernel)
{
  MhKernelInfo *head=nullptr,*tail=nullptr;
  for (const MhKernelInfo *k=kernel; k != nullptr; k=k->next)
    {
      MhKernelInfo *c=clone_one(k);
      if (c == nullptr)
        {
          destroy_chain(head);
          return nullptr;
        }
      if (head == nullptr) head=c; else tail->next=c;
      tail=c;
    }
  return head;
}

MH_API void MhScaleKernelInfo(MhKernelInfo *kernel,double scaling_factor,unsigned flags)
{
  if (kernel != nullptr)
    scale_kernel(kernel,scaling_factor,flags);
}

} // extern "C"
