// Is the window of tie_watch.hpp wide enough?  The first filter of ResizeImage over adversarial
// frames, twice on the host: in the one-launch kernels' order (alpha-premultiplied samples, fused
// multiply-adds over the window slots, a reciprocal and three products) and in the reference's
// (MagickCore/resize.c:3494-3530: alpha = weight*QuantumScale*alpha_j, pixel += alpha*p, gamma +=
// alpha, PerceptibleReciprocal, ClampToQuantum — every operation rounded on its own).  Wherever the
// two ROUNDED intermediates differ, the kernel's test (TieWatch: plain window for the alpha
// channel, the window that widens with the reciprocal of the alpha sum for the colours; clamped
// sums that are not exact zeros) must have reported the pixel — and it should report little else.
// Test infrastructure: built (-ffp-contract=off) and run by tests/test_tie_watch.py (no GPU).
#include "tie_watch.hpp"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

static const double kQS=1.0/65535.0,kEps=1.0e-12;

static double sinc(double x) { return x == 0.0 ? 1.0 : std::sin(M_PI*x)/(M_PI*x); }
static double lanczos3(double d) { return std::fabs(d) < 3.0 ? sinc(d)*sinc(d/3.0) : 0.0; }
static double triangle(double d) { return std::fabs(d) < 1.0 ? 1.0-std::fabs(d) : 0.0; }
static double box(double d) { return std::fabs(d) <= 0.5 ? 1.0 : 0.0; }
static double catrom(double d)
{
  d=std::fabs(d);
  if (d < 1.0) return 1.0+d*d*(-2.5+1.5*d);
  if (d < 2.0) return 2.0+d*(-4.0+d*(2.5-0.5*d));
  return 0.0;
}

struct Lists { std::vector<int> start,count; std::vector<std::vector<double>> w; };

// contribution lists the way resize.c:3404-3443 forms them (normalised)
static void build(Lists &t,int in,int out,double (*filter)(double),double filter_support)
{
  const double factor=(double) out/in;
  const double scale=std::max(1.0/factor,1.0),support=std::max(scale*filter_support,0.5);
  t.start.assign(out,0); t.count.assign(out,0); t.w.assign(out,{});
  for (int x=0; x < out; x++)
    {
      const double bisect=(x+0.5)/factor+1e-12;
      const int start=(int) std::max(bisect-support+0.5,0.0);
      const int stop=(int) std::min(bisect+support+0.5,(double) in);
      double density=0.0;
      for (int n=0; n < stop-start; n++)
        {
          const double v=filter(((start+n)-bisect+0.5)/scale);
          t.w[x].push_back(v);
          density+=v;
        }
      if ((density != 0.0) && (density != 1.0))
        for (double &v : t.w[x])
          v*=1.0/density;                          // PerceptibleReciprocal(density), resize.c:3437-3443
      t.start[x]=start;
      t.count[x]=stop-start;
    }
}

static uint16_t clamp_q16(double v)
{
  if (std::isnan(v) || (v <= 0.0)) return 0;
  if (v >= 65535.0) return 65535;
  return (uint16_t) (v+0.5);
}

struct Tally { long long pixels=0,differ=0,reported=0,missed=0; };

// second = the filter whose result leaves the operator: its input is a rounded intermediate both orders
// share, nothing is tested against rounding boundaries, a difference of one level / ULP is the
// contract; reported are alpha sums below the limit and clamped sums that are not exact zeros
template<bool kFloat,bool second=false>
static void run(const char *name,int in,int out,double (*filter)(double),double support,int kind,Tally &tally)
{
  Lists t;
  build(t,in,out,filter,support);
  std::mt19937_64 rng(1000u*(unsigned) in+(unsigned) out+(unsigned) kind);
  const int columns=kFloat ? 600 : 1200;
  // a column of `in` RGBA pixels per trial
  std::vector<double> px((size_t) in*4);
  for (int column=0; column < columns; column++)
    {
      for (int y=0; y < in; y++)
        {
          for (int c=0; c < 3; c++)
            px[(size_t) y*4+c]=(double) (rng() % 65536u);
          double a;
          switch (kind)
          {
            case 0: a=(double) (rng() % 4u); break;                          // tiny alpha
            case 1: a=(rng() & 1u) ? 65535.0 : 0.0; break;                   // binary alpha
            case 2: a=(double) (rng() % 65536u); break;                      // any alpha
            default: a=(double) (1u+rng() % 65535u); break;                  // no transparent pixel
          }
          px[(size_t) y*4+3]=a;
          if (kFloat && (kind == 3))
            for (int c=0; c < 4; c++)
              px[(size_t) y*4+c]=(double) (float) (px[(size_t) y*4+c]*(0.25+(double) (rng() % 1000u)/1000.0));
        }
      for (int y=0; y < out; y++)
        {
          const int start=t.start[y],count=t.count[y];
          if ((count <= 0) || (count > 8))
            continue;
          // ---- the kernels' order: premultiplied samples in window slots (source row mod 8), fused
          double s[4]={0.0,0.0,0.0,0.0};
          double slot_w[8]={0,0,0,0,0,0,0,0},slot_p[8][4]={};
          for (int k=0; k < count; k++)
            {
              const int slot=(start+k) % 8;
              slot_w[slot]=t.w[y][k];
              const double a=px[(size_t) (start+k)*4+3];
              slot_p[slot][3]=a;
              for (int c=0; c < 3; c++)
                slot_p[slot][c]=a*px[(size_t) (start+k)*4+c];
            }
          for (int j=0; j < 8; j++)
            for (int c=0; c < 4; c++)
              s[c]=std::fma(slot_w[j],slot_p[j][c],s[c]);
          const double sa=s[3];
          bool reported=false;
          double fused[4];
          mh::TieWatchBits<kFloat> plain,colour;
          plain.plain();
          colour.plain();
          if (!(std::fabs(sa)*kQS >= kEps))
            {
              const double scale=(sa < 0.0 ? -1.0/kEps : 1.0/kEps)*kQS;
              for (int c=0; c < 3; c++)
                fused[c]=s[c]*scale;
              fused[3]=sa;
              reported=(s[0] != 0.0) || (s[1] != 0.0) || (s[2] != 0.0) || (s[3] != 0.0);
            }
          else
            {
              const double r=1.0/sa;
              colour.quotient(r);
              for (int c=0; c < 3; c++)
                fused[c]=s[c]*r;
              fused[3]=sa;
              if (second)
                reported=std::fabs(sa) < (kFloat ? 0.05 : 1.0e-3);
              else
                reported=colour.near(fused[0]) || colour.near(fused[1]) || colour.near(fused[2]) || plain.near(fused[3]);
            }
          // ---- the reference's order
          double pixel[4]={0.0,0.0,0.0,0.0},gamma=0.0;
          for (int k=0; k < count; k++)
            {
              const double w=t.w[y][k];
              const double alpha=w*kQS*px[(size_t) (start+k)*4+3];
              for (int c=0; c < 3; c++)
                pixel[c]+=alpha*px[(size_t) (start+k)*4+c];
              gamma+=alpha;
              pixel[3]+=w*px[(size_t) (start+k)*4+3];
            }
          const double sign=gamma < 0.0 ? -1.0 : 1.0;
          const double reciprocal=(sign*gamma) >= kEps ? 1.0/gamma : sign/kEps;
          double reference[4]={reciprocal*pixel[0],reciprocal*pixel[1],reciprocal*pixel[2],pixel[3]};
          bool differ=false;
          for (int c=0; c < 4; c++)
            {
              if (kFloat)
                {
                  const float a=(float) fused[c],b=(float) reference[c];
                  differ=differ || (second ? !((a == b) || (std::nextafterf(a,b) == b)) : !(a == b));
                }
              else
                {
                  const int a=clamp_q16(fused[c]),b=clamp_q16(reference[c]);
                  differ=differ || (second ? std::abs(a-b) > 1 : a != b);
                }
            }
          tally.pixels++;
          tally.differ+=differ ? 1 : 0;
          tally.reported+=reported ? 1 : 0;
          if (differ && !reported)
            {
              tally.missed++;
              if (tally.missed < 6)
                std::printf("MISSED %s %d -> %d kind %d row %d: fused %.17g %.17g %.17g %.17g | reference %.17g %.17g %.17g %.17g\n",name,
                  in,out,kind,y,fused[0],fused[1],fused[2],fused[3],reference[0],reference[1],reference[2],reference[3]);
            }
        }
    }
}

int main()
{
  struct Filter { const char *name; double (*f)(double); double support; };
  const Filter filters[]={{"Triangle",triangle,1.0},{"Catrom",catrom,2.0},{"Lanczos",lanczos3,3.0},{"Box",box,0.5}};
  const int geometry[][2]={{40,80},{40,120},{41,164},{50,137},{64,256},{33,100}};
  int failed=0;
  for (const Filter &f : filters)
    {
      Tally q16,flt;
      for (const auto &g : geometry)
        {
          for (int kind=0; kind < 4; kind++)
            run<false>(f.name,g[0],g[1],f.f,f.support,kind,q16);
          run<true>(f.name,g[0],g[1],f.f,f.support,3,flt);        // float: no transparent pixel, nothing cancels
        }
      std::printf("%-8s Q16: %lld pixels, %lld differ between the two orders, %lld reported (%.4f %%), %lld differ unreported | "
                  "float: %lld pixels, %lld differ, %lld reported (%.4f %%), %lld unreported\n",f.name,q16.pixels,q16.differ,q16.reported,
                  100.0*(double) q16.reported/(double) q16.pixels,q16.missed,flt.pixels,flt.differ,flt.reported,
                  100.0*(double) flt.reported/(double) flt.pixels,flt.missed);
      failed+=(q16.missed != 0) || (flt.missed != 0) ? 1 : 0;
      // the second filter over the same kinds of rows (rounded intermediates are whole levels: the frames are)
      Tally out16,outf;
      for (const auto &g : geometry)
        {
          for (int kind=0; kind < 4; kind++)
            run<false,true>(f.name,g[0],g[1],f.f,f.support,kind,out16);
          run<true,true>(f.name,g[0],g[1],f.f,f.support,3,outf);
        }
      std::printf("%-8s second filter, Q16: %lld pixels, %lld more than a level apart, %lld reported (%.4f %%), %lld unreported | "
                  "float: %lld pixels, %lld more than an ULP apart, %lld reported, %lld unreported\n",f.name,out16.pixels,out16.differ,
                  out16.reported,100.0*(double) out16.reported/(double) out16.pixels,out16.missed,outf.pixels,outf.differ,outf.reported,
                  outf.missed);
      failed+=(out16.missed != 0) || (outf.missed != 0) ? 1 : 0;
    }
  std::printf(failed == 0 ? "ALL OK\n" : "FAILED\n");
  return failed == 0 ? 0 : 1;
}
