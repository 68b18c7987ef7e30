// CPU emulation of resize_stream.hip's walk over the tables of resize_stream_plan.hpp: a lane per
// source column, the window's kRows source rows with dense scalar weights, the wave's row of the
// intermediate with nt neighbours and f x nt scalar weights, the listed columns near the image
// edges with their own dense weights, the strips of 64-(nt-1) columns — checked against the plain
// two-pass evaluation of the same contribution lists.  Test infrastructure: built and run by
// tests/test_resize_stream_plan.py (no GPU).
//   g++ -O2 -std=c++17 -I imagemagick_amd/csrc tests/cpu/resize_stream_plan_test.cpp -o /tmp/stream_plan_test
#include "resize_stream_plan.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Table
{
  int out_size=0,max_taps=0;
  std::vector<int> start,count;
  std::vector<double> weight;     // [tap][out]
};

static double sinc(double x) { return x == 0.0 ? 1.0 : std::sin(M_PI*x)/(M_PI*x); }
static double lanczos3(double d) { return std::fabs(d) < 3.0 ? sinc(d)*sinc(d/3.0) : 0.0; }
static double triangle(double d) { return std::fabs(d) < 1.0 ? 1.0-std::fabs(d) : 0.0; }
static double catrom(double d)
{
  d=std::fabs(d);
  if (d < 1.0) return 1.0+d*d*(-2.5+1.5*d);
  if (d < 2.0) return 2.0+d*(-4.0+d*(2.5-0.5*d));
  return 0.0;
}

// contribution lists the way resize.c:3418-3443 forms them (normalised)
static void build(Table &t,int in,int out,double (*filter)(double),double filter_support)
{
  const double factor=(double) out/in;
  const double scale=std::max(1.0/factor,1.0),support=std::max(scale*filter_support,0.5);
  t.out_size=out;
  t.start.resize(out); t.count.resize(out);
  std::vector<std::vector<double>> w(out);
  t.max_taps=0;
  for (int x=0; x < out; x++)
    {
      const double bisect=(x+0.5)/factor+1e-12;
      const int start=(int) std::max(bisect-support+0.5,0.0);
      const int stop=(int) std::min(bisect+support+0.5,(double) in);
      double density=0.0;
      for (int n=0; n < stop-start; n++)
        {
          const double v=filter(((start+n)-bisect+0.5)/scale);
          w[x].push_back(v);
          density+=v;
        }
      for (double &v : w[x])
        v/=density;
      t.start[x]=start;
      t.count[x]=stop-start;
      t.max_taps=std::max(t.max_taps,stop-start);
    }
  t.weight.assign((size_t) t.max_taps*out,0.0);
  for (int x=0; x < out; x++)
    for (int j=0; j < t.count[x]; j++)
      t.weight[(size_t) j*out+x]=w[x][j];
}

static int check(int H,int W,int OH,int OW,double (*filter)(double),double support,bool expect_ok,const char *name)
{
  Table vt,ht;
  build(vt,H,OH,filter,support);
  build(ht,W,OW,filter,support);
  mh::StreamResizePlan p;
  const bool ok=mh::build_stream_resize_plan(p,vt,ht,W,H);
  if (ok != expect_ok)
    {
      std::printf("FAIL %s %dx%d -> %dx%d: plan %s, expected %s\n",name,W,H,OW,OH,ok ? "built" : "declined",
        expect_ok ? "built" : "declined");
      return 1;
    }
  if (!ok)
    {
      std::printf("ok   %s %dx%d -> %dx%d declined\n",name,W,H,OW,OH);
      return 0;
    }
  // a smooth single-channel frame (the four channels are treated alike)
  std::vector<double> src((size_t) H*W);
  for (int y=0; y < H; y++)
    for (int x=0; x < W; x++)
      src[(size_t) y*W+x]=1000.0+37.0*std::sin(0.37*x+0.11*y)+(double) ((x*131+y*71) % 97);
  // plain two passes
  std::vector<double> mid((size_t) OH*W),want((size_t) OH*OW),got((size_t) OH*OW,-1.0e300);
  for (int y=0; y < OH; y++)
    for (int x=0; x < W; x++)
      {
        double s=0.0;
        for (int k=0; k < vt.count[y]; k++)
          s+=vt.weight[(size_t) k*OH+y]*src[(size_t) (vt.start[y]+k)*W+x];
        mid[(size_t) y*W+x]=s;
      }
  for (int y=0; y < OH; y++)
    for (int x=0; x < OW; x++)
      {
        double s=0.0;
        for (int k=0; k < ht.count[x]; k++)
          s+=ht.weight[(size_t) k*OW+x]*mid[(size_t) y*W+ht.start[x]+k];
        want[(size_t) y*OW+x]=s;
      }
  // the kernel's walk
  constexpr int KR=mh::StreamResizePlan::kRows,PAD=8;
  const int ROWS=p.window_rows();
  const int f=p.f,nt=p.nt,lo=p.lo;
  const int strips=(int) p.strip_first.size(),rows_per_chunk=23;
  const int chunks=(OH+rows_per_chunk-1)/rows_per_chunk;
  for (int chunk=0; chunk < chunks; chunk++)
    for (int strip=0; strip < strips; strip++)
      {
        const int c0=p.strip_first[strip]+lo,nvl=p.strip_count[strip];
        const double *hw=&p.strip_hw[(size_t) strip*mh::StreamResizePlan::kMaxDense];
        const int y0=chunk*rows_per_chunk,y1=std::min(OH,y0+rows_per_chunk);
        int base=p.vbase[y0];
        double win[64][KR];
        auto fetch=[&](int lane,int row) { const int c=std::min(std::max(c0+lane,0),W-1); return src[(size_t) std::min(row,H-1)*W+c]; };
        for (int lane=0; lane < 64; lane++)
          for (int j=0; j < ROWS; j++)
            win[lane][(base+j) % ROWS]=fetch(lane,base+j);       // source row r in slot r % ROWS
        for (int y=y0; y < y1; y++)
          {
            if (base < p.vbase[y])
              {
                for (int lane=0; lane < 64; lane++)
                  win[lane][base % ROWS]=fetch(lane,base+ROWS);
                base++;
              }
            if (base != p.vbase[y])
              {
                std::printf("FAIL %s: the window fell behind at row %d\n",name,y);
                return 1;
              }
            double row[64+2*PAD];
            for (int i=0; i < 64+2*PAD; i++)
              row[i]=1.0e300;                     // whatever lies beside the wave's row
            for (int lane=0; lane < 64; lane++)
              {
                double s=0.0;
                for (int j=0; j < ROWS; j++)
                  s+=p.vdense[(size_t) y*KR+j]*win[lane][j];
                row[PAD+lane]=s;
              }
            for (int lane=-lo; lane < nvl-lo; lane++)
              {
                const int c=c0+lane;
                const bool listed=(c < p.edge_left) || (c >= p.edge_right);
                const int entry=c < p.edge_left ? c : mh::StreamResizePlan::kListed+(c-p.edge_right);
                for (int q=0; q < f; q++)
                  {
                    double s=0.0;
                    for (int j=mh::StreamResizePlan::phase_first(f,q); j < mh::StreamResizePlan::phase_first(f,q)+nt-1; j++)
                      {
                        const double w=listed ? p.listed[(size_t) entry*mh::StreamResizePlan::kMaxDense+q*nt+j] : hw[q*nt+j];
                        if (w != 0.0)              // (0 * the 1e300 marker would still be 0; an Inf would not)
                          s+=w*row[PAD+lane+lo+j];
                      }
                    got[(size_t) y*OW+(size_t) f*c+q]=s;
                  }
              }
          }
      }
  double worst=0.0;
  for (size_t i=0; i < want.size(); i++)
    worst=std::max(worst,std::fabs(got[i]-want[i])/std::max(1.0,std::fabs(want[i])));
  // (the same weights; the window's slots are summed in slot order, not tap order)
  const bool pass=worst < 1.0e-13;
  std::printf("%s %s %dx%d -> %dx%d f=%d nt=%d lo=%d listed %d+%d: worst relative difference %.3g\n",pass ? "ok  " : "FAIL",
    name,W,H,OW,OH,f,nt,lo,p.edge_left,W-p.edge_right,worst);
  return pass ? 0 : 1;
}

int main()
{
  int failures=0;
  failures+=check(37,53,148,212,lanczos3,3.0,true,"lanczos 4x");
  failures+=check(64,300,256,1200,lanczos3,3.0,true,"lanczos 4x");
  failures+=check(300,61,701,244,lanczos3,3.0,true,"lanczos 4x / 2.34x");
  failures+=check(20,116,20,464,lanczos3,3.0,true,"lanczos 4x / 1x");
  // (the representative column itself is clipped: declined)
  failures+=check(9,5,36,20,lanczos3,3.0,false,"lanczos 4x, 5 columns");
  failures+=check(7,3,7,6,lanczos3,3.0,false,"lanczos 2x, 3 columns");
  failures+=check(12,58,30,116,lanczos3,3.0,true,"lanczos 2x");
  failures+=check(90,100,180,200,catrom,2.0,true,"catrom 2x");
  // 3x: the middle output of a column sits at whole-number distances (minus MagickEpsilon) from its
  // taps — the zeros of every interpolating filter: weights born of cancellation, 1e-12 +- 1e-15 from
  // binade to binade but bit-identical inside one (the strips' own weights)
  failures+=check(12,58,30,174,catrom,2.0,true,"catrom 3x");
  failures+=check(12,300,30,900,lanczos3,3.0,true,"lanczos 3x");
  failures+=check(12,58,30,232,catrom,2.0,true,"catrom 4x");
  failures+=check(64,2048,64,8192,lanczos3,3.0,true,"lanczos 4x wide");
  failures+=check(41,50,164,150,triangle,1.0,true,"triangle 3x");
  failures+=check(29,33,290,330,lanczos3,3.0,false,"lanczos 10x");
  failures+=check(150,70,600,141,lanczos3,3.0,false,"lanczos 2.01x");
  // weights_denominator: the first (vertical) filter's weights as fractions with a small denominator — Triangle at
  // 2x: quarters (minus MagickEpsilon), at 3x: thirds; Lanczos: none
  {
    auto denominator=[&](int H,int OH,double (*filter)(double),double support) -> int
    {
      Table vt,ht;
      build(vt,H,OH,filter,support);
      build(ht,128,256,filter,support);
      mh::StreamResizePlan plan;
      if (!mh::build_stream_resize_plan(plan,vt,ht,128,H))
        return -1;
      return plan.weights_denominator;
    };
    const int t2=denominator(60,120,triangle,1.0),t4=denominator(60,240,triangle,1.0),t3=denominator(60,180,triangle,1.0);
    const int l2=denominator(60,120,lanczos3,3.0),c2=denominator(60,120,catrom,2.0);
    std::printf("weights denominators: triangle 2x %d, 4x %d, 3x %d; lanczos 2x %d; catrom 2x %d\n",t2,t4,t3,l2,c2);
    if ((t2 != 4) || (t4 != 8) || (t3 != 3) || (l2 != 0) || (c2 != 128))
      failures++;
  }
  std::printf(failures == 0 ? "ALL OK\n" : "%d FAILURES\n",failures);
  return failures == 0 ? 0 : 1;
}
