// CPU emulation of resize_mfma.hip's walk over the tables of resize_mfma_plan.hpp: the same
// strips / blocks / K-block ring / lane layouts (v_mfma_f64_16x16x4_f64: A lane = (i=lane&15,
// k=lane>>4), B lane = (k=lane>>4, j=lane&15), D register r of lane = (i=(lane>>4)+4r, j=lane&15)),
// checked against the plain two-pass evaluation of the same contribution lists.  Test
// infrastructure: built and run by tests/test_resize_mfma_plan.py (no GPU).
//   g++ -O2 -std=c++17 -I imagemagick_amd/csrc tests/cpu/resize_mfma_plan_test.cpp -o /tmp/plan_test
#include "resize_mfma_plan.hpp"
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

// contribution lists the way resize.c:3418-3443 forms them (Lanczos-3 weights, normalised)
static void build(Table &t,int in,int out)
{
  const double factor=(double) out/in;
  const double scale=std::max(1.0/factor,1.0),support=std::max(scale*3.0,0.5);
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
          const double d=((start+n)-bisect+0.5)/scale;
          const double v=std::fabs(d) < 3.0 ? sinc(d)*sinc(d/3.0) : 0.0;
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

typedef double Frag[64];
static void mfma(const Frag a,const Frag b,double d[64][4])
{
  for (int lane=0; lane < 64; lane++)
    for (int r=0; r < 4; r++)
      {
        const int i=(lane >> 4)+4*r,j=lane & 15;
        double s=d[lane][r];
        for (int k=0; k < 4; k++)
          s+=a[i+16*k]*b[j+16*k];          // A lane (i,k) = i+16k, B lane (k,j) = j+16k
        d[lane][r]=s;
      }
}

static int run(int W,int H,int OW,int OH,int tps,bool expect_ok,int waves=4)
{
  Table vt,ht;
  build(vt,H,OH);
  build(ht,W,OW);
  mh::MfmaResizePlan p;
  const bool ok=mh::build_mfma_resize_plan(p,vt,ht,tps,waves);
  if (ok != expect_ok)
    {
      std::printf("FAIL %dx%d -> %dx%d: plan ok=%d, expected %d\n",W,H,OW,OH,(int) ok,(int) expect_ok);
      return 1;
    }
  if (!ok)
    return 0;
  std::vector<double> src((size_t) W*H);
  srand(W*7+H);
  for (double &v : src)
    v=(rand()%65536)/1.0;
  // reference: two passes in double
  std::vector<double> mid((size_t) W*OH),want((size_t) OW*OH),got((size_t) OW*OH,-1.0);
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
        for (int j=0; j < ht.count[x]; j++)
          s+=ht.weight[(size_t) j*OW+x]*mid[(size_t) y*W+ht.start[x]+j];
        want[(size_t) y*OW+x]=s;
      }
  // the kernel's walk
  const int pitch=16*p.nvb_max;
  std::vector<double> patch((size_t) p.patch_rows_max*pitch);
  for (int strip=0; strip < p.nstrips; strip++)
    for (int rg0=0; rg0 < p.nrg; rg0+=p.waves)
      {
        const int col_lo=p.strip_col_lo[strip],nvb=p.strip_nvb[strip],pc=16*nvb;
        const int prow_lo=p.rg_row_lo[rg0];
        int prow_hi=prow_lo;
        for (int i=0; (i < p.waves) && (rg0+i < p.nrg); i++)
          prow_hi=std::max(prow_hi,p.rg_row_lo[rg0+i]+4*p.rg_nvk[rg0+i]);
        if ((prow_hi-prow_lo > p.patch_rows_max) || (nvb > p.nvb_max) || (p.strip_wcount[strip] > p.wblocks_max) || (p.nk > 5))
          { std::printf("FAIL: patch or weights exceed the plan's maxima\n"); return 1; }
        for (int r=0; r < prow_hi-prow_lo; r++)
          for (int i=0; i < pc; i++)
            patch[(size_t) r*pc+i]=src[(size_t) std::min(prow_lo+r,H-1)*W+std::min(col_lo+i,W-1)];
        for (int wave=0; wave < p.waves; wave++)
          {
            const int rg=rg0+wave;
            if (rg >= p.nrg)
              continue;
            const int rrow=p.rg_row_lo[rg]-prow_lo,nvk=p.rg_nvk[rg];
            double ring[8][64];
            for (auto &s : ring) for (double &v : s) v=0.0;
            int tdone=0;
            for (int vb=0; vb < nvb; vb++)
              {
                double acc[64][4]={};
                for (int kb=0; kb < nvk; kb++)
                  {
                    Frag a,b;
                    for (int lane=0; lane < 64; lane++)
                      {
                        const int g=lane >> 4,n=lane & 15;
                        a[lane]=patch[(size_t) (rrow+4*kb+g)*pc+16*vb+n];
                        b[lane]=p.wv[((size_t) p.rg_woff[rg]+kb)*64+lane];
                      }
                    mfma(a,b,acc);
                  }
                for (int s=0; s < 4; s++)
                  for (int lane=0; lane < 64; lane++)
                    ring[s][lane]=ring[s+4][lane];
                for (int r=0; r < 4; r++)
                  for (int lane=0; lane < 64; lane++)
                    ring[4+r][lane]=acc[lane][r];
                for (; (tdone < std::min(p.tps,p.ntiles-strip*p.tps)) && ((int) (p.tile_meta[strip*p.tps+tdone] >> 8) <= vb); tdone++)
                  {
                    const int t=strip*p.tps+tdone;
                    const unsigned meta=p.tile_meta[t];
                    const int sl0=(int) (meta & 255u),sl1=sl0+p.nk;
                    const int woff=tdone*p.nk;
                    if (sl0 != p.tile_kb0[t]-4*(vb-1))
                      { std::printf("FAIL: tile %d meta word disagrees with the tables\n",t); return 1; }
                    if ((sl0 < 0) || (sl1 > 8))
                      { std::printf("FAIL: tile %d outside the ring (slots %d..%d at block %d)\n",t,sl0,sl1,vb); return 1; }
                    double o[64][4]={};
                    for (int s=0; s < 8; s++)
                      if ((s >= sl0) && (s < sl1))
                        {
                          Frag b;
                          for (int lane=0; lane < 64; lane++)
                            b[lane]=p.wh[((size_t) p.strip_wbase[strip]+woff+(s-sl0))*64+lane];
                          mfma(ring[s],b,o);
                        }
                    for (int lane=0; lane < 64; lane++)
                      for (int r=0; r < 4; r++)
                        {
                          const int x=16*t+(lane & 15),y=16*rg+(lane >> 4)+4*r;
                          if ((x < OW) && (y < OH))
                            got[(size_t) y*OW+x]=o[lane][r];
                        }
                  }
              }
            if (tdone != std::min(p.tps,p.ntiles-strip*p.tps))
              { std::printf("FAIL: strip %d finished %d tiles\n",strip,tdone); return 1; }
          }
      }
  double worst=0.0;
  for (size_t i=0; i < want.size(); i++)
    worst=std::max(worst,std::fabs(want[i]-got[i])/(1.0+std::fabs(want[i])));
  if (!(worst < 1e-12))
    {
      std::printf("FAIL %dx%d -> %dx%d tps %d: worst relative difference %g\n",W,H,OW,OH,tps,worst);
      return 1;
    }
  std::printf("ok   %dx%d -> %dx%d tps %d: %d strips, nvb<=%d, nvk<=%d, patch rows<=%d, %d K-blocks a tile, worst %g\n",
    W,H,OW,OH,tps,p.nstrips,p.nvb_max,p.nvk_max,p.patch_rows_max,p.nk,worst);
  return 0;
}

int main()
{
  int bad=0;
  bad+=run(64,48,256,192,16,true);          // the 4x of config C3, whole strips
  bad+=run(53,37,212,148,16,true);          // ragged: partial tiles, strips and row groups
  bad+=run(300,23,1200,92,8,true);
  bad+=run(100,90,150,135,4,true);          // 1.5x
  bad+=run(41,50,164,150,16,true);          // 4x by 3x
  bad+=run(40,40,41,43,16,false);           // barely an enlargement: windows wider than the ring, two passes
  bad+=run(50,60,100,120,16,true);          // 2x
  bad+=run(33,29,330,290,2,true);           // 10x: several tiles per K-block
  bad+=run(17,9,16*17,16*9,16,true);        // 16x
  bad+=run(1,1,40,40,16,true);              // one source pixel
  bad+=run(600,23,2400,92,16,true);
  bad+=run(64,100,256,400,16,true,6);       // six waves a workgroup
  bad+=run(53,37,212,148,16,true,6);
  if (bad == 0)
    std::printf("ALL OK\n");
  return bad != 0;
}
