// Host-side geometry of the one-launch resize on the fp64 matrix pipe (resize_mfma.hip).
//
// ResizeImage for an enlargement runs VerticalFilter, then HorizontalFilter
// (MagickCore/resize.c:3846-3861, :3549-3759, :3333-3547).  Both are banded matrix products:
//     I[x][y]   = sum_k P[x][k] * Wv[k][y]        (x source column, k source row, y output row)
//     O[y][xo]  = sum_k I'[y][k] * Wh[k][xo]      (k source column, xo output column)
// with P / I' the alpha-premultiplied samples (alpha*colour, alpha).  v_mfma_f64_16x16x4_f64
// evaluates a 16x16 block of either product per K-block of 4 source rows / columns, and the
// result layout of the first product (lane = y, register r of lane group g = source column
// g+4r) IS the A-operand layout of the second one for K-block r — the intermediate stays in
// registers.  This header flattens the contribution lists into the per-lane weight blocks and the
// per-strip / per-tile / per-row-group indices the kernel walks.  Pure host C++ (no HIP), so the
// CPU test tests/cpu/resize_mfma_plan_test.cpp can emulate the kernel's walk with the same tables.
//
//   out tile     16 output columns; its source window [lo,hi) in K-blocks relative to the strip
//   strip        `tps` consecutive out tiles = the columns one workgroup owns; its source columns
//                [col_lo, col_lo+16*nvb) are produced by the vertical product in `nvb` blocks of 16
//   row group    16 output rows = one wave; its source rows [row_lo, row_lo+4*nvk)
//   weight block 64 doubles, one per lane: lane (g=lane>>4, n=lane&15) holds
//                Wh[col_lo+4*(kb0+j)+g][16*t+n]  resp.  Wv[row_lo+4*kb+g][16*rg+n]
#pragma once

#include <algorithm>
#include <cstddef>
#include <vector>

namespace mh {

struct MfmaResizePlan
{
  static constexpr int kTile=16;       // output columns per tile, output rows per row group
  int waves=4;                         // row groups (waves) per workgroup step
  static constexpr int kMaxVK=6;       // K-blocks of 4 source rows per row group the kernel unrolls
  int tps=16;                          // tiles per strip
  int out_columns=0,out_rows=0;
  int ntiles=0,nstrips=0,nrg=0;
  int nvb_max=1,nvk_max=1,patch_rows_max=1,wblocks_max=1;
  std::vector<int> strip_col_lo,strip_nvb,strip_wbase,strip_wcount;   // [nstrips]
  // Every tile takes exactly `nk` K-blocks (the widest window's count; narrower windows are padded
  // with zero-weight blocks), so the kernel's matrix chain is straight-line code: tile i of a strip
  // owns weight blocks [i*nk, (i+1)*nk) of the strip and its first K-block sits in ring slot sl0.
  int nk=1;
  std::vector<int> tile_kb0;                                          // [ntiles] first K-block (may be < 0)
  // per tile, the word the kernel keeps in LDS: sl0 | vb<<8 (vb: the block after which the tile's
  // window is complete; the ring then holds blocks vb-1 (slots 0..3) and vb (slots 4..7))
  std::vector<unsigned> tile_meta;                                    // [ntiles]
  std::vector<double> wh;              // [blocks][64]
  std::vector<int> rg_row_lo,rg_nvk,rg_woff;                          // [nrg]
  std::vector<double> wv;              // [blocks][64]
};

// Table: {int out_size; std::vector<int> start,count; std::vector<double> weight /* [tap][out] */;}
// Returns false when the geometry does not fit the kernel's walk (the caller runs two passes):
// an output without contributions, a tile window that does not fit two 16-column blocks, a row
// group that needs more than kMaxVK K-blocks.
template<class Table>
static bool build_mfma_resize_plan(MfmaResizePlan &p,const Table &vt,const Table &ht,int tps,int waves)
{
  constexpr int T=MfmaResizePlan::kTile;
  p.tps=tps;
  p.waves=waves;
  p.out_columns=ht.out_size;
  p.out_rows=vt.out_size;
  if ((p.out_columns <= 0) || (p.out_rows <= 0) || (tps < 1) || (waves < 1))
    return false;
  for (int i=0; i < ht.out_size; i++)
    if (ht.count[(size_t) i] <= 0)
      return false;
  for (int i=0; i < vt.out_size; i++)
    if (vt.count[(size_t) i] <= 0)
      return false;
  p.ntiles=(p.out_columns+T-1)/T;
  p.nstrips=(p.ntiles+tps-1)/tps;
  p.nrg=(p.out_rows+T-1)/T;

  // ---- horizontal: tiles and strips
  std::vector<int> tlo((size_t) p.ntiles),thi((size_t) p.ntiles);
  for (int t=0; t < p.ntiles; t++)
    {
      int lo=0x7fffffff,hi=0;
      for (int x=t*T; (x < (t+1)*T) && (x < p.out_columns); x++)
        {
          lo=std::min(lo,ht.start[(size_t) x]);
          hi=std::max(hi,ht.start[(size_t) x]+ht.count[(size_t) x]);
        }
      tlo[(size_t) t]=lo;
      thi[(size_t) t]=hi;
      if ((t > 0) && ((lo < tlo[(size_t) t-1]) || (hi < thi[(size_t) t-1])))
        return false;                   // the walk needs monotone windows
    }
  p.strip_col_lo.assign((size_t) p.nstrips,0);
  p.strip_nvb.assign((size_t) p.nstrips,0);
  p.strip_wbase.assign((size_t) p.nstrips,0);
  p.strip_wcount.assign((size_t) p.nstrips,0);
  p.tile_kb0.assign((size_t) p.ntiles,0);
  p.tile_meta.assign((size_t) p.ntiles,0u);
  p.nvb_max=1;
  p.nk=1;
  std::vector<int> tile_vb((size_t) p.ntiles,0);
  for (int s=0; s < p.nstrips; s++)
    {
      const int t0=s*tps,t1=std::min(p.ntiles,t0+tps);
      const int col_lo=tlo[(size_t) t0];
      const int col_hi=thi[(size_t) t1-1];
      const int nvb=(col_hi-col_lo+15)/16;
      p.strip_col_lo[(size_t) s]=col_lo;
      p.strip_nvb[(size_t) s]=nvb;
      p.nvb_max=std::max(p.nvb_max,nvb);
      for (int t=t0; t < t1; t++)
        {
          const int kb0=(tlo[(size_t) t]-col_lo)/4;
          const int kbe=(thi[(size_t) t]-1-col_lo)/4;
          const int vb=kbe/4;
          // the ring holds the K-blocks of vertical blocks vb-1 and vb
          if ((kb0 < 4*(vb-1)) || (vb > 254))
            return false;
          p.tile_kb0[(size_t) t]=kb0;
          tile_vb[(size_t) t]=vb;
          p.nk=std::max(p.nk,kbe-kb0+1);
        }
    }
  int wtotal=0;
  for (int s=0; s < p.nstrips; s++)
    {
      const int t0=s*tps,t1=std::min(p.ntiles,t0+tps);
      p.strip_wbase[(size_t) s]=wtotal;
      p.strip_wcount[(size_t) s]=(t1-t0)*p.nk;
      wtotal+=(t1-t0)*p.nk;
      for (int t=t0; t < t1; t++)
        {
          // pad to nk blocks: move the first block down until the last one is the ring's slot 7 at most
          const int vb=tile_vb[(size_t) t];
          const int kb0=std::min(p.tile_kb0[(size_t) t],4*vb+4-p.nk);
          p.tile_kb0[(size_t) t]=kb0;
          p.tile_meta[(size_t) t]=(unsigned) (kb0-4*(vb-1)) | ((unsigned) vb << 8);
        }
    }
  p.wblocks_max=tps*p.nk;
  p.wh.assign((size_t) wtotal*64,0.0);
  for (int s=0; s < p.nstrips; s++)
    {
      const int t0=s*tps,t1=std::min(p.ntiles,t0+tps);
      const int col_lo=p.strip_col_lo[(size_t) s];
      for (int t=t0; t < t1; t++)
        for (int j=0; j < p.nk; j++)
          {
            double *blk=&p.wh[((size_t) p.strip_wbase[(size_t) s]+(size_t) (t-t0)*(size_t) p.nk+(size_t) j)*64];
            for (int lane=0; lane < 64; lane++)
              {
                const int g=lane >> 4,n=lane & 15;
                const int x=t*T+n;
                const int col=col_lo+4*(p.tile_kb0[(size_t) t]+j)+g;
                if (x >= p.out_columns)
                  continue;
                const int k=col-ht.start[(size_t) x];
                if ((k >= 0) && (k < ht.count[(size_t) x]))
                  blk[lane]=ht.weight[(size_t) k*(size_t) ht.out_size+(size_t) x];
              }
          }
    }

  // ---- vertical: row groups
  p.rg_row_lo.assign((size_t) p.nrg,0);
  p.rg_nvk.assign((size_t) p.nrg,0);
  p.rg_woff.assign((size_t) p.nrg,0);
  p.nvk_max=1;
  int vtotal=0;
  for (int rg=0; rg < p.nrg; rg++)
    {
      int lo=0x7fffffff,hi=0;
      for (int y=rg*T; (y < (rg+1)*T) && (y < p.out_rows); y++)
        {
          lo=std::min(lo,vt.start[(size_t) y]);
          hi=std::max(hi,vt.start[(size_t) y]+vt.count[(size_t) y]);
        }
      if ((rg > 0) && (lo < p.rg_row_lo[(size_t) rg-1]))
        return false;
      const int nvk=(hi-lo+3)/4;
      if (nvk > MfmaResizePlan::kMaxVK)
        return false;
      p.rg_row_lo[(size_t) rg]=lo;
      p.rg_nvk[(size_t) rg]=nvk;
      p.rg_woff[(size_t) rg]=vtotal;
      p.nvk_max=std::max(p.nvk_max,nvk);
      vtotal+=nvk;
    }
  p.wv.assign((size_t) vtotal*64,0.0);
  for (int rg=0; rg < p.nrg; rg++)
    for (int kb=0; kb < p.rg_nvk[(size_t) rg]; kb++)
      {
        double *blk=&p.wv[((size_t) p.rg_woff[(size_t) rg]+(size_t) kb)*64];
        for (int lane=0; lane < 64; lane++)
          {
            const int g=lane >> 4,n=lane & 15;
            const int y=rg*T+n;
            const int row=p.rg_row_lo[(size_t) rg]+4*kb+g;
            if (y >= p.out_rows)
              continue;
            const int k=row-vt.start[(size_t) y];
            if ((k >= 0) && (k < vt.count[(size_t) y]))
              blk[lane]=vt.weight[(size_t) k*(size_t) vt.out_size+(size_t) y];
          }
      }
  // source rows one workgroup step (`waves` row groups) stages
  p.patch_rows_max=1;
  for (int rg0=0; rg0 < p.nrg; rg0+=waves)
    {
      int hi=0;
      for (int rg=rg0; (rg < rg0+waves) && (rg < p.nrg); rg++)
        hi=std::max(hi,p.rg_row_lo[(size_t) rg]+4*p.rg_nvk[(size_t) rg]);
      p.patch_rows_max=std::max(p.patch_rows_max,hi-p.rg_row_lo[(size_t) rg0]);
    }
  return true;
}

} // namespace mh
