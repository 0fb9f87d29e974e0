// Window geometry shared by the Swin token kernels (swin_tokens.hip, swin_gemm.hip).
#pragma once
#include "common.hpp"
#include "../../include/dua_hip.h"

namespace dua {

struct WinGeom {
  int B, D, H, W, C;
  int wd, wh, ww;        // window extents (already clipped to the map, attention.py:225-251)
  int sd, sh, sw;        // shift (0 on clipped axes)
  int Dp, Hp, Wp;        // padded extents (multiples of the window)
  int nwh, nww, nw, n;   // windows along h, w; windows per sample; tokens per window
};

// window token (sample b, window wi, token t) -> voxel of the padded, un-rolled map; false when it is padding
__device__ __forceinline__ bool window_to_voxel(const WinGeom& g, int wi, int t, int& d, int& h, int& w) {
  const int wz = wi / (g.nwh * g.nww), wy = (wi / g.nww) % g.nwh, wx = wi % g.nww;
  const int tz = t / (g.wh * g.ww), ty = (t / g.ww) % g.wh, tx = t % g.ww;
  d = (wz * g.wd + tz + g.sd) % g.Dp;
  h = (wy * g.wh + ty + g.sh) % g.Hp;
  w = (wx * g.ww + tx + g.sw) % g.Wp;
  return d < g.D && h < g.H && w < g.W;
}

static inline bool geom_ok(const dua_window_geom* p) {
  return p && p->B > 0 && p->D > 0 && p->H > 0 && p->W > 0 && p->C > 0 && p->wd > 0 && p->wh > 0 && p->ww > 0 &&
         p->wd <= p->D && p->wh <= p->H && p->ww <= p->W && p->sd >= 0 && p->sh >= 0 && p->sw >= 0 && p->sd < p->wd &&
         p->sh < p->wh && p->sw < p->ww;
}

static inline WinGeom make_geom(const dua_window_geom* p) {
  WinGeom g;
  g.B = p->B; g.D = p->D; g.H = p->H; g.W = p->W; g.C = p->C;
  g.wd = p->wd; g.wh = p->wh; g.ww = p->ww; g.sd = p->sd; g.sh = p->sh; g.sw = p->sw;
  const int nwd = (p->D + p->wd - 1) / p->wd;
  g.nwh = (p->H + p->wh - 1) / p->wh; g.nww = (p->W + p->ww - 1) / p->ww;
  g.Dp = nwd * p->wd; g.Hp = g.nwh * p->wh; g.Wp = g.nww * p->ww;
  g.nw = nwd * g.nwh * g.nww; g.n = p->wd * p->wh * p->ww;
  return g;
}

}  // namespace dua
