// cellmask.h - conservative per-cell cull mask of one (tile, Gaussian) entry.
//
// The 16x16 tile is cut into 16 CELLS of 4x4 pixels (cell c = cy * 4 + cx).  A pixel can pass the
// blend's  alpha >= 1/255  test only inside the ellipse
//     q(d) = ca dx^2 + 2 cb dx dy + cc dy^2 <= tau,   tau = 2 ln(255 op),   d = pixel - mean.
// Bit c of the mask is set iff that ellipse meets the rectangle spanned by the cell's pixel centres.
// The test is exact up to its safety margins: for each of the four 4-pixel-high bands the
// ellipse-and-band region is convex, so its projection on x is ONE interval [L, R] whose ends are
// the ellipse's leftmost / rightmost points clamped into the band; a cell of the band is hit iff
// its x range meets [L, R].  Margins (tau inflated, 1e-3 px on the intervals) absorb fp32
// rounding: a set bit never changes a result, a cleared bit must be provably empty.
//
// Plain C++ (no HIP types) so that the SAME function is compiled into the sort kernel (binning.hip)
// and into the host-side checker tests/test_cellmask_cpu.py drives against brute force.
#pragma once
#include <math.h>
#include <stdint.h>

#ifndef HGS_HD
#ifdef __HIPCC__
#define HGS_HD __host__ __device__ __forceinline__
#else
#define HGS_HD inline
#endif
#endif

#define HGS_CELL 4              // pixels per cell edge
#define HGS_CELLS_PER_TILE 16

HGS_HD uint32_t hgs_cell_mask(float mx, float my, float ca, float cb, float cc, float op, float x0, float y0) {
  const float a255 = 255.0f * op;
  if (!(a255 >= 0.999f)) return 0u;                     // alpha <= op < 1/255 everywhere
  const float det = ca * cc - cb * cb;
  if (!(det > 0.0f && ca > 0.0f && cc > 0.0f)) return 0xffffu;   // degenerate conic: never cull
  const float tau = 2.0f * logf(fmaxf(a255, 1.0f)) * 1.002f + 0.03f;
  const float idet = 1.0f / det;
  const float ex = sqrtf(tau * cc * idet), ey = sqrtf(tau * ca * idet);   // half extents of the ellipse
  const float eps = 2e-3f;
  const float ica = 1.0f / ca;
  const float dyR = -cb * ex / cc;                      // dy of the rightmost point; the leftmost one has -dyR
  const float bca = cb * ica, k0 = tau * ica, k1 = det * ica * ica;
  uint32_t mask = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const float d0 = (y0 + 4.0f * (float)b) - my, d1 = d0 + 3.0f;
    const float lo = fmaxf(d0, -ey), hi = fminf(d1, ey);
    if (!(lo <= hi + eps)) continue;                    // band misses the ellipse
    const float yr = fminf(fmaxf(dyR, lo), hi), yl = fminf(fmaxf(-dyR, lo), hi);
    const float R = -bca * yr + sqrtf(fmaxf(0.0f, k0 - k1 * yr * yr)) + eps;
    const float Lx = -bca * yl - sqrtf(fmaxf(0.0f, k0 - k1 * yl * yl)) - eps;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float c0 = (x0 + 4.0f * (float)c) - mx, c1 = c0 + 3.0f;
      if (Lx <= c1 && R >= c0) mask |= 1u << (4 * b + c);
    }
  }
  return mask;
}

// Tile-level version of the same test: can the Gaussian reach alpha >= 1/255 anywhere on the tile's 16 x 16 pixel
// centres?  Exact minimum of q over the rectangle (the minimum of a convex quadratic over a box is 0 if the centre is
// inside, else it lies on an edge, where the free coordinate's optimum is the clamped 1-D minimiser), with the same
// inflated tau as hgs_cell_mask and the rectangle grown by the same eps: a SUPERSET of "hgs_cell_mask != 0".
// The binning stage drops (Gaussian, tile) pairs that fail it (22 % of upstream's entries on an avatar): they
// cannot change any pixel.  Evaluated from the GeomRec fields by preprocess_fwd (count), fill (scatter) and
// preprocess_bwd (which rows exist): the same inputs, the same decision.
HGS_HD bool hgs_tile_hit(float mx, float my, float ca, float cb, float cc, float op, float x0, float y0) {
  const float a255 = 255.0f * op;
  if (!(a255 >= 0.999f)) return false;
  const float det = ca * cc - cb * cb;
  if (!(det > 0.0f && ca > 0.0f && cc > 0.0f)) return true;
  const float tau = 2.0f * logf(fmaxf(a255, 1.0f)) * 1.002f + 0.03f;
  const float eps = 4e-3f;
  const float xa = x0 - eps, xb = x0 + 15.0f + eps, ya = y0 - eps, yb = y0 + 15.0f + eps;
  const float bc = cb / cc, ba = cb / ca;
  float best;
  {
    const float dx = fminf(fmaxf(mx, xa), xb) - mx, dy = fminf(fmaxf(my, ya), yb) - my;      // 0 when the centre is inside
    best = ca * dx * dx + 2.0f * cb * dx * dy + cc * dy * dy;
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    {  // vertical edges x = xa / xb
      const float dx = (e ? xb : xa) - mx;
      const float dy = fminf(fmaxf(my - bc * dx, ya), yb) - my;
      best = fminf(best, ca * dx * dx + 2.0f * cb * dx * dy + cc * dy * dy);
    }
    {  // horizontal edges y = ya / yb
      const float dy = (e ? yb : ya) - my;
      const float dx = fminf(fmaxf(mx - ba * dy, xa), xb) - mx;
      best = fminf(best, ca * dx * dx + 2.0f * cb * dx * dy + cc * dy * dy);
    }
  }
  return best <= tau * 1.0005f + 1e-3f * best;
}
