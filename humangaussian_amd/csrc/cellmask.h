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
#ifndef __HIPCC__
#include <algorithm>
using std::max;
using std::min;
#endif

#ifndef HGS_HD
#ifdef __HIPCC__
#define HGS_HD __host__ __device__ __forceinline__
#else
#define HGS_HD inline
#endif
#endif

// sqrt / reciprocal / log / clamp / fma of the MASK (not of hgs_alpha_rect, whose result decides which list entries exist): on
// the device the raw 1-ulp instructions (v_sqrt_f32, v_rcp_f32, v_log_f32), v_med3_f32 and v_fma_f32.  HIP's `sqrtf` and `/`
// are correctly rounded by default - a ~15-instruction fix-up sequence per sqrt and ~10 per division, two thirds of this
// function's former ~320 instructions - and the function sits in the per-tile latency chain of the sort kernel (a build
// with a constant mask ran that kernel 6.8 us shorter; the raw forms gave 3 us of it back at one view, 10 us at 8 views).
// The margins below (tau inflated by 0.2 % + 0.03, eps on the intervals) are 1e3 x an ulp of anything compared here;
// tests/test_cellmask_cpu.py runs its brute-force check also on host builds whose sqrt / rcp / log results are pushed
// 3 ulp in either direction (HGS_CM_SKEW_*).
#if defined(__HIP_DEVICE_COMPILE__)
#define HGS_CM_SQRT(x) __builtin_amdgcn_sqrtf(x)
#define HGS_CM_RCP(x) __builtin_amdgcn_rcpf(x)
#define HGS_CM_LN(x) (__builtin_amdgcn_logf(x) * 0.69314718056f)
#define HGS_CM_CLAMP(x, lo, hi) __builtin_amdgcn_fmed3f((x), (lo), (hi))      // (lo <= hi wherever the result is used)
#else
#ifndef HGS_CM_SKEW_SQRT
#define HGS_CM_SKEW_SQRT 1.0f
#define HGS_CM_SKEW_RCP 1.0f
#define HGS_CM_SKEW_LN 1.0f
#endif
#define HGS_CM_SQRT(x) (sqrtf(x) * HGS_CM_SKEW_SQRT)
#define HGS_CM_RCP(x) ((1.0f / (x)) * HGS_CM_SKEW_RCP)
#define HGS_CM_LN(x) (logf(x) * HGS_CM_SKEW_LN)
#define HGS_CM_CLAMP(x, lo, hi) fminf(fmaxf((x), (lo)), (hi))
#endif

#define HGS_CELL 4              // pixels per cell edge
#define HGS_CELLS_PER_TILE 16

// ca cc - cb^2 with ONE rounding error (Kahan's difference of products): for a very elongated Gaussian the plain fp32
// expression cancels - at an anisotropy of 1e-3 it lost every digit, the ellipse's extents came out too small and cleared
// bits hid live pixels (found by brute force at sigma_major ~ 1000 px).
HGS_HD float hgs_conic_det(float ca, float cb, float cc) {
  const float w = cb * cb;
  const float e = fmaf(-cb, cb, w);                     // w - cb^2, exactly
  return fmaf(ca, cc, -w) + e;
}
// what the compensated determinant cannot repair (|det| within a few ulp of ca cc: anisotropy beyond ~3e-4): never cull
HGS_HD bool hgs_conic_cullable(float ca, float cc, float det) {
  return det > 0.0f && ca > 0.0f && cc > 0.0f && det > 1e-6f * (ca * cc);
}

HGS_HD uint32_t hgs_cell_mask(float mx, float my, float ca, float cb, float cc, float op, float x0, float y0) {
#ifdef HGS_DEBUG_NO_CUTS       // (tests only: tests/test_gpu_random_cameras.py builds the library once without the two cuts)
  return 0xffffu;
#endif
  const float a255 = 255.0f * op;
  if (!(a255 >= 0.999f)) return 0u;                     // alpha <= op < 1/255 everywhere
  const float det = hgs_conic_det(ca, cb, cc);
  if (!hgs_conic_cullable(ca, cc, det)) return 0xffffu; // degenerate / extremely elongated conic: never cull
  const float tau = 2.0f * HGS_CM_LN(fmaxf(a255, 1.0f)) * 1.002f + 0.03f;
  const float idet = HGS_CM_RCP(det);
  const float ex = HGS_CM_SQRT(tau * cc * idet), ey = HGS_CM_SQRT(tau * ca * idet);   // half extents of the ellipse
  const float eps = 4e-3f;
  const float ica = HGS_CM_RCP(ca);
  const float dyR = -cb * ex * HGS_CM_RCP(cc);          // dy of the rightmost point; the leftmost one has -dyR
  const float bca = cb * ica, k0 = tau * ica, k1 = det * ica * ica;
  uint32_t mask = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {                         // (branch-free: a band the ellipse misses contributes no bit)
    const float d0 = (y0 + 4.0f * (float)b) - my, d1 = d0 + 3.0f;
    const float lo = fmaxf(d0, -ey), hi = fminf(d1, ey);
    const bool band = lo <= hi + eps;
    const float yr = HGS_CM_CLAMP(dyR, lo, hi), yl = HGS_CM_CLAMP(-dyR, lo, hi);
    const float R = fmaf(-bca, yr, HGS_CM_SQRT(fmaxf(0.0f, fmaf(-k1 * yr, yr, k0)))) + eps;
    const float Lx = fmaf(-bca, yl, -HGS_CM_SQRT(fmaxf(0.0f, fmaf(-k1 * yl, yl, k0)))) - eps;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float c0 = (x0 + 4.0f * (float)c) - mx, c1 = c0 + 3.0f;
      if (band && Lx <= c1 && R >= c0) mask |= 1u << (4 * b + c);
    }
  }
  return mask;
}

// The tile rect that gets list entries: upstream's rect (a 3-sigma circle of the larger eigenvalue, [tmin, tmax) in tile
// units, passed in) cut down to the tiles that the axis-aligned box of the ellipse  q(d) <= tau  can touch - only inside
// that ellipse can a pixel reach alpha >= 1/255, so the dropped (Gaussian, tile) pairs cannot change any pixel (19-22 %
// of upstream's entries on an avatar: flat and faint Gaussians).  Same margins and the same determinant as the cell
// mask; used by hgs_k_preprocess_fwd and, compiled for the host, by tests/test_cellmask_cpu.py.
HGS_HD void hgs_alpha_rect(float mx, float my, float ca, float cb, float cc, float op, int& tminx, int& tminy, int& tmaxx,
                           int& tmaxy) {
#ifdef HGS_DEBUG_NO_CUTS
  return;
#endif
  const float a255 = 255.0f * op;
  const float qdet = hgs_conic_det(ca, cb, cc);
  if (!(a255 >= 0.999f)) {
    tmaxx = tminx; tmaxy = tminy;                        // alpha <= op < 1/255 everywhere
  } else if (hgs_conic_cullable(ca, cc, qdet)) {
    const float tau = 2.0f * logf(fmaxf(a255, 1.0f)) * 1.002f + 0.03f;
    const float ex = sqrtf(tau * cc / qdet) * 1.0005f + 4e-3f, ey = sqrtf(tau * ca / qdet) * 1.0005f + 4e-3f;
    // tile t holds the pixel centres 16 t .. 16 t + 15
    tminx = max(tminx, (int)ceilf((mx - ex - 15.0f) * 0.0625f));
    tminy = max(tminy, (int)ceilf((my - ey - 15.0f) * 0.0625f));
    tmaxx = min(tmaxx, (int)floorf((mx + ex) * 0.0625f) + 1);
    tmaxy = min(tmaxy, (int)floorf((my + ey) * 0.0625f) + 1);
    if (tmaxx < tminx) tmaxx = tminx;
    if (tmaxy < tminy) tmaxy = tminy;
  }
}
