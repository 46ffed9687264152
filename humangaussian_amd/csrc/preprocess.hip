// preprocess.hip - per-Gaussian forward (stage F1 of SURVEY.md 2.3(B)) and its backward
// (B2 + B3 fused, plus the deterministic gather of the per-entry gradient rows).
//
// Replaces upstream's preprocessCUDA / computeCov2DCUDA / computeCov3D / computeColorFromSH
// (un-vendored; behaviour specified in SURVEY.md Appendix A.2 / A.6).  The arithmetic is
// written in the exact left-to-right order of oracle/gs_oracle.py::preprocess and the
// library is built with -ffp-contract=off, so radii / rects match the oracle bit-for-bit.
//
// Batched: the forward runs over the B*P (view, Gaussian) instances of a call; the backward runs
// one thread per Gaussian that walks its B views in order and writes every parameter gradient
// ONCE as the sum over the views (deterministic; dL/dmeans2D stays per view).
//
// Roofline: pure streaming, HBM-bound.  Forward reads 44+12*M' B and writes 64+4 B per
// (view, Gaussian) (M' = active SH coefficients); backward reads 44+12*M' + B*(64 +
// 48*tiles_touched) and writes 44+12*M + 12*B bytes per Gaussian.
#include "hgs_common.h"

namespace {

__device__ __constant__ const float SH_C0 = 0.28209479177387814f;
__device__ __constant__ const float SH_C1 = 0.4886025119029199f;
__device__ __constant__ const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f,
                                                0.31539156525252005f, -1.0925484305920792f,
                                                0.5462742152960396f};
__device__ __constant__ const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f,
                                                -0.4570457994644658f, 0.3731763325901154f,
                                                -0.4570457994644658f, 1.445305721320277f,
                                                -0.5900435899266435f};

struct Cov3 { float c0, c1, c2, c3, c4, c5; };   // xx xy xz yy yz zz

struct RotScale {
  float R00, R01, R02, R10, R11, R12, R20, R21, R22;
  float sx, sy, sz;
};

// fused activations (HGS_ACT_*): what GaussianModel.get_opacity / get_scaling / get_rotation compute
__device__ __forceinline__ float act_opacity(float raw, int act) {
  return (act & HGS_ACT_OPACITY_SIGMOID) ? 1.0f / (1.0f + expf(-raw)) : raw;
}
__device__ __forceinline__ void act_scale(const float* __restrict__ scales, int i, int act, float& s0, float& s1, float& s2) {
  s0 = scales[3 * i + 0]; s1 = scales[3 * i + 1]; s2 = scales[3 * i + 2];
  if (act & HGS_ACT_SCALE_EXP) { s0 = expf(s0); s1 = expf(s1); s2 = expf(s2); }
}
// returns the quaternion the kernels use and, for the backward, 1 / max(|q_raw|, 1e-12) (F.normalize)
__device__ __forceinline__ float4 act_rotation(const float* __restrict__ rots, int i, int act, float& inv_norm) {
  float4 q = reinterpret_cast<const float4*>(rots)[i];
  inv_norm = 1.0f;
  if (act & HGS_ACT_ROTATION_NORMALIZE) {
    inv_norm = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    q.x *= inv_norm; q.y *= inv_norm; q.z *= inv_norm; q.w *= inv_norm;
  }
  return q;
}

__device__ __forceinline__ RotScale make_rotscale(const float* __restrict__ scales,
                                                  const float* __restrict__ rots, int i,
                                                  float mod, int act) {
  RotScale o;
  float s0, s1, s2, inv_norm;
  act_scale(scales, i, act, s0, s1, s2);
  o.sx = mod * s0;
  o.sy = mod * s1;
  o.sz = mod * s2;
  const float4 q = act_rotation(rots, i, act, inv_norm);
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  o.R00 = 1.0f - 2.0f * (y * y + z * z);
  o.R01 = 2.0f * (x * y - r * z);
  o.R02 = 2.0f * (x * z + r * y);
  o.R10 = 2.0f * (x * y + r * z);
  o.R11 = 1.0f - 2.0f * (x * x + z * z);
  o.R12 = 2.0f * (y * z - r * x);
  o.R20 = 2.0f * (x * z - r * y);
  o.R21 = 2.0f * (y * z + r * x);
  o.R22 = 1.0f - 2.0f * (x * x + y * y);
  return o;
}

__device__ __forceinline__ Cov3 cov3d_from(const RotScale& q) {
  const float L00 = q.R00 * q.sx, L01 = q.R01 * q.sy, L02 = q.R02 * q.sz;
  const float L10 = q.R10 * q.sx, L11 = q.R11 * q.sy, L12 = q.R12 * q.sz;
  const float L20 = q.R20 * q.sx, L21 = q.R21 * q.sy, L22 = q.R22 * q.sz;
  Cov3 c;
  c.c0 = L00 * L00 + L01 * L01 + L02 * L02;
  c.c1 = L00 * L10 + L01 * L11 + L02 * L12;
  c.c2 = L00 * L20 + L01 * L21 + L02 * L22;
  c.c3 = L10 * L10 + L11 * L11 + L12 * L12;
  c.c4 = L10 * L20 + L11 * L21 + L12 * L22;
  c.c5 = L20 * L20 + L21 * L21 + L22 * L22;
  return c;
}

// Everything the EWA projection produces that forward and backward both need.
struct Proj2D {
  float tx0, ty0, tz;          // view-space mean
  float tx, ty;                // after the frustum clamp
  bool clx, cly;
  float M00, M01, M02, M10, M11, M12;
  float u0, u1, u2, w0, w1, w2;
  float a, b, c, det;
};

__device__ __forceinline__ void view_point(const float* __restrict__ V, float x, float y,
                                           float z, float& tx, float& ty, float& tz) {
  tx = V[0] * x + V[4] * y + V[8] * z + V[12];
  ty = V[1] * x + V[5] * y + V[9] * z + V[13];
  tz = V[2] * x + V[6] * y + V[10] * z + V[14];
}

__device__ __forceinline__ void project_cov(const Cam& v, const float* __restrict__ V,
                                            const Cov3& s, Proj2D& o) {
  const float limx = 1.3f * v.tanfovx, limy = 1.3f * v.tanfovy;
  const float txtz = o.tx0 / o.tz, tytz = o.ty0 / o.tz;
  o.clx = (txtz < -limx) || (txtz > limx);
  o.cly = (tytz < -limy) || (tytz > limy);
  o.tx = fminf(limx, fmaxf(-limx, txtz)) * o.tz;
  o.ty = fminf(limy, fmaxf(-limy, tytz)) * o.tz;
  const float J00 = v.focal_x / o.tz;
  const float J02 = -(v.focal_x * o.tx) / (o.tz * o.tz);
  const float J11 = v.focal_y / o.tz;
  const float J12 = -(v.focal_y * o.ty) / (o.tz * o.tz);
  o.M00 = J00 * V[0] + J02 * V[2];
  o.M01 = J00 * V[4] + J02 * V[6];
  o.M02 = J00 * V[8] + J02 * V[10];
  o.M10 = J11 * V[1] + J12 * V[2];
  o.M11 = J11 * V[5] + J12 * V[6];
  o.M12 = J11 * V[9] + J12 * V[10];
  o.u0 = o.M00 * s.c0 + o.M01 * s.c1 + o.M02 * s.c2;
  o.u1 = o.M00 * s.c1 + o.M01 * s.c3 + o.M02 * s.c4;
  o.u2 = o.M00 * s.c2 + o.M01 * s.c4 + o.M02 * s.c5;
  o.w0 = o.M10 * s.c0 + o.M11 * s.c1 + o.M12 * s.c2;
  o.w1 = o.M10 * s.c1 + o.M11 * s.c3 + o.M12 * s.c4;
  o.w2 = o.M10 * s.c2 + o.M11 * s.c4 + o.M12 * s.c5;
  o.a = (o.u0 * o.M00 + o.u1 * o.M01 + o.u2 * o.M02) + 0.3f;
  o.b = o.u0 * o.M10 + o.u1 * o.M11 + o.u2 * o.M12;
  o.c = (o.w0 * o.M10 + o.w1 * o.M11 + o.w2 * o.M12) + 0.3f;
  o.det = o.a * o.c - o.b * o.b;
}

// One Gaussian's (M,3) SH block as 16 B loads into registers.  A thread's block is 12 M bytes
// (192 B at degree 3), so scalar loads cost 48 uncoalescable instructions per thread; when 3 M is
// a multiple of 4 the block is 16 B aligned and twelve dwordx4 loads do (SH degree 3:
// preprocess_fwd 90 -> ~45 us at 500k Gaussians).  `sh48` must be indexed with constants only.
__device__ __forceinline__ bool sh_block_vectorisable(int M) { return ((M * 3) & 3) == 0 && M <= 16; }
__device__ __forceinline__ void load_sh_block(const float* __restrict__ src, int M, float (&sh48)[48]) {
  const float4* s4 = reinterpret_cast<const float4*>(src);
  const int nq = (M * 3) >> 2;
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < nq) t = s4[q];
    sh48[4 * q + 0] = t.x; sh48[4 * q + 1] = t.y; sh48[4 * q + 2] = t.z; sh48[4 * q + 3] = t.w;
  }
}

// ---- SH blocks through LDS (M >= 4) in the single-view per-Gaussian BACKWARD.  A thread's (M,3) block is 12 M contiguous
// bytes, so per-thread 16 B accesses touch 64 different cache lines per wave instruction - for the gradient block that
// is 12 partial-line stores per line.  The 256 blocks of a chunk are ONE contiguous, 16 B-aligned range: the workgroup
// moves it with fully coalesced dwordx4 accesses through LDS rows of an ODD number of float4 (conflict-free b128 row
// accesses), in (shs) and out (dL_dshs): hgs_k_preprocess_bwd_s3 96 -> 77 us at 500k Gaussians, bit-identical results.
// (The same staging of the FORWARD's SH loads lost - preprocess_fwd 54 -> 68 us at 500k: its loads were not what it
// waits for, the LDS round trip and two more barriers per chunk were pure cost.  EXPERIMENTS.md, round 5.)
__host__ __device__ __forceinline__ int hgs_sh_row_f4(int M) { const int q = (3 * M + 3) / 4; return (q & 1) ? q : q + 1; }
__host__ __device__ __forceinline__ bool hgs_sh_staged(int M) { return M >= 4 && M <= 16; }

// stage the SH blocks of Gaussians [i0, i0 + cnt) of one view-independent tensor into `stage` ([256][row_f4 * 4] floats)
__device__ __forceinline__ void stage_sh_chunk(const float* __restrict__ shs, int M, int i0, int cnt, float* __restrict__ stage) {
  const int m3 = 3 * M, rowf = hgs_sh_row_f4(M) * 4;
  const float* __restrict__ src = shs + (size_t)i0 * m3;                 // (i0 is a multiple of 256: 16 B aligned for every M)
  const int nfl = cnt * m3, nfull = nfl >> 2;
  const float4* __restrict__ s4 = reinterpret_cast<const float4*>(src);
  for (int f = (int)threadIdx.x; f < nfull; f += HGS_BLOCK) {
    const float4 t = s4[f];
    int g = (4 * f) / m3, c = 4 * f - g * m3;
    const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      stage[g * rowf + c] = tv[j];
      if (++c == m3) { c = 0; ++g; }
    }
  }
  if ((int)threadIdx.x < (nfl & 3)) {                                    // the last 1-3 floats of the tensor's last chunk
    const int e = 4 * nfull + (int)threadIdx.x;
    stage[(e / m3) * rowf + (e % m3)] = src[e];
  }
}

// a thread's staged row -> registers (`sh48` must be indexed with constants only)
__device__ __forceinline__ void load_sh_row_lds(const float* __restrict__ row, int M, float (&sh48)[48]) {
  const float4* r4 = reinterpret_cast<const float4*>(row);
  const int nq = (3 * M + 3) >> 2;
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < nq) t = r4[q];
    sh48[4 * q + 0] = t.x; sh48[4 * q + 1] = t.y; sh48[4 * q + 2] = t.z; sh48[4 * q + 3] = t.w;
  }
}

// the inverse of stage_sh_chunk: the rows of a FULL chunk (256 Gaussians) leave LDS as one contiguous, coalesced block
__device__ __forceinline__ void unstage_sh_chunk(float* __restrict__ dst_all, int M, int i0, const float* __restrict__ stage) {
  const int m3 = 3 * M, rowf = hgs_sh_row_f4(M) * 4;
  float4* __restrict__ d4 = reinterpret_cast<float4*>(dst_all + (size_t)i0 * m3);
  const int nfull = (HGS_BLOCK * m3) >> 2;                                // (256 * 3 M is a multiple of 4)
  for (int f = (int)threadIdx.x; f < nfull; f += HGS_BLOCK) {
    int g = (4 * f) / m3, c = 4 * f - g * m3;
    float tv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tv[j] = stage[g * rowf + c];
      if (++c == m3) { c = 0; ++g; }
    }
    d4[f] = make_float4(tv[0], tv[1], tv[2], tv[3]);
  }
}
// does the single-view per-Gaussian backward of this workgroup move its SH blocks through LDS?  (host: LDS size; device:
// workgroup-uniform - full chunks only, so that no thread of a staging workgroup is idle)
__host__ __device__ __forceinline__ size_t hgs_pre_bwd_stage_bytes(int M, int deg, bool has_sh) {
  return (deg > 0 && has_sh && hgs_sh_staged(M)) ? (size_t)HGS_BLOCK * hgs_sh_row_f4(M) * 16 : 0;
}

// SH basis evaluation for one Gaussian; sh points at its (M,3) block.
__device__ __forceinline__ void eval_sh(int deg, const float* __restrict__ sh, float x,
                                        float y, float z, float out[3]) {
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float res = SH_C0 * sh[ch];
    if (deg > 0) {
      res = res - SH_C1 * y * sh[3 + ch] + SH_C1 * z * sh[6 + ch] - SH_C1 * x * sh[9 + ch];
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z;
        const float xy = x * y, yz = y * z, xz = x * z;
        res = res + SH_C2[0] * xy * sh[12 + ch] + SH_C2[1] * yz * sh[15 + ch] +
              SH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + ch] + SH_C2[3] * xz * sh[21 + ch] +
              SH_C2[4] * (xx - yy) * sh[24 + ch];
        if (deg > 2) {
          res = res + SH_C3[0] * y * (3.0f * xx - yy) * sh[27 + ch] +
                SH_C3[1] * xy * z * sh[30 + ch] +
                SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + ch] +
                SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + ch] +
                SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + ch] +
                SH_C3[5] * z * (xx - yy) * sh[42 + ch] +
                SH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + ch];
        }
      }
    }
    out[ch] = res;
  }
}

}  // namespace

// ------------------------------------------------------------------------------ forward
namespace {

// Everything stage F1 computes for Gaussian i.  Returns tiles_touched.
__device__ __forceinline__ uint32_t preprocess_one(
    const View& v, const Cam& cam, int i, const float* __restrict__ means3D, const float* __restrict__ shs,
    const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
    const float* __restrict__ scales, const float* __restrict__ rotations,
    const float* __restrict__ cov3D_precomp, GeomRec& rec) {
  const float* __restrict__ V = cam.viewmatrix;
  const float* __restrict__ PM = cam.projmatrix;
  rec.mx = rec.my = rec.ca = rec.cb = rec.cc = rec.op = 0.f;
  rec.r = rec.g = rec.b = rec.depth = 0.f;
  rec.rect_lo = rec.rect_hi = rec.offset = 0u;
  rec.radius = 0;
  rec.clamped = rec.flags = 0u;
  const float x = means3D[3 * i + 0], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
  Proj2D pj;
  view_point(V, x, y, z, pj.tx0, pj.ty0, pj.tz);
  if (!(pj.tz > HGS_NEAR_Z)) return 0;
  const float hx = PM[0] * x + PM[4] * y + PM[8] * z + PM[12];
  const float hy = PM[1] * x + PM[5] * y + PM[9] * z + PM[13];
  const float hw = PM[3] * x + PM[7] * y + PM[11] * z + PM[15];
  const float pw = 1.0f / (hw + 0.0000001f);
  const float projx = hx * pw, projy = hy * pw;
  Cov3 s;
  if (cov3D_precomp) {
    const float* c = cov3D_precomp + 6 * (size_t)i;
    s.c0 = c[0]; s.c1 = c[1]; s.c2 = c[2]; s.c3 = c[3]; s.c4 = c[4]; s.c5 = c[5];
  } else {
    s = cov3d_from(make_rotscale(scales, rotations, i, v.scale_modifier, v.act));
  }
  project_cov(cam, V, s, pj);
  if (pj.det == 0.0f) return 0;
  const float det_inv = 1.0f / pj.det;
  const float mid = 0.5f * (pj.a + pj.c);
  const float root = sqrtf(fmaxf(0.1f, mid * mid - pj.det));
  const float lam = fmaxf(mid + root, mid - root);
  const float radf = ceilf(3.0f * sqrtf(lam));
  const float mx = ((projx + 1.0f) * (float)v.W - 1.0f) * 0.5f;
  const float my = ((projy + 1.0f) * (float)v.H - 1.0f) * 0.5f;
  const int gxm = v.grid_x, gym = v.grid_y;
  const int rminx = min(gxm, max(0, (int)((mx - radf) / 16.0f)));
  const int rminy = min(gym, max(0, (int)((my - radf) / 16.0f)));
  const int rmaxx = min(gxm, max(0, (int)((mx + radf + 15.0f) / 16.0f)));
  const int rmaxy = min(gym, max(0, (int)((my + radf + 15.0f) / 16.0f)));
  const int area = (rmaxx - rminx) * (rmaxy - rminy);
  if (area <= 0) return 0;
  rec.mx = mx; rec.my = my;
  rec.ca = pj.c * det_inv; rec.cb = -pj.b * det_inv; rec.cc = pj.a * det_inv;
  rec.op = act_opacity(opacities[i], v.act);
  rec.depth = pj.tz;
  // The tile rect that gets list entries: upstream's rect cut down to the box of the alpha >= 1/255 ellipse (cellmask.h);
  // radii / visibility stay upstream's.
  int tminx = rminx, tminy = rminy, tmaxx = rmaxx, tmaxy = rmaxy;
  hgs_alpha_rect(mx, my, rec.ca, rec.cb, rec.cc, rec.op, tminx, tminy, tmaxx, tmaxy);
  rec.rect_lo = (uint32_t)tminx | ((uint32_t)tminy << 16);
  rec.rect_hi = (uint32_t)tmaxx | ((uint32_t)tmaxy << 16);
  rec.radius = (int)radf;
  const int tight_area = (tmaxx - tminx) * (tmaxy - tminy);
  rec.flags = (pj.clx ? 1u : 0u) | (pj.cly ? 2u : 0u);
  if (colors_precomp) {
    rec.r = colors_precomp[3 * i + 0];
    rec.g = colors_precomp[3 * i + 1];
    rec.b = colors_precomp[3 * i + 2];
  } else {
    const float* cp = cam.campos;
    const float ddx = x - cp[0], ddy = y - cp[1], ddz = z - cp[2];
    const float n = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
    float col[3];
    if (sh_block_vectorisable(v.M)) {
      float sh48[48];
      load_sh_block(shs + (size_t)i * v.M * 3, v.M, sh48);
      eval_sh(v.D, sh48, ddx / n, ddy / n, ddz / n, col);
    } else {
      eval_sh(v.D, shs + (size_t)i * v.M * 3, ddx / n, ddy / n, ddz / n, col);
    }
    uint32_t cl = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      col[ch] = col[ch] + 0.5f;
      if (col[ch] < 0.0f) { cl |= (1u << ch); col[ch] = 0.0f; }
    }
    rec.r = col[0]; rec.g = col[1]; rec.b = col[2];
    rec.clamped = cl;
  }
  return (uint32_t)tight_area;
}

__device__ __forceinline__ void store_geom(GeomRec* dstp, const GeomRec& rec) {
  uint4* dst = reinterpret_cast<uint4*>(dstp);
  const uint4* src = reinterpret_cast<const uint4*>(&rec);
  dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
}

}  // namespace

// Entry ids: a Gaussian's tiles_touched entries get CONTIGUOUS ids = chunk base + exclusive prefix
// inside its 256-Gaussian chunk.  The prefix is formed here; the chunk bases are bump-allocated by
// hgs_k_tiles from the chunk sums written here (where a range lives is irrelevant: the rows of
// one Gaussian are contiguous and summed in a fixed order by hgs_k_preprocess_bwd).
__device__ __forceinline__ uint32_t chunk_prefix(uint32_t tt, uint32_t* wtot, uint32_t* chunk_sum) {
  uint32_t total;
  const uint32_t ex = hgs_block_excl_scan<HGS_BLOCK>(tt, wtot, total);
  if (threadIdx.x == 0) *chunk_sum = total;
  return ex;
}

// the first workgroup of the first kernel of a forward call clears the call's counters
__device__ __forceinline__ void zero_counters(const Layout& L) {
  if (blockIdx.x == 0)
    for (uint32_t i = threadIdx.x; i < sizeof(Counters) / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(L.ctr)[i] = 0u;
}

// LDS-histogram variant (T*4 bytes of dynamic LDS <= 64 KB).  Workgroup (view b, g) owns the
// Gaussian chunks [g*cpw, (g+1)*cpw) of view b (256 Gaussians each); it counts its tile hits with
// LDS atomics and writes its histogram ROW hist[b*nwg + g][0..T) - no global atomics per entry.
extern "C" __global__ void __launch_bounds__(HGS_BLOCK)
hgs_k_preprocess_fwd(View v, Layout L, const float* __restrict__ means3D,
                     const float* __restrict__ shs, const float* __restrict__ colors_precomp,
                     const float* __restrict__ opacities, const float* __restrict__ scales,
                     const float* __restrict__ rotations,
                     const float* __restrict__ cov3D_precomp, int32_t* __restrict__ radii, float* __restrict__ zero_leaf) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_hist[];
  __shared__ uint32_t wtot[HGS_BLOCK / 64];
  zero_counters(L);
  const int b = (int)blockIdx.x / v.nwg, lw = (int)blockIdx.x % v.nwg;
  const Cam cam = v.cam[b];
  for (int t = threadIdx.x; t < v.T; t += HGS_BLOCK) lds_hist[t] = 0u;
  __syncthreads();
  const int gxm = v.grid_x;
  for (int c = 0; c < v.cpw; ++c) {
    const int chunk = lw * v.cpw + c;
    if (chunk >= v.nblk) break;
    const int i = chunk * HGS_BLOCK + threadIdx.x;
    uint32_t tt = 0;
    if (i < v.P) {
      GeomRec rec;
      tt = preprocess_one(v, cam, i, means3D, shs, colors_precomp, opacities, scales, rotations,
                          cov3D_precomp, rec);
      radii[(size_t)b * v.P + i] = rec.radius;
      if (zero_leaf) {         // the caller's screen-space leaf [B][P][3] (hgs_forward_batch_act_leaf): zeros, no fill launch
        float* z = zero_leaf + ((size_t)b * v.P + i) * 3;
        z[0] = 0.0f; z[1] = 0.0f; z[2] = 0.0f;
      }
      store_geom(&L.geom[(size_t)b * v.P + i], rec);      // the record leaves the registers now ...
      if (tt) {
        const int minx = rec.rect_lo & 0xffffu, miny = rec.rect_lo >> 16;
        const int maxx = rec.rect_hi & 0xffffu, maxy = rec.rect_hi >> 16;
        for (int ty = miny; ty < maxy; ++ty)
          for (int tx = minx; tx < maxx; ++tx) atomicAdd(&lds_hist[ty * gxm + tx], 1u);
      }
    }
    const uint32_t off = chunk_prefix(tt, wtot, &L.chunk_sums[(size_t)b * v.nblk + chunk]);
    if (i < v.P && tt) L.geom[(size_t)b * v.P + i].offset = off;      // ... its entry-id prefix follows
  }
  __syncthreads();
  uint32_t* row = L.hist + (size_t)blockIdx.x * v.T;
  for (int t = threadIdx.x; t < v.T; t += HGS_BLOCK) row[t] = lds_hist[t];
}

// Fallback for very large images (T*4 > 64 KB): one chunk per workgroup, one global
// atomic per touched tile.
extern "C" __global__ void __launch_bounds__(HGS_BLOCK)
hgs_k_preprocess_fwd_ga(View v, Layout L, const float* __restrict__ means3D,
                        const float* __restrict__ shs, const float* __restrict__ colors_precomp,
                        const float* __restrict__ opacities, const float* __restrict__ scales,
                        const float* __restrict__ rotations,
                        const float* __restrict__ cov3D_precomp, int32_t* __restrict__ radii, float* __restrict__ zero_leaf) {
  __shared__ uint32_t wtot[HGS_BLOCK / 64];
  zero_counters(L);
  const int b = (int)blockIdx.x / v.nblk, chunk = (int)blockIdx.x % v.nblk;
  const Cam cam = v.cam[b];
  const int i = chunk * HGS_BLOCK + threadIdx.x;
  uint32_t tt = 0;
  if (i < v.P) {
    GeomRec rec;
    tt = preprocess_one(v, cam, i, means3D, shs, colors_precomp, opacities, scales, rotations,
                        cov3D_precomp, rec);
    radii[(size_t)b * v.P + i] = rec.radius;
    if (zero_leaf) {
      float* z = zero_leaf + ((size_t)b * v.P + i) * 3;
      z[0] = 0.0f; z[1] = 0.0f; z[2] = 0.0f;
    }
    store_geom(&L.geom[(size_t)b * v.P + i], rec);
    if (tt) {
      const int minx = rec.rect_lo & 0xffffu, miny = rec.rect_lo >> 16;
      const int maxx = rec.rect_hi & 0xffffu, maxy = rec.rect_hi >> 16;
      uint32_t* cnt = L.tile_count + (size_t)b * v.T;
      for (int ty = miny; ty < maxy; ++ty)
        for (int tx = minx; tx < maxx; ++tx) atomicAdd(&cnt[ty * v.grid_x + tx], 1u);
    }
  }
  const uint32_t off = chunk_prefix(tt, wtot, &L.chunk_sums[blockIdx.x]);
  if (i < v.P && tt) L.geom[(size_t)b * v.P + i].offset = off;
}

// ----------------------------------------------------------------------------- backward
// One thread per Gaussian.  For every view of the batch, in view order: sums the Gaussian's
// tiles_touched gradient rows (contiguous, fixed order => deterministic), then chains through
// conic -> cov2D -> (cov3D, mean), projection, depth, SH and Sigma = R S^2 R^T, and ADDS the
// view's parameter gradients to running sums.  Every output element is written exactly once;
// dL/dmeans2D is written per view.  Instantiated per active SH degree so that the (M,3)
// gradient block accumulates in registers with constant indices.
namespace {

// MODE 0: one thread per Gaussian loops over the views of the call; 1: one view (no loop-carried sums);
// 2: one thread per (Gaussian, view) - a workgroup of B waves owns 64 Gaussians, wave b computes view b's
// gradients exactly like the single-view kernel (the camera stays wave-uniform), the workgroup exchanges
// them through LDS and wave 0 adds them in view order (the same sequence of fp32 additions as the loop of
// mode 0) and writes the outputs.  B times the threads of mode 0, none of
// which carries sums across a loop: the 8-view backward was a latency chain of 1.5 waves per SIMD.
template <int DEG, int MODE>
__device__ __forceinline__ void preprocess_bwd_body(
    const View& v, const Layout& L, const hgs_status* __restrict__ status,
    const float* __restrict__ grad_rows, const float* __restrict__ means3D,
    const float* __restrict__ shs, const float* __restrict__ colors_precomp,
    const float* __restrict__ opacities_raw,
    const float* __restrict__ scales, const float* __restrict__ rotations,
    const float* __restrict__ cov3D_precomp, float* __restrict__ dL_dmeans3D,
    float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dshs, float* __restrict__ dL_dcolors,
    float* __restrict__ dL_dopac, float* __restrict__ dL_dscales, float* __restrict__ dL_drots,
    float* __restrict__ dL_dcov3D, float* __restrict__ pack, int pack_F) {
  // `pack` (hgs_backward_batch_packed): the gradients of Gaussian i go into ONE row of pack_F = 15 + 3 M floats
  //   [means3D 3 | means2D 3, summed over the call's views | sh 3 M | opacity 1 | scales 3 | rotations 4 | radii 1, max over the views]
  // - the layout the view-parallel step all-gathers (view_parallel.py) - instead of into six tensors that a second kernel
  // would have to read back and interleave.  The caller passes dL_dshs / dL_dscales / dL_drots as non-null markers (any
  // value) for the parts it wants; dL_dmeans2D (per view) may still be given.
  constexpr int NC = (DEG + 1) * (DEG + 1);       // active SH coefficients
  constexpr bool SINGLE = MODE != 0;              // this thread handles exactly one view
  constexpr bool VPAR = MODE == 2;
  constexpr int NV = 23 + 3 * NC;                 // values a thread hands over in mode 2 (the last three: means2D x, y - sums - and the radius - a max)
  extern __shared__ float red[];                  // mode 2: [NV][threads of the workgroup]
  int i = blockIdx.x * HGS_BLOCK + threadIdx.x, bview = 0, il = threadIdx.x;
  if (VPAR) {                                     // wave = view (wave-uniform camera), lane = Gaussian
    il = (int)threadIdx.x & 63;
    bview = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    i = blockIdx.x * 64 + il;
  }
  const bool mine = (i < v.P) && (bview < v.B);
  // single view, SH degree >= 1: the chunk's SH blocks come in (and the gradient blocks go out) through LDS as coalesced
  // ranges (stage_sh_chunk: per-thread 16 B pieces at a 12 M byte stride cost 8x the L2 <-> L1 traffic).  Full chunks only.
  const bool staged = MODE == 1 && DEG > 0 && dL_dshs != nullptr && shs != nullptr && hgs_sh_staged(v.M) &&
                      (int)(blockIdx.x + 1) * HGS_BLOCK <= v.P && pack == nullptr;
  if (staged) {
    stage_sh_chunk(shs, v.M, (int)blockIdx.x * HGS_BLOCK, HGS_BLOCK, red);
    __syncthreads();
  }
  if (!VPAR && !mine) return;
  if (!mine) i = 0;                               // mode 2: idle threads stay for the barrier; they read Gaussian 0 of
  if (bview >= v.B) bview = 0;                    // view 0 (the view stays wave-uniform) and write nothing
  const bool ok = status->overflow == 0;

  // ---- view-independent inputs.  One view: loaded once.  Several views: RE-loaded at the top of every
  // view iteration behind a compiler barrier - they come from L1/L2, and keeping them (and what the
  // compiler derives from them) live across the loop next to the running sums cost 176-256 VGPRs
  // (1-2 waves/SIMD) instead of ~100-130.
  float x, y, z;
  Cov3 s;
  RotScale rs;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  float q_inv_norm = 1.0f;
  const bool want_sh = dL_dshs != nullptr && shs != nullptr;
  float sh48[48];
  auto load_inputs = [&]() {
    x = means3D[3 * i + 0]; y = means3D[3 * i + 1]; z = means3D[3 * i + 2];
    if (cov3D_precomp) {
      const float* c = cov3D_precomp + 6 * (size_t)i;
      s.c0 = c[0]; s.c1 = c[1]; s.c2 = c[2]; s.c3 = c[3]; s.c4 = c[4]; s.c5 = c[5];
    } else {
      rs = make_rotscale(scales, rotations, i, v.scale_modifier, v.act);
      s = cov3d_from(rs);
      q = act_rotation(rotations, i, v.act, q_inv_norm);
    }
    if (DEG > 0 && want_sh) {
      if (staged) {
        load_sh_row_lds(red + (int)threadIdx.x * (hgs_sh_row_f4(v.M) * 4), v.M, sh48);
      } else if (sh_block_vectorisable(v.M)) {
        load_sh_block(shs + (size_t)i * v.M * 3, v.M, sh48);
      } else {
        const float* shp = shs + (size_t)i * v.M * 3;
#pragma unroll
        for (int k = 3; k < 48; ++k) sh48[k] = (k < 3 * NC) ? shp[k] : 0.f;   // degree >= 1 terms only
      }
    }
  };
  if (SINGLE) load_inputs();

  // ---- sums over the views
  float a_mean[3] = {0.f, 0.f, 0.f}, a_sc[3] = {0.f, 0.f, 0.f}, a_rot[4] = {0.f, 0.f, 0.f, 0.f};
  float a_cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, a_col[3] = {0.f, 0.f, 0.f}, a_op = 0.f;
  float a_sh[3 * NC];
#pragma unroll
  for (int k = 0; k < 3 * NC; ++k) a_sh[k] = 0.f;
  float a_m2x = 0.f, a_m2y = 0.f, a_rad = 0.f;   // packed output: screen-space gradient summed over the views, largest radius

  const int b_begin = VPAR ? bview : 0;
  const int b_end = SINGLE ? b_begin + 1 : v.B;   // modes 1, 2: exactly one iteration, known at compile time (no loop-carried sums)
  for (int b = b_begin; b < b_end; ++b) {
    if (!SINGLE) {
      asm volatile("" ::: "memory");          // nothing loaded below may be carried over from the last view
      load_inputs();
    }
    const Cam cam = v.cam[b];
    const float* __restrict__ V = cam.viewmatrix;
    const float* __restrict__ PM = cam.projmatrix;
    const GeomRec g = L.geom[(size_t)b * v.P + i];
    float gmx = 0.f, gmy = 0.f;
    if ((g.radius > 0) && ok && mine) {
      float gA = 0.f, gB = 0.f, gC = 0.f, gop = 0.f, gr = 0.f, gg = 0.f, gb = 0.f, gdep = 0.f;
      const int rw = (int)(g.rect_hi & 0xffffu) - (int)(g.rect_lo & 0xffffu);
      const int rh = (int)(g.rect_hi >> 16) - (int)(g.rect_lo >> 16);
      const int tt = rw * rh;
      const float4* rows = reinterpret_cast<const float4*>(grad_rows) +
                           HGS_GROW_F4 * (size_t)(L.chunk_base[(size_t)b * v.nblk + (i >> 8)] + g.offset);
#define HGS_ROWS_UNROLL_MANY 4
      // Several views: the rows of RU tiles are fetched together (independent loads: one latency round per group
      // instead of one per row - a Gaussian touches 3.3 tiles on average), then added in tile order; 8 views:
      // 95.8 -> 81.8 us.  One view: row by row (grouping measured 15.7 -> 16.3 us: one wave per SIMD has nothing to
      // overlap the wider loads with).
      constexpr int RU = SINGLE ? 1 : HGS_ROWS_UNROLL_MANY;      // (view-parallel form: 1 / 4 / 8 measured, 68.9 / 67.7 / 72.9 us)
      for (int k0 = 0; k0 < tt; k0 += RU) {
        float4 r0[RU], r1[RU], r2[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int k = min(k0 + u, tt - 1);
          r0[u] = rows[HGS_GROW_F4 * k + 0]; r1[u] = rows[HGS_GROW_F4 * k + 1]; r2[u] = rows[HGS_GROW_F4 * k + 2];
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          if (k0 + u < tt) {
            gmx += r0[u].x; gmy += r0[u].y; gA += r0[u].z; gB += r0[u].w;
            gC += r1[u].x; gop += r1[u].y; gr += r1[u].z; gg += r1[u].w;
            gb += r2[u].x; gdep += r2[u].y;
          }
        }
      }
      a_op += gop;
      a_col[0] += gr; a_col[1] += gg; a_col[2] += gb;

      Proj2D pj;
      view_point(V, x, y, z, pj.tx0, pj.ty0, pj.tz);
      project_cov(cam, V, s, pj);

      // ---- conic -> cov2D (a, b, c)
      const float a = pj.a, bq_ = pj.b, c = pj.c;
      const float inv2 = 1.0f / (pj.det * pj.det);
      const float dLa = inv2 * (-c * c * gA + bq_ * c * gB - bq_ * bq_ * gC);
      const float dLb = inv2 * (2.0f * bq_ * c * gA - (pj.det + 2.0f * bq_ * bq_) * gB + 2.0f * a * bq_ * gC);
      const float dLc = inv2 * (-bq_ * bq_ * gA + a * bq_ * gB - a * a * gC);

      // ---- cov2D -> packed cov3D
      const float M0[3] = {pj.M00, pj.M01, pj.M02}, M1[3] = {pj.M10, pj.M11, pj.M12};
      float dcov[6];
      dcov[0] = dLa * M0[0] * M0[0] + dLb * M0[0] * M1[0] + dLc * M1[0] * M1[0];
      dcov[3] = dLa * M0[1] * M0[1] + dLb * M0[1] * M1[1] + dLc * M1[1] * M1[1];
      dcov[5] = dLa * M0[2] * M0[2] + dLb * M0[2] * M1[2] + dLc * M1[2] * M1[2];
      dcov[1] = 2.f * dLa * M0[0] * M0[1] + dLb * (M0[0] * M1[1] + M0[1] * M1[0]) + 2.f * dLc * M1[0] * M1[1];
      dcov[2] = 2.f * dLa * M0[0] * M0[2] + dLb * (M0[0] * M1[2] + M0[2] * M1[0]) + 2.f * dLc * M1[0] * M1[2];
      dcov[4] = 2.f * dLa * M0[1] * M0[2] + dLb * (M0[1] * M1[2] + M0[2] * M1[1]) + 2.f * dLc * M1[1] * M1[2];
#pragma unroll
      for (int k = 0; k < 6; ++k) a_cov[k] += dcov[k];

      // ---- cov2D -> M -> J -> t
      const float u[3] = {pj.u0, pj.u1, pj.u2}, w[3] = {pj.w0, pj.w1, pj.w2};
      float dM0[3], dM1[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        dM0[k] = 2.f * dLa * u[k] + dLb * w[k];
        dM1[k] = 2.f * dLc * w[k] + dLb * u[k];
      }
      const float dJ00 = dM0[0] * V[0] + dM0[1] * V[4] + dM0[2] * V[8];
      const float dJ02 = dM0[0] * V[2] + dM0[1] * V[6] + dM0[2] * V[10];
      const float dJ11 = dM1[0] * V[1] + dM1[1] * V[5] + dM1[2] * V[9];
      const float dJ12 = dM1[0] * V[2] + dM1[1] * V[6] + dM1[2] * V[10];
      const float tzi = 1.0f / pj.tz, tz2 = tzi * tzi, tz3 = tz2 * tzi;
      const float fx = cam.focal_x, fy = cam.focal_y;
      float dt[3];
      dt[0] = pj.clx ? 0.f : -fx * tz2 * dJ02;
      dt[1] = pj.cly ? 0.f : -fy * tz2 * dJ12;
      dt[2] = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + 2.f * fx * pj.tx * tz3 * dJ02 +
              2.f * fy * pj.ty * tz3 * dJ12;
      // depth head: depth = t.z
      dt[2] += gdep;
      float dmean[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int bq = 0; bq < 3; ++bq)
        dmean[bq] += V[4 * bq + 0] * dt[0] + V[4 * bq + 1] * dt[1] + V[4 * bq + 2] * dt[2];

      // ---- pixel mean -> NDC -> homogeneous -> mean
      const float dpx = gmx * 0.5f * (float)v.W, dpy = gmy * 0.5f * (float)v.H;
      const float hx = PM[0] * x + PM[4] * y + PM[8] * z + PM[12];
      const float hy = PM[1] * x + PM[5] * y + PM[9] * z + PM[13];
      const float hw = PM[3] * x + PM[7] * y + PM[11] * z + PM[15];
      const float pw = 1.0f / (hw + 0.0000001f);
      const float dhx = dpx * pw, dhy = dpy * pw;
      const float dhw = -(dpx * hx + dpy * hy) * pw * pw;
#pragma unroll
      for (int bq = 0; bq < 3; ++bq)
        dmean[bq] += PM[4 * bq + 0] * dhx + PM[4 * bq + 1] * dhy + PM[4 * bq + 3] * dhw;

      // ---- Sigma -> scale / rotation
      if (!cov3D_precomp) {
        const float G00 = dcov[0], G11 = dcov[3], G22 = dcov[5];
        const float G01 = 0.5f * dcov[1], G02 = 0.5f * dcov[2], G12 = 0.5f * dcov[4];
        const float Lm[3][3] = {{rs.R00 * rs.sx, rs.R01 * rs.sy, rs.R02 * rs.sz},
                                {rs.R10 * rs.sx, rs.R11 * rs.sy, rs.R12 * rs.sz},
                                {rs.R20 * rs.sx, rs.R21 * rs.sy, rs.R22 * rs.sz}};
        const float Gm[3][3] = {{G00, G01, G02}, {G01, G11, G12}, {G02, G12, G22}};
        const float Rm[3][3] = {{rs.R00, rs.R01, rs.R02}, {rs.R10, rs.R11, rs.R12},
                                {rs.R20, rs.R21, rs.R22}};
        const float sv[3] = {rs.sx, rs.sy, rs.sz};
        float D[3][3];   // dL/dR
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          float ds = 0.f;
#pragma unroll
          for (int ii = 0; ii < 3; ++ii) {
            const float dLik = 2.f * (Gm[ii][0] * Lm[0][k] + Gm[ii][1] * Lm[1][k] + Gm[ii][2] * Lm[2][k]);
            ds += dLik * Rm[ii][k];
            D[ii][k] = dLik * sv[k];
          }
          // the fork returns dL/d(mod * scale) here (its computeCov3D backward drops the modifier's factor); the true
          // derivative only on request (HGS_GRAD_SCALE_TRUE_DERIVATIVE) - identical at scale_modifier 1
          a_sc[k] += (v.act & HGS_GRAD_SCALE_TRUE_DERIVATIVE) ? ds * v.scale_modifier : ds;
        }
        const float r = q.x, qx = q.y, qy = q.z, qz = q.w;
        a_rot[0] += 2.f * (-qz * D[0][1] + qy * D[0][2] + qz * D[1][0] - qx * D[1][2] - qy * D[2][0] + qx * D[2][1]);
        a_rot[1] += 2.f * (qy * D[0][1] + qz * D[0][2] + qy * D[1][0] - 2.f * qx * D[1][1] - r * D[1][2] +
                           qz * D[2][0] + r * D[2][1] - 2.f * qx * D[2][2]);
        a_rot[2] += 2.f * (-2.f * qy * D[0][0] + qx * D[0][1] + r * D[0][2] + qx * D[1][0] + qz * D[1][2] -
                           r * D[2][0] + qz * D[2][1] - 2.f * qy * D[2][2]);
        a_rot[3] += 2.f * (-2.f * qz * D[0][0] - r * D[0][1] + qx * D[0][2] + r * D[1][0] - 2.f * qz * D[1][1] +
                           qy * D[1][2] + qx * D[2][0] + qy * D[2][1]);
      }

      // ---- colour -> SH coefficients and the view direction (into the mean)
      if (want_sh) {
        float draw[3];                   // gradient wrt the pre-clamp colour
        draw[0] = (g.clamped & 1u) ? 0.f : gr;
        draw[1] = (g.clamped & 2u) ? 0.f : gg;
        draw[2] = (g.clamped & 4u) ? 0.f : gb;
        const float* cp = cam.campos;
        const float ox = x - cp[0], oy = y - cp[1], oz = z - cp[2];
        const float n = sqrtf(ox * ox + oy * oy + oz * oz);
        const float dx = ox / n, dy = oy / n, dz = oz / n;
        float basis[NC];
        basis[0] = SH_C0;
        float ddir[3] = {0.f, 0.f, 0.f};
        if constexpr (DEG > 0) {
          basis[1] = -SH_C1 * dy; basis[2] = SH_C1 * dz; basis[3] = -SH_C1 * dx;
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            ddir[0] += draw[ch] * (-SH_C1 * sh48[9 + ch]);
            ddir[1] += draw[ch] * (-SH_C1 * sh48[3 + ch]);
            ddir[2] += draw[ch] * (SH_C1 * sh48[6 + ch]);
          }
        }
        if constexpr (DEG > 1) {
          const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
          const float xy = dx * dy, yz = dy * dz, xz = dx * dz;
          basis[4] = SH_C2[0] * xy; basis[5] = SH_C2[1] * yz;
          basis[6] = SH_C2[2] * (2.f * zz - xx - yy);
          basis[7] = SH_C2[3] * xz; basis[8] = SH_C2[4] * (xx - yy);
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            const float s4 = sh48[12 + ch], s5 = sh48[15 + ch], s6 = sh48[18 + ch], s7 = sh48[21 + ch], s8 = sh48[24 + ch];
            ddir[0] += draw[ch] * (SH_C2[0] * dy * s4 + SH_C2[2] * -2.f * dx * s6 + SH_C2[3] * dz * s7 + SH_C2[4] * 2.f * dx * s8);
            ddir[1] += draw[ch] * (SH_C2[0] * dx * s4 + SH_C2[1] * dz * s5 + SH_C2[2] * -2.f * dy * s6 + SH_C2[4] * -2.f * dy * s8);
            ddir[2] += draw[ch] * (SH_C2[1] * dy * s5 + SH_C2[2] * 4.f * dz * s6 + SH_C2[3] * dx * s7);
          }
          if constexpr (DEG > 2) {
            basis[9] = SH_C3[0] * dy * (3.f * xx - yy);
            basis[10] = SH_C3[1] * xy * dz;
            basis[11] = SH_C3[2] * dy * (4.f * zz - xx - yy);
            basis[12] = SH_C3[3] * dz * (2.f * zz - 3.f * xx - 3.f * yy);
            basis[13] = SH_C3[4] * dx * (4.f * zz - xx - yy);
            basis[14] = SH_C3[5] * dz * (xx - yy);
            basis[15] = SH_C3[6] * dx * (xx - 3.f * yy);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
              const float s9 = sh48[27 + ch], s10 = sh48[30 + ch], s11 = sh48[33 + ch], s12 = sh48[36 + ch];
              const float s13 = sh48[39 + ch], s14 = sh48[42 + ch], s15 = sh48[45 + ch];
              ddir[0] += draw[ch] * (SH_C3[0] * s9 * 6.f * xy + SH_C3[1] * s10 * yz + SH_C3[2] * s11 * -2.f * xy +
                                     SH_C3[3] * s12 * -6.f * xz + SH_C3[4] * s13 * (4.f * zz - 3.f * xx - yy) +
                                     SH_C3[5] * s14 * 2.f * xz + SH_C3[6] * s15 * 3.f * (xx - yy));
              ddir[1] += draw[ch] * (SH_C3[0] * s9 * 3.f * (xx - yy) + SH_C3[1] * s10 * xz +
                                     SH_C3[2] * s11 * (4.f * zz - xx - 3.f * yy) + SH_C3[3] * s12 * -6.f * yz +
                                     SH_C3[4] * s13 * -2.f * xy + SH_C3[5] * s14 * -2.f * yz + SH_C3[6] * s15 * -6.f * xy);
              ddir[2] += draw[ch] * (SH_C3[1] * s10 * xy + SH_C3[2] * s11 * 8.f * yz +
                                     SH_C3[3] * s12 * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * s13 * 8.f * xz +
                                     SH_C3[5] * s14 * (xx - yy));
            }
          }
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          a_sh[3 * k + 0] += basis[k] * draw[0];
          a_sh[3 * k + 1] += basis[k] * draw[1];
          a_sh[3 * k + 2] += basis[k] * draw[2];
        }
        // d(normalize)/d(dir_orig)
        const float dot = dx * ddir[0] + dy * ddir[1] + dz * ddir[2];
        dmean[0] += (ddir[0] - dx * dot) / n;
        dmean[1] += (ddir[1] - dy * dot) / n;
        dmean[2] += (ddir[2] - dz * dot) / n;
      }
      a_mean[0] += dmean[0]; a_mean[1] += dmean[1]; a_mean[2] += dmean[2];
    }
    if (dL_dmeans2D && mine) {
      float* o = dL_dmeans2D + ((size_t)b * v.P + i) * 3;
      o[0] = gmx * 0.5f * (float)v.W;
      o[1] = gmy * 0.5f * (float)v.H;
      o[2] = 0.f;
    }
    a_m2x += gmx * 0.5f * (float)v.W;
    a_m2y += gmy * 0.5f * (float)v.H;
    a_rad = fmaxf(a_rad, (float)g.radius);
  }

  if (VPAR) {
    // hand the view's gradients over: value-major, so that the writes and the reads are conflict-free
    float vals[NV];
    {
      int vi = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) vals[vi++] = a_mean[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) vals[vi++] = a_sc[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) vals[vi++] = a_rot[k];
#pragma unroll
      for (int k = 0; k < 6; ++k) vals[vi++] = a_cov[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) vals[vi++] = a_col[k];
      vals[vi++] = a_op;
#pragma unroll
      for (int k = 0; k < 3 * NC; ++k) vals[vi++] = a_sh[k];
      vals[vi++] = a_m2x; vals[vi++] = a_m2y; vals[vi++] = a_rad;
    }
    const int nthr = (int)blockDim.x;
#pragma unroll
    for (int k = 0; k < NV; ++k) red[k * nthr + threadIdx.x] = vals[k];
    __syncthreads();
    if (bview != 0 || !mine) return;
    // view-ordered sums: ((0 + c_0) + c_1) + ... - the additions the loop of mode 0 makes
#pragma unroll
    for (int k = 0; k < NV; ++k) vals[k] = 0.f;
#pragma unroll 1
    for (int bb = 0; bb < v.B; ++bb) {
      const float* col = red + bb * 64 + il;
#pragma unroll
      for (int k = 0; k < NV - 1; ++k) vals[k] += col[k * nthr];
      vals[NV - 1] = fmaxf(vals[NV - 1], col[(NV - 1) * nthr]);       // (the radius: a max)
    }
    {
      int vi = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) a_mean[k] = vals[vi++];
#pragma unroll
      for (int k = 0; k < 3; ++k) a_sc[k] = vals[vi++];
#pragma unroll
      for (int k = 0; k < 4; ++k) a_rot[k] = vals[vi++];
#pragma unroll
      for (int k = 0; k < 6; ++k) a_cov[k] = vals[vi++];
#pragma unroll
      for (int k = 0; k < 3; ++k) a_col[k] = vals[vi++];
      a_op = vals[vi++];
#pragma unroll
      for (int k = 0; k < 3 * NC; ++k) a_sh[k] = vals[vi++];
      a_m2x = vals[vi++]; a_m2y = vals[vi++]; a_rad = vals[vi++];
    }
  }
  if (pack) {
    // ---- packed output: one row, fused activations' chain rule like below
    float* __restrict__ row = pack + (size_t)i * pack_F;
    row[0] = a_mean[0]; row[1] = a_mean[1]; row[2] = a_mean[2];
    row[3] = a_m2x; row[4] = a_m2y; row[5] = 0.f;
    if (dL_dshs) {
#pragma unroll
      for (int k = 0; k < 3 * NC; ++k)
        if (k < 3 * v.M) row[6 + k] = a_sh[k];
      for (int k = 3 * NC; k < 3 * v.M; ++k) row[6 + k] = 0.f;
    }
    float* __restrict__ tail = row + 6 + 3 * v.M;
    float gop = a_op;
    if (v.act & HGS_ACT_OPACITY_SIGMOID) {
      const float y = act_opacity(opacities_raw[i], v.act);
      gop = gop * ((1.0f - y) * y);
    }
    tail[0] = gop;
    if (dL_dscales) {
      if (v.act & HGS_ACT_SCALE_EXP) {
        float s0, s1, s2;
        act_scale(scales, i, v.act, s0, s1, s2);
        a_sc[0] *= s0; a_sc[1] *= s1; a_sc[2] *= s2;
      }
      tail[1] = a_sc[0]; tail[2] = a_sc[1]; tail[3] = a_sc[2];
    }
    if (dL_drots) {
      if (!SINGLE && !cov3D_precomp) q = act_rotation(rotations, i, v.act, q_inv_norm);
      if (v.act & HGS_ACT_ROTATION_NORMALIZE) {
        const float dot = q.x * a_rot[0] + q.y * a_rot[1] + q.z * a_rot[2] + q.w * a_rot[3];
        a_rot[0] = (a_rot[0] - q.x * dot) * q_inv_norm;
        a_rot[1] = (a_rot[1] - q.y * dot) * q_inv_norm;
        a_rot[2] = (a_rot[2] - q.z * dot) * q_inv_norm;
        a_rot[3] = (a_rot[3] - q.w * dot) * q_inv_norm;
      }
      tail[4] = a_rot[0]; tail[5] = a_rot[1]; tail[6] = a_rot[2]; tail[7] = a_rot[3];
    }
    tail[8] = a_rad;                                 // (an exact integer < 2^24)
    return;
  }

  // ---- one write per output element
  if (dL_dshs && staged) {                       // the thread's row -> LDS, the chunk's rows -> one coalesced block
    float* row = red + (int)threadIdx.x * (hgs_sh_row_f4(v.M) * 4);
#pragma unroll
    for (int k = 0; k < 3 * NC; ++k)
      if (k < 3 * v.M) row[k] = a_sh[k];
    for (int k = 3 * NC; k < 3 * v.M; ++k) row[k] = 0.f;
    __syncthreads();
    unstage_sh_chunk(dL_dshs, v.M, (int)blockIdx.x * HGS_BLOCK, red);
  } else if (dL_dshs) {
    float* out = dL_dshs + (size_t)i * v.M * 3;
    if (sh_block_vectorisable(v.M)) {            // 16 B stores of the (M,3) gradient block
      float4* o4 = reinterpret_cast<float4*>(out);
      const int nq = (v.M * 3) >> 2;
#pragma unroll
      for (int qd = 0; qd < 12; ++qd)
        if (qd < nq) {
          float4 t;
          t.x = (4 * qd + 0 < 3 * NC) ? a_sh[(4 * qd + 0) % (3 * NC)] : 0.f;
          t.y = (4 * qd + 1 < 3 * NC) ? a_sh[(4 * qd + 1) % (3 * NC)] : 0.f;
          t.z = (4 * qd + 2 < 3 * NC) ? a_sh[(4 * qd + 2) % (3 * NC)] : 0.f;
          t.w = (4 * qd + 3 < 3 * NC) ? a_sh[(4 * qd + 3) % (3 * NC)] : 0.f;
          o4[qd] = t;
        }
    } else {
#pragma unroll
      for (int k = 0; k < 3 * NC; ++k)
        if (k < 3 * v.M) out[k] = a_sh[k];
      for (int k = 3 * NC; k < 3 * v.M; ++k) out[k] = 0.f;
    }
  }
  if (dL_dmeans3D) {
    dL_dmeans3D[3 * i + 0] = a_mean[0]; dL_dmeans3D[3 * i + 1] = a_mean[1]; dL_dmeans3D[3 * i + 2] = a_mean[2];
  }
  if (dL_dcolors) {
    dL_dcolors[3 * i + 0] = a_col[0]; dL_dcolors[3 * i + 1] = a_col[1]; dL_dcolors[3 * i + 2] = a_col[2];
  }
  // fused activations: chain rule on the view-summed gradients (sigmoid' = y (1 - y) with y as the
  // forward stored it in geom.op is not needed: recompute from the raw input, like autograd does)
  if (dL_dopac) {
    float g = a_op;
    if (v.act & HGS_ACT_OPACITY_SIGMOID) {
      const float y = act_opacity(opacities_raw[i], v.act);
      g = g * ((1.0f - y) * y);
    }
    dL_dopac[i] = g;
  }
  if (dL_dscales) {
    if (v.act & HGS_ACT_SCALE_EXP) {       // d exp(x) = exp(x) = the activated scale (= rs.s / scale_modifier's input)
      float s0, s1, s2;
      act_scale(scales, i, v.act, s0, s1, s2);
      a_sc[0] *= s0; a_sc[1] *= s1; a_sc[2] *= s2;
    }
    dL_dscales[3 * i + 0] = a_sc[0]; dL_dscales[3 * i + 1] = a_sc[1]; dL_dscales[3 * i + 2] = a_sc[2];
  }
  if (dL_drots) {
    if (!SINGLE && !cov3D_precomp) q = act_rotation(rotations, i, v.act, q_inv_norm);
    if (v.act & HGS_ACT_ROTATION_NORMALIZE) {   // d normalize: (g - q_hat (q_hat . g)) / |q|
      const float dot = q.x * a_rot[0] + q.y * a_rot[1] + q.z * a_rot[2] + q.w * a_rot[3];
      a_rot[0] = (a_rot[0] - q.x * dot) * q_inv_norm;
      a_rot[1] = (a_rot[1] - q.y * dot) * q_inv_norm;
      a_rot[2] = (a_rot[2] - q.z * dot) * q_inv_norm;
      a_rot[3] = (a_rot[3] - q.w * dot) * q_inv_norm;
    }
    reinterpret_cast<float4*>(dL_drots)[i] = make_float4(a_rot[0], a_rot[1], a_rot[2], a_rot[3]);
  }
  if (dL_dcov3D) {
    float* o = dL_dcov3D + 6 * (size_t)i;
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = a_cov[k];
  }
}

}  // namespace

#define HGS_PRE_BWD_KERNEL(DEG, NAME, MODE, THREADS)                                                     \
  extern "C" __global__ void __launch_bounds__(THREADS) NAME(                                       \
      View v, Layout L, const hgs_status* __restrict__ status, const float* __restrict__ grad_rows, \
      const float* __restrict__ means3D, const float* __restrict__ shs,                             \
      const float* __restrict__ colors_precomp, const float* __restrict__ opacities_raw,            \
      const float* __restrict__ scales,                                                             \
      const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,                 \
      float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D, float* __restrict__ dL_dshs, \
      float* __restrict__ dL_dcolors, float* __restrict__ dL_dopac, float* __restrict__ dL_dscales,  \
      float* __restrict__ dL_drots, float* __restrict__ dL_dcov3D, float* __restrict__ pack, int pack_F) { \
    preprocess_bwd_body<DEG, MODE>(v, L, status, grad_rows, means3D, shs, colors_precomp, opacities_raw, scales, \
                             rotations, cov3D_precomp, dL_dmeans3D, dL_dmeans2D, dL_dshs,           \
                             dL_dcolors, dL_dopac, dL_dscales, dL_drots, dL_dcov3D, pack, pack_F);  \
  }
HGS_PRE_BWD_KERNEL(0, hgs_k_preprocess_bwd_d0, 0, HGS_BLOCK)         // thread per Gaussian, loop over the views
HGS_PRE_BWD_KERNEL(1, hgs_k_preprocess_bwd_d1, 0, HGS_BLOCK)
HGS_PRE_BWD_KERNEL(2, hgs_k_preprocess_bwd_d2, 0, HGS_BLOCK)
HGS_PRE_BWD_KERNEL(3, hgs_k_preprocess_bwd_d3, 0, HGS_BLOCK)
HGS_PRE_BWD_KERNEL(0, hgs_k_preprocess_bwd_s0, 1, HGS_BLOCK)         // single-view instantiations
HGS_PRE_BWD_KERNEL(1, hgs_k_preprocess_bwd_s1, 1, HGS_BLOCK)
HGS_PRE_BWD_KERNEL(2, hgs_k_preprocess_bwd_s2, 1, HGS_BLOCK)
HGS_PRE_BWD_KERNEL(3, hgs_k_preprocess_bwd_s3, 1, HGS_BLOCK)
HGS_PRE_BWD_KERNEL(0, hgs_k_preprocess_bwd_p0, 2, 1024)   // thread per (Gaussian, view): up to 16 views (SH degree >= 2: 8, registers / LDS)
HGS_PRE_BWD_KERNEL(1, hgs_k_preprocess_bwd_p1, 2, 1024)
HGS_PRE_BWD_KERNEL(2, hgs_k_preprocess_bwd_p2, 2, 512)
HGS_PRE_BWD_KERNEL(3, hgs_k_preprocess_bwd_p3, 2, 512)

// ------------------------------------------------------------------------- mark visible
extern "C" __global__ void __launch_bounds__(HGS_BLOCK)
hgs_k_mark_visible(const float* __restrict__ V, int P, const float* __restrict__ means3D,
                   uint8_t* __restrict__ present) {
  const int i = blockIdx.x * HGS_BLOCK + threadIdx.x;
  if (i >= P) return;
  const float x = means3D[3 * i + 0], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
  const float tz = V[2] * x + V[6] * y + V[10] * z + V[14];
  present[i] = tz > HGS_NEAR_Z ? 1 : 0;
}
