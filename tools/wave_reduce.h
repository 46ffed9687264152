// wave_reduce.h - reduce-scatter of 10 per-lane values across a wave64 (gfx950).
//
// Sums each of 10 values over the 64 lanes in 28 VALU instructions instead of 60 (6 DPP adds
// per value): halve the lanes and double the values per register first,
//   xor 32:  v_permlane32_swap  (one swap + one add folds TWO values: 5 + 5)
//   xor 16:  v_permlane16_swap  (one swap + one add folds two registers:  3 + 3)
//   xor 8,4,2,1 inside rows of 16 lanes with row DPP adds (3 registers x 4)
// Result: register r, row w (lanes 16w..16w+15, every lane of the row) holds the full sum of
// value HGS_RED_SLOT(r, w).  The summation tree is fixed => bitwise reproducible.
#pragma once
#include <hip/hip_runtime.h>

namespace hgsred {

__device__ __forceinline__ void swap32(float& a, float& b) {
  // after: a = [a.lo32 | b.lo32], b = [a.hi32 | b.hi32]   (lo32 = lanes 0..31)
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
  a = __int_as_float(r[0]);
  b = __int_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& a, float& b) {
  // rows of 16 lanes: after: a = [a.r0, b.r0, a.r2, b.r2], b = [a.r1, b.r1, a.r3, b.r3]
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(a), __float_as_int(b), false, false);
  a = __int_as_float(r[0]);
  b = __int_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of each row, result in every lane of the row
__device__ __forceinline__ float row_allsum(float v) {
  v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]  (xor 1)
  v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]  (xor 2)
  v = dpp_add<0x141>(v);   // row_half_mirror      (i <-> 7-i inside 8 lanes)
  v = dpp_add<0x140>(v);   // row_mirror           (i <-> 15-i)
  return v;
}

// in: x[0..9] per-lane values.  out: o[0..2]; o[r] row w holds the sum of value slot(r,w):
//   o0: rows {0,1,2,3} -> values {0, 2, 1, 3}
//   o1: rows {0,1,2,3} -> values {4, 6, 5, 7}
//   o2: rows {0,1,2,3} -> values {8, -, 9, -}
__device__ __forceinline__ void reduce10(const float (&x)[10], float (&o)[3]) {
  float a0 = x[0], a1 = x[1], a2 = x[2], a3 = x[3], a4 = x[4];
  float a5 = x[5], a6 = x[6], a7 = x[7], a8 = x[8], a9 = x[9];
  swap32(a0, a1); const float p01 = a0 + a1;      // lo: value0, hi: value1   (32 partials each)
  swap32(a2, a3); const float p23 = a2 + a3;
  swap32(a4, a5); const float p45 = a4 + a5;
  swap32(a6, a7); const float p67 = a6 + a7;
  swap32(a8, a9); const float p89 = a8 + a9;
  float b0 = p01, b1 = p23, b2 = p45, b3 = p67, b4 = p89, b5 = 0.0f;
  swap16(b0, b1); const float q0 = b0 + b1;       // rows: v0, v2, v1, v3
  swap16(b2, b3); const float q1 = b2 + b3;       // rows: v4, v6, v5, v7
  swap16(b4, b5); const float q2 = b4 + b5;       // rows: v8, 0, v9, 0
  o[0] = row_allsum(q0);
  o[1] = row_allsum(q1);
  o[2] = row_allsum(q2);
}
// value index held by (register r, row w); -1 = unused
__device__ __forceinline__ int slot_of(int r, int w) {
  const int base = r * 4;
  const int idx = (w == 0) ? 0 : (w == 1) ? 2 : (w == 2) ? 1 : 3;
  const int v = base + idx;
  return (r == 2 && (w & 1)) ? -1 : v;
}

}  // namespace hgsred
