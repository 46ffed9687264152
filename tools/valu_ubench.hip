// Micro-benchmark: cycles per wave64 instruction for the VALU ops the blend kernels use.
// hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o tools/valu_ubench && ./tools/valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

#define REP 256
template <int MODE>
__global__ void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f;
  float a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
  float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const float c = seed * 0.999f, d = seed * 1e-3f;
  const float2v cc = {c, c}, dd = {d, d};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (MODE == 0) {  // v_fma_f32, 8 independent chains
        a0 = __builtin_fmaf(a0, c, d); a1 = __builtin_fmaf(a1, c, d); a2 = __builtin_fmaf(a2, c, d); a3 = __builtin_fmaf(a3, c, d);
        a4 = __builtin_fmaf(a4, c, d); a5 = __builtin_fmaf(a5, c, d); a6 = __builtin_fmaf(a6, c, d); a7 = __builtin_fmaf(a7, c, d);
      } else if (MODE == 1) {  // v_pk_fma_f32, 4 independent chains (8 per REP/8 -> count 4 pk per 8)
        p0 = __builtin_elementwise_fma(p0, cc, dd); p1 = __builtin_elementwise_fma(p1, cc, dd);
        p2 = __builtin_elementwise_fma(p2, cc, dd); p3 = __builtin_elementwise_fma(p3, cc, dd);
        p0 = __builtin_elementwise_fma(p0, cc, dd); p1 = __builtin_elementwise_fma(p1, cc, dd);
        p2 = __builtin_elementwise_fma(p2, cc, dd); p3 = __builtin_elementwise_fma(p3, cc, dd);
      } else if (MODE == 2) {  // v_exp_f32
        a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
        a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
      } else if (MODE == 3) {  // v_mul_f32
        a0 *= c; a1 *= c; a2 *= c; a3 *= c; a4 *= c; a5 *= c; a6 *= c; a7 *= c;
      } else if (MODE == 4) {  // v_cndmask (select on compare)
        a0 = a0 > d ? a1 : a0; a1 = a1 > d ? a2 : a1; a2 = a2 > d ? a3 : a2; a3 = a3 > d ? a4 : a3;
        a4 = a4 > d ? a5 : a4; a5 = a5 > d ? a6 : a5; a6 = a6 > d ? a7 : a6; a7 = a7 > d ? a0 : a7;
      } else if (MODE == 5) {  // DPP wave_shr mov
        a0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a0), 0x138, 0xf, 0xf, false));
        a1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a1), 0x138, 0xf, 0xf, false));
        a2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a2), 0x138, 0xf, 0xf, false));
        a3 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a3), 0x138, 0xf, 0xf, false));
        a4 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a4), 0x138, 0xf, 0xf, false));
        a5 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a5), 0x138, 0xf, 0xf, false));
        a6 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a6), 0x138, 0xf, 0xf, false));
        a7 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a7), 0x138, 0xf, 0xf, false));
      } else if (MODE == 6) {  // v_rcp_f32
        a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
        a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
      } else if (MODE == 7) {  // v_pk_mul_f32
        p0 *= cc; p1 *= cc; p2 *= cc; p3 *= cc; p0 *= cc; p1 *= cc; p2 *= cc; p3 *= cc;
      } else if (MODE == 8) {  // dependent v_fma chain (latency)
        a0 = __builtin_fmaf(a0, c, d); a0 = __builtin_fmaf(a0, c, d); a0 = __builtin_fmaf(a0, c, d); a0 = __builtin_fmaf(a0, c, d);
        a0 = __builtin_fmaf(a0, c, d); a0 = __builtin_fmaf(a0, c, d); a0 = __builtin_fmaf(a0, c, d); a0 = __builtin_fmaf(a0, c, d);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
double run(int waves_per_simd, float* out) {
  const int blocks = 256 * waves_per_simd;    // 256 threads = 4 waves = 1 per SIMD of a CU
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = (double)iters * REP;       // "instruction slots" as written above
  const double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * waves_per_simd);
  return ns_per_instr_per_simd;
}

int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 4);
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32 (per pk instr)", "v_exp_f32", "v_mul_f32", "v_cmp+v_cndmask (per pair)",
                         "v_mov_dpp wave_shr", "v_rcp_f32", "v_pk_mul_f32 (per pk instr)", "dependent v_fma chain"};
  for (int wps : {1, 2, 4, 8}) {
    double r[9];
    r[0] = run<0>(wps, out); r[1] = run<1>(wps, out); r[2] = run<2>(wps, out); r[3] = run<3>(wps, out);
    r[4] = run<4>(wps, out); r[5] = run<5>(wps, out); r[6] = run<6>(wps, out); r[7] = run<7>(wps, out); r[8] = run<8>(wps, out);
    printf("waves/SIMD=%d  (ns per wave-instruction per SIMD; x clock GHz = cycles)\n", wps);
    for (int i = 0; i < 9; ++i) printf("   %-32s %7.3f ns  (%.2f cyc @2.4GHz, %.2f @2.0)\n", names[i], r[i], r[i] * 2.4, r[i] * 2.0);
  }
  return 0;
}
