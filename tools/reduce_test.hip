// hipcc --offload-arch=gfx950 -O3 tools/reduce_test.hip -o tools/reduce_test && ./tools/reduce_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "wave_reduce.h"
__global__ void k(const float* in, float* out) {
  const int lane = threadIdx.x;
  float x[10], o[3];
  for (int i = 0; i < 10; ++i) x[i] = in[i * 64 + lane];
  hgsred::reduce10(x, o);
  for (int r = 0; r < 3; ++r) out[r * 64 + lane] = o[r];
}
int main() {
  float h[640], *d, *dout, ho[192];
  for (int i = 0; i < 10; ++i) for (int l = 0; l < 64; ++l) h[i * 64 + l] = (float)((i + 1) * 1000 + l * (i + 3)) * 0.001f;
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&dout, sizeof(ho));
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
  (void)hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 0; r < 3; ++r) for (int w = 0; w < 4; ++w) {
    const int base = r * 4; const int idx = (w == 0) ? 0 : (w == 1) ? 2 : (w == 2) ? 1 : 3;
    const int v = (r == 2 && (w & 1)) ? -1 : base + idx;
    double ref = 0; if (v >= 0) for (int l = 0; l < 64; ++l) ref += h[v * 64 + l];
    for (int l = 16 * w; l < 16 * w + 16; ++l) {
      const float got = ho[r * 64 + l];
      if (std::fabs(got - ref) > 1e-3 * std::fabs(ref) + 1e-4) { if (bad < 8) printf("MISMATCH r%d row%d lane%d value%d got %f ref %f\n", r, w, l, v, got, ref); ++bad; }
    }
  }
  printf(bad ? "reduce10 FAILED (%d)\n" : "reduce10 OK\n", bad);
  return bad != 0;
}
