// Host build of csrc/cellmask.h for tests/test_cellmask_cpu.py (test infrastructure, g++ only).
#include "../humangaussian_amd/csrc/cellmask.h"
extern "C" void hgs_cell_mask_host(int n, const float* mx, const float* my, const float* ca, const float* cb,
                                   const float* cc, const float* op, const float* x0, const float* y0, uint32_t* out) {
  for (int i = 0; i < n; ++i) out[i] = hgs_cell_mask(mx[i], my[i], ca[i], cb[i], cc[i], op[i], x0[i], y0[i]);
}
extern "C" void hgs_tile_hit_host(int n, const float* mx, const float* my, const float* ca, const float* cb,
                                  const float* cc, const float* op, const float* x0, const float* y0, uint32_t* out) {
  for (int i = 0; i < n; ++i) out[i] = hgs_tile_hit(mx[i], my[i], ca[i], cb[i], cc[i], op[i], x0[i], y0[i]) ? 1u : 0u;
}
