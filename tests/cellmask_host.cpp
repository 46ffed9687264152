// Host build of csrc/cellmask.h for tests/test_cellmask_cpu.py (test infrastructure, g++ only).
#include "../humangaussian_amd/csrc/cellmask.h"
extern "C" void hgs_cell_mask_host(int n, const float* mx, const float* my, const float* ca, const float* cb,
                                   const float* cc, const float* op, const float* x0, const float* y0, uint32_t* out) {
  for (int i = 0; i < n; ++i) out[i] = hgs_cell_mask(mx[i], my[i], ca[i], cb[i], cc[i], op[i], x0[i], y0[i]);
}
// rect = [minx, miny, maxx, maxy) in tile units, in: upstream's rect, out: the cut one
extern "C" void hgs_alpha_rect_host(int n, const float* mx, const float* my, const float* ca, const float* cb,
                                    const float* cc, const float* op, int32_t* rect) {
  for (int i = 0; i < n; ++i) {
    int a = rect[4 * i], b = rect[4 * i + 1], c = rect[4 * i + 2], d = rect[4 * i + 3];
    hgs_alpha_rect(mx[i], my[i], ca[i], cb[i], cc[i], op[i], a, b, c, d);
    rect[4 * i] = a; rect[4 * i + 1] = b; rect[4 * i + 2] = c; rect[4 * i + 3] = d;
  }
}
