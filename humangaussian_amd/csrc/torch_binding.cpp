// torch_binding.cpp - the `_C` module of the reference's extension, re-done on top of the C ABI.
//
// Upstream binds `rasterize_gaussians`, `rasterize_gaussians_backward` and `mark_visible` with
// pybind11 and wraps them in a Python autograd.Function (diff_gaussian_rasterization/__init__.py
// of the ashawkey fork, imported at
// /root/reference/gaussiansplatting/gaussian_renderer/__init__.py:14).  Here the autograd node
// itself lives in C++ (torch::autograd::Function): at 0.3 ms per training-style step the Python
// autograd.Function machinery (apply, ctypes marshalling, GIL hand-off into the engine's device
// thread for backward) cost as much host time as the GPU needs for the whole step.
//
// This file is PLUMBING: it owns tensors, the current HIP stream, the per-device capacity
// estimates and the one host wait per forward.  All arithmetic is behind include/hgs_rast.h
// (libhgs_rast.so); nothing here launches a kernel itself.  Built by g++ (no device code).
#include <torch/extension.h>
#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hgs_rast.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

constexpr int RING = 8;

// Switch the current HIP device only when it is not already the tensors' device.  (torch's own
// HIPGuard types are keyed on DeviceType::HIP, which the ROCm build masquerades as CUDA.)
struct DeviceSwitch {
  c10::DeviceIndex prev = -1;
  explicit DeviceSwitch(c10::DeviceIndex want) {
    const c10::DeviceIndex cur = c10::hip::current_device();
    if (cur != want) { prev = cur; c10::hip::set_device(want); }
  }
  ~DeviceSwitch() { if (prev >= 0) c10::hip::set_device(prev); }
};

void hip_ok(hipError_t e, const char* what) {
  if (e != hipSuccess)
    throw std::runtime_error(std::string("humangaussian_amd: ") + what + ": " + hipGetErrorString(e));
}

struct Pending {
  hipEvent_t event;
  int slot;
  int64_t cap;
  int32_t hint;
};

// Per-device grow-only estimates (entry capacity, longest tile list) and a small ring of
// pinned, device-mapped status mirrors the scan kernel stores into directly.
struct DevState {
  int64_t capacity = 0;
  int32_t tile_hint = 0;      // longest tile list of the last call (with margin); 0 = unknown
  int64_t max_R = 0;
  int64_t max_tile = 0;
  Tensor status_ring;         // pinned int32 [RING][8]
  hgs_status* ring = nullptr;
  hipEvent_t status_event = nullptr;          // recorded by the library right behind the scan stage
  hipEvent_t pend_events[RING] = {};
  int ring_pos = 0;
  std::deque<Pending> pending;
  int64_t synced_calls = 0;
  int64_t wait_ns = 0;        // host time blocked in the per-forward status wait (diagnostic)
  std::mutex mu;
};

std::mutex g_mu;
std::vector<DevState*> g_states(64, nullptr);
bool g_async = false;
std::vector<void*> g_stage_fwd, g_stage_bwd;

DevState& state_for(int dev) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (dev < 0 || dev >= (int)g_states.size()) throw std::runtime_error("bad device index");
  if (!g_states[dev]) {
    auto* st = new DevState();
    st->status_ring = at::zeros({RING, 8}, at::TensorOptions().dtype(at::kInt)).pin_memory();
    static_assert(sizeof(hgs_status) == 32, "hgs_status is 8 words");
    st->ring = reinterpret_cast<hgs_status*>(st->status_ring.data_ptr<int32_t>());
    hip_ok(hipEventCreateWithFlags(&st->status_event, hipEventDisableTiming), "hipEventCreate");
    for (auto& e : st->pend_events) hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    g_states[dev] = st;
  }
  return *g_states[dev];
}

int64_t round_capacity(int64_t n) { return std::max<int64_t>(1 << 16, (n + 0xFFFF) & ~int64_t(0xFFFF)); }
int64_t al256(size_t n) { return (int64_t)((n + 255) & ~size_t(255)); }

Tensor f32c(const Tensor& t, const c10::Device& dev, const char* name) {
  if (t.device() != dev)
    throw std::runtime_error(std::string("expected ") + name + " on " + dev.str() + ", got " + t.device().str());
  Tensor r = t.scalar_type() == at::kFloat ? t : t.to(at::kFloat);
  return r.is_contiguous() ? r : r.contiguous();
}

const float* fptr(const Tensor& t) { return t.defined() && t.numel() > 0 ? t.data_ptr<float>() : nullptr; }
float* fptr_mut(Tensor& t) { return t.defined() && t.numel() > 0 ? t.data_ptr<float>() : nullptr; }

struct Settings {
  hgs_settings s;
  Tensor bg, vm, pm, cp;    // keep-alive
};

Settings make_settings(const Tensor& bg, const Tensor& vm, const Tensor& pm, const Tensor& cp, int64_t H,
                       int64_t W, double tanfovx, double tanfovy, double scale_modifier, int64_t sh_degree,
                       bool prefiltered, bool debug, const c10::Device& dev) {
  Settings r;
  r.bg = f32c(bg, dev, "bg");
  r.vm = f32c(vm, dev, "viewmatrix");
  r.pm = f32c(pm, dev, "projmatrix");
  r.cp = f32c(cp, dev, "campos");
  if (r.bg.numel() != 3 || r.vm.numel() != 16 || r.pm.numel() != 16 || r.cp.numel() != 3)
    throw std::runtime_error("bg/campos must have 3 elements, viewmatrix/projmatrix 16");
  r.s.image_height = (int32_t)H;
  r.s.image_width = (int32_t)W;
  r.s.tanfovx = (float)tanfovx;
  r.s.tanfovy = (float)tanfovy;
  r.s.bg = r.bg.data_ptr<float>();
  r.s.scale_modifier = (float)scale_modifier;
  r.s.viewmatrix = r.vm.data_ptr<float>();
  r.s.projmatrix = r.pm.data_ptr<float>();
  r.s.sh_degree = (int32_t)sh_degree;
  r.s.campos = r.cp.data_ptr<float>();
  r.s.prefiltered = prefiltered ? 1 : 0;
  r.s.debug = debug ? 1 : 0;
  return r;
}

void observe(DevState& st, const hgs_status& h) {
  st.max_R = std::max<int64_t>(st.max_R, h.num_rendered);
  st.max_tile = std::max<int64_t>(st.max_tile, h.reserved[1]);
}

// Inspect the status of earlier async forwards whose status has landed.
void drain_pending(DevState& st, bool block) {
  while (!st.pending.empty()) {
    Pending p = st.pending.front();
    if (block) {
      hip_ok(hipEventSynchronize(p.event), "hipEventSynchronize");
    } else {
      hipError_t q = hipEventQuery(p.event);
      if (q == hipErrorNotReady) break;
      hip_ok(q, "hipEventQuery");
    }
    st.pending.pop_front();
    const hgs_status h = st.ring[p.slot];
    observe(st, h);
    if (h.overflow) {
      st.capacity = std::max(st.capacity, round_capacity(2 * (int64_t)h.num_rendered));
      st.tile_hint = 0;
      throw std::runtime_error(
          "humangaussian_amd (async mode): an earlier render overflowed its buffers (num_rendered=" +
          std::to_string(h.num_rendered) + ", capacity=" + std::to_string(p.cap) + ", longest tile list=" +
          std::to_string(h.reserved[1]) + ", hint=" + std::to_string(p.hint) +
          "); its outputs and gradients were invalid.  Capacity has been raised; re-run the step "
          "(or disable async mode).");
    }
  }
}

void check_rc(int rc, const char* what) {
  if (rc == HGS_ESHAPE)
    throw std::runtime_error("inconsistent optional inputs (shs/colors_precomp, scales+rotations/cov3D_precomp)");
  if (rc != HGS_OK) throw std::runtime_error(std::string("libhgs_rast: ") + what + " failed with code " + std::to_string(rc));
}

// What the backward call needs, prepared while the GPU runs the forward.
struct BwdPlan : torch::CustomClassHolder {
  Settings settings;
  Tensor work, work2;                 // [geom | img | bin | rows] (work2: bin | rows after a retry)
  char *geom = nullptr, *bin = nullptr, *img = nullptr, *rows = nullptr;
  int64_t cap = 0;
  bool have_status = false;
  hgs_status status{};
  Tensor d_means3D, d_means2D, d_sh, d_cp, d_opac, d_sc, d_ro, d_cv;
  int32_t P = 0, M = 0;
  bool has_sh = false, has_cp = false, has_sr = false, has_cv = false;
};

struct Rasterize : public torch::autograd::Function<Rasterize> {
  static variable_list forward(AutogradContext* ctx, const Tensor& means3D, const Tensor& means2D,
                               const Tensor& sh, const Tensor& colors_precomp, const Tensor& opacities,
                               const Tensor& scales, const Tensor& rotations, const Tensor& cov3D,
                               const Tensor& bg, const Tensor& viewmatrix, const Tensor& projmatrix,
                               const Tensor& campos, int64_t H, int64_t W, double tanfovx, double tanfovy,
                               double scale_modifier, int64_t sh_degree, bool prefiltered, bool debug,
                               bool want_grad) {
    (void)means2D;
    const c10::Device dev = means3D.device();
    if (!dev.is_cuda())
      throw std::runtime_error("humangaussian_amd: tensors must live on a HIP device (torch device type 'cuda'); "
                               "there is no CPU path");
    const int64_t P = means3D.size(0);
    if (P != 0 && (means3D.dim() != 2 || means3D.size(1) != 3))
      throw std::runtime_error("means3D must have dimensions (num_points, 3)");
    DeviceSwitch guard(dev.index());

    auto plan = c10::make_intrusive<BwdPlan>();
    plan->settings = make_settings(bg, viewmatrix, projmatrix, campos, H, W, tanfovx, tanfovy, scale_modifier,
                                   sh_degree, prefiltered, debug, dev);
    const Tensor m3 = f32c(means3D, dev, "means3D");
    const bool has_sh = sh.defined() && sh.numel() > 0, has_cp = colors_precomp.defined() && colors_precomp.numel() > 0;
    const bool has_sc = scales.defined() && scales.numel() > 0, has_ro = rotations.defined() && rotations.numel() > 0;
    const bool has_cv = cov3D.defined() && cov3D.numel() > 0;
    const Tensor sh_ = has_sh ? f32c(sh, dev, "shs") : Tensor();
    const Tensor cp_ = has_cp ? f32c(colors_precomp, dev, "colors_precomp") : Tensor();
    const Tensor sc_ = has_sc ? f32c(scales, dev, "scales") : Tensor();
    const Tensor ro_ = has_ro ? f32c(rotations, dev, "rotations") : Tensor();
    const Tensor cv_ = has_cv ? f32c(cov3D, dev, "cov3D_precomp") : Tensor();
    const Tensor op_ = f32c(opacities, dev, "opacities");
    if (has_sh && (sh_.dim() != 3 || sh_.size(0) != P || sh_.size(2) != 3))
      throw std::runtime_error("shs must have dimensions (num_points, M, 3)");
    const int32_t M = has_sh ? (int32_t)sh_.size(1) : 0;

    const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    Tensor color = at::empty({3, H, W}, fopt), depth = at::empty({1, H, W}, fopt), alpha = at::empty({1, H, W}, fopt);
    Tensor radii = at::empty({P}, fopt.dtype(at::kInt));

    DevState& st = state_for(dev.index());
    std::lock_guard<std::mutex> lk(st.mu);
    hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
    if (!st.pending.empty()) drain_pending(st, false);
    const bool go_async = g_async && want_grad && P > 0 && st.synced_calls >= 2;
    int64_t cap;
    int32_t hint;
    if (go_async) {
      cap = std::max(st.capacity, round_capacity(2 * st.max_R));
      hint = (int32_t)std::max<int64_t>(1024, 2 * st.max_tile + 64);
    } else {
      cap = P > 0 ? std::max(st.capacity, round_capacity(4 * P)) : 0;
      hint = st.tile_hint;
    }
    // one allocation for the four opaque regions [geom | img | bin | backward rows] (the fork keeps
    // three such byte tensors for its backward); a capacity retry re-allocates only the last two
    const int64_t g_sz = al256(hgs_geom_bytes((int32_t)P, (int32_t)H, (int32_t)W)), i_sz = al256(hgs_img_bytes((int32_t)H, (int32_t)W));
    int64_t b_sz = al256(hgs_bin_bytes(cap)), s_sz = want_grad ? al256(hgs_bwd_scratch_bytes(cap)) : 0;
    const auto bopt = at::TensorOptions().dtype(at::kByte).device(dev);
    plan->work = at::empty({g_sz + i_sz + b_sz + s_sz}, bopt);
    char* base = static_cast<char*>(plan->work.data_ptr());
    plan->geom = base;
    plan->img = base + g_sz;
    plan->bin = base + g_sz + i_sz;
    plan->rows = base + g_sz + i_sz + b_sz;

    bool have_status = false;
    hgs_status h{};
    bool grads_ready = false;
    int attempt = 0;
    for (; attempt < 4; ++attempt) {
      const int slot = st.ring_pos;
      st.ring_pos = (st.ring_pos + 1) % RING;
      const int rc = hgs_forward(&plan->settings.s, (int32_t)P, M, fptr(m3), fptr(sh_), fptr(cp_), fptr(op_), fptr(sc_),
                                 fptr(ro_), fptr(cv_), fptr_mut(color), fptr_mut(depth), fptr_mut(alpha),
                                 P > 0 ? radii.data_ptr<int32_t>() : nullptr, plan->geom, plan->bin, cap, plan->img,
                                 want_grad ? 1 : 0, hint, &st.ring[slot], /*mapped=*/1,
                                 go_async ? nullptr : st.status_event,
                                 g_stage_fwd.empty() ? nullptr : g_stage_fwd.data(), stream);
      check_rc(rc, "hgs_forward");
      // `debug=True` is upstream's switch for surfacing device errors at the call that caused
      // them (std::runtime_error): synchronise and report
      if (debug) hip_ok(hipStreamSynchronize(stream), "device error in the rasterizer forward (debug=True)");
      // host work that does not depend on the result runs HERE, while the GPU is busy
      if (want_grad && !grads_ready) {
        plan->d_means3D = at::empty({P, 3}, fopt);
        plan->d_means2D = at::empty({P, 3}, fopt);
        plan->d_opac = at::empty(opacities.sizes(), fopt);
        if (has_sh) plan->d_sh = at::empty({P, M, 3}, fopt);
        if (has_cp) plan->d_cp = at::empty({P, 3}, fopt);
        if (has_sc) { plan->d_sc = at::empty({P, 3}, fopt); plan->d_ro = at::empty({P, 4}, fopt); }
        if (has_cv) plan->d_cv = at::empty({P, 6}, fopt);
        grads_ready = true;
      }
      if (go_async) {
        Pending p{st.pend_events[slot], slot, cap, hint};
        hip_ok(hipEventRecord(p.event, stream), "hipEventRecord");
        st.pending.push_back(p);
        if ((int)st.pending.size() >= RING - 1) drain_pending(st, true);   // never let the ring wrap
        break;
      }
      // One host wait per forward, like upstream's blocking read of num_rendered - but only for
      // the status (stored by the scan kernel): fill, sort and blend are already enqueued and keep
      // the GPU busy while the host goes on to autograd and the backward launch.
      const auto tw = std::chrono::steady_clock::now();
      hip_ok(hipEventSynchronize(st.status_event), "hipEventSynchronize");
      st.wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tw).count();
      h = st.ring[slot];
      have_status = true;
      if (!h.overflow) break;
      if (h.overflow & 1u) {                    // R exceeded the capacity: grow, re-run
        cap = round_capacity((int64_t)(h.num_rendered * 1.25) + 1);
        b_sz = al256(hgs_bin_bytes(cap));
        s_sz = want_grad ? al256(hgs_bwd_scratch_bytes(cap)) : 0;
        plan->work2 = at::empty({b_sz + s_sz}, bopt);
        plan->bin = static_cast<char*>(plan->work2.data_ptr());
        plan->rows = plan->bin + b_sz;
      }
      if (h.overflow & 2u) hint = 0;            // a tile list outgrew the hint
    }
    if (attempt == 4) throw std::runtime_error("libhgs_rast: entry capacity did not converge");
    if (have_status) {
      observe(st, h);
      st.synced_calls += 1;
      st.capacity = std::max(st.capacity, cap);
      st.tile_hint = (int32_t)std::max<int64_t>(1024, (int64_t)(h.reserved[1] * 1.5) + 64);
    }
    ctx->set_materialize_grads(false);
    if (want_grad) {
      plan->cap = cap;
      plan->have_status = have_status;
      plan->status = h;
      plan->P = (int32_t)P;
      plan->M = M;
      plan->has_sh = has_sh; plan->has_cp = has_cp; plan->has_sr = has_sc; plan->has_cv = has_cv;
      ctx->saved_data["plan"] = c10::IValue::make_capsule(plan);
      // inputs and outputs go through save_for_backward (version checks, no reference cycle
      // through the outputs); the opaque work buffers ride in the plan
      variable_list saved = {m3, op_, radii, color, depth, alpha};
      for (const Tensor* t : {&sh_, &cp_, &sc_, &ro_, &cv_})
        if (t->defined()) saved.push_back(*t);
      ctx->save_for_backward(saved);
    }
    ctx->mark_non_differentiable({radii});
    return {color, radii, depth, alpha};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto holder = ctx->saved_data["plan"].toCapsule();
    BwdPlan* plan = static_cast<BwdPlan*>(holder.get());
    const auto saved = ctx->get_saved_variables();
    const Tensor &m3 = saved[0], &op_ = saved[1], &radii = saved[2], &color = saved[3], &depth = saved[4], &alpha = saved[5];
    size_t k = 6;
    const Tensor sh_ = plan->has_sh ? saved[k++] : Tensor();
    const Tensor cp_ = plan->has_cp ? saved[k++] : Tensor();
    const Tensor sc_ = plan->has_sr ? saved[k++] : Tensor();
    const Tensor ro_ = plan->has_sr ? saved[k++] : Tensor();
    const Tensor cv_ = plan->has_cv ? saved[k++] : Tensor();
    const c10::Device dev = m3.device();
    DeviceSwitch guard(dev.index());
    const Tensor gc = grads[0].defined() ? f32c(grads[0], dev, "grad_color") : at::zeros_like(color);
    const Tensor gd = grads[2].defined() ? f32c(grads[2], dev, "grad_depth") : at::zeros_like(depth);
    const Tensor ga = grads[3].defined() ? f32c(grads[3], dev, "grad_alpha") : at::zeros_like(alpha);
    hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
    const int rc = hgs_backward(
        &plan->settings.s, plan->P, plan->M, fptr(m3), fptr(sh_), fptr(cp_), fptr(op_), fptr(sc_), fptr(ro_), fptr(cv_),
        plan->P > 0 ? radii.data_ptr<int32_t>() : nullptr, fptr(color), fptr(depth), fptr(alpha), fptr(gc), fptr(gd),
        fptr(ga), plan->geom, plan->bin, plan->img, plan->have_status ? &plan->status : nullptr, plan->cap, plan->rows,
        fptr_mut(plan->d_means3D), fptr_mut(plan->d_means2D), fptr_mut(plan->d_sh), fptr_mut(plan->d_cp),
        fptr_mut(plan->d_opac), fptr_mut(plan->d_sc), fptr_mut(plan->d_ro), fptr_mut(plan->d_cv),
        g_stage_bwd.empty() ? nullptr : g_stage_bwd.data(), stream);
    check_rc(rc, "hgs_backward");
    if (plan->settings.s.debug) hip_ok(hipStreamSynchronize(stream), "device error in the rasterizer backward (debug=True)");
    variable_list out(21);
    out[0] = plan->d_means3D; out[1] = plan->d_means2D; out[2] = plan->d_sh; out[3] = plan->d_cp;
    out[4] = plan->d_opac; out[5] = plan->d_sc; out[6] = plan->d_ro; out[7] = plan->d_cv;
    ctx->saved_data.erase("plan");
    return out;
  }
};

// None optionals travel through autograd as one shared empty tensor (upstream does the same:
// "None inputs become empty tensors"); undefined tensors are not valid apply() inputs.
Tensor opt(const c10::optional<Tensor>& t) {
  static const Tensor empty = at::empty({0}, at::TensorOptions().dtype(at::kFloat));
  return t.has_value() ? *t : empty;
}

std::vector<Tensor> rasterize(const Tensor& means3D, const Tensor& means2D, const c10::optional<Tensor>& sh,
                              const c10::optional<Tensor>& colors_precomp, const Tensor& opacities,
                              const c10::optional<Tensor>& scales, const c10::optional<Tensor>& rotations,
                              const c10::optional<Tensor>& cov3D, const Tensor& bg, const Tensor& viewmatrix,
                              const Tensor& projmatrix, const Tensor& campos, int64_t H, int64_t W, double tanfovx,
                              double tanfovy, double scale_modifier, int64_t sh_degree, bool prefiltered, bool debug,
                              bool want_grad) {
  return Rasterize::apply(means3D, means2D, opt(sh), opt(colors_precomp), opacities, opt(scales), opt(rotations),
                          opt(cov3D), bg, viewmatrix, projmatrix, campos, H, W, tanfovx, tanfovy, scale_modifier,
                          sh_degree, prefiltered, debug, want_grad);
}

Tensor mark_visible(const Tensor& positions, const Tensor& bg, const Tensor& viewmatrix, const Tensor& projmatrix,
                    const Tensor& campos, int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier,
                    int64_t sh_degree, bool prefiltered, bool debug) {
  at::NoGradGuard ng;
  const c10::Device dev = positions.device();
  if (!dev.is_cuda()) throw std::runtime_error("humangaussian_amd: tensors must live on a HIP device");
  DeviceSwitch guard(dev.index());
  Settings s = make_settings(bg, viewmatrix, projmatrix, campos, H, W, tanfovx, tanfovy, scale_modifier, sh_degree,
                             prefiltered, debug, dev);
  const Tensor pos = f32c(positions, dev, "positions");
  const int64_t P = pos.size(0);
  Tensor present = at::zeros({P}, at::TensorOptions().dtype(at::kByte).device(dev));
  if (P > 0) {
    const int rc = hgs_mark_visible(&s.s, (int32_t)P, pos.data_ptr<float>(), present.data_ptr<uint8_t>(),
                                    c10::hip::getCurrentHIPStream(dev.index()).stream());
    check_rc(rc, "hgs_mark_visible");
  }
  return present.to(at::kBool);
}

// replaces simple_knn._C.distCUDA2: mean squared distance to the 3 nearest neighbours
Tensor knn_mean_dist2(const Tensor& points) {
  at::NoGradGuard ng;
  const c10::Device dev = points.device();
  if (!dev.is_cuda()) throw std::runtime_error("humangaussian_amd: tensors must live on a HIP device");
  if (points.dim() != 2 || points.size(1) != 3) throw std::runtime_error("points must have dimensions (num_points, 3)");
  DeviceSwitch guard(dev.index());
  const Tensor pts = f32c(points, dev, "points");
  const int64_t P = pts.size(0);
  Tensor out = at::empty({P}, at::TensorOptions().dtype(at::kFloat).device(dev));
  if (P > 0) {
    const int rc = hgs_knn_mean_dist2((int32_t)P, pts.data_ptr<float>(), out.data_ptr<float>(),
                                      c10::hip::getCurrentHIPStream(dev.index()).stream());
    check_rc(rc, "hgs_knn_mean_dist2");
  }
  return out;
}

// one rank's (P, 15 + 3M) pack for the view-parallel all-gather
Tensor pack_view_contribution(const Tensor& g_means3D, const Tensor& g_means2D, const Tensor& g_sh,
                              const Tensor& g_opac, const Tensor& g_scales, const Tensor& g_rot,
                              const Tensor& radii) {
  at::NoGradGuard ng;
  const c10::Device dev = g_means3D.device();
  if (!dev.is_cuda()) throw std::runtime_error("humangaussian_amd: tensors must live on a HIP device");
  DeviceSwitch guard(dev.index());
  const int64_t P = g_means3D.size(0);
  const Tensor a = f32c(g_means3D, dev, "means3D grad"), b = f32c(g_means2D, dev, "means2D grad");
  const Tensor c = f32c(g_sh, dev, "sh grad"), d = f32c(g_opac, dev, "opacity grad");
  const Tensor e = f32c(g_scales, dev, "scales grad"), f = f32c(g_rot, dev, "rotations grad");
  if (radii.device() != dev || radii.scalar_type() != at::kInt) throw std::runtime_error("radii must be int32 on the same device");
  const Tensor r = radii.contiguous();
  const int64_t M = P > 0 ? c.numel() / (3 * P) : 0;
  if (a.numel() != 3 * P || b.numel() != 3 * P || c.numel() != 3 * M * P || d.numel() != P || e.numel() != 3 * P ||
      f.numel() != 4 * P || r.numel() != P)
    throw std::runtime_error("pack_view_contribution: inconsistent shapes");
  Tensor out = at::empty({P, 15 + 3 * M}, a.options());
  const int rc = hgs_pack_view_contribution((int32_t)P, (int32_t)M, fptr(a), fptr(b), fptr(c), fptr(d), fptr(e), fptr(f),
                                            P > 0 ? r.data_ptr<int32_t>() : nullptr, fptr_mut(out),
                                            c10::hip::getCurrentHIPStream(dev.index()).stream());
  check_rc(rc, "hgs_pack_view_contribution");
  return out;
}

// view-parallel reduction of the all-gathered packs: (world, P, F) -> (P, F)
Tensor reduce_view_packs(const Tensor& gathered) {
  at::NoGradGuard ng;
  const c10::Device dev = gathered.device();
  if (!dev.is_cuda()) throw std::runtime_error("humangaussian_amd: tensors must live on a HIP device");
  if (gathered.dim() != 3) throw std::runtime_error("gathered packs must have dimensions (world, num_points, F)");
  DeviceSwitch guard(dev.index());
  const Tensor g = f32c(gathered, dev, "gathered");
  Tensor out = at::empty({g.size(1), g.size(2)}, g.options());
  const int rc = hgs_reduce_view_packs((int32_t)g.size(0), g.size(1), (int32_t)g.size(2), g.data_ptr<float>(),
                                       out.data_ptr<float>(), c10::hip::getCurrentHIPStream(dev.index()).stream());
  check_rc(rc, "hgs_reduce_view_packs");
  return out;
}

void set_stage_events(const c10::optional<std::vector<int64_t>>& fwd, const c10::optional<std::vector<int64_t>>& bwd) {
  g_stage_fwd.clear();
  g_stage_bwd.clear();
  if (fwd) for (int64_t hnd : *fwd) g_stage_fwd.push_back(reinterpret_cast<void*>(hnd));
  if (bwd) for (int64_t hnd : *bwd) g_stage_bwd.push_back(reinterpret_cast<void*>(hnd));
  if (!g_stage_fwd.empty() && g_stage_fwd.size() != HGS_FWD_STAGES) throw std::runtime_error("fwd needs HGS_FWD_STAGES events");
  if (!g_stage_bwd.empty() && g_stage_bwd.size() != HGS_BWD_STAGES) throw std::runtime_error("bwd needs HGS_BWD_STAGES events");
}

py::dict device_state(int64_t dev) {
  DevState& st = state_for((int)dev);
  std::lock_guard<std::mutex> lk(st.mu);
  py::dict d;
  d["capacity"] = st.capacity;
  d["tile_hint"] = st.tile_hint;
  d["max_R"] = st.max_R;
  d["max_tile"] = st.max_tile;
  d["pending"] = (int64_t)st.pending.size();
  d["synced_calls"] = st.synced_calls;
  d["wait_ns"] = st.wait_ns;
  return d;
}

void drain(int64_t dev, bool block) {
  DevState& st = state_for((int)dev);
  std::lock_guard<std::mutex> lk(st.mu);
  drain_pending(st, block);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "torch binding of libhgs_rast.so (include/hgs_rast.h): autograd node, capacity logic, the one host wait";
  m.def("rasterize", &rasterize, py::call_guard<py::gil_scoped_release>());
  m.def("mark_visible", &mark_visible, py::call_guard<py::gil_scoped_release>());
  m.def("knn_mean_dist2", &knn_mean_dist2, py::call_guard<py::gil_scoped_release>());
  m.def("reduce_view_packs", &reduce_view_packs, py::call_guard<py::gil_scoped_release>());
  m.def("pack_view_contribution", &pack_view_contribution, py::call_guard<py::gil_scoped_release>());
  m.def("set_async", [](bool on) { g_async = on; });
  m.def("get_async", []() { return g_async; });
  m.def("set_stage_events", &set_stage_events);
  m.def("device_state", &device_state);
  m.def("drain_pending", &drain, py::call_guard<py::gil_scoped_release>());
  m.def("abi_version", []() { return hgs_abi_version(); });
}
