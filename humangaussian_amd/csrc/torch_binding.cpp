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
#include <atomic>
#include <chrono>
#include <thread>
#include <map>
#include <tuple>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hgs_rast.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

constexpr int RING = 64;      // status slots (pinned, 32 B each): a slot stays untouched for the next RING - 1 forwards

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

// Workload estimate for one (views, height, width) shape on one device: entry capacity of the bin
// buffer and the longest tile list (sort-class / segment hint).  Both follow the workload with a
// DECAYING maximum, so alternating cameras (wide / head zoom, train / eval resolution) neither
// trip the device-side overflow check every other call nor pin the buffers at a one-off peak.
struct Estimate {
  int64_t capacity = 0;       // entries; 0 = unknown
  int64_t tile_hint = 0;      // 0 = unknown
};

struct DevState {
  std::map<std::tuple<int64_t, int64_t, int64_t>, Estimate> est;   // (B, H, W) -> estimate
  int64_t max_R = 0;          // of the last call
  int64_t max_tile = 0;       // of the last call
  int64_t last_capacity = 0, last_hint = 0;
  Tensor status_ring;         // pinned int32 [RING][8]
  hgs_status* ring = nullptr;
  hipEvent_t status_event = nullptr;          // recorded by the library right behind the tiles stage
  int ring_pos = 0;
  int64_t ring_issued = 0;    // slots handed out so far
  int64_t last_scratch = 0, last_pairs = 0;   // of the last backward (diagnostic)
  int64_t calls = 0;
  int64_t retries = 0;        // forwards that had to be re-run (capacity or hint exceeded)
  int64_t wait_ns = 0;        // host time blocked in the per-forward status wait (diagnostic)
  // host time per phase (diagnostic, bench.py): forward: checks + allocations | hgs_forward (launches) | autograd state
  // in the shadow of the wait | the wait | after it; backward: before | hgs_backward (launches) | after
  std::atomic<int64_t> host_ns[8] = {};
  std::mutex mu;
};

std::mutex g_mu;
std::vector<DevState*> g_states(64, nullptr);
std::vector<void*> g_stage_fwd, g_stage_bwd;

DevState& state_for(int dev) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (dev < 0 || dev >= (int)g_states.size()) throw std::runtime_error("bad device index");
  if (!g_states[dev]) {
    DeviceSwitch guard((c10::DeviceIndex)dev);          // the event belongs to `dev`, whatever device is current
    auto* st = new DevState();
    st->status_ring = at::zeros({RING, 8}, at::TensorOptions().dtype(at::kInt)).pin_memory();
    static_assert(sizeof(hgs_status) == 32, "hgs_status is 8 words");
    st->ring = reinterpret_cast<hgs_status*>(st->status_ring.data_ptr<int32_t>());
    hip_ok(hipEventCreateWithFlags(&st->status_event, hipEventDisableTiming), "hipEventCreate");
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

// B camera blocks: settings[b] points into the batched tensors (kept alive here)
struct Settings {
  std::vector<hgs_settings> s;
  Tensor bg, vm, pm, cp;    // (B,3), (B,16), (B,16), (B,3)
};

Settings make_settings(const Tensor& bg, const Tensor& vm, const Tensor& pm, const Tensor& cp, int64_t B, int64_t H,
                       int64_t W, const double* tanfovx, const double* tanfovy, double scale_modifier,
                       int64_t sh_degree, bool prefiltered, bool debug, const c10::Device& dev) {
  if (B < 1 || B > HGS_MAX_VIEWS)
    throw std::runtime_error("a batched call takes 1.." + std::to_string(HGS_MAX_VIEWS) + " views, got " + std::to_string(B));
  Settings r;
  r.vm = f32c(vm, dev, "viewmatrix");
  r.pm = f32c(pm, dev, "projmatrix");
  r.cp = f32c(cp, dev, "campos");
  Tensor bgc = f32c(bg, dev, "bg");
  if (bgc.numel() == 3 && B > 1) bgc = bgc.reshape({1, 3}).expand({B, 3}).contiguous();
  r.bg = bgc;
  if (r.bg.numel() != 3 * B || r.vm.numel() != 16 * B || r.pm.numel() != 16 * B || r.cp.numel() != 3 * B)
    throw std::runtime_error("bg/campos must have 3 elements (per view), viewmatrix/projmatrix 16");
  r.s.resize(B);
  for (int64_t b = 0; b < B; ++b) {
    hgs_settings& s = r.s[b];
    s.image_height = (int32_t)H;
    s.image_width = (int32_t)W;
    s.tanfovx = (float)tanfovx[b];
    s.tanfovy = (float)tanfovy[b];
    s.bg = r.bg.data_ptr<float>() + 3 * b;
    s.scale_modifier = (float)scale_modifier;
    s.viewmatrix = r.vm.data_ptr<float>() + 16 * b;
    s.projmatrix = r.pm.data_ptr<float>() + 16 * b;
    s.sh_degree = (int32_t)sh_degree;
    s.campos = r.cp.data_ptr<float>() + 3 * b;
    s.prefiltered = prefiltered ? 1 : 0;
    s.debug = debug ? 1 : 0;
  }
  return r;
}

void check_rc(int rc, const char* what) {
  if (rc == HGS_ESHAPE)
    throw std::runtime_error("inconsistent optional inputs (shs/colors_precomp, scales+rotations/cov3D_precomp)");
  if (rc != HGS_OK) throw std::runtime_error(std::string("libhgs_rast: ") + what + " failed with code " + std::to_string(rc));
}

void need_numel(const Tensor& t, int64_t n, const char* name, const char* shape) {
  if (t.numel() != n)
    throw std::runtime_error(std::string(name) + " must have dimensions " + shape + " (got " + std::to_string(t.numel()) +
                             " elements, expected " + std::to_string(n) + ")");
}

// What the backward call needs.  The plan stays alive as long as the autograd node does, so the
// backward can run more than once (retain_graph=True, torch.autograd.grad per loss term) like
// upstream's Python autograd.Function; gradient tensors are allocated per backward call.
struct BwdPlan : torch::CustomClassHolder {
  Settings settings;
  Tensor work, work2;                 // [geom | img | bin | rows] (work2: bin | rows after a retry)
  char *geom = nullptr, *bin = nullptr, *img = nullptr, *rows = nullptr;
  int64_t cap = 0;
  hgs_status status{};
  // the pinned status slot of the forward call and its issue number: the blend forward writes num_pairs there once the
  // sort has run; backward() sizes its pair rows by it while the slot is known not to have been handed out again
  const hgs_status* slot = nullptr;
  int64_t slot_issue = 0;
  int32_t B = 1, P = 0, M = 0, act = 0;
  bool batched = false;
  bool has_sh = false, has_cp = false, has_sr = false, has_cv = false;
  std::vector<int64_t> opac_sizes;
  // gradient tensors of the FIRST backward call, allocated by forward() while it waits for the device-side status
  // (host time that is otherwise idle; in backward() the same allocations sit on the step's critical host path)
  std::vector<Tensor> pre_grads;
  void* pre_stream = nullptr;

  void alloc_grads(std::vector<Tensor>& g, const c10::Device& dev) const {
    const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    g.assign(8, Tensor());
    g[0] = at::empty({(int64_t)P, 3}, fopt);                                                             // means3D
    g[1] = batched ? at::empty({(int64_t)B, (int64_t)P, 3}, fopt) : at::empty({(int64_t)P, 3}, fopt);   // means2D
    if (has_sh) g[2] = at::empty({(int64_t)P, (int64_t)M, 3}, fopt);
    if (has_cp) g[3] = at::empty({(int64_t)P, 3}, fopt);
    g[4] = at::empty(opac_sizes, fopt);
    if (has_sr) { g[5] = at::empty({(int64_t)P, 3}, fopt); g[6] = at::empty({(int64_t)P, 4}, fopt); }
    if (has_cv) g[7] = at::empty({(int64_t)P, 6}, fopt);
  }
};

// Packed backward (view_parallel.py): while the request is set, a backward whose inputs are the SH / scale / rotation
// configuration writes ONE (P, 15 + 3M) pack (hgs_backward_batch_packed) - the tensor the rank sends - and hands autograd
// strided views of it; `take_packed()` collects the pack.  Thread-local: the request belongs to the thread that runs
// autograd.grad (torch's engine runs a CPU-initiated backward of one device on the calling thread's device worker; the
// flag is read through a process-wide atomic for that reason, the pack returned per process).
std::atomic<int> g_pack_request{0};
std::mutex g_pack_mu;
Tensor g_last_pack;

struct Rasterize : public torch::autograd::Function<Rasterize> {
  // 22 arguments after ctx (backward returns one slot per argument).  `tanfov` is a CPU double
  // tensor [2][B] (x row, y row); `batch` = 0 for the single-view API, else the number of views;
  // `act` = HGS_ACT_* bits (raw parameters in, activations fused into the per-Gaussian kernels).
  static variable_list forward(AutogradContext* ctx, const Tensor& means3D, const Tensor& means2D,
                               const Tensor& sh, const Tensor& colors_precomp, const Tensor& opacities,
                               const Tensor& scales, const Tensor& rotations, const Tensor& cov3D,
                               const Tensor& bg, const Tensor& viewmatrix, const Tensor& projmatrix,
                               const Tensor& campos, const Tensor& tanfov, int64_t H, int64_t W,
                               double scale_modifier, int64_t sh_degree, bool prefiltered, bool debug,
                               bool want_grad, int64_t batch, int64_t act_in) {
    // bit 16 of `act` is the binding's own (never handed to the library): `means2D` is UNINITIALISED storage that the
    // forward's per-Gaussian kernel zero-fills (ABI v16: hgs_forward_batch_act_leaf) - what renderer.render() passes
    // instead of launching torch.zeros
    const bool zero_leaf = ((act_in >> 16) & 1) != 0;
    const int64_t act = act_in & 0xffff;
    const auto th0 = std::chrono::steady_clock::now();
    const c10::Device dev = means3D.device();
    if (!dev.is_cuda())
      throw std::runtime_error("humangaussian_amd: tensors must live on a HIP device (torch device type 'cuda'); "
                               "there is no CPU path");
    const bool batched = batch > 0;
    const int64_t B = batched ? batch : 1;
    const int64_t P = means3D.size(0);
    if (P != 0 && (means3D.dim() != 2 || means3D.size(1) != 3))
      throw std::runtime_error("means3D must have dimensions (num_points, 3)");
    if (tanfov.device().is_cuda() || tanfov.scalar_type() != at::kDouble || tanfov.numel() != 2 * B || !tanfov.is_contiguous())
      throw std::runtime_error("tanfov must be a contiguous CPU double tensor of shape (2, views)");
    DeviceSwitch guard(dev.index());

    auto plan = c10::make_intrusive<BwdPlan>();
    const double* tf = tanfov.data_ptr<double>();
    plan->settings = make_settings(bg, viewmatrix, projmatrix, campos, B, H, W, tf, tf + B, scale_modifier, sh_degree,
                                   prefiltered, debug, dev);
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
    // the kernels index every per-Gaussian tensor by P: a stale tensor (captured before a
    // densification / pruning step) must fail here, not read out of bounds on the device
    need_numel(op_, P, "opacities", "(num_points, 1)");
    if (has_cp) need_numel(cp_, 3 * P, "colors_precomp", "(num_points, 3)");
    if (has_sc) need_numel(sc_, 3 * P, "scales", "(num_points, 3)");
    if (has_ro) need_numel(ro_, 4 * P, "rotations", "(num_points, 4)");
    if (has_cv) need_numel(cv_, 6 * P, "cov3D_precomp", "(num_points, 6)");

    const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    Tensor color, depth, alpha, radii;
    if (batched) {
      color = at::empty({B, 3, H, W}, fopt); depth = at::empty({B, 1, H, W}, fopt); alpha = at::empty({B, 1, H, W}, fopt);
      radii = at::empty({B, P}, fopt.dtype(at::kInt));
    } else {
      color = at::empty({3, H, W}, fopt); depth = at::empty({1, H, W}, fopt); alpha = at::empty({1, H, W}, fopt);
      radii = at::empty({P}, fopt.dtype(at::kInt));
    }

    float* leaf_ptr = nullptr;
    if (zero_leaf && means2D.defined() && means2D.numel() > 0) {
      if (means2D.numel() != B * P * 3) throw std::runtime_error("means2D must have (views x) num_points x 3 elements");
      if (means2D.device() == dev && means2D.scalar_type() == at::kFloat && means2D.is_contiguous()) {
        leaf_ptr = static_cast<float*>(means2D.data_ptr());
      } else {                       // (a leaf the kernel cannot write: filled the ordinary way)
        at::NoGradGuard ng;
        const_cast<Tensor&>(means2D).zero_();
      }
    }

    DevState& st = state_for(dev.index());
    std::lock_guard<std::mutex> lk(st.mu);
    hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
    Estimate& est = st.est[std::make_tuple(B, H, W)];
    int64_t cap = P > 0 ? (est.capacity > 0 ? est.capacity : round_capacity(3 * P * B)) : 0;
    int32_t hint = (int32_t)est.tile_hint;
    // one allocation for the four opaque regions [geom | img | bin | backward rows] (the fork keeps
    // three such byte tensors for its backward); a capacity retry re-allocates only the last two
    const int64_t g_sz = al256(hgs_geom_bytes_batch((int32_t)B, (int32_t)P, (int32_t)H, (int32_t)W));
    const int64_t i_sz = al256(hgs_img_bytes_batch((int32_t)B, (int32_t)H, (int32_t)W));
    // (the backward's pair rows - 16 x 40 B per entry - are allocated by backward() for its own duration: they would
    // otherwise be held by every view of a step until its backward runs)
    int64_t b_sz = al256(hgs_bin_bytes(cap));
    const int64_t s_sz = 0;
    const auto bopt = at::TensorOptions().dtype(at::kByte).device(dev);
    plan->work = at::empty({g_sz + i_sz + b_sz + s_sz}, bopt);
    char* base = static_cast<char*>(plan->work.data_ptr());
    plan->geom = base;
    plan->img = base + g_sz;
    plan->bin = base + g_sz + i_sz;
    plan->rows = base + g_sz + i_sz + b_sz;

    hgs_status h{};
    int attempt = 0;
    for (; attempt < 4; ++attempt) {
      const int slot = st.ring_pos;
      st.ring_pos = (st.ring_pos + 1) % RING;
      st.ring_issued += 1;
      plan->slot = &st.ring[slot];
      plan->slot_issue = st.ring_issued;
      // The library stores the status into this pinned slot from the fill launch and sets reserved[2] = 1 LAST (behind a
      // system-scope fence): the host polls that word.  (An event recorded behind the fill stage cost the GPU ~6 us of
      // idle time per forward: the record is a barrier packet with a system-scope release in the middle of the chain.)
      volatile uint32_t* ready = &st.ring[slot].reserved[2];
      *ready = 0u;
      std::atomic_thread_fence(std::memory_order_seq_cst);
      const auto th1 = std::chrono::steady_clock::now();
      if (attempt == 0) st.host_ns[0] += std::chrono::duration_cast<std::chrono::nanoseconds>(th1 - th0).count();
      const int rc = hgs_forward_batch_act_leaf(plan->settings.s.data(), (int32_t)B, (int32_t)P, M, fptr(m3), fptr(sh_), fptr(cp_),
                                                fptr(op_), fptr(sc_), fptr(ro_), fptr(cv_), fptr_mut(color), fptr_mut(depth),
                                                fptr_mut(alpha), P > 0 ? radii.data_ptr<int32_t>() : nullptr, plan->geom,
                                                plan->bin, cap, plan->img, want_grad ? 1 : 0, hint, &st.ring[slot], /*mapped=*/1,
                                                /*status_event=*/nullptr, g_stage_fwd.empty() ? nullptr : g_stage_fwd.data(),
                                                (int32_t)act, leaf_ptr, stream);
      check_rc(rc, "hgs_forward_batch");
      const auto th2 = std::chrono::steady_clock::now();
      st.host_ns[1] += std::chrono::duration_cast<std::chrono::nanoseconds>(th2 - th1).count();
      // `debug=True` is upstream's switch for surfacing device errors at the call that caused
      // them (std::runtime_error): synchronise and report
      if (debug) hip_ok(hipStreamSynchronize(stream), "device error in the rasterizer forward (debug=True)");
      // Host work that does not depend on the status goes HERE, in the shadow of the wait below (with an idle GPU
      // that wait is the ~25 us the first two kernels take): the autograd bookkeeping and the gradient tensors of
      // the backward call.  (Measured: a step is ~172 us of kernels and ~170 us of host work; whatever sits behind
      // the wait is on the critical path whenever the host is the slower of the two.)
      if (attempt == 0 && want_grad) {
        plan->B = (int32_t)B; plan->P = (int32_t)P; plan->M = M;
        plan->batched = batched;
        plan->act = (int32_t)act;
        plan->has_sh = has_sh; plan->has_cp = has_cp; plan->has_sr = has_sc; plan->has_cv = has_cv;
        plan->opac_sizes = opacities.sizes().vec();
        // inputs and outputs go through save_for_backward (version checks: the backward reads the
        // forward's outputs, so they must not be modified in place before backward()); the opaque
        // work buffers ride in the plan
        variable_list saved = {m3, op_, radii, color, depth, alpha};
        for (const Tensor* t : {&sh_, &cp_, &sc_, &ro_, &cv_})
          if (t->defined()) saved.push_back(*t);
        ctx->save_for_backward(saved);
        plan->alloc_grads(plan->pre_grads, dev);
        plan->pre_stream = (void*)stream;
      }
      // One host wait per forward, like upstream's blocking read of num_rendered - but only for the status: sort and
      // blend are already enqueued and keep the GPU busy while the host goes on to autograd and the backward launch.
      const auto tw = std::chrono::steady_clock::now();
      st.host_ns[2] += std::chrono::duration_cast<std::chrono::nanoseconds>(tw - th2).count();
      for (uint64_t spins = 0; *ready == 0u; ++spins) {
        if ((spins & 0x3ff) == 0x3ff) {
          if (std::chrono::steady_clock::now() - tw > std::chrono::seconds(2)) {       // (a device error: surface it)
            hip_ok(hipStreamSynchronize(stream), "rasterizer forward");
            if (*ready == 0u) throw std::runtime_error("libhgs_rast: the forward finished without publishing its status");
            break;
          }
          std::this_thread::yield();
        }
      }
      std::atomic_thread_fence(std::memory_order_seq_cst);
      const int64_t waited = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tw).count();
      st.wait_ns += waited;
      st.host_ns[3] += waited;
      h = st.ring[slot];
      if (!h.overflow) break;
      st.retries += 1;
      if (h.overflow & 1u) {                    // R exceeded the capacity: grow, re-run
        cap = round_capacity((int64_t)(h.num_rendered * 1.25) + 1);
        b_sz = al256(hgs_bin_bytes(cap));
        plan->work2 = at::empty({b_sz + s_sz}, bopt);
        plan->bin = static_cast<char*>(plan->work2.data_ptr());
        plan->rows = plan->bin + b_sz;
      }
      if (h.overflow & 2u) hint = 0;            // a tile list outgrew the hint
    }
    if (attempt == 4) throw std::runtime_error("libhgs_rast: entry capacity did not converge");
    st.calls += 1;
    st.max_R = h.num_rendered;
    st.max_tile = h.reserved[1];
    st.last_capacity = cap;
    st.last_hint = hint;
    if (P > 0) {
      // decaying maxima: follow the workload down slowly, up at once
      est.capacity = std::max<int64_t>(round_capacity((int64_t)(h.num_rendered * 1.25) + 1),
                                       est.capacity > 0 ? round_capacity((int64_t)(est.capacity * 0.9)) : 0);
      est.tile_hint = std::max<int64_t>(std::max<int64_t>(1024, (int64_t)(h.reserved[1] * 1.5) + 64),
                                        (int64_t)(est.tile_hint * 0.9));
    }
    ctx->set_materialize_grads(false);
    if (want_grad) {
      plan->cap = cap;
      plan->status = h;
      ctx->saved_data["plan"] = c10::IValue::make_capsule(plan);
    }
    ctx->mark_non_differentiable({radii});
    st.host_ns[4] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - th0).count();   // whole forward()
    return {color, radii, depth, alpha};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto tb0 = std::chrono::steady_clock::now();
    auto it = ctx->saved_data.find("plan");
    if (it == ctx->saved_data.end() || it->second.isNone())
      throw std::runtime_error("humangaussian_amd: the rasterizer's backward state is gone (the forward ran without "
                               "gradient tracking)");
    auto holder = it->second.toCapsule();
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
    const Tensor gc = grads[0].defined() ? f32c(grads[0], dev, "grad_color") : Tensor();
    const Tensor gd = grads[2].defined() ? f32c(grads[2], dev, "grad_depth") : Tensor();
    const Tensor ga = grads[3].defined() ? f32c(grads[3], dev, "grad_alpha") : Tensor();
    hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
    std::vector<Tensor> g;
    if (!plan->pre_grads.empty() && plan->pre_stream == (void*)stream) g = std::move(plan->pre_grads);
    plan->pre_grads.clear();
    if (g.empty()) plan->alloc_grads(g, dev);
    Tensor &d_means3D = g[0], &d_means2D = g[1], &d_sh = g[2], &d_cp = g[3], &d_opac = g[4], &d_sc = g[5], &d_ro = g[6], &d_cv = g[7];
    // pair rows: 16 per entry in the worst case, ~4.4 on an avatar - counted once the forward's sort has run (the
    // blend forward then publishes hgs_status.num_pairs into the call's pinned slot; nothing is waited for here)
    int64_t pairs = 0;
    {
      DevState& st = state_for(dev.index());
      std::lock_guard<std::mutex> lk(st.mu);
      if (plan->slot && st.ring_issued - plan->slot_issue < RING - 1)
        pairs = (int64_t)*reinterpret_cast<const volatile uint32_t*>(&plan->slot->num_pairs);
      st.last_pairs = pairs;
      st.last_scratch = al256(hgs_bwd_scratch_bytes_pairs((int64_t)plan->status.num_rendered, pairs));
    }
    Tensor scratch = at::empty({al256(hgs_bwd_scratch_bytes_pairs((int64_t)plan->status.num_rendered, pairs))},
                               at::TensorOptions().dtype(at::kByte).device(dev));
    plan->rows = static_cast<char*>(scratch.data_ptr());
    // tell the library what the scratch holds: its kernels check the count against the one the sort left on the device and
    // write nothing (NaN gradients) if the slot delivered a stale number (hgs_rast.h: hgs_backward)
    plan->status.num_pairs = (uint32_t)pairs;
    const auto tb1 = std::chrono::steady_clock::now();
    const bool packed = g_pack_request.load(std::memory_order_relaxed) != 0 && plan->has_sh && plan->has_sr && !plan->has_cp &&
                        !plan->has_cv && plan->M >= 1 && plan->P > 0;
    if (packed) {
      const int64_t P = plan->P, M = plan->M, F = 15 + 3 * M;
      Tensor pack = at::empty({P, F}, at::TensorOptions().dtype(at::kFloat).device(dev));
      const int rc = hgs_backward_batch_packed(
          plan->settings.s.data(), plan->B, plan->P, plan->M, fptr(m3), fptr(sh_), fptr(op_), fptr(sc_), fptr(ro_),
          radii.data_ptr<int32_t>(), fptr(color), fptr(depth), fptr(alpha), fptr(gc), fptr(gd), fptr(ga), plan->geom, plan->bin,
          plan->img, &plan->status, plan->cap, plan->rows, pack.data_ptr<float>(),
          plan->batched ? fptr_mut(d_means2D) : nullptr, g_stage_bwd.empty() ? nullptr : g_stage_bwd.data(), plan->act, stream);
      check_rc(rc, "hgs_backward_batch_packed");
      variable_list out(22);
      out[0] = pack.as_strided({P, 3}, {F, 1}, 0);
      out[1] = plan->batched ? d_means2D : pack.as_strided({P, 3}, {F, 1}, 3);
      out[2] = pack.as_strided({P, M, 3}, {F, 3, 1}, 6);
      {
        std::vector<int64_t> os = plan->opac_sizes, ostr(plan->opac_sizes.size(), 1);
        if (!ostr.empty()) ostr[0] = F;
        out[4] = pack.as_strided(os, ostr, 6 + 3 * M);
      }
      out[5] = pack.as_strided({P, 3}, {F, 1}, 7 + 3 * M);
      out[6] = pack.as_strided({P, 4}, {F, 1}, 10 + 3 * M);
      {
        std::lock_guard<std::mutex> lk(g_pack_mu);
        g_last_pack = pack;
      }
      DevState& st = state_for(dev.index());
      const auto tb2 = std::chrono::steady_clock::now();
      st.host_ns[5] += std::chrono::duration_cast<std::chrono::nanoseconds>(tb1 - tb0).count();
      st.host_ns[6] += std::chrono::duration_cast<std::chrono::nanoseconds>(tb2 - tb1).count();
      st.host_ns[7] += std::chrono::duration_cast<std::chrono::nanoseconds>(tb2 - tb0).count();
      return out;
    }
    const int rc = hgs_backward_batch_act(
        plan->settings.s.data(), plan->B, plan->P, plan->M, fptr(m3), fptr(sh_), fptr(cp_), fptr(op_), fptr(sc_), fptr(ro_),
        fptr(cv_), plan->P > 0 ? radii.data_ptr<int32_t>() : nullptr, fptr(color), fptr(depth), fptr(alpha), fptr(gc),
        fptr(gd), fptr(ga), plan->geom, plan->bin, plan->img, &plan->status, plan->cap, plan->rows, fptr_mut(d_means3D),
        fptr_mut(d_means2D), fptr_mut(d_sh), fptr_mut(d_cp), fptr_mut(d_opac), fptr_mut(d_sc), fptr_mut(d_ro),
        fptr_mut(d_cv), g_stage_bwd.empty() ? nullptr : g_stage_bwd.data(), plan->act, stream);
    check_rc(rc, "hgs_backward_batch");
    const auto tb2 = std::chrono::steady_clock::now();
    if (plan->settings.s[0].debug) hip_ok(hipStreamSynchronize(stream), "device error in the rasterizer backward (debug=True)");
    variable_list out(22);
    out[0] = d_means3D; out[1] = d_means2D; out[2] = d_sh; out[3] = d_cp;
    out[4] = d_opac; out[5] = d_sc; out[6] = d_ro; out[7] = d_cv;
    {
      DevState& st = state_for(dev.index());
      st.host_ns[5] += std::chrono::duration_cast<std::chrono::nanoseconds>(tb1 - tb0).count();
      st.host_ns[6] += std::chrono::duration_cast<std::chrono::nanoseconds>(tb2 - tb1).count();
      st.host_ns[7] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tb0).count();   // whole backward()
    }
    return out;
  }
};

// None optionals travel through autograd as one shared empty tensor (upstream does the same:
// "None inputs become empty tensors"); undefined tensors are not valid apply() inputs.
Tensor opt(const c10::optional<Tensor>& t) {
  static const Tensor empty = at::empty({0}, at::TensorOptions().dtype(at::kFloat));
  return t.has_value() ? *t : empty;
}

Tensor tanfov_tensor(const std::vector<double>& x, const std::vector<double>& y) {
  if (x.size() != y.size() || x.empty()) throw std::runtime_error("tanfovx / tanfovy: one value per view");
  Tensor t = at::empty({2, (int64_t)x.size()}, at::TensorOptions().dtype(at::kDouble));
  double* d = t.data_ptr<double>();
  for (size_t i = 0; i < x.size(); ++i) { d[i] = x[i]; d[x.size() + i] = y[i]; }
  return t;
}

// single view: the signature of upstream's _C.rasterize_gaussians call
std::vector<Tensor> rasterize(const Tensor& means3D, const Tensor& means2D, const c10::optional<Tensor>& sh,
                              const c10::optional<Tensor>& colors_precomp, const Tensor& opacities,
                              const c10::optional<Tensor>& scales, const c10::optional<Tensor>& rotations,
                              const c10::optional<Tensor>& cov3D, const Tensor& bg, const Tensor& viewmatrix,
                              const Tensor& projmatrix, const Tensor& campos, int64_t H, int64_t W, double tanfovx,
                              double tanfovy, double scale_modifier, int64_t sh_degree, bool prefiltered, bool debug,
                              bool want_grad, bool zero_means2D) {
  return Rasterize::apply(means3D, means2D, opt(sh), opt(colors_precomp), opacities, opt(scales), opt(rotations),
                          opt(cov3D), bg, viewmatrix, projmatrix, campos, tanfov_tensor({tanfovx}, {tanfovy}), H, W,
                          scale_modifier, sh_degree, prefiltered, debug, want_grad, (int64_t)0,
                          (int64_t)(zero_means2D ? (1 << 16) : 0));
}

// B views in one launch set: bg (3) or (B,3), viewmatrix / projmatrix (B,4,4), campos (B,3), means2D (B,P,3);
// returns color (B,3,H,W), radii (B,P), depth (B,1,H,W), alpha (B,1,H,W)
std::vector<Tensor> rasterize_batch(const Tensor& means3D, const Tensor& means2D, const c10::optional<Tensor>& sh,
                                    const c10::optional<Tensor>& colors_precomp, const Tensor& opacities,
                                    const c10::optional<Tensor>& scales, const c10::optional<Tensor>& rotations,
                                    const c10::optional<Tensor>& cov3D, const Tensor& bg, const Tensor& viewmatrix,
                                    const Tensor& projmatrix, const Tensor& campos, int64_t H, int64_t W,
                                    const std::vector<double>& tanfovx, const std::vector<double>& tanfovy,
                                    double scale_modifier, int64_t sh_degree, bool prefiltered, bool debug,
                                    bool want_grad, int64_t activation_flags) {
  const int64_t B = (int64_t)tanfovx.size();
  if (means2D.defined() && means2D.numel() > 0 && means2D.numel() != B * means3D.size(0) * 3)
    throw std::runtime_error("means2D must have dimensions (views, num_points, 3) for a batched call");
  return Rasterize::apply(means3D, means2D, opt(sh), opt(colors_precomp), opacities, opt(scales), opt(rotations),
                          opt(cov3D), bg, viewmatrix, projmatrix, campos, tanfov_tensor(tanfovx, tanfovy), H, W,
                          scale_modifier, sh_degree, prefiltered, debug, want_grad, B, activation_flags);
}

Tensor mark_visible(const Tensor& positions, const Tensor& bg, const Tensor& viewmatrix, const Tensor& projmatrix,
                    const Tensor& campos, int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier,
                    int64_t sh_degree, bool prefiltered, bool debug) {
  at::NoGradGuard ng;
  const c10::Device dev = positions.device();
  if (!dev.is_cuda()) throw std::runtime_error("humangaussian_amd: tensors must live on a HIP device");
  DeviceSwitch guard(dev.index());
  Settings s = make_settings(bg, viewmatrix, projmatrix, campos, 1, H, W, &tanfovx, &tanfovy, scale_modifier, sh_degree,
                             prefiltered, debug, dev);
  const Tensor pos = f32c(positions, dev, "positions");
  const int64_t P = pos.size(0);
  Tensor present = at::zeros({P}, at::TensorOptions().dtype(at::kByte).device(dev));
  if (P > 0) {
    const int rc = hgs_mark_visible(s.s.data(), (int32_t)P, pos.data_ptr<float>(), present.data_ptr<uint8_t>(),
                                    c10::hip::getCurrentHIPStream(dev.index()).stream());
    check_rc(rc, "hgs_mark_visible");
  }
  return present.to(at::kBool);
}

// replaces simple_knn._C.distCUDA2: mean squared distance to the 3 nearest neighbours
Tensor knn_mean_dist2(const Tensor& points, bool brute_force) {
  at::NoGradGuard ng;
  const c10::Device dev = points.device();
  if (!dev.is_cuda()) throw std::runtime_error("humangaussian_amd: tensors must live on a HIP device");
  if (points.dim() != 2 || points.size(1) != 3) throw std::runtime_error("points must have dimensions (num_points, 3)");
  DeviceSwitch guard(dev.index());
  const Tensor pts = f32c(points, dev, "points");
  const int64_t P = pts.size(0);
  Tensor out = at::empty({P}, at::TensorOptions().dtype(at::kFloat).device(dev));
  if (P > 0x7fffffffll / 4) throw std::runtime_error("too many points");
  if (P > 0 && brute_force) {
    const int rc = hgs_knn_mean_dist2((int32_t)P, pts.data_ptr<float>(), out.data_ptr<float>(),
                                      c10::hip::getCurrentHIPStream(dev.index()).stream());
    check_rc(rc, "hgs_knn_mean_dist2");
  } else if (P > 0) {      // the grid form (near-linear); its scratch lives for the duration of the call's stream work
    Tensor scratch = at::empty({(int64_t)hgs_knn_scratch_bytes((int32_t)P)}, at::TensorOptions().dtype(at::kByte).device(dev));
    const int rc = hgs_knn_mean_dist2_grid((int32_t)P, pts.data_ptr<float>(), out.data_ptr<float>(), scratch.data_ptr(),
                                           c10::hip::getCurrentHIPStream(dev.index()).stream());
    check_rc(rc, "hgs_knn_mean_dist2_grid");
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

// view-parallel reduction of the all-gathered packs: (world, P, F) [+ the running total (P, F) of the step's earlier
// collectives] -> (P, F), one left-to-right chain
Tensor reduce_view_packs(const Tensor& gathered, const c10::optional<Tensor>& acc_in) {
  at::NoGradGuard ng;
  const c10::Device dev = gathered.device();
  if (!dev.is_cuda()) throw std::runtime_error("humangaussian_amd: tensors must live on a HIP device");
  if (gathered.dim() != 3) throw std::runtime_error("gathered packs must have dimensions (world, num_points, F)");
  DeviceSwitch guard(dev.index());
  const Tensor g = f32c(gathered, dev, "gathered");
  Tensor a;
  if (acc_in.has_value() && acc_in->defined()) {
    a = f32c(*acc_in, dev, "running total");
    if (a.numel() != g.size(1) * g.size(2)) throw std::runtime_error("running total must have dimensions (num_points, F)");
  }
  Tensor out = at::empty({g.size(1), g.size(2)}, g.options());
  const int rc = hgs_reduce_view_packs_acc((int32_t)g.size(0), g.size(1), (int32_t)g.size(2), g.data_ptr<float>(),
                                           a.defined() ? a.data_ptr<float>() : nullptr, out.data_ptr<float>(),
                                           c10::hip::getCurrentHIPStream(dev.index()).stream());
  check_rc(rc, "hgs_reduce_view_packs_acc");
  return out;
}

void set_packed_backward(bool on) {
  g_pack_request.store(on ? 1 : 0, std::memory_order_relaxed);
  if (!on) return;
  std::lock_guard<std::mutex> lk(g_pack_mu);
  g_last_pack = Tensor();
}

// the pack the last packed backward wrote (None if the backward was not eligible: the caller packs the six tensors itself)
c10::optional<Tensor> take_packed() {
  std::lock_guard<std::mutex> lk(g_pack_mu);
  Tensor t = g_last_pack;
  g_last_pack = Tensor();
  if (!t.defined()) return c10::nullopt;
  return t;
}

// the LAST reduction of a view-parallel step with the unpack fused in: (world, P, 15 + 3M) [+ running total (P, F)] ->
// [means3D (P,3), means2D (P,3), sh (P,M,3), opacity (P,1), scales (P,3), rotations (P,4), radii (P,) int32]
std::vector<Tensor> reduce_view_packs_unpack(const Tensor& gathered, const c10::optional<Tensor>& acc_in) {
  at::NoGradGuard ng;
  const c10::Device dev = gathered.device();
  if (!dev.is_cuda()) throw std::runtime_error("humangaussian_amd: tensors must live on a HIP device");
  if (gathered.dim() != 3 || gathered.size(2) < 15 || (gathered.size(2) - 15) % 3)
    throw std::runtime_error("gathered packs must have dimensions (world, num_points, 15 + 3 M)");
  DeviceSwitch guard(dev.index());
  const Tensor g = f32c(gathered, dev, "gathered");
  const int64_t P = g.size(1), F = g.size(2), M = (F - 15) / 3;
  Tensor a;
  if (acc_in.has_value() && acc_in->defined()) {
    a = f32c(*acc_in, dev, "running total");
    if (a.numel() != P * F) throw std::runtime_error("running total must have dimensions (num_points, F)");
  }
  const auto fopt = g.options();
  Tensor m3 = at::empty({P, 3}, fopt), m2 = at::empty({P, 3}, fopt), sh = at::empty({P, M, 3}, fopt);
  Tensor op = at::empty({P, 1}, fopt), sc = at::empty({P, 3}, fopt), ro = at::empty({P, 4}, fopt);
  Tensor radii = at::empty({P}, fopt.dtype(at::kInt));
  const int rc = hgs_reduce_view_packs_unpack((int32_t)g.size(0), P, (int32_t)M, g.data_ptr<float>(),
                                              a.defined() ? a.data_ptr<float>() : nullptr, fptr_mut(m3), fptr_mut(m2),
                                              fptr_mut(sh), fptr_mut(op), fptr_mut(sc), fptr_mut(ro),
                                              P > 0 ? radii.data_ptr<int32_t>() : nullptr,
                                              c10::hip::getCurrentHIPStream(dev.index()).stream());
  check_rc(rc, "hgs_reduce_view_packs_unpack");
  return {m3, m2, sh, op, sc, ro, radii};
}

// ---- bookkeeping either side of the path (include/hgs_rast.h: hgs_densify_*, hgs_compact_*, hgs_reanchor)
void need_dev(const Tensor& t, const c10::Device& dev, at::ScalarType ty, const char* name) {
  if (t.device() != dev || t.scalar_type() != ty || !t.is_contiguous())
    throw std::runtime_error(std::string(name) + ": expected a contiguous tensor of the right dtype on " + dev.str());
}

// in-place update of xyz_gradient_accum / denom / max_radii2D; returns (radii_max int32 [P], visibility bool [P])
std::vector<Tensor> densify_stats(const Tensor& grad_means2D, const Tensor& radii, const c10::optional<Tensor>& keep,
                                  Tensor accum, Tensor denom, Tensor max_radii2D) {
  at::NoGradGuard ng;
  const c10::Device dev = radii.device();
  if (!dev.is_cuda()) throw std::runtime_error("humangaussian_amd: tensors must live on a HIP device");
  DeviceSwitch guard(dev.index());
  const Tensor r = radii.dim() == 1 ? radii.unsqueeze(0) : radii;
  const int64_t B = r.size(0), P = r.size(1);
  const Tensor g = f32c(grad_means2D, dev, "viewspace gradient");
  const Tensor rc = r.contiguous();
  if (rc.scalar_type() != at::kInt) throw std::runtime_error("radii must be int32");
  if (g.numel() != B * P * 3) throw std::runtime_error("viewspace gradient must have dimensions (views, num_points, 3)");
  need_dev(accum, dev, at::kFloat, "xyz_gradient_accum");
  need_dev(denom, dev, at::kFloat, "denom");
  need_dev(max_radii2D, dev, at::kFloat, "max_radii2D");
  if (accum.numel() != P || denom.numel() != P || max_radii2D.numel() != P)
    throw std::runtime_error("xyz_gradient_accum / denom / max_radii2D must have num_points elements");
  Tensor keep_u8;
  if (keep.has_value() && keep->defined()) {
    if (keep->device() != dev) throw std::runtime_error("keep mask must live on the device of radii");
    keep_u8 = keep->to(at::kByte).contiguous();
    if (keep_u8.numel() != P) throw std::runtime_error("keep mask must have num_points elements on the device");
  }
  Tensor rmax = at::empty({P}, rc.options());
  Tensor vis = at::empty({P}, rc.options().dtype(at::kByte));
  const int rcode = hgs_densify_stats((int32_t)B, (int32_t)P, fptr(g), P > 0 ? rc.data_ptr<int32_t>() : nullptr,
                                      keep_u8.defined() && P > 0 ? keep_u8.data_ptr<uint8_t>() : nullptr, fptr_mut(accum),
                                      fptr_mut(denom), fptr_mut(max_radii2D), P > 0 ? rmax.data_ptr<int32_t>() : nullptr,
                                      P > 0 ? vis.data_ptr<uint8_t>() : nullptr,
                                      c10::hip::getCurrentHIPStream(dev.index()).stream());
  check_rc(rcode, "hgs_densify_stats");
  return {rmax, vis.to(at::kBool)};
}

// returns (clone bool [P], split bool [P], prune bool [P], counts int32 [3] on the device)
std::vector<Tensor> densify_masks(const Tensor& accum, const Tensor& denom, const Tensor& scales, bool scales_are_log,
                                  const Tensor& opacity, bool opacity_is_logit, const Tensor& max_radii2D,
                                  double grad_threshold, double percent_dense, double extent, double min_opacity,
                                  double max_screen_size, double size_thresh) {
  at::NoGradGuard ng;
  const c10::Device dev = accum.device();
  if (!dev.is_cuda()) throw std::runtime_error("humangaussian_amd: tensors must live on a HIP device");
  DeviceSwitch guard(dev.index());
  const Tensor a = f32c(accum, dev, "xyz_gradient_accum"), d = f32c(denom, dev, "denom"), sc = f32c(scales, dev, "scales");
  const Tensor op = f32c(opacity, dev, "opacity"), mr = f32c(max_radii2D, dev, "max_radii2D");
  const int64_t P = a.numel();
  if (d.numel() != P || sc.numel() != 3 * P || op.numel() != P || mr.numel() != P)
    throw std::runtime_error("densify_masks: inconsistent shapes");
  const auto bopt = at::TensorOptions().dtype(at::kByte).device(dev);
  Tensor cl = at::empty({P}, bopt), sp = at::empty({P}, bopt), pr = at::empty({P}, bopt);
  Tensor counts = at::empty({3}, bopt.dtype(at::kInt));
  const int rc = hgs_densify_masks((int32_t)P, fptr(a), fptr(d), fptr(sc), scales_are_log ? 1 : 0, fptr(op),
                                   opacity_is_logit ? 1 : 0, fptr(mr), (float)grad_threshold, (float)percent_dense,
                                   (float)extent, (float)min_opacity, (float)max_screen_size, (float)size_thresh,
                                   P > 0 ? cl.data_ptr<uint8_t>() : nullptr, P > 0 ? sp.data_ptr<uint8_t>() : nullptr,
                                   P > 0 ? pr.data_ptr<uint8_t>() : nullptr,
                                   reinterpret_cast<uint32_t*>(counts.data_ptr<int32_t>()),
                                   c10::hip::getCurrentHIPStream(dev.index()).stream());
  check_rc(rc, "hgs_densify_masks");
  return {cl.to(at::kBool), sp.to(at::kBool), pr.to(at::kBool), counts};
}

// keeps the rows where keep[i] of every tensor in `tensors` (all [P, ...] fp32), order preserved: the
// pruning of all parameters and of both Adam moments with ONE index computation (one D2H read: the count)
std::vector<Tensor> compact_rows(const Tensor& keep, const std::vector<Tensor>& tensors) {
  at::NoGradGuard ng;
  const c10::Device dev = keep.device();
  if (!dev.is_cuda()) throw std::runtime_error("humangaussian_amd: tensors must live on a HIP device");
  DeviceSwitch guard(dev.index());
  for (const Tensor& t : tensors) {
    if (t.device() != dev) throw std::runtime_error("compact_rows: every tensor must live on the device of `keep`");
    if (t.scalar_type() != at::kFloat)
      throw std::runtime_error("compact_rows: fp32 tensors only (got " + std::string(c10::toString(t.scalar_type())) +
                               "): gather other dtypes with index_select on the returned row count");
    if (t.dim() < 1) throw std::runtime_error("compact_rows: tensors need a leading num_points dimension");
  }
  const Tensor k = keep.to(at::kByte).contiguous();
  const int64_t P = k.numel();
  hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
  Tensor idx = at::empty({P}, at::TensorOptions().dtype(at::kInt).device(dev));
  Tensor cnt = at::empty({1}, at::TensorOptions().dtype(at::kInt).device(dev));
  Tensor scratch = at::empty({(int64_t)hgs_compact_scratch_bytes((int32_t)P)}, at::TensorOptions().dtype(at::kByte).device(dev));
  check_rc(hgs_compact_index((int32_t)P, P > 0 ? k.data_ptr<uint8_t>() : nullptr, P > 0 ? idx.data_ptr<int32_t>() : nullptr,
                             reinterpret_cast<uint32_t*>(cnt.data_ptr<int32_t>()), scratch.data_ptr(), stream),
           "hgs_compact_index");
  const int64_t n = cnt.item<int32_t>();
  std::vector<Tensor> out;
  for (const Tensor& t : tensors) {
    if (t.size(0) != P) throw std::runtime_error("compact_rows: every tensor needs num_points rows");
    const Tensor src = f32c(t, dev, "tensor");
    const int64_t rf = P > 0 ? src.numel() / P : 1;
    std::vector<int64_t> shape = src.sizes().vec();
    shape[0] = n;
    Tensor dst = at::empty(shape, src.options());
    check_rc(hgs_gather_rows(n, (int32_t)std::max<int64_t>(rf, 1), n > 0 ? idx.data_ptr<int32_t>() : nullptr, fptr(src),
                             fptr_mut(dst), stream), "hgs_gather_rows");
    out.push_back(dst);
  }
  return out;
}

Tensor reanchor(const Tensor& vertices, const Tensor& faces, const Tensor& mapping_face, const Tensor& mapping_uvw,
                const Tensor& mapping_dist) {
  at::NoGradGuard ng;
  const c10::Device dev = vertices.device();
  if (!dev.is_cuda()) throw std::runtime_error("humangaussian_amd: tensors must live on a HIP device");
  DeviceSwitch guard(dev.index());
  const Tensor v = f32c(vertices, dev, "vertices"), uvw = f32c(mapping_uvw, dev, "mapping_uvw"), dist = f32c(mapping_dist, dev, "mapping_dist");
  const Tensor f = faces.to(at::kInt).contiguous(), mf = mapping_face.to(at::kInt).contiguous();
  if (f.device() != dev || mf.device() != dev) throw std::runtime_error("faces / mapping_face must live on the same device");
  const int64_t P = mf.numel();
  if (uvw.numel() != 3 * P || dist.numel() != P || v.numel() % 3 || f.numel() % 3) throw std::runtime_error("reanchor: inconsistent shapes");
  Tensor xyz = at::empty({P, 3}, v.options());
  check_rc(hgs_reanchor((int32_t)P, fptr(v), f.numel() ? f.data_ptr<int32_t>() : nullptr, P > 0 ? mf.data_ptr<int32_t>() : nullptr,
                        fptr(uvw), fptr(dist), fptr_mut(xyz), c10::hip::getCurrentHIPStream(dev.index()).stream()),
           "hgs_reanchor");
  return xyz;
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
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (dev < 0 || dev >= (int64_t)g_states.size()) throw std::runtime_error("bad device index");
    if (!g_states[dev]) return py::dict();             // nothing has run on this device yet: no state is created here
  }
  DevState& st = state_for((int)dev);
  std::lock_guard<std::mutex> lk(st.mu);
  py::dict d;
  d["capacity"] = st.last_capacity;
  d["tile_hint"] = st.last_hint;
  d["max_R"] = st.max_R;
  d["max_tile"] = st.max_tile;
  d["calls"] = st.calls;
  d["retries"] = st.retries;
  d["wait_ns"] = st.wait_ns;
  d["bwd_pairs_last"] = st.last_pairs;          // pair count the last backward sized its scratch by (0: worst case)
  d["bwd_scratch_last"] = st.last_scratch;
  {
    static const char* names[8] = {"fwd_pre", "fwd_launch", "fwd_shadow", "fwd_wait", "fwd_total", "bwd_pre", "bwd_launch", "bwd_total"};
    py::dict hn;
    for (int i = 0; i < 8; ++i) hn[names[i]] = st.host_ns[i].load();
    d["host_ns"] = hn;
  }
  py::dict e;
  for (const auto& kv : st.est)
    e[py::make_tuple(std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first))] =
        py::make_tuple(kv.second.capacity, kv.second.tile_hint);
  d["estimates"] = e;
  return d;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "torch binding of libhgs_rast.so (include/hgs_rast.h): autograd node, capacity logic, the one host wait";
  m.def("rasterize", &rasterize, py::call_guard<py::gil_scoped_release>());
  m.def("rasterize_batch", &rasterize_batch, py::call_guard<py::gil_scoped_release>());
  m.def("mark_visible", &mark_visible, py::call_guard<py::gil_scoped_release>());
  m.def("knn_mean_dist2", &knn_mean_dist2, py::arg("points"), py::arg("brute_force") = false,
        py::call_guard<py::gil_scoped_release>());
  m.def("reduce_view_packs", &reduce_view_packs, py::arg("gathered"), py::arg("acc_in") = py::none(),
        py::call_guard<py::gil_scoped_release>());
  m.def("pack_view_contribution", &pack_view_contribution, py::call_guard<py::gil_scoped_release>());
  m.def("reduce_view_packs_unpack", &reduce_view_packs_unpack, py::arg("gathered"), py::arg("acc_in") = py::none(),
        py::call_guard<py::gil_scoped_release>());
  m.def("set_packed_backward", &set_packed_backward);
  m.def("take_packed", &take_packed);
  m.def("densify_stats", &densify_stats, py::call_guard<py::gil_scoped_release>());
  m.def("densify_masks", &densify_masks, py::call_guard<py::gil_scoped_release>());
  m.def("compact_rows", &compact_rows, py::call_guard<py::gil_scoped_release>());
  m.def("reanchor", &reanchor, py::call_guard<py::gil_scoped_release>());
  m.def("set_stage_events", &set_stage_events);
  m.def("device_state", &device_state);
  m.def("abi_version", []() { return hgs_abi_version(); });
}
