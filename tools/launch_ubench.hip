// host cost of N direct kernel launches vs one hipGraphLaunch of the same N-kernel graph (ROCm, one stream)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { char pad[1100]; };
__global__ void k(Big b, float* p, int n) { if (threadIdx.x == 0 && blockIdx.x == 0 && n < 0) p[0] = b.pad[0]; }
int main() {
  float* p; hipMalloc(&p, 4); hipStream_t s; hipStreamCreate(&s); Big b{};
  for (int N : {3, 5, 8}) {
    auto direct = [&]() { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, s, b, p, i); };
    for (int i = 0; i < 200; ++i) direct();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; ++i) { direct(); if ((i & 63) == 63) hipStreamSynchronize(s); }
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    hipGraph_t g; hipGraphExec_t e;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal); direct(); hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    for (int i = 0; i < 200; ++i) hipGraphLaunch(e, s);
    hipStreamSynchronize(s);
    auto t2 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; ++i) { hipGraphLaunch(e, s); if ((i & 63) == 63) hipStreamSynchronize(s); }
    auto t3 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    printf("N=%d direct %.2f us per batch (%.2f per launch), graph %.2f us per launch\n", N,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000, std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000 / N,
           std::chrono::duration<double, std::micro>(t3 - t2).count() / 2000);
  }
  return 0;
}
