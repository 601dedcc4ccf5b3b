// Launch cost on this runtime + part, for the question "would HIP-graph replay of the prover's launch chains help?" (round 5).
// A chain of N small DEPENDENT kernels (what a commitment chain or a round's field work is) is issued three ways:
//   stream  : N hipLaunchKernelGGL calls on one stream
//   graph   : the same chain captured once (stream capture), instantiated, replayed with hipGraphLaunch
//   graph+u : replay after hipGraphExecKernelNodeSetParams on every node (what a chain whose by-value arguments change per proof needs)
// For each: host time until the last call returns (enqueue cost), GPU time first-kernel-start .. last-kernel-end by events, and wall time to completion.
// Kernel sizes: "tiny" (one block, ~2 us) and "small" (1024 blocks x 256 lanes of a short loop, ~8-10 us: the prover's field kernels).
// Also: T host threads launching tiny kernels on T streams at once (aggregate launches per second: the lockstep prover's bottleneck).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <chrono>
#include <thread>
#include <atomic>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void __launch_bounds__(256) k_step(uint32_t* buf, uint32_t n, uint32_t iters, uint32_t salt) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t v = buf[i] + salt;
  for (uint32_t t = 0; t < iters; ++t) v = v * 1664525u + 1013904223u;
  buf[i] = v;
}

struct Res { double host_us, gpu_us, wall_us; };

static Res run_stream(hipStream_t s, uint32_t* d, uint32_t n, uint32_t iters, int N, hipEvent_t e0, hipEvent_t e1) {
  CK(hipStreamSynchronize(s));
  const double t0 = now_us();
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_step, dim3((n + 255) / 256), dim3(256), 0, s, d, n, iters, (uint32_t)i);
  CK(hipEventRecord(e1, s));
  const double t1 = now_us();
  CK(hipStreamSynchronize(s));
  const double t2 = now_us();
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return {t1 - t0, ms * 1e3, t2 - t0};
}

int main(int argc, char** argv) {
  const int reps = 30;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint32_t* d; CK(hipMalloc(&d, 1 << 22)); CK(hipMemset(d, 1, 1 << 22));
  struct Shape { const char* name; uint32_t n, iters; } shapes[] = {{"tiny (1 block)", 256, 64}, {"small (1024 blocks)", 1u << 18, 256}};
  for (const Shape& sh : shapes) for (int N : {10, 40, 200}) {
    // stream launches
    Res a{0, 0, 0};
    for (int r = 0; r < reps + 3; ++r) { Res q = run_stream(s, d, sh.n, sh.iters, N, e0, e1); if (r >= 3) { a.host_us += q.host_us; a.gpu_us += q.gpu_us; a.wall_us += q.wall_us; } }
    // graph: capture once
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_step, dim3((sh.n + 255) / 256), dim3(256), 0, s, d, sh.n, sh.iters, (uint32_t)i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
    Res b{0, 0, 0}, c{0, 0, 0};
    for (int upd = 0; upd < 2; ++upd) for (int r = 0; r < reps + 3; ++r) {
      CK(hipStreamSynchronize(s));
      const double t0 = now_us();
      if (upd) {
        for (size_t i = 0; i < nn; ++i) {
          uint32_t salt = (uint32_t)(r * 1000 + i); uint32_t nv = sh.n, it = sh.iters; void* args[4] = {&d, &nv, &it, &salt};
          hipKernelNodeParams p{}; p.func = (void*)k_step; p.gridDim = dim3((sh.n + 255) / 256); p.blockDim = dim3(256); p.sharedMemBytes = 0; p.kernelParams = args; p.extra = nullptr;
          CK(hipGraphExecKernelNodeSetParams(ge, nodes[i], &p));
        }
      }
      CK(hipEventRecord(e0, s));
      CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      const double t1 = now_us();
      CK(hipStreamSynchronize(s));
      const double t2 = now_us();
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      Res& o = upd ? c : b;
      if (r >= 3) { o.host_us += t1 - t0; o.gpu_us += ms * 1e3; o.wall_us += t2 - t0; }
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    printf("%-20s N=%3d | stream: host %7.1f us gpu %7.1f wall %7.1f | graph: host %7.1f gpu %7.1f wall %7.1f | graph+setparams: host %7.1f gpu %7.1f wall %7.1f  (per kernel: %.2f / %.2f / %.2f us of GPU time)\n",
           sh.name, N, a.host_us / reps, a.gpu_us / reps, a.wall_us / reps, b.host_us / reps, b.gpu_us / reps, b.wall_us / reps, c.host_us / reps, c.gpu_us / reps, c.wall_us / reps,
           a.gpu_us / reps / N, b.gpu_us / reps / N, c.gpu_us / reps / N);
    fflush(stdout);
  }
  // aggregate launch rate from T threads, one stream each
  for (int T : {1, 2, 4, 8}) {
    std::vector<hipStream_t> st(T); for (auto& x : st) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    const int per = 2000; std::atomic<int> go{0};
    std::vector<std::thread> th; std::vector<double> host(T, 0.0);
    const double t0 = now_us();
    for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
      (void)hipSetDevice(0);
      const double a = now_us();
      for (int i = 0; i < per; ++i) hipLaunchKernelGGL(k_step, dim3(1), dim3(256), 0, st[t], d + 256 * t, 256u, 16u, (uint32_t)i);
      host[t] = now_us() - a;
      (void)hipStreamSynchronize(st[t]);
    });
    for (auto& x : th) x.join();
    const double wall = now_us() - t0;
    double hmax = 0; for (double h : host) hmax = h > hmax ? h : hmax;
    printf("threads %d x %d tiny launches: enqueue %.2f us per launch per thread, aggregate %.2f us per launch to completion (%.0f K launches/s)\n", T, per, hmax / per, wall / (per * T), per * T / wall * 1e3);
    for (auto& x : st) CK(hipStreamDestroy(x));
  }
  return 0;
}
