// Host-side cost of HIP launch primitives on this box (design input for the frame pipeline).
// hipcc --offload-arch=gfx950 -O2 tools/hip_overheads.cpp -o /tmp/hip_overheads
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void tiny(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  float *d; CK(hipMalloc(&d, 1 << 20));
  float *h; CK(hipHostMalloc(&h, 1 << 20, hipHostMallocDefault));
  hipStream_t s0, s1, s2; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, e2; CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  const int K = 2000;
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s0, d, 16384);
  CK(hipStreamSynchronize(s0));
  double t0 = now();
  for (int i = 0; i < K; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s0, d, 16384);
  double t1 = now(); CK(hipStreamSynchronize(s0)); double t2 = now();
  printf("eager launch: host %.2f us/launch, total %.2f us/launch\n", (t1 - t0) / K, (t2 - t0) / K);
  t0 = now();
  for (int i = 0; i < K; ++i) { CK(hipEventRecord(e0, s0)); CK(hipStreamWaitEvent(s1, e0, 0)); }
  t1 = now(); CK(hipDeviceSynchronize()); t2 = now();
  printf("record+wait pair: host %.2f us, total %.2f us\n", (t1 - t0) / K, (t2 - t0) / K);
  t0 = now();
  for (int i = 0; i < K; ++i) { hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s0, d, 16384); CK(hipEventRecord(e0, s0)); CK(hipStreamWaitEvent(s1, e0, 0)); hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s1, d + 65536, 16384); CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s0, e1, 0)); }
  t1 = now(); CK(hipDeviceSynchronize()); t2 = now();
  printf("ping-pong 2 kernels + 2 record/wait: host %.2f us, total %.2f us\n", (t1 - t0) / K, (t2 - t0) / K);
  t0 = now();
  for (int i = 0; i < K; ++i) CK(hipMemcpyAsync(h, d, 48000, hipMemcpyDeviceToHost, s0));
  t1 = now(); CK(hipStreamSynchronize(s0)); t2 = now();
  printf("D2H 48KB pinned: host %.2f us, total %.2f us\n", (t1 - t0) / K, (t2 - t0) / K);
  t0 = now();
  for (int i = 0; i < K; ++i) CK(hipMemcpyAsync(d + 131072, d, 2073600 / 8, hipMemcpyDeviceToDevice, s0));
  t1 = now(); CK(hipStreamSynchronize(s0)); t2 = now();
  printf("D2D 256KB: host %.2f us, total %.2f us\n", (t1 - t0) / K, (t2 - t0) / K);
  // linear graph of N kernels
  for (int N : {1, 6, 8}) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s0, d, 16384);
    CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s0));
    CK(hipStreamSynchronize(s0));
    t0 = now();
    for (int i = 0; i < K; ++i) CK(hipGraphLaunch(ge, s0));
    t1 = now(); CK(hipStreamSynchronize(s0)); t2 = now();
    printf("linear graph %d kernels: host %.2f us/graph, total %.2f us/graph\n", N, (t1 - t0) / K, (t2 - t0) / K);
  }
  {  // forked graph: s0 -> {s1: 2 kernels, s2: 1 kernel} -> join
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s0, d, 16384);
    CK(hipEventRecord(e0, s0)); CK(hipStreamWaitEvent(s1, e0, 0)); CK(hipStreamWaitEvent(s2, e0, 0));
    hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s1, d + 65536, 16384);
    hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s1, d + 65536, 16384);
    hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s2, d + 131072, 16384);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s0, d, 16384);
    CK(hipEventRecord(e1, s1)); CK(hipEventRecord(e2, s2)); CK(hipStreamWaitEvent(s0, e1, 0)); CK(hipStreamWaitEvent(s0, e2, 0));
    CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s0));
    CK(hipStreamSynchronize(s0));
    t0 = now();
    for (int i = 0; i < K; ++i) CK(hipGraphLaunch(ge, s0));
    t1 = now(); CK(hipStreamSynchronize(s0)); t2 = now();
    printf("forked graph 7 kernels/3 branches: host %.2f us/graph, total %.2f us/graph\n", (t1 - t0) / K, (t2 - t0) / K);
  }
  {  // kernel writing straight to pinned host memory
    float *hd; CK(hipHostGetDevicePointer((void **)&hd, h, 0));
    t0 = now();
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL(tiny, dim3(47), dim3(256), 0, s0, hd, 12000);
    t1 = now(); CK(hipStreamSynchronize(s0)); t2 = now();
    printf("kernel RMW 48KB in pinned host mem: host %.2f us, total %.2f us\n", (t1 - t0) / K, (t2 - t0) / K);
  }
  return 0;
}
