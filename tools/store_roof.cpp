// Store-bandwidth ceiling for the level-0 kernel's output pattern: 16 frames x 1080 x 1920 texels of
// 12 bytes (+ a quarter-size fp32 plane), written with the access shapes the level kernels use.
// Build: hipcc --offload-arch=gfx950 -O3 tools/store_roof.cpp -o tools/bin/store_roof
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct T3 { float x, y, z; };
template <bool NT>
__global__ void k_tex(T3 *__restrict__ t, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    if (NT) { __builtin_nontemporal_store(v, &t[i].x); __builtin_nontemporal_store(v + 1, &t[i].y); __builtin_nontemporal_store(v + 2, &t[i].z); }
    else { t[i] = T3{v, v + 1, v + 2}; }
  }
}
__global__ void k_f4(float4 *__restrict__ t, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) t[i] = float4{v, v, v, v};
}
__global__ void k_read(const uint4 *__restrict__ s, size_t n, unsigned *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < n; i += step) { uint4 q = s[i]; acc += q.x ^ q.y ^ q.z ^ q.w; }
  if (acc == 0x12345678u) *out = acc;
}
int main() {
  const size_t npx = (size_t)16 * 1080 * 1920, bytes = npx * 12;
  T3 *d; unsigned *o;
  hipMalloc(&d, bytes * 2); hipMalloc(&o, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](const char *name, auto f, double bts) {
    for (int w = 0; w < 3; ++w) f();
    hipEventRecord(a); for (int r = 0; r < 10; ++r) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-34s %8.1f us  %6.2f TB/s\n", name, ms * 100, bts / (ms * 1e-4) / 1e12);
  };
  for (int blocks : {2048, 8192, 32768}) {
    printf("grid %d x 256\n", blocks);
    run("texel 12 B stores, nontemporal", [&] { hipLaunchKernelGGL(k_tex<true>, dim3(blocks), dim3(256), 0, 0, d, npx, 1.f); }, (double)bytes);
    run("texel 12 B stores", [&] { hipLaunchKernelGGL(k_tex<false>, dim3(blocks), dim3(256), 0, 0, d, npx, 1.f); }, (double)bytes);
    run("float4 stores", [&] { hipLaunchKernelGGL(k_f4, dim3(blocks), dim3(256), 0, 0, (float4 *)d, bytes / 16, 1.f); }, (double)bytes);
    run("uint4 reads", [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, (const uint4 *)d, bytes / 16, o); }, (double)bytes);
  }
  run("hipMemsetAsync", [&] { hipMemsetAsync(d, 0, bytes, 0); }, (double)bytes);
  run("hipMemcpy D2D (r+w)", [&] { hipMemcpyAsync((char *)d + bytes, d, bytes, hipMemcpyDeviceToDevice, 0); }, 2.0 * bytes);
  return 0;
}
