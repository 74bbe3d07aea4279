// Store bandwidth for the access shape of the level kernels: every wavefront owns a strip of COLS
// texels (12 bytes each) and writes it row by row (row stride 1920 texels), ROWS rows per wavefront,
// optionally with a dependent-ALU delay between rows.
// Build: hipcc --offload-arch=gfx950 -O3 tools/store_pattern.cpp -o tools/bin/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
struct T3 { float x, y, z; };
template <int PER, bool NT>
__global__ __launch_bounds__(64) void k_strip(T3 *__restrict__ t, int w, int h, int cols, int rows, int spin) {
  const int strip = blockIdx.x, seg = blockIdx.y, frame = blockIdx.z, lane = threadIdx.x;
  T3 *base = t + (size_t)frame * w * h;
  float v = lane;
  for (int r = seg * rows; r < min(seg * rows + rows, h); ++r) {
    for (int s = 0; s < spin; ++s) v = v * 1.0001f + 0.5f;
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int x = strip * cols + p * 64 + lane;
      if (p * 64 + lane < cols && x < w) {
        T3 *q = base + (size_t)r * w + x;
        if (NT) { __builtin_nontemporal_store(v, &q->x); __builtin_nontemporal_store(v, &q->y); __builtin_nontemporal_store(v, &q->z); }
        else { *q = T3{v, v, v}; }
      }
    }
  }
}
int main() {
  const int W = 1920, H = 1080, F = 16;
  const size_t bytes = (size_t)F * W * H * 12;
  T3 *d; (void)hipMalloc(&d, bytes);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  auto run = [&](const char *name, auto f) {
    for (int i = 0; i < 3; ++i) f();
    (void)hipEventRecord(a); for (int r = 0; r < 10; ++r) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-52s %8.1f us  %5.2f TB/s\n", name, ms * 100, bytes / (ms * 1e-4) / 1e12);
  };
  char nm[128];
  for (int spin : {0}) for (int rows : {120}) for (int cols : {64, 96, 104, 112, 114, 116, 120, 128}) {
    const int per = (cols + 63) / 64, ns = (W + cols - 1) / cols, nseg = (H + rows - 1) / rows;
    for (int nt = 0; nt < 2; ++nt) {
      snprintf(nm, sizeof nm, "cols %3d rows %3d spin %3d %s (%d waves)", cols, rows, spin, nt ? "nt" : "  ", ns * nseg * F);
      run(nm, [&] {
        dim3 g(ns, nseg, F);
        if (per == 1) { if (nt) hipLaunchKernelGGL((k_strip<1, true>), g, dim3(64), 0, 0, d, W, H, cols, rows, spin); else hipLaunchKernelGGL((k_strip<1, false>), g, dim3(64), 0, 0, d, W, H, cols, rows, spin); }
        else if (per == 2) { if (nt) hipLaunchKernelGGL((k_strip<2, true>), g, dim3(64), 0, 0, d, W, H, cols, rows, spin); else hipLaunchKernelGGL((k_strip<2, false>), g, dim3(64), 0, 0, d, W, H, cols, rows, spin); }
        else { if (nt) hipLaunchKernelGGL((k_strip<4, true>), g, dim3(64), 0, 0, d, W, H, cols, rows, spin); else hipLaunchKernelGGL((k_strip<4, false>), g, dim3(64), 0, 0, d, W, H, cols, rows, spin); }
      });
    }
  }
  return 0;
}
