// Host-side cost of gklt_pipeline_step without Python: C2-sized synthetic frames (random bytes),
// device-resident, 4000 features.  Prints host enqueue time and total time per step.
// build: hipcc -O2 -std=c++17 tools/pipe_host.cpp -Iinclude -Lgslam_amd -lgklt -Wl,-rpath,$PWD/gslam_amd -o tools/bin/pipe_host
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gklt.h"

#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, gklt_last_error()); return 1; } } while (0)

int main() {
  const int W = getenv("PW") ? atoi(getenv("PW")) : 1920, H = getenv("PH") ? atoi(getenv("PH")) : 1080, N = getenv("PN") ? atoi(getenv("PN")) : 4000; constexpr int NF = 8;
  gklt_params p;
  CK(gklt_default_params(&p));
  p.window_width = p.window_height = 7;
  p.nPyramidLevels = 4;
  p.subsampling = 2;
  CK(gklt_update_tc_border(&p));
  gklt_ctx *ctx;
  CK(gklt_ctx_create(0, &ctx));
  CK(gklt_ctx_set_params(ctx, &p));
  std::vector<uint8_t> img((size_t)W * H);
  void *dframes[NF];
  for (int f = 0; f < NF; ++f) {
    for (auto &b : img) b = (uint8_t)(rand() & 0xff);
    CK(gklt_device_malloc(ctx, img.size(), &dframes[f]));
    CK(gklt_copy_to_device(ctx, dframes[f], img.data(), img.size()));
  }
  std::vector<float> x(N), y(N);
  std::vector<int> v(N, 0);
  for (int i = 0; i < N; ++i) { x[i] = 100 + (i % 80) * 21.5f; y[i] = 100 + (i / 80) * 17.5f; }
  gklt_pipeline *pipe;
  CK(gklt_pipeline_create(ctx, W, H, N, &pipe));
  CK(gklt_pipeline_upload_features(pipe, N, x.data(), y.data(), v.data()));
  CK(gklt_pipeline_start(pipe, (uint8_t *)dframes[0], (uint8_t *)dframes[1], W, 1));
  void *out;
  CK(gklt_host_malloc_pinned(ctx, 3 * sizeof(float) * N, &out));
  int t = 1;
  for (int rep = 0; rep < 4; ++rep) {
    const int K = 2000;
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < K; ++k, ++t) {
      if (t % 32 == 0) CK(gklt_pipeline_upload_features(pipe, N, x.data(), y.data(), v.data()));
      CK(gklt_pipeline_step(pipe, (uint8_t *)dframes[(t + 1) % NF], W, 1, out));
    }
    auto t1 = std::chrono::steady_clock::now();
    CK(gklt_pipeline_sync(pipe));
    auto t2 = std::chrono::steady_clock::now();
    printf("C host: enqueue %.1f us/step, total %.1f us/step\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / K,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / K);
  }
  gklt_pipeline_destroy(pipe);
  gklt_ctx_destroy(ctx);
  return 0;
}
