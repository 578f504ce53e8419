// Plain C/C++ host program on the C ABI of include/kp2d.h — no Python, no torch: what a non-Python maintainer binds.
//   hipcc -O2 -Iinclude examples/c_abi_forward.cpp -Lnano-vs-slam_amd/csrc -lkp2d_hip -Wl,-rpath,$PWD/nano-vs-slam_amd/csrc -o /tmp/c_abi_forward
// Builds KP2DTiny-S (V2, 28 classes), fills every state-dict tensor from a tiny LCG keyed by the tensor's index,
// runs forward + post_processing + top-k on LCG frames and prints checksums; tests/test_c_abi_example.py rebuilds the
// same tensors in numpy, runs the Python host layer and compares.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kp2d.h"

#define CK(expr)                                                              \
  do {                                                                        \
    int rc_ = (expr);                                                         \
    if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, kp2d_last_error()); return 1; } \
  } while (0)
#define HK(expr)                                                              \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); return 1; } \
  } while (0)

// uniform in [0,1): 32-bit LCG (Numerical Recipes constants), top 24 bits
static float lcg(uint32_t& s) {
  s = s * 1664525u + 1013904223u;
  return (float)(s >> 8) * (1.0f / 16777216.0f);
}

static bool ends_with(const std::string& s, const char* suf) {
  const size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 2, H = 64, W = 96;
  kp2d_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = (int32_t)sizeof cfg;
  cfg.version = 2;
  const int dims[6] = {16, 32, 32, 64, 64, 128};
  for (int i = 0; i < 6; ++i) cfg.channel_dims[i] = dims[i];
  cfg.nfeatures = 32; cfg.n_classes = 28; cfg.num_clusters = 64; cfg.encoder_dim = 64; cfg.downsample = 2;
  cfg.leaky_relu = 1; cfg.global_descriptor = KP2D_GD_NETVLAD; cfg.upscale_method = KP2D_UP_PIXELSHUFFLE;
  kp2d_model* m = nullptr;
  CK(kp2d_create(&cfg, &m));
  const int nw = kp2d_num_weights(m);
  for (int i = 0; i < nw; ++i) {
    const char* key; int64_t shape[4]; int nd;
    CK(kp2d_weight_info(m, i, &key, shape, &nd));
    size_t n = 1;
    for (int d = 0; d < nd; ++d) n *= (size_t)shape[d];
    size_t fan = 1;
    for (int d = 1; d < nd; ++d) fan *= (size_t)shape[d];
    std::vector<float> w(n);
    uint32_t s = 12345u + 977u * (uint32_t)i;
    const std::string k(key);
    const bool positive = ends_with(k, "running_var") || (ends_with(k, "bn.weight"));
    const float amp = nd > 1 ? 2.0f * sqrtf(3.0f / (float)fan) : 0.4f;     // centred uniform with std sqrt(2/fan_in)
    for (size_t e = 0; e < n; ++e) w[e] = positive ? 0.5f + lcg(s) : (lcg(s) - 0.5f) * amp;
    CK(kp2d_set_weight(m, key, w.data(), shape, nd));
  }
  CK(kp2d_finalize_weights(m));

  const int Hc = H / 4, Wc = W / 4, H2 = H / 2, W2 = W / 2;
  const size_t nx = (size_t)B * 3 * H * W, ncell = (size_t)Hc * Wc, vd = kp2d_vlad_dim(m, H, W);
  std::vector<float> hx(nx);
  uint32_t s = 777u;
  for (size_t e = 0; e < nx; ++e) hx[e] = lcg(s) * 2.0f - 1.0f;
  float *x, *score, *shift, *feat, *seg, *vlad, *score_o, *coord, *desc, *val;
  int64_t* seg_ids; int32_t *idx, *cnt; void* ws;
  const int k = 300;
  HK(hipMalloc((void**)&x, nx * 4));
  HK(hipMalloc((void**)&score, B * ncell * 4)); HK(hipMalloc((void**)&shift, B * 2 * ncell * 4));
  HK(hipMalloc((void**)&feat, (size_t)B * 32 * H2 * W2 * 4)); HK(hipMalloc((void**)&seg, (size_t)B * 28 * H2 * W2 * 4));
  HK(hipMalloc((void**)&vlad, B * vd * 4));
  HK(hipMalloc((void**)&score_o, B * ncell * 4)); HK(hipMalloc((void**)&coord, B * 2 * ncell * 4));
  HK(hipMalloc((void**)&desc, B * 32 * ncell * 4)); HK(hipMalloc((void**)&seg_ids, (size_t)B * H2 * W2 * 8));
  HK(hipMalloc((void**)&idx, (size_t)B * k * 4)); HK(hipMalloc((void**)&val, (size_t)B * k * 4)); HK(hipMalloc((void**)&cnt, B * 4));
  const size_t wsb = kp2d_workspace_bytes(m, B, H, W);
  HK(hipMalloc(&ws, wsb));
  HK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  HK(hipStreamCreate(&st));
  CK(kp2d_forward(m, x, B, H, W, KP2D_FWD_EVAL, score, shift, feat, seg, vlad, nullptr, ws, wsb, st));
  CK(kp2d_post(m, score, shift, feat, seg, B, H, W, Hc, Wc, 32, H2, W2, 28, H2, W2, score_o, coord, desc, seg_ids, 0, st));
  CK(kp2d_select_topk(score_o, B, (int)ncell, k, -INFINITY, idx, val, cnt, st));
  HK(hipStreamSynchronize(st));

  auto sum = [&](const float* dptr, size_t n, double* s1, double* s2) -> int {
    std::vector<float> h(n);
    if (hipMemcpy(h.data(), dptr, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    *s1 = 0; *s2 = 0;
    for (float v : h) { *s1 += v; *s2 += (double)v * v; }
    return 0;
  };
  double a, b2;
  if (sum(score, B * ncell, &a, &b2)) return 1;
  printf("score_sum %.6f score_sq %.6f\n", a, b2);
  if (sum(shift, B * 2 * ncell, &a, &b2)) return 1;
  printf("shift_sum %.6f shift_sq %.6f\n", a, b2);
  if (sum(vlad, B * vd, &a, &b2)) return 1;
  printf("vlad_sum %.6f vlad_sq %.6f\n", a, b2);
  if (sum(desc, B * 32 * ncell, &a, &b2)) return 1;
  printf("desc_sum %.6f desc_sq %.6f\n", a, b2);
  if (sum(coord, B * 2 * ncell, &a, &b2)) return 1;
  printf("coord_sum %.4f\n", a);
  std::vector<int32_t> hidx((size_t)B * k);
  HK(hipMemcpy(hidx.data(), idx, hidx.size() * 4, hipMemcpyDeviceToHost));
  long isum = 0;
  for (int32_t v : hidx) isum += v;
  printf("topk_first %d topk_idx_sum %ld\n", hidx[0], isum);
  kp2d_destroy(m);
  return 0;
}
