// Probe (measurement tool, not product code): numerics and rate of v_mfma_f32_32x32x16_f16 on gfx950,
// to decide whether a split-fp16 (hi + lo) operand scheme can reproduce fp32 convolutions.
//   1. operand lane map check with exact integers
//   2. are fp16 SUBNORMAL operands honoured or flushed?
//   3. accuracy of x*w ~ xh*wh + xh*wl + xl*wh (3 MFMAs, one accumulator) vs fp64, next to the exact-fp32 MFMA
//   4. issue rate: cycles per MFMA, one wave per SIMD
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f16_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// C[32x32] = A[32xK] * B[Kx32]; A row-major [32][K], B row-major [K][32], halves
__global__ void gemm_f16(const _Float16* A, const _Float16* B, float* C, int K) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[r * K + k0 + 8 * h + j]; b[j] = B[(k0 + 8 * h + j) * 32 + r]; }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

// split scheme: A,B fp32; xs = activation pre-scale (power of 2), ws = weight pre-scale
__global__ void gemm_split(const float* A, const float* B, float* C, int K, float xs, float ws, int mode) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    half8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) {
      const float x = A[r * K + k0 + 8 * h + j] * xs;
      const float w = B[(k0 + 8 * h + j) * 32 + r] * ws;
      const _Float16 xh = (_Float16)x, wh = (_Float16)w;
      ah[j] = xh; al[j] = (_Float16)(x - (float)xh);
      bh[j] = wh; bl[j] = (_Float16)(w - (float)wh);
    }
    if (mode >= 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    if (mode >= 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  }
  const float inv = 1.f / (xs * ws);
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i] * inv;
}

__global__ void gemm_f32(const float* A, const float* B, float* C, int K) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc = {0};
  for (int k0 = 0; k0 < K; k0 += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k0 + h], B[(k0 + h) * 32 + r], acc, 0, 0, 0);
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

template <int WHICH>
__global__ void rate(float* out, long long* cyc, int iters) {
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f - threadIdx.x * 0.002f); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (WHICH == 0) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    } else {
      d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d3, 0, 0, 0);
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  for (int i = 0; i < 4; ++i) s += d0[i] + d1[i] + d2[i] + d3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int K = 576;
  std::vector<float> A(32 * K), B(K * 32), C(1024);
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
  // 1. lane map with integers (asymmetric B)
  {
    std::vector<_Float16> Ah(32 * 16), Bh(16 * 32);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) Ah[i * 16 + k] = (_Float16)((i * 3 + k) % 7 - 3);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) Bh[k * 32 + j] = (_Float16)((k * 5 + 2 * j) % 9 - 4);
    _Float16 *hA, *hB; hipMalloc(&hA, Ah.size() * 2); hipMalloc(&hB, Bh.size() * 2);
    hipMemcpy(hA, Ah.data(), Ah.size() * 2, hipMemcpyHostToDevice); hipMemcpy(hB, Bh.data(), Bh.size() * 2, hipMemcpyHostToDevice);
    gemm_f16<<<1, 64>>>(hA, hB, dC, 16);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      float ref = 0; for (int k = 0; k < 16; ++k) ref += (float)Ah[i * 16 + k] * (float)Bh[k * 32 + j];
      if (ref != C[i * 32 + j]) ++bad;
    }
    printf("[1] lane map 32x32x16 f16 (bf16 map assumed): %d mismatches of 1024\n", bad);
    // 2. subnormal operands: A = 2^-20 (subnormal in fp16: min normal 2^-14), B = 1
    for (auto& v : Ah) v = (_Float16)ldexpf(1.f, -20);
    for (auto& v : Bh) v = (_Float16)1.f;
    hipMemcpy(hA, Ah.data(), Ah.size() * 2, hipMemcpyHostToDevice); hipMemcpy(hB, Bh.data(), Bh.size() * 2, hipMemcpyHostToDevice);
    gemm_f16<<<1, 64>>>(hA, hB, dC, 16);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    printf("[2] subnormal A (2^-20) x 1.0, K=16: got %.6e, exact %.6e -> %s\n", C[0], 16 * ldexp(1.0, -20),
           C[0] == 16 * ldexpf(1.f, -20) ? "subnormals HONOURED" : "subnormals FLUSHED/changed");
    for (auto& v : Ah) v = (_Float16)1.f;
    for (auto& v : Bh) v = (_Float16)ldexpf(1.f, -22);
    hipMemcpy(hA, Ah.data(), Ah.size() * 2, hipMemcpyHostToDevice); hipMemcpy(hB, Bh.data(), Bh.size() * 2, hipMemcpyHostToDevice);
    gemm_f16<<<1, 64>>>(hA, hB, dC, 16);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    printf("[2] 1.0 x subnormal B (2^-22), K=16: got %.6e, exact %.6e\n", C[0], 16 * ldexp(1.0, -22));
  }
  // 3. accuracy
  srand(1);
  auto rnd = []() { float u = 0; for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX; return u - 6.f; };
  for (int trial = 0; trial < 3; ++trial) {
    const float xmag = trial == 0 ? 1.f : (trial == 1 ? 0.02f : 8.f);
    for (auto& v : A) { float g = rnd() * xmag; v = g < 0 ? g * 0.01f : g; }          // post-LeakyReLU-like activations
    for (auto& v : B) v = rnd() * 0.059f;                                              // ~N(0, 2/576)
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(1024);
    double refmax = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * 32 + j];
      ref[i * 32 + j] = s; refmax = fmax(refmax, fabs(s));
    }
    auto err = [&](const char* name) {
      hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
      double e = 0, rms = 0; for (int i = 0; i < 1024; ++i) { double d = fabs(C[i] - ref[i]); e = fmax(e, d); rms += d * d; }
      printf("    %-46s max abs err %.3e  rms %.3e  (|ref|max %.2f)\n", name, e, sqrt(rms / 1024), refmax);
    };
    printf("[3] activations scale %.2f, K=%d\n", xmag, K);
    gemm_f32<<<1, 64>>>(dA, dB, dC, K); err("exact fp32 MFMA 32x32x2");
    gemm_split<<<1, 64>>>(dA, dB, dC, K, 1.f, 1.f, 1); err("fp16 hi only (1 MFMA)");
    gemm_split<<<1, 64>>>(dA, dB, dC, K, 1.f, 1.f, 3); err("split 3 MFMA, no scaling");
    gemm_split<<<1, 64>>>(dA, dB, dC, K, 1.f, 2048.f, 3); err("split 3 MFMA, weights x2^11");
    gemm_split<<<1, 64>>>(dA, dB, dC, K, 256.f, 2048.f, 3); err("split 3 MFMA, weights x2^11, acts x2^8");
    gemm_split<<<1, 64>>>(dA, dB, dC, K, 1.f, 2048.f, 2); err("split 2 MFMA (xh*wh + xh*wl), weights x2^11");
  }
  // 4. rate
  {
    float* o; long long* c; hipMalloc(&o, 256 * 1024 * 4); hipMalloc(&c, 1024 * 8);
    std::vector<long long> cy(1024);
    const int iters = 20000;
    for (int which = 0; which < 2; ++which) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (which == 0) rate<0><<<256, 256>>>(o, c, iters); else rate<1><<<256, 256>>>(o, c, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(cy.data(), c, 256 * 8, hipMemcpyDeviceToHost);
      const double flop = 256.0 * 4 * iters * 4 * (which == 0 ? 32.0 * 32 * 16 * 2 : 16.0 * 16 * 32 * 2);
      printf("[4] %s: %.1f cycles/MFMA/SIMD (clock64), %.1f TFLOP/s over 256 CUs x 4 waves (%.3f ms)\n",
             which == 0 ? "32x32x16 f16" : "16x16x32 f16", (double)cy[0] / (iters * 4.0), flop / (ms * 1e-3) / 1e12, ms);
    }
  }
  return 0;
}
