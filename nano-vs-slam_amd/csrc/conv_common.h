// Shared pieces of the convolution kernels: activation selector and the accumulator epilogue
// (per-channel affine, activation, and the store modes of kp2d_kernels.h::Store).
#pragma once
#include "kp2d_kernels.h"

// Timing ablations (KP2D_DBG bits, DESIGN.md "phase ablations") exist only in a -DKP2D_ABLATE build: as run-time
// tests they cost a scalar branch per staged granule in the production kernel.
#ifdef KP2D_ABLATE
#define KP2D_DBG_ON(bit) ((a.dbg & (bit)) != 0)
#else
#define KP2D_DBG_ON(bit) false
#endif

namespace kp2d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int TILE = 16;

// kept out of line: it is inlined 64x per thread otherwise and bloats the kernels past the instruction cache
__device__ __attribute__((noinline)) float act_apply(float v, int act, int ch) {
  switch (act) {
    case ACT_LEAKY: return v >= 0.f ? v : v * 0.01f;
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case ACT_TANH: return tanhf(v);
    case ACT_SIGMOID0_TANH: return ch == 0 ? 1.f / (1.f + expf(-v)) : tanhf(v);
    case ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    default: return v;
  }
}

// Exact-form GELU 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. a
// GELU error below |x| * 1e-7): 15 instructions inline, against ~60 plus a call for libm's erff through act_apply —
// the MixFFN's GELU layer spent more time in its epilogue than in its matrix product.
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);
  const float erf_abs = fmaf(-p * t, e, 1.0f);
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// One pixel of the frame front-end (src/evaluation/visual_odometry.py:77-87: kornia.image_to_tensor / 255, kornia bilinear
// resize with align_corners=False, .sub(0.5).mul(2)) from uint8 HWC frames: ONE definition for the three kernels that compute
// it (post.hip preprocess_kernel, conv3x3.hip conv1a_u8_kernel, conv3x3_f16.hip conv1a_mfma_kernel<true>), with floating-point
// contraction off — left to the compiler, the same source line was fused into FMAs in one kernel and not in another, and
// "forward_frames(frames) == forward(preprocess(frames))" held by luck.
__device__ __forceinline__ void frame_pixel(const unsigned char* __restrict__ img, int Hs, int Ws, int H, int W, int y, int x, float v[3]) {
#pragma clang fp contract(off)
  if (Hs == H && Ws == W) {
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = (float)img[((size_t)y * Ws + x) * 3 + c] / 255.0f;
  } else {
    const float sy = fmaxf(((float)y + 0.5f) * ((float)Hs / (float)H) - 0.5f, 0.f);
    const float sx = fmaxf(((float)x + 0.5f) * ((float)Ws / (float)W) - 0.5f, 0.f);
    const int y0 = min((int)sy, Hs - 1), x0 = min((int)sx, Ws - 1);
    const int y1 = min(y0 + 1, Hs - 1), x1 = min(x0 + 1, Ws - 1);
    const float wy = sy - (float)y0, wx = sx - (float)x0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float a00 = (float)img[((size_t)y0 * Ws + x0) * 3 + c] / 255.0f, a01 = (float)img[((size_t)y0 * Ws + x1) * 3 + c] / 255.0f;
      const float a10 = (float)img[((size_t)y1 * Ws + x0) * 3 + c] / 255.0f, a11 = (float)img[((size_t)y1 * Ws + x1) * 3 + c] / 255.0f;
      const float top = (1.f - wx) * a00 + wx * a01, bot = (1.f - wx) * a10 + wx * a11;
      v[c] = (1.f - wy) * top + wy * bot;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) v[c] = (v[c] - 0.5f) * 2.0f;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// fp32 pair -> (hi, lo) fp16 pairs with hi = rn(x), lo = rn(x - hi): v_cvt_pk_f16_f32, then v_fma_mixlo_f16 /
// v_fma_mixhi_f16 (read the fp16 half directly, form the exact fp32 difference x - float(hi) and round it into one half
// of the destination): three instructions per two values
__device__ __forceinline__ void split2(float x, float y, f16x2& hi, f16x2& lo) {
  const f32x2 v = {x, y};
  hi = __builtin_convertvector(v, f16x2);
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(x));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(y));
  lo = __builtin_bit_cast(f16x2, l);
}

}  // namespace kp2d
