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
