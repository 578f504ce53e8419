// LightGlue matcher kernels (reference: lightglue/lightglue.py; inference path, flash = False, no early stopping /
// point pruning).  Descriptor width D <= 64 (configs S / A: 32, F: 64), 4 heads, <= 1024 keypoints per image: the
// whole matcher is ~2 GFLOP per image pair and launch/latency bound (23 launches of 5-34 us per forward).  The kernels
// are row-wise: both images' tokens live in ONE row-major buffer [B*M rows of image 0 | B*N rows of image 1] so that
// every per-token layer (Linear, rotary, LayerNorm, GELU, residual) is a single launch over all tokens of the batch.
// D = 32: the token-wise products run on the matrix cores in split-fp16 (fp32-grade, as the attention products of
// attention.hip); D = 64 and the input projection: plain fp32 FMAs.
//
//  lg_posenc_kernel     normalize_keypoints (:137-149) + LearnableFourierPositionalEncoding (:168-173)
//  lg_tail_mfma_kernel  D = 32: out_proj / to_out + ffn + residual + the NEXT projection (cross to_qk | to_v, next Wqkv with
//                       rotary, final_proj) of a transformer block in one launch (:247-261, :303-327)
//  lg_linear_kernel     Y = X W^T + b with fused epilogues: rotary on the q|k columns (:158-159, :253-257),
//                       LayerNorm + GELU (ffn.1, ffn.2), residual add (x + ffn(...), :261)
//  (attention)          attention.hip: softmax(q k^T / sqrt(d)) v, streaming, split-fp16 matrix cores (:208-224, :312-321)
//  lg_sim_kernel        sim = f0 f1^T (:391) + per-tile row / column log-sum-exp partials
//  lg_finalize_kernel   sigmoid_log_double_softmax (:363-376) + per-tile row / column max, argmax (filter_matches :403-404)
//  lg_filter_kernel     mutual check, threshold, match scores (:405-416)
#include <cstdlib>
#include "kp2d_kernels.h"
#include "device_guard.h"

namespace kp2d {

// keypoints [n][2] -> cs[row][hd] = (cos f0..f_{hd/2-1} | sin ...).  Grid (batch item, image set, slice): every
// workgroup finds the keypoint extent itself (n <= a few thousand values, L2-resident) and writes one slice of the
// table — one workgroup per (item, set) was a 12-us serial chain of sincos at 1024 keypoints
__global__ __launch_bounds__(256) void lg_posenc_kernel(const LgPosArgs a) {
  __shared__ float red[4][4];
  const int set = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
  const int n = set ? a.N : a.M;
  const float* k = (set ? a.k1 : a.k0) + (size_t)b * n * 2;
  const float* sz = set ? a.size1 : a.size0;
  float sx, sy;
  if (sz) {
    sx = sz[b * 2]; sy = sz[b * 2 + 1];
  } else {   // size = 1 + max - min over the keypoints (:140-141)
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = tid; i < n; i += 256) {
      const float x = k[2 * i], y = k[2 * i + 1];
      mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mny = fminf(mny, y); mxy = fmaxf(mxy, y);
    }
    for (int o = 32; o > 0; o >>= 1) {
      mnx = fminf(mnx, __shfl_xor(mnx, o)); mxx = fmaxf(mxx, __shfl_xor(mxx, o));
      mny = fminf(mny, __shfl_xor(mny, o)); mxy = fmaxf(mxy, __shfl_xor(mxy, o));
    }
    if ((tid & 63) == 0) { red[tid >> 6][0] = mnx; red[tid >> 6][1] = mxx; red[tid >> 6][2] = mny; red[tid >> 6][3] = mxy; }
    __syncthreads();
    mnx = fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0]));
    mxx = fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1]));
    mny = fminf(fminf(red[0][2], red[1][2]), fminf(red[2][2], red[3][2]));
    mxy = fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3]));
    sx = 1.f + mxx - mnx; sy = 1.f + mxy - mny;
  }
  const float scale = fmaxf(sx, sy) / 2.f;
  const int hf = a.hd >> 1;
  float* cs = a.cs + ((size_t)(set ? a.B * a.M : 0) + (size_t)b * n) * a.hd;
  for (int e = blockIdx.z * 256 + tid; e < n * hf; e += 256 * gridDim.z) {
    const int i = e / hf, f = e - i * hf;
    const float x = (k[2 * i] - sx / 2.f) / scale, y = (k[2 * i + 1] - sy / 2.f) / scale;
    const float pr = a.wr[2 * f] * x + a.wr[2 * f + 1] * y;
    cs[(size_t)i * a.hd + f] = cosf(pr);
    cs[(size_t)i * a.hd + hf + f] = sinf(pr);
  }
}

int launch_lg_posenc(const LgPosArgs& a, hipStream_t s) {
  if (a.hd < 2 || (a.hd & 1)) return -1800;
  const int work = ((a.M > a.N ? a.M : a.N) * (a.hd >> 1) + 255) / 256;
  hipLaunchKernelGGL(lg_posenc_kernel, dim3(a.B, 2, work < 16 ? work : 16), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Row-wise linear layer.  Workgroup = 64 rows; thread (rg = tid/16, cg = tid%16) owns rows 4rg..4rg+3 and the
// column pairs {2cg + 32j, 2cg + 1 + 32j}, j < nout/32 (a rotary pair is two adjacent columns of one thread).
// X^T and W^T ([k][row] / [k][col]) sit in LDS, so one k step is one 16-byte broadcast read of the 4 rows and one
// 8-byte read per column pair.
// ---------------------------------------------------------------------------------------------
constexpr int LG_NJ = 6;        // nout <= 192 (Wqkv of the 64-wide config)
constexpr int LG_ROWS = 64;

__global__ __launch_bounds__(256) void lg_linear_kernel(const LgLinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int K = a.k0 + a.k1, nout = a.nout, nj = nout >> 5;
  float* xt = sm;                       // [K][64]
  float* wt = sm + (size_t)K * LG_ROWS; // [K][nout]
  const int tid = threadIdx.x, rg = tid >> 4, cg = tid & 15;
  const int row0 = blockIdx.x * LG_ROWS;
  // staging with 16-byte loads (nout % 32 == 0; the launcher checks k0, k1 and the row strides are multiples of 4)
  for (int e = tid; e < (K * nout) >> 2; e += 256)
    reinterpret_cast<float4*>(wt)[e] = reinterpret_cast<const float4*>(a.w)[e];
  const int K4 = K >> 2;
  for (int e = tid; e < LG_ROWS * K4; e += 256) {
    const int r = e / K4, k = 4 * (e - r * K4);
    const int row = row0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < a.rows)
      v = k < a.k0 ? *reinterpret_cast<const float4*>(a.x0 + (size_t)row * a.xs0 + k)
                   : *reinterpret_cast<const float4*>(a.x1 + (size_t)row * a.xs1 + (k - a.k0));
    xt[k * LG_ROWS + r] = v.x; xt[(k + 1) * LG_ROWS + r] = v.y; xt[(k + 2) * LG_ROWS + r] = v.z; xt[(k + 3) * LG_ROWS + r] = v.w;
  }
  __syncthreads();
  float acc[4][2 * LG_NJ];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 2 * LG_NJ; ++c) acc[r][c] = 0.f;
  for (int k = 0; k < K; ++k) {
    const float4 xv = *reinterpret_cast<const float4*>(&xt[k * LG_ROWS + 4 * rg]);
    const float xr[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int j = 0; j < LG_NJ; ++j) {
      if (j < nj) {
        const float2 wv = *reinterpret_cast<const float2*>(&wt[k * nout + 2 * cg + 32 * j]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[r][2 * j] = fmaf(xr[r], wv.x, acc[r][2 * j]);
          acc[r][2 * j + 1] = fmaf(xr[r], wv.y, acc[r][2 * j + 1]);
        }
      }
    }
  }
  // bias
#pragma unroll
  for (int j = 0; j < LG_NJ; ++j)
    if (j < nj) {
      const int c = 2 * cg + 32 * j;
      const float b0 = a.bias ? a.bias[c] : 0.f, b1 = a.bias ? a.bias[c + 1] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[r][2 * j] += b0; acc[r][2 * j + 1] += b1; }
    }
  if (a.epi == LG_EPI_ROTARY) {
    // columns [0, rot_cols) are q|k in (head, dim) order: pair (2i, 2i+1) of a head rotates by frequency i
    const int hf = a.hd >> 1;
#pragma unroll
    for (int j = 0; j < LG_NJ; ++j)
      if (j < nj) {
        const int c = 2 * cg + 32 * j;
        if (c < a.rot_cols) {
          const int f = (c % a.hd) >> 1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = row0 + 4 * rg + r;
            if (row < a.rows) {
              const float co = a.cs[(size_t)row * a.hd + f], si = a.cs[(size_t)row * a.hd + hf + f];
              const float y0 = acc[r][2 * j], y1 = acc[r][2 * j + 1];
              acc[r][2 * j] = y0 * co - y1 * si;        // t*cos + rotate_half(t)*sin, rotate_half = (-t1, t0)
              acc[r][2 * j + 1] = y1 * co + y0 * si;
            }
          }
        }
      }
  } else if (a.epi == LG_EPI_LNGELU) {
    // LayerNorm over the nout columns of each row (eps 1e-5, biased variance), then exact (erf) GELU
    const float inv = 1.f / (float)nout;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 2 * LG_NJ; ++c) if (c < 2 * nj) s += acc[r][c];
      for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
      const float mu = s * inv;
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < 2 * LG_NJ; ++c) if (c < 2 * nj) { const float d = acc[r][c] - mu; q = fmaf(d, d, q); }
      for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o);
      const float rs = 1.f / sqrtf(q * inv + 1e-5f);
#pragma unroll
      for (int j = 0; j < LG_NJ; ++j)
        if (j < nj) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int c = 2 * cg + 32 * j + u;
            const float y = (acc[r][2 * j + u] - mu) * rs * a.ln_g[c] + a.ln_b[c];
            acc[r][2 * j + u] = 0.5f * y * (1.f + erff(y * 0.70710678118654752f));
          }
        }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * rg + r;
    if (row >= a.rows) continue;
#pragma unroll
    for (int j = 0; j < LG_NJ; ++j)
      if (j < nj) {
        const int c = 2 * cg + 32 * j;
        if (c >= a.nvalid) continue;          // padded output columns are not stored
        float y0 = acc[r][2 * j], y1 = acc[r][2 * j + 1];
        if (a.epi == LG_EPI_RESID) {
          y0 += a.res[(size_t)row * a.rs + c];
          y1 += a.res[(size_t)row * a.rs + c + 1];
        }
        float* o = a.out + (size_t)row * a.os + a.oo + c;
        o[0] = y0;
        if (c + 1 < a.nvalid) o[1] = y1;
      }
  }
}

int launch_lg_linear(const LgLinArgs& a, hipStream_t s) {
  const int K = a.k0 + a.k1;
  if (a.nout < 32 || (a.nout & 31) || a.nout > 32 * LG_NJ || K < 4 || K > 128) return -1801;
  if ((a.k0 & 3) || (a.k1 & 3) || (a.xs0 & 3) || (a.k1 && (a.xs1 & 3))) return -1801;     // 16-byte staging loads
  if (a.epi == LG_EPI_ROTARY && ((a.hd & 1) || (a.rot_cols & 1) || !a.cs)) return -1802;
  const size_t lds = (size_t)K * (LG_ROWS + a.nout) * sizeof(float);
  static PerDeviceOnce lds_once;      // per device: a handle may live on any visible device
  if (int e = lds_opt_in(lds_once, reinterpret_cast<const void*>(&lg_linear_kernel))) return e;
  hipLaunchKernelGGL(lg_linear_kernel, dim3((a.rows + LG_ROWS - 1) / LG_ROWS), dim3(256), lds, s, a);
  return (int)hipGetLastError();
}

constexpr int LG_TAIL_VEC = 352;      // floats of bias / LayerNorm vectors staged by lg_tail_mfma_kernel

// ---------------------------------------------------------------------------------------------
// Fused block tail for D = 32: message = out_proj(ctx); h = GELU(LayerNorm(W1 [x | message] + b1)); x += W2 h + b2;
// then, optionally, the NEXT token-wise projection of the updated rows (the cross block's [to_qk | to_v], the next
// layer's Wqkv with rotary, the final projection).  All of it is row-local: one launch instead of five, and no HBM
// round trip of the message / hidden tensors.
// On the matrix cores.  A wave owns 16 rows and never meets another wave after the weights are
// staged.  Every product is computed transposed, Y^T = W X^T, with v_mfma_f32_16x16x32_f16 in split-fp16 arithmetic
// (x w = xh wh + xh wl + xl wh, fp32 accumulation: fp32-grade, as the attention products): W tiles are the A operand
// (pre-split images, lightglue_api.cpp mfma_image()), the activations the B operand.  An accumulator tile holds, for
// the lane's row (lane % 16), features 16 t + 4 (lane / 16) + r; the images order their k elements exactly that way,
// so an output is the next product's B operand after a split in registers: no LDS round trip, no barrier, no k loop.
// (The first version did these products with fp32 FMAs, transposed tiles in LDS: 192 dependent k steps per row group,
// 8 us (16-row groups) to 18 us (64-row groups) of compute behind a 5-7 us launch + staging floor,
// profiles/r3_ab_lg_tail_stages.txt; matcher 0.55 -> 0.45 ms for 8 pairs, 0.24 -> 0.20 ms for one.)
// a.ctx == nullptr: projection only (the first layer's Wqkv) — x goes straight to stage 4.
// ---------------------------------------------------------------------------------------------
typedef _Float16 lh8 __attribute__((ext_vector_type(8)));
typedef _Float16 lh2 __attribute__((ext_vector_type(2)));
typedef float lf4 __attribute__((ext_vector_type(4)));
typedef float lf2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lg_split8(const float* x, lh8& hi, lh8& lo) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const lf2 v = {x[j], x[j + 1]};
    const lh2 h = __builtin_convertvector(v, lh2);
    unsigned l;       // lo = fp16(x - float(hi)), three instructions per two values (attention.hip att_split2)
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(x[j]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(x[j + 1]));
    const lh2 lw = __builtin_bit_cast(lh2, l);
    hi[j] = h[0]; hi[j + 1] = h[1]; lo[j] = lw[0]; lo[j + 1] = lw[1];
  }
}
// acc += W(t, s) . b for one image block (hi at +0, lo at +512 halves): small terms first
__device__ __forceinline__ lf4 lg_mma(const _Float16* blk, int lane, const lh8 bh, const lh8 bl, lf4 acc) {
  const lh8 wh = *reinterpret_cast<const lh8*>(blk + lane * 8);
  const lh8 wl = *reinterpret_cast<const lh8*>(blk + 512 + lane * 8);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh, acc, 0, 0, 0);
}

constexpr int LG_IMG_O = 2 * 1024, LG_IMG_1 = 8 * 1024, LG_IMG_2 = 4 * 1024, LG_IMG_N = 6 * 1024;      // halves

template <int NW>   // waves per workgroup (16 rows each)
__global__ __launch_bounds__(64 * NW) void lg_tail_mfma_kernel(const LgTailArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  _Float16* io = reinterpret_cast<_Float16*>(sm);
  _Float16* i1 = io + LG_IMG_O;
  _Float16* i2 = i1 + LG_IMG_1;
  _Float16* in = i2 + LG_IMG_2;
  float* vec = reinterpret_cast<float*>(in + LG_IMG_N);      // bo[32] | b1[64] | ln_g[64] | ln_b[64] | b2[32] | bn[96]
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = (blockIdx.x * NW + wave) * 16 + (lane & 15);
  const bool ok = row < a.rows;
  const bool full = a.ctx != nullptr;
  // every global read of the kernel, before the only barrier
  auto copy16 = [&](_Float16* dst, const void* src, int halves) {
    for (int e = tid; e < halves >> 3; e += 64 * NW) reinterpret_cast<uint4*>(dst)[e] = reinterpret_cast<const uint4*>(src)[e];
  };
  if (full) { copy16(io, a.io, LG_IMG_O); copy16(i1, a.i1, LG_IMG_1); copy16(i2, a.i2, LG_IMG_2); }
  if (a.nn) copy16(in, a.in, a.nn * 64);                     // (nn / 16 tiles of 1024 halves)
  for (int e = tid; e < LG_TAIL_VEC; e += 64 * NW) {
    float v = 0.f;
    if (e < 256) {
      if (full) v = e < 32 ? a.bo[e] : e < 96 ? a.b1[e - 32] : e < 160 ? a.ln_g[e - 96] : e < 224 ? a.ln_b[e - 160] : a.b2[e - 224];
    } else if (a.nn && a.bn && e - 256 < a.nn) {
      v = a.bn[e - 256];
    }
    vec[e] = v;
  }
  // this lane's eight features of its row: 4 g .. 4 g + 3 and 16 + 4 g .. + 3 (the accumulator order)
  float xv[8] = {}, cv[8] = {};
  if (ok) {
    const float* xp = a.x + (size_t)row * 32 + 4 * g;
    *reinterpret_cast<float4*>(&xv[0]) = *reinterpret_cast<const float4*>(xp);
    *reinterpret_cast<float4*>(&xv[4]) = *reinterpret_cast<const float4*>(xp + 16);
    if (full) {
      const float* cp = a.ctx + (size_t)row * 32 + 4 * g;
      *reinterpret_cast<float4*>(&cv[0]) = *reinterpret_cast<const float4*>(cp);
      *reinterpret_cast<float4*>(&cv[4]) = *reinterpret_cast<const float4*>(cp + 16);
    }
  }
  // rotary factors of the lane's two column pairs: column 16 t + 4 g + 2 p has frequency ((4 g + 2 p) % hd) / 2 for
  // every tile t (hd divides 16)
  float rco[2] = {1.f, 1.f}, rsi[2] = {0.f, 0.f};
  if (a.nn && a.cs && ok) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int f = ((4 * g + 2 * p) % a.hd) >> 1;
      rco[p] = a.cs[(size_t)row * a.hd + f];
      rsi[p] = a.cs[(size_t)row * a.hd + (a.hd >> 1) + f];
    }
  }
  __syncthreads();
  auto bias4 = [&](int off) { return *reinterpret_cast<const lf4*>(&vec[off + 4 * g]); };
  float xn[8];
  if (full) {
    // ---- stage 1: message = ctx Wo^T + bo ----
    lh8 bh, bl;
    float msg[8];
    lg_split8(cv, bh, bl);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const lf4 acc = lg_mma(io + t * 1024, lane, bh, bl, bias4(16 * t));
#pragma unroll
      for (int r = 0; r < 4; ++r) msg[4 * t + r] = acc[r];
    }
    // ---- stage 2: h = GELU(LayerNorm([x | message] W1^T + b1)) ----
    lh8 xh, xl, mh, ml;
    lg_split8(xv, xh, xl);
    lg_split8(msg, mh, ml);
    float h[16];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      lf4 acc = lg_mma(i1 + (2 * t) * 1024, lane, xh, xl, bias4(32 + 16 * t));
      acc = lg_mma(i1 + (2 * t + 1) * 1024, lane, mh, ml, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) h[4 * t + r] = acc[r];
    }
    {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) s += h[c];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const float mu = s * (1.f / 64.f);
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) { const float d = h[c] - mu; q = fmaf(d, d, q); }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      const float rs = 1.f / sqrtf(q * (1.f / 64.f) + 1e-5f);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const lf4 gg = bias4(96 + 16 * t), be = bias4(160 + 16 * t);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y = (h[4 * t + r] - mu) * rs * gg[r] + be[r];
          h[4 * t + r] = 0.5f * y * (1.f + erff(y * 0.70710678118654752f));
        }
      }
    }
    // ---- stage 3: x += h W2^T + b2 ----
    lh8 hh[2], hl[2];
    lg_split8(&h[0], hh[0], hl[0]);
    lg_split8(&h[8], hh[1], hl[1]);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      lf4 acc = lg_mma(i2 + (2 * t) * 1024, lane, hh[0], hl[0], bias4(224 + 16 * t));
      acc = lg_mma(i2 + (2 * t + 1) * 1024, lane, hh[1], hl[1], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) xn[4 * t + r] = xv[4 * t + r] + acc[r];
    }
    if (ok) {
      float* xp = a.x + (size_t)row * 32 + 4 * g;
      *reinterpret_cast<float4*>(xp) = *reinterpret_cast<const float4*>(&xn[0]);
      *reinterpret_cast<float4*>(xp + 16) = *reinterpret_cast<const float4*>(&xn[4]);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) xn[c] = xv[c];
  }
  if (!a.nn) return;
  // ---- stage 4: the next projection of the updated rows, rotary on its leading columns ----
  lh8 nh, nl;
  lg_split8(xn, nh, nl);
  const int nt = a.nn >> 4;
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    if (t >= nt) break;
    lf4 y = lg_mma(in + t * 1024, lane, nh, nl, bias4(256 + 16 * t));
    const int c = 16 * t + 4 * g;
    if (a.cs && c < a.rot_cols) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float t0 = y[2 * p], t1 = y[2 * p + 1];
        y[2 * p] = t0 * rco[p] - t1 * rsi[p];        // t*cos + rotate_half(t)*sin, rotate_half = (-t1, t0)
        y[2 * p + 1] = t1 * rco[p] + t0 * rsi[p];
      }
    }
    if (!ok) continue;
    float* o = a.on + (size_t)row * a.nos + c;
    if (c + 3 < a.nvalid) {
      *reinterpret_cast<lf4*>(o) = y;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (c + r < a.nvalid) o[r] = y[r];
    }
  }
}

int launch_lg_tail(const LgTailArgs& a, hipStream_t s) {
  if (a.D != 32) return -1804;
  if (a.nn && (a.nn != 64 && a.nn != 96)) return -1805;
  if (a.nn && (!a.in || !a.on || a.nvalid < 1 || a.nvalid > a.nn || (a.nos & 3) ||
               (a.cs && ((a.hd & 1) || 16 % a.hd || (a.rot_cols & 15)))))
    return -1805;
  if (a.ctx && (!a.io || !a.i1 || !a.i2)) return -1807;
  if (!a.ctx && !a.nn) return -1807;
  // four waves per workgroup also at one image pair (2048 rows = 32 workgroups): one-wave workgroups spread wider but
  // each stages the 42 KB of operands with 64 lanes — 0.236 vs 0.204 ms per forward
  static const int forced = getenv("KP2D_LG_TAIL_NW") ? atoi(getenv("KP2D_LG_TAIL_NW")) : 0;
  const int nw = forced ? forced : 4;
  const size_t lds = (size_t)(LG_IMG_O + LG_IMG_1 + LG_IMG_2 + LG_IMG_N) * 2 + LG_TAIL_VEC * sizeof(float);
  const dim3 grid((a.rows + 16 * nw - 1) / (16 * nw));
  if (nw == 4) hipLaunchKernelGGL(lg_tail_mfma_kernel<4>, grid, dim3(256), lds, s, a);
  else if (nw == 1) hipLaunchKernelGGL(lg_tail_mfma_kernel<1>, grid, dim3(64), lds, s, a);
  else return -1806;
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Assignment.  scores is the reference's [B][M+1][N+1] log-assignment tensor; its inner block first holds sim.
// Three launches over 64 x 64 tiles of the inner block (the first version walked whole rows and whole columns in
// six: its two column walks alone were 33 + 25 us of a 96-us stage at one pair, profiles/r3_lightglue_kernels.txt):
//   lg_sim_kernel       sim tile -> scores, plus the tile's partial log-sum-exp of every row and column
//   lg_finalize_kernel  merges the partials of its rows / columns, writes the final tile, the border row / column,
//                       and the tile's partial max / argmax of every row and column
//   lg_filter_kernel    merges the argmax partials it needs and applies filter_matches
// A partial is indexed [b][tile][row or column]; merges run in increasing tile order, so ties resolve to the lowest
// index as torch.max does and the result does not depend on the grid.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
  const float mm = fmaxf(m, m2);
  if (mm == -INFINITY) { m = mm; s = 0.f; return; }
  s = s * expf(m - mm) + s2 * expf(m2 - mm);
  m = mm;
}

constexpr int LG_TP = 68;      // pitch (floats) of the LDS tile that global rows go through

// thread (ty = tid/16, tx = tid%16) owns rows i0 + 4 ty + r, columns j0 + 4 tx + c of the tile
__global__ __launch_bounds__(256) void lg_sim_kernel(const LgAssignArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lg_sm[];             // at[D][64] | bt[D][64] | tile[64][LG_TP]
  const int b = blockIdx.z, i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int tid = threadIdx.x, D = a.D, M = a.M, N = a.N;
  const int Mv = a.cnt0 ? min(M, a.cnt0[b]) : M, Nv = a.cnt1 ? min(N, a.cnt1[b]) : N;      // keypoints that exist (the rest is padding)
  float* const at = lg_sm;                   // [k][row]
  float* const bt = at + 64 * D;
  float* const tile = bt + 64 * D;
  const float* f0 = a.fz + (size_t)b * M * a.fs;
  const float* f1 = a.fz + ((size_t)a.B * M + (size_t)b * N) * a.fs;
  // (a run-time-bounded loop with the loads inside is not unrolled: its round trips to L2 ran one after the other,
  // and one workgroup took 10 us for 0.26 MFLOP.  All loads of a thread are issued before the first LDS write.)
  {
    const int q = D >> 2;                    // 16-byte granules per row (D = 32 or 64: 2 or 4 per thread and matrix)
    float4 va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u, r = e / q, k = 4 * (e - r * q);
      va[u] = vb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < 64 * q) {
        if (i0 + r < M) va[u] = *reinterpret_cast<const float4*>(f0 + (size_t)(i0 + r) * a.fs + k);
        if (j0 + r < N) vb[u] = *reinterpret_cast<const float4*>(f1 + (size_t)(j0 + r) * a.fs + k);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u, r = e / q, k = 4 * (e - r * q);
      if (e < 64 * q) {
        at[k * 64 + r] = va[u].x; at[(k + 1) * 64 + r] = va[u].y; at[(k + 2) * 64 + r] = va[u].z; at[(k + 3) * 64 + r] = va[u].w;
        bt[k * 64 + r] = vb[u].x; bt[(k + 1) * 64 + r] = vb[u].y; bt[(k + 2) * 64 + r] = vb[u].z; bt[(k + 3) * 64 + r] = vb[u].w;
      }
    }
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  float acc[4][4] = {};
  for (int k = 0; k < D; ++k) {
    const float4 av = *reinterpret_cast<const float4*>(&at[k * 64 + 4 * ty]);
    const float4 bv = *reinterpret_cast<const float4*>(&bt[k * 64 + 4 * tx]);
    const float ar[4] = {av.x, av.y, av.z, av.w}, br[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(ar[r], br[c], acc[r][c]);
  }
  // The rows of scores are N + 1 floats: no 16-byte alignment, and a thread's 4 x 4 block stored directly is sixteen
  // 4-byte stores at a 16-byte lane stride (a quarter of every 64-byte segment used).  Through an LDS tile instead:
  // every wave store is 64 consecutive floats of one row.
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      tile[(4 * ty + r) * LG_TP + 4 * tx + c] = acc[r][c];
      const int i = i0 + 4 * ty + r, j = j0 + 4 * tx + c;
      if (i >= Mv || j >= Nv) acc[r][c] = -INFINITY;          // outside the inner block / padding rows: no weight in the sums below
    }
  // rows: the 16 threads of a row group are 16 adjacent lanes
  const int TN = gridDim.x, TM = gridDim.y;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float m = fmaxf(fmaxf(acc[r][0], acc[r][1]), fmaxf(acc[r][2], acc[r][3]));
    for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    if (m != -INFINITY) {
#pragma unroll
      for (int c = 0; c < 4; ++c) s += expf(acc[r][c] - m);
    }
    for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
    const int i = i0 + 4 * ty + r;
    if (tx == 0 && i < M) {
      const size_t e = ((size_t)b * TN + blockIdx.x) * M + i;
      a.rp_m[e] = m; a.rp_s[e] = s;
    }
  }
  // columns: the 16 row groups of a column are lane bits 4-5 of the four waves
  __syncthreads();                           // at / bt are free: reuse their first words for the cross-wave step
  float* cm = at;                            // [4 waves][64 columns]
  float* cs = at + 256;
  const int wave = tid >> 6, q = (tid >> 4) & 3;
  {
    float* sc = a.scores + (size_t)b * (M + 1) * (N + 1);
    const int j = j0 + (tid & 63);
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + 16 * wave + r;
      if (i < M && j < N) sc[(size_t)i * (N + 1) + j] = tile[(16 * wave + r) * LG_TP + (tid & 63)];
    }
  }
  float m4[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float m = fmaxf(fmaxf(acc[0][c], acc[1][c]), fmaxf(acc[2][c], acc[3][c]));
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    if (q == 0) cm[wave * 64 + 4 * tx + c] = m;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = 4 * tx + c;
    const float m = fmaxf(fmaxf(cm[col], cm[64 + col]), fmaxf(cm[128 + col], cm[192 + col]));
    m4[c] = m;
    float s = 0.f;
    if (m != -INFINITY) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s += expf(acc[r][c] - m);
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (q == 0) cs[wave * 64 + col] = s;
  }
  __syncthreads();
  if (wave == 0 && q == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = 4 * tx + c, j = j0 + col;
      if (j < N) {
        const size_t e = ((size_t)b * TM + blockIdx.y) * N + j;
        a.cp_m[e] = m4[c];
        a.cp_s[e] = ((cs[col] + cs[64 + col]) + cs[128 + col]) + cs[192 + col];
      }
    }
  }
}

// merge n per-tile partials (stride apart) in tile order; sixteen pairs of loads in flight at a time
__device__ __forceinline__ void lse_merge_partials(const float* pm, const float* ps, int n, size_t stride, float& m, float& s) {
  m = -INFINITY; s = 0.f;
  for (int t0 = 0; t0 < n; t0 += 16) {
    float vm[16], vs[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      vm[u] = -INFINITY; vs[u] = 0.f;
      if (t0 + u < n) { vm[u] = pm[(size_t)(t0 + u) * stride]; vs[u] = ps[(size_t)(t0 + u) * stride]; }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (t0 + u < n) lse_merge(m, s, vm[u], vs[u]);
  }
}
// max / argmax over n per-tile partials in tile order (strict >: the lowest tile wins ties)
__device__ __forceinline__ int best_of_partials(const float* pm, const int* pa, int n, size_t stride, float& mx) {
  mx = -INFINITY;
  int arg = 0;
  for (int t0 = 0; t0 < n; t0 += 16) {
    float vm[16];
    int va[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      vm[u] = -INFINITY; va[u] = 0;
      if (t0 + u < n) { vm[u] = pm[(size_t)(t0 + u) * stride]; va[u] = pa[(size_t)(t0 + u) * stride]; }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (vm[u] > mx) { mx = vm[u]; arg = va[u]; }
  }
  return arg;
}

__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

// one 64 x 64 tile of scores: inner block <- log_softmax rows + log_softmax cols + certainties
// (sigmoid_log_double_softmax, :363-376); the tiles of the first tile column / row also write the border column / row
// (logsigmoid(-z)); plus the tile's row / column max and argmax of the final values for filter_matches (:403-404)
__global__ __launch_bounds__(256) void lg_finalize_kernel(const LgAssignArgs a) {
  __shared__ float s_base[64], s_col[64], s_ls1[64];
  __shared__ float s_cm[4][64];
  __shared__ int s_ci[4][64];
  __shared__ float tile[64 * LG_TP];
  const int b = blockIdx.z, i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int tid = threadIdx.x, M = a.M, N = a.N, TN = gridDim.x, TM = gridDim.y;
  const int Mv = a.cnt0 ? min(M, a.cnt0[b]) : M, Nv = a.cnt1 ? min(N, a.cnt1[b]) : N;
  float* sc = a.scores + (size_t)b * (M + 1) * (N + 1);
  const float* z0p = a.fz + (size_t)b * M * a.fs + a.D;                          // matchability logits of image 0
  const float* z1p = a.fz + ((size_t)a.B * M + (size_t)b * N) * a.fs + a.D;      // ... of image 1
  // sim tile in: whole 64-float row segments per wave load (see lg_sim_kernel), all sixteen issued before anything
  // waits — the merges below then run under their latency
  float tin[16];
  {
    const int j = j0 + (tid & 63), w = tid >> 6;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + 16 * w + r;
      tin[r] = (i < M && j < N) ? sc[(size_t)i * (N + 1) + j] : 0.f;
    }
  }
  if (tid < 64) {
    const int i = i0 + tid;
    float base = 0.f;
    if (i < M) {
      float m, s;
      lse_merge_partials(a.rp_m + (size_t)b * TN * M + i, a.rp_s + (size_t)b * TN * M + i, TN, M, m, s);
      const float z0 = z0p[(size_t)i * a.fs];
      base = log_sigmoid(z0) - (m + logf(s));
      if (blockIdx.x == 0) sc[(size_t)i * (N + 1) + N] = log_sigmoid(-z0);
    }
    s_base[tid] = base;
  } else if (tid < 128) {
    const int c = tid - 64, j = j0 + c;
    float cl = 0.f, ls1 = 0.f;
    if (j < N) {
      float m, s;
      lse_merge_partials(a.cp_m + (size_t)b * TM * N + j, a.cp_s + (size_t)b * TM * N + j, TM, N, m, s);
      cl = m + logf(s);
      const float z1 = z1p[(size_t)j * a.fs];
      ls1 = log_sigmoid(z1);
      if (blockIdx.y == 0) sc[(size_t)M * (N + 1) + j] = log_sigmoid(-z1);
    }
    s_col[c] = cl; s_ls1[c] = ls1;
  } else if (tid == 128 && blockIdx.x == 0 && blockIdx.y == 0) {
    sc[(size_t)M * (N + 1) + N] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) tile[(16 * (tid >> 6) + r) * LG_TP + (tid & 63)] = tin[r];
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  float v[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + 4 * ty + r;
    const float base = s_base[4 * ty + r];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = j0 + 4 * tx + c;
      float x = -INFINITY;
      if (i < M && j < N) {
        float* p = &tile[(4 * ty + r) * LG_TP + 4 * tx + c];
        // (a padding row / column: -inf — no assignment mass, never a maximum)
        if (i < Mv && j < Nv) x = 2.f * *p + base - s_col[4 * tx + c] + s_ls1[4 * tx + c];
        *p = x;
      }
      v[r][c] = x;
    }
  }
  // row max / argmax over the tile's columns (lowest column on ties)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float mx = v[r][0];
    int arg = j0 + 4 * tx;
#pragma unroll
    for (int c = 1; c < 4; ++c)
      if (v[r][c] > mx) { mx = v[r][c]; arg = j0 + 4 * tx + c; }
    for (int o = 1; o < 16; o <<= 1) {
      const float m2 = __shfl_xor(mx, o);
      const int a2 = __shfl_xor(arg, o);
      if (m2 > mx || (m2 == mx && a2 < arg)) { mx = m2; arg = a2; }
    }
    const int i = i0 + 4 * ty + r;
    if (tx == 0 && i < M) {
      const size_t e = ((size_t)b * TN + blockIdx.x) * M + i;
      a.rmax[e] = mx; a.rarg[e] = arg;
    }
  }
  // column max / argmax over the tile's rows (lowest row on ties)
  const int wave = tid >> 6, q = (tid >> 4) & 3;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float mx = v[0][c];
    int arg = i0 + 4 * ty;
#pragma unroll
    for (int r = 1; r < 4; ++r)
      if (v[r][c] > mx) { mx = v[r][c]; arg = i0 + 4 * ty + r; }
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      const float m2 = __shfl_xor(mx, o);
      const int a2 = __shfl_xor(arg, o);
      if (m2 > mx || (m2 == mx && a2 < arg)) { mx = m2; arg = a2; }
    }
    if (q == 0) { s_cm[wave][4 * tx + c] = mx; s_ci[wave][4 * tx + c] = arg; }
  }
  __syncthreads();
  {   // final tile out
    const int j = j0 + (tid & 63);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + 16 * wave + r;
      if (i < M && j < N) sc[(size_t)i * (N + 1) + j] = tile[(16 * wave + r) * LG_TP + (tid & 63)];
    }
  }
  if (tid < 64 && j0 + tid < N) {
    float mx = s_cm[0][tid];
    int arg = s_ci[0][tid];
    for (int w = 1; w < 4; ++w)
      if (s_cm[w][tid] > mx) { mx = s_cm[w][tid]; arg = s_ci[w][tid]; }      // waves hold increasing rows: strict > keeps the lowest
    const size_t e = ((size_t)b * TM + blockIdx.y) * N + j0 + tid;
    a.cmax[e] = mx; a.carg[e] = arg;
  }
}

// filter_matches (:401-416): thread e < M handles row e, thread e >= M column e - M
__global__ __launch_bounds__(256) void lg_filter_kernel(const LgAssignArgs a) {
  const int b = blockIdx.y, M = a.M, N = a.N;
  const int TN = (N + 63) >> 6, TM = (M + 63) >> 6;
  const int e = blockIdx.x * 256 + threadIdx.x;
  auto row_best = [&](int i, float& mx) {
    return best_of_partials(a.rmax + (size_t)b * TN * M + i, a.rarg + (size_t)b * TN * M + i, TN, M, mx);
  };
  auto col_best = [&](int j) {
    float mx;
    return best_of_partials(a.cmax + (size_t)b * TM * N + j, a.carg + (size_t)b * TM * N + j, TM, N, mx);
  };
  const int Mv = a.cnt0 ? min(M, a.cnt0[b]) : M, Nv = a.cnt1 ? min(N, a.cnt1[b]) : N;
  if (e < M && (e >= Mv || Nv == 0)) {              // a padding row, or nothing to match against
    a.mscores0[(size_t)b * M + e] = 0.f;
    a.matches0[(size_t)b * M + e] = -1;
  } else if (e >= M && e < M + N && (e - M >= Nv || Mv == 0)) {
    a.mscores1[(size_t)b * N + e - M] = 0.f;
    a.matches1[(size_t)b * N + e - M] = -1;
  } else if (e < M) {
    float mx;
    const int j = row_best(e, mx);
    const bool mutual = col_best(j) == e;
    const float ms = mutual ? expf(mx) : 0.f;
    a.mscores0[(size_t)b * M + e] = ms;
    a.matches0[(size_t)b * M + e] = (mutual && ms > a.th) ? (int64_t)j : (int64_t)-1;
  } else if (e < M + N) {
    const int j = e - M, i = col_best(j);
    float mx;
    const bool mutual1 = row_best(i, mx) == j;       // then m1[m0[i]] == i as well: row i is mutual with column j
    const float ms0 = mutual1 ? expf(mx) : 0.f;
    a.mscores1[(size_t)b * N + j] = ms0;
    a.matches1[(size_t)b * N + j] = (mutual1 && ms0 > a.th) ? (int64_t)i : (int64_t)-1;
  }
}

int launch_lg_assign(const LgAssignArgs& a, hipStream_t s) {
  if (a.D < 8 || a.D > 64 || (a.D & 3) || a.M < 1 || a.N < 1) return -1803;
  const dim3 tiles((a.N + 63) / 64, (a.M + 63) / 64, a.B);
  // (D >= 8: the cross-wave scratch of the column sums reuses the first 512 floats of at)
  hipLaunchKernelGGL(lg_sim_kernel, tiles, dim3(256), (size_t)(2 * 64 * a.D + 64 * LG_TP) * sizeof(float), s, a);
  hipLaunchKernelGGL(lg_finalize_kernel, tiles, dim3(256), 0, s, a);
  hipLaunchKernelGGL(lg_filter_kernel, dim3((a.M + a.N + 255) / 256, a.B), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace kp2d
