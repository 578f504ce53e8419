// LightGlue matcher kernels (reference: lightglue/lightglue.py; inference path, flash = False, no early stopping /
// point pruning).  Descriptor width D <= 64 (configs S / A: 32, F: 64), 4 heads, <= 1024 keypoints per image: the
// whole matcher is ~2 GFLOP per image pair and launch/latency bound, so the token-wise layers are plain fp32 FMAs (the
// attention products use the fp32-grade split-fp16 MFMA kernel of attention.hip) and the kernels are row-wise: both images' tokens live in ONE row-major buffer [B*M rows of image 0 | B*N rows of image 1] so that every
// per-token layer (Linear, rotary, LayerNorm, GELU, residual) is a single launch over all tokens of the batch.
//
//  lg_posenc_kernel     normalize_keypoints (:137-149) + LearnableFourierPositionalEncoding (:168-173)
//  lg_linear_kernel     Y = X W^T + b with fused epilogues: rotary on the q|k columns (:158-159, :253-257),
//                       LayerNorm + GELU (ffn.1, ffn.2), residual add (x + ffn(...), :261)
//  (attention)          attention.hip: softmax(q k^T / sqrt(d)) v, streaming, split-fp16 matrix cores (:208-224, :312-321)
//  lg_sim_kernel        sim = f0 f1^T (:391) + per-tile row / column log-sum-exp partials
//  lg_finalize_kernel   sigmoid_log_double_softmax (:363-376) + per-tile row / column max, argmax (filter_matches :403-404)
//  lg_filter_kernel     mutual check, threshold, match scores (:405-416)
#include <cstdlib>
#include <type_traits>
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

constexpr int LG_TAIL_VEC = 352;      // floats of bias / LayerNorm vectors staged by lg_tail_kernel
template <int RT> __device__ __forceinline__ void lg_rows(const float* p, float (&x)[RT]) {      // RT adjacent LDS floats
  if constexpr (RT == 4) { const float4 v = *reinterpret_cast<const float4*>(p); x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
  else if constexpr (RT == 2) { const float2 v = *reinterpret_cast<const float2*>(p); x[0] = v.x; x[1] = v.y; }
  else x[0] = *p;
}

// ---------------------------------------------------------------------------------------------
// Fused block tail for D = 32: message = out_proj(ctx); h = GELU(LayerNorm(W1 [x | message] + b1)); x += W2 h + b2.
// All three products are row-local, so one workgroup carries its 16 RT rows through them with the intermediate tiles
// (transposed, [k][row]) in LDS: 3 launches and 2 HBM round trips of the message / hidden tensors less per block.
// Thread mapping of lg_linear_kernel (thread = RT rows x column pairs {2cg + 32j, +1}); a row's arithmetic does not
// depend on RT.  One image pair is 2048 rows: at 64 rows per workgroup (RT = 4) that was 32 workgroups walking the
// 192 k-steps of the four products on 32 of the 256 CUs, 20 us per launch (RT = 1: 128 workgroups, 0.404 -> 0.348 ms per
// forward, profiles/r3_lightglue_kernels.txt).
// ---------------------------------------------------------------------------------------------
template <int RT>   // rows per thread: a workgroup carries 16 * RT rows
__global__ __launch_bounds__(256) void lg_tail_kernel(const LgTailArgs a) {
  constexpr int D = 32, D2 = 64, R = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wo = sm;                  // [D][D]
  float* w1 = wo + D * D;          // [D2][D2]
  float* w2 = w1 + D2 * D2;        // [D2][D]
  float* wn = w2 + D2 * D;         // [D][96]  the next projection (unused without one)
  float* vec = wn + D * 96;        // bo[32] | b1[64] | ln_g[64] | ln_b[64] | b2[32] | bn[96]
  float* ct = vec + LG_TAIL_VEC;   // [D][R]   ctx^T
  float* xt = ct + D * R;          // [D2][R]  (x | message)^T, later h^T
  const int tid = threadIdx.x, rg = tid >> 4, cg = tid & 15;
  const int row0 = blockIdx.x * R;
  for (int e = tid; e < (D * D) >> 2; e += 256) reinterpret_cast<float4*>(wo)[e] = reinterpret_cast<const float4*>(a.wo)[e];
  for (int e = tid; e < (D2 * D2) >> 2; e += 256) reinterpret_cast<float4*>(w1)[e] = reinterpret_cast<const float4*>(a.w1)[e];
  for (int e = tid; e < (D2 * D) >> 2; e += 256) reinterpret_cast<float4*>(w2)[e] = reinterpret_cast<const float4*>(a.w2)[e];
  // Every global read of the kernel is issued here, before the first barrier: with one or two waves per SIMD nothing
  // hides a load, and the first version fetched its biases, the next projection's weights, the old x and the rotary
  // table stage by stage — six exposed round trips, 56 % of the wave-cycles waiting (profiles/r3_pmc_lg8_summary.txt)
  if (a.nn)
    for (int e = tid; e < (D * a.nn) >> 2; e += 256) reinterpret_cast<float4*>(wn)[e] = reinterpret_cast<const float4*>(a.wn)[e];
  for (int e = tid; e < LG_TAIL_VEC; e += 256) {
    float v = 0.f;
    if (e < 32) v = a.bo[e];
    else if (e < 96) v = a.b1[e - 32];
    else if (e < 160) v = a.ln_g[e - 96];
    else if (e < 224) v = a.ln_b[e - 160];
    else if (e < 256) v = a.b2[e - 224];
    else if (a.nn && a.bn && e - 256 < a.nn) v = a.bn[e - 256];
    vec[e] = v;
  }
  // rotary factors of this thread's rows: column pair c = 2 cg + 32 j has frequency (c % hd) / 2, the same for every j
  // when hd divides 32 (the launcher checks)
  float rco[RT], rsi[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    rco[r] = 1.f; rsi[r] = 0.f;
    const int row = row0 + RT * rg + r;
    if (a.nn && a.cs && row < a.rows) {
      const int f = ((2 * cg) % a.hd) >> 1;
      rco[r] = a.cs[(size_t)row * a.hd + f];
      rsi[r] = a.cs[(size_t)row * a.hd + (a.hd >> 1) + f];
    }
  }
  for (int e = tid; e < R * (D >> 2); e += 256) {
    const int r = e >> 3, k = 4 * (e & 7);
    const int row = row0 + r;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f), x = c;
    if (row < a.rows) {
      c = *reinterpret_cast<const float4*>(a.ctx + (size_t)row * D + k);
      x = *reinterpret_cast<const float4*>(a.x + (size_t)row * D + k);
    }
    ct[k * R + r] = c.x; ct[(k + 1) * R + r] = c.y; ct[(k + 2) * R + r] = c.z; ct[(k + 3) * R + r] = c.w;
    xt[k * R + r] = x.x; xt[(k + 1) * R + r] = x.y; xt[(k + 2) * R + r] = x.z; xt[(k + 3) * R + r] = x.w;
  }
  __syncthreads();
#if defined(LG_ABL) && LG_ABL == 4
  if (a.rows > 0) return;
#endif
  // ---- stage 1: message[64 x 32] = ctx Wo^T + bo -> xt rows D..2D ----
  {
    float acc[RT][2] = {};
    for (int k = 0; k < D; ++k) {
      float xr[RT];
      lg_rows<RT>(&ct[k * R + RT * rg], xr);
      const float2 wv = *reinterpret_cast<const float2*>(&wo[k * D + 2 * cg]);
#pragma unroll
      for (int r = 0; r < RT; ++r) { acc[r][0] = fmaf(xr[r], wv.x, acc[r][0]); acc[r][1] = fmaf(xr[r], wv.y, acc[r][1]); }
    }
    const float b0 = vec[2 * cg], b1 = vec[2 * cg + 1];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      xt[(D + 2 * cg) * R + RT * rg + r] = acc[r][0] + b0;
      xt[(D + 2 * cg + 1) * R + RT * rg + r] = acc[r][1] + b1;
    }
  }
  __syncthreads();
#if defined(LG_ABL) && LG_ABL == 5
  if (a.rows > 0) return;
#endif
  // ---- stage 2: h[64 x 64] = GELU(LayerNorm([x | message] W1^T + b1)) ----
  float h[RT][4];
  float xold[RT][2];               // this thread's elements of x, for the residual of stage 3
#pragma unroll
  for (int r = 0; r < RT; ++r) { xold[r][0] = xt[(2 * cg) * R + RT * rg + r]; xold[r][1] = xt[(2 * cg + 1) * R + RT * rg + r]; }
  {
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) h[r][c] = 0.f;
#if defined(LG_ABL) && (LG_ABL == 2 || LG_ABL == 3)
    for (int k = 0; k < (a.rows < 0 ? D2 : 1); ++k) {
#else
    for (int k = 0; k < D2; ++k) {
#endif
      float xr[RT];
      lg_rows<RT>(&xt[k * R + RT * rg], xr);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float2 wv = *reinterpret_cast<const float2*>(&w1[k * D2 + 2 * cg + 32 * j]);
#pragma unroll
        for (int r = 0; r < RT; ++r) { h[r][2 * j] = fmaf(xr[r], wv.x, h[r][2 * j]); h[r][2 * j + 1] = fmaf(xr[r], wv.y, h[r][2 * j + 1]); }
      }
    }
    float bb[4], gg[4], be[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = 2 * cg + 32 * (c >> 1) + (c & 1);
      bb[c] = vec[32 + col]; gg[c] = vec[96 + col]; be[c] = vec[160 + col];
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) { h[r][c] += bb[c]; s += h[r][c]; }
      for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
      const float mu = s * (1.f / D2);
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) { const float d = h[r][c] - mu; q = fmaf(d, d, q); }
      for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o);
      const float rs = 1.f / sqrtf(q * (1.f / D2) + 1e-5f);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float y = (h[r][c] - mu) * rs * gg[c] + be[c];
#if defined(LG_ABL) && LG_ABL == 1
        h[r][c] = y;
#else
        h[r][c] = 0.5f * y * (1.f + erff(y * 0.70710678118654752f));
#endif
      }
    }
  }
#if defined(LG_ABL) && LG_ABL == 6
  if (a.rows > 0) { if (h[0][0] == 123.f) a.x[0] = 0.f; return; }
#endif
  __syncthreads();                 // every thread is done reading (x | message)^T and W1
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) xt[(2 * cg + 32 * (c >> 1) + (c & 1)) * R + RT * rg + r] = h[r][c];
  __syncthreads();
  // ---- stage 3: x += h W2^T + b2 ----
  {
    float acc[RT][2] = {};
    for (int k = 0; k < D2; ++k) {
      float xr[RT];
      lg_rows<RT>(&xt[k * R + RT * rg], xr);
      const float2 wv = *reinterpret_cast<const float2*>(&w2[k * D + 2 * cg]);
#pragma unroll
      for (int r = 0; r < RT; ++r) { acc[r][0] = fmaf(xr[r], wv.x, acc[r][0]); acc[r][1] = fmaf(xr[r], wv.y, acc[r][1]); }
    }
    const float b0 = vec[224 + 2 * cg], b1 = vec[224 + 2 * cg + 1];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const int row = row0 + RT * rg + r;
      float2 nx = make_float2(0.f, 0.f);
      if (row < a.rows) {
        nx = make_float2(xold[r][0] + acc[r][0] + b0, xold[r][1] + acc[r][1] + b1);
        *reinterpret_cast<float2*>(a.x + (size_t)row * D + 2 * cg) = nx;
      }
      if (a.nn) {                    // updated x, transposed, for stage 4 (ctx^T is no longer needed)
        ct[(2 * cg) * R + RT * rg + r] = nx.x;
        ct[(2 * cg + 1) * R + RT * rg + r] = nx.y;
      }
    }
  }
#if defined(LG_ABL) && LG_ABL == 7
  if (a.rows > 0) return;
#endif
  if (!a.nn) return;
  __syncthreads();
  // ---- stage 4: the next projection of the updated rows (same arithmetic and order as lg_linear_kernel) ----
  // (compile-time width: with nn a run-time value the k loop kept its address arithmetic and three predicated reads per
  // step and took 3.5 us of the kernel's 25 for a fifth of its FMAs, profiles/r3_ab_lg_tail_stages.txt)
  auto stage4 = [&](auto njc) {
    constexpr int NJ = decltype(njc)::value, NN = 32 * NJ;
    float acc[RT][2 * NJ];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int c = 0; c < 2 * NJ; ++c) acc[r][c] = 0.f;
#pragma unroll 8
    for (int k = 0; k < D; ++k) {
      float xr[RT];
      lg_rows<RT>(&ct[k * R + RT * rg], xr);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float2 wv = *reinterpret_cast<const float2*>(&wn[k * NN + 2 * cg + 32 * j]);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          acc[r][2 * j] = fmaf(xr[r], wv.x, acc[r][2 * j]);
          acc[r][2 * j + 1] = fmaf(xr[r], wv.y, acc[r][2 * j + 1]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = 2 * cg + 32 * j;
      const float b0 = vec[256 + c], b1 = vec[256 + c + 1];
      const bool rot = a.cs && c < a.rot_cols;
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int row = row0 + RT * rg + r;
        if (row >= a.rows || c >= a.nvalid) continue;
        float y0 = acc[r][2 * j] + b0, y1 = acc[r][2 * j + 1] + b1;
        if (rot) {
          const float co = rco[r], si = rsi[r];
          const float t0 = y0, t1 = y1;
          y0 = t0 * co - t1 * si;
          y1 = t1 * co + t0 * si;
        }
        float* o = a.on + (size_t)row * a.nos + c;
        if (c + 1 < a.nvalid) *reinterpret_cast<float2*>(o) = make_float2(y0, y1);      // (nos and c are even)
        else o[0] = y0;
      }
    }
  };
  if (a.nn == 96) stage4(std::integral_constant<int, 3>{});
  else stage4(std::integral_constant<int, 2>{});
}

int launch_lg_tail(const LgTailArgs& a, hipStream_t s) {
  if (a.D != 32) return -1804;
  if (a.nn && (a.nn != 64 && a.nn != 96)) return -1805;
  if (a.nn && (!a.wn || !a.on || a.nvalid < 1 || a.nvalid > a.nn || (a.cs && ((a.hd & 1) || (a.rot_cols & 1) || 32 % a.hd)) || (a.nos & 1))) return -1805;
  static const int forced = getenv("KP2D_LG_TAIL_RT") ? atoi(getenv("KP2D_LG_TAIL_RT")) : 0;
  const int rt = forced ? forced : a.rows >= 64 * 256 ? 4 : a.rows >= 32 * 256 ? 2 : 1;      // (8 pairs x 2048 rows: 0.547 / 0.595 / 0.569 ms per forward at RT = 4 / 2 / 1)
  const size_t lds = (size_t)(32 * 32 + 64 * 64 + 64 * 32 + 32 * 96 + LG_TAIL_VEC + (32 + 64) * 16 * rt) * sizeof(float);
  static PerDeviceOnce once[3];       // (67 KB at RT = 4: above the 64 KB a kernel gets without opting in)
  const void* fn = rt == 4 ? (const void*)&lg_tail_kernel<4> : rt == 2 ? (const void*)&lg_tail_kernel<2> : (const void*)&lg_tail_kernel<1>;
  if (int e = lds_opt_in(once[rt >> 1], fn)) return e;
  const dim3 grid((a.rows + 16 * rt - 1) / (16 * rt));
  switch (rt) {
    case 4: hipLaunchKernelGGL(lg_tail_kernel<4>, grid, dim3(256), lds, s, a); break;
    case 2: hipLaunchKernelGGL(lg_tail_kernel<2>, grid, dim3(256), lds, s, a); break;
    case 1: hipLaunchKernelGGL(lg_tail_kernel<1>, grid, dim3(256), lds, s, a); break;
    default: return -1806;
  }
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

// thread (ty = tid/16, tx = tid%16) owns rows i0 + 4 ty + r, columns j0 + 4 tx + c of the tile
__global__ __launch_bounds__(256) void lg_sim_kernel(const LgAssignArgs a) {
  __shared__ __attribute__((aligned(16))) float at[64 * 64], bt[64 * 64];   // [k][row]
  const int b = blockIdx.z, i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int tid = threadIdx.x, D = a.D, M = a.M, N = a.N;
  const float* f0 = a.fz + (size_t)b * M * a.fs;
  const float* f1 = a.fz + ((size_t)a.B * M + (size_t)b * N) * a.fs;
  for (int e = tid; e < 64 * D; e += 256) {
    const int r = e / D, k = e - r * D;
    at[k * 64 + r] = i0 + r < M ? f0[(size_t)(i0 + r) * a.fs + k] : 0.f;
    bt[k * 64 + r] = j0 + r < N ? f1[(size_t)(j0 + r) * a.fs + k] : 0.f;
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
  float* sc = a.scores + (size_t)b * (M + 1) * (N + 1);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = i0 + 4 * ty + r, j = j0 + 4 * tx + c;
      if (i < M && j < N) sc[(size_t)i * (N + 1) + j] = acc[r][c];
      else acc[r][c] = -INFINITY;            // outside the inner block: no weight in the sums below
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

__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

// one 64 x 64 tile of scores: inner block <- log_softmax rows + log_softmax cols + certainties
// (sigmoid_log_double_softmax, :363-376); the tiles of the first tile column / row also write the border column / row
// (logsigmoid(-z)); plus the tile's row / column max and argmax of the final values for filter_matches (:403-404)
__global__ __launch_bounds__(256) void lg_finalize_kernel(const LgAssignArgs a) {
  __shared__ float s_base[64], s_col[64], s_ls1[64];
  __shared__ float s_cm[4][64];
  __shared__ int s_ci[4][64];
  const int b = blockIdx.z, i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int tid = threadIdx.x, M = a.M, N = a.N, TN = gridDim.x, TM = gridDim.y;
  float* sc = a.scores + (size_t)b * (M + 1) * (N + 1);
  const float* z0p = a.fz + (size_t)b * M * a.fs + a.D;                          // matchability logits of image 0
  const float* z1p = a.fz + ((size_t)a.B * M + (size_t)b * N) * a.fs + a.D;      // ... of image 1
  if (tid < 64) {
    const int i = i0 + tid;
    float base = 0.f;
    if (i < M) {
      float m = -INFINITY, s = 0.f;
      for (int t = 0; t < TN; ++t) {
        const size_t e = ((size_t)b * TN + t) * M + i;
        lse_merge(m, s, a.rp_m[e], a.rp_s[e]);
      }
      const float z0 = z0p[(size_t)i * a.fs];
      base = log_sigmoid(z0) - (m + logf(s));
      if (blockIdx.x == 0) sc[(size_t)i * (N + 1) + N] = log_sigmoid(-z0);
    }
    s_base[tid] = base;
  } else if (tid < 128) {
    const int c = tid - 64, j = j0 + c;
    float cl = 0.f, ls1 = 0.f;
    if (j < N) {
      float m = -INFINITY, s = 0.f;
      for (int t = 0; t < TM; ++t) {
        const size_t e = ((size_t)b * TM + t) * N + j;
        lse_merge(m, s, a.cp_m[e], a.cp_s[e]);
      }
      cl = m + logf(s);
      const float z1 = z1p[(size_t)j * a.fs];
      ls1 = log_sigmoid(z1);
      if (blockIdx.y == 0) sc[(size_t)M * (N + 1) + j] = log_sigmoid(-z1);
    }
    s_col[c] = cl; s_ls1[c] = ls1;
  } else if (tid == 128 && blockIdx.x == 0 && blockIdx.y == 0) {
    sc[(size_t)M * (N + 1) + N] = 0.f;
  }
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
        float* p = sc + (size_t)i * (N + 1) + j;
        x = 2.f * *p + base - s_col[4 * tx + c] + s_ls1[4 * tx + c];
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
    mx = -INFINITY;
    int arg = 0;
    for (int t = 0; t < TN; ++t) {
      const size_t k = ((size_t)b * TN + t) * M + i;
      const float v = a.rmax[k];
      if (v > mx) { mx = v; arg = a.rarg[k]; }
    }
    return arg;
  };
  auto col_best = [&](int j) {
    float mx = -INFINITY;
    int arg = 0;
    for (int t = 0; t < TM; ++t) {
      const size_t k = ((size_t)b * TM + t) * N + j;
      const float v = a.cmax[k];
      if (v > mx) { mx = v; arg = a.carg[k]; }
    }
    return arg;
  };
  if (e < M) {
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
  if (a.D < 1 || a.D > 64 || a.M < 1 || a.N < 1) return -1803;
  const dim3 tiles((a.N + 63) / 64, (a.M + 63) / 64, a.B);
  hipLaunchKernelGGL(lg_sim_kernel, tiles, dim3(256), 0, s, a);
  hipLaunchKernelGGL(lg_finalize_kernel, tiles, dim3(256), 0, s, a);
  hipLaunchKernelGGL(lg_filter_kernel, dim3((a.M + a.N + 255) / 256, a.B), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace kp2d
