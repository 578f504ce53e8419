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
//  lg_sim_kernel        sim = f0 f1^T (:391)
//  lg_rowlse / lg_col   row / column log-sum-exp of sim, column argmax of the final scores
//  lg_finalize_kernel   sigmoid_log_double_softmax (:363-376) + row max/argmax (filter_matches :403-404)
//  lg_filter_kernel     mutual check, threshold, match scores (:405-416)
#include "kp2d_kernels.h"
#include "device_guard.h"

namespace kp2d {

// one workgroup per (image set, batch item): keypoints [n][2] -> cs[row][hd] = (cos f0..f_{hd/2-1} | sin ...)
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
  for (int e = tid; e < n * hf; e += 256) {
    const int i = e / hf, f = e - i * hf;
    const float x = (k[2 * i] - sx / 2.f) / scale, y = (k[2 * i + 1] - sy / 2.f) / scale;
    const float pr = a.wr[2 * f] * x + a.wr[2 * f + 1] * y;
    cs[(size_t)i * a.hd + f] = cosf(pr);
    cs[(size_t)i * a.hd + hf + f] = sinf(pr);
  }
}

int launch_lg_posenc(const LgPosArgs& a, hipStream_t s) {
  if (a.hd < 2 || (a.hd & 1)) return -1800;
  hipLaunchKernelGGL(lg_posenc_kernel, dim3(a.B, 2), dim3(256), 0, s, a);
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

// ---------------------------------------------------------------------------------------------
// Fused block tail for D = 32: message = out_proj(ctx); h = GELU(LayerNorm(W1 [x | message] + b1)); x += W2 h + b2.
// All three products are row-local, so one workgroup carries its 64 rows through them with the intermediate tiles
// (transposed, [k][row]) in LDS: 3 launches and 2 HBM round trips of the message / hidden tensors less per block.
// Same thread mapping as lg_linear_kernel (thread = 4 rows x column pairs {2cg + 32j, +1}).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lg_tail_kernel(const LgTailArgs a) {
  constexpr int D = 32, D2 = 64, R = LG_ROWS;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wo = sm;                  // [D][D]
  float* w1 = wo + D * D;          // [D2][D2]
  float* w2 = w1 + D2 * D2;        // [D2][D]
  float* ct = w2 + D2 * D;         // [D][R]   ctx^T
  float* xt = ct + D * R;          // [D2][R]  (x | message)^T, later h^T
  const int tid = threadIdx.x, rg = tid >> 4, cg = tid & 15;
  const int row0 = blockIdx.x * R;
  for (int e = tid; e < (D * D) >> 2; e += 256) reinterpret_cast<float4*>(wo)[e] = reinterpret_cast<const float4*>(a.wo)[e];
  for (int e = tid; e < (D2 * D2) >> 2; e += 256) reinterpret_cast<float4*>(w1)[e] = reinterpret_cast<const float4*>(a.w1)[e];
  for (int e = tid; e < (D2 * D) >> 2; e += 256) reinterpret_cast<float4*>(w2)[e] = reinterpret_cast<const float4*>(a.w2)[e];
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
  // ---- stage 1: message[64 x 32] = ctx Wo^T + bo -> xt rows D..2D ----
  {
    float acc[4][2] = {};
    for (int k = 0; k < D; ++k) {
      const float4 xv = *reinterpret_cast<const float4*>(&ct[k * R + 4 * rg]);
      const float2 wv = *reinterpret_cast<const float2*>(&wo[k * D + 2 * cg]);
      const float xr[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[r][0] = fmaf(xr[r], wv.x, acc[r][0]); acc[r][1] = fmaf(xr[r], wv.y, acc[r][1]); }
    }
    const float b0 = a.bo[2 * cg], b1 = a.bo[2 * cg + 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xt[(D + 2 * cg) * R + 4 * rg + r] = acc[r][0] + b0;
      xt[(D + 2 * cg + 1) * R + 4 * rg + r] = acc[r][1] + b1;
    }
  }
  __syncthreads();
  // ---- stage 2: h[64 x 64] = GELU(LayerNorm([x | message] W1^T + b1)) ----
  float h[4][4];
  {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) h[r][c] = 0.f;
    for (int k = 0; k < D2; ++k) {
      const float4 xv = *reinterpret_cast<const float4*>(&xt[k * R + 4 * rg]);
      const float xr[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float2 wv = *reinterpret_cast<const float2*>(&w1[k * D2 + 2 * cg + 32 * j]);
#pragma unroll
        for (int r = 0; r < 4; ++r) { h[r][2 * j] = fmaf(xr[r], wv.x, h[r][2 * j]); h[r][2 * j + 1] = fmaf(xr[r], wv.y, h[r][2 * j + 1]); }
      }
    }
    float bb[4], gg[4], be[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = 2 * cg + 32 * (c >> 1) + (c & 1);
      bb[c] = a.b1[col]; gg[c] = a.ln_g[col]; be[c] = a.ln_b[col];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
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
        h[r][c] = 0.5f * y * (1.f + erff(y * 0.70710678118654752f));
      }
    }
  }
  __syncthreads();                 // every thread is done reading (x | message)^T and W1
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) xt[(2 * cg + 32 * (c >> 1) + (c & 1)) * R + 4 * rg + r] = h[r][c];
  // the next projection's weights take W1's place (D x nn <= 32 x 96 floats)
  if (a.nn)
    for (int e = tid; e < (D * a.nn) >> 2; e += 256) reinterpret_cast<float4*>(w1)[e] = reinterpret_cast<const float4*>(a.wn)[e];
  __syncthreads();
  // ---- stage 3: x += h W2^T + b2 ----
  {
    float acc[4][2] = {};
    for (int k = 0; k < D2; ++k) {
      const float4 xv = *reinterpret_cast<const float4*>(&xt[k * R + 4 * rg]);
      const float2 wv = *reinterpret_cast<const float2*>(&w2[k * D + 2 * cg]);
      const float xr[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[r][0] = fmaf(xr[r], wv.x, acc[r][0]); acc[r][1] = fmaf(xr[r], wv.y, acc[r][1]); }
    }
    const float b0 = a.b2[2 * cg], b1 = a.b2[2 * cg + 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 4 * rg + r;
      float2 nx = make_float2(0.f, 0.f);
      if (row < a.rows) {
        float2* xp = reinterpret_cast<float2*>(a.x + (size_t)row * D + 2 * cg);
        const float2 old = *xp;
        nx = make_float2(old.x + acc[r][0] + b0, old.y + acc[r][1] + b1);
        *xp = nx;
      }
      if (a.nn) {                    // updated x, transposed, for stage 4 (ctx^T is no longer needed)
        ct[(2 * cg) * R + 4 * rg + r] = nx.x;
        ct[(2 * cg + 1) * R + 4 * rg + r] = nx.y;
      }
    }
  }
  if (!a.nn) return;
  __syncthreads();
  // ---- stage 4: the next projection of the updated rows (same arithmetic and order as lg_linear_kernel) ----
  {
    const int nn = a.nn, nj = nn >> 5;
    float acc[4][6];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[r][c] = 0.f;
    for (int k = 0; k < D; ++k) {
      const float4 xv = *reinterpret_cast<const float4*>(&ct[k * R + 4 * rg]);
      const float xr[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (j < nj) {
          const float2 wv = *reinterpret_cast<const float2*>(&w1[k * nn + 2 * cg + 32 * j]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[r][2 * j] = fmaf(xr[r], wv.x, acc[r][2 * j]);
            acc[r][2 * j + 1] = fmaf(xr[r], wv.y, acc[r][2 * j + 1]);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (j < nj) {
        const int c = 2 * cg + 32 * j;
        const float b0 = a.bn ? a.bn[c] : 0.f, b1 = a.bn ? a.bn[c + 1] : 0.f;
        const bool rot = a.cs && c < a.rot_cols;
        const int hf = a.hd >> 1, f = rot ? (c % a.hd) >> 1 : 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + 4 * rg + r;
          if (row >= a.rows || c >= a.nvalid) continue;
          float y0 = acc[r][2 * j] + b0, y1 = acc[r][2 * j + 1] + b1;
          if (rot) {
            const float co = a.cs[(size_t)row * a.hd + f], si = a.cs[(size_t)row * a.hd + hf + f];
            const float t0 = y0, t1 = y1;
            y0 = t0 * co - t1 * si;
            y1 = t1 * co + t0 * si;
          }
          float* o = a.on + (size_t)row * a.nos + c;
          o[0] = y0;
          if (c + 1 < a.nvalid) o[1] = y1;
        }
      }
  }
}

int launch_lg_tail(const LgTailArgs& a, hipStream_t s) {
  if (a.D != 32) return -1804;
  if (a.nn && (a.nn != 64 && a.nn != 96)) return -1805;
  if (a.nn && (!a.wn || !a.on || a.nvalid < 1 || a.nvalid > a.nn || (a.cs && ((a.hd & 1) || (a.rot_cols & 1))))) return -1805;
  const size_t lds = (size_t)(32 * 32 + 64 * 64 + 64 * 32 + 32 * LG_ROWS + 64 * LG_ROWS) * sizeof(float);
  hipLaunchKernelGGL(lg_tail_kernel, dim3((a.rows + LG_ROWS - 1) / LG_ROWS), dim3(256), lds, s, a);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Assignment.  scores is the reference's [B][M+1][N+1] log-assignment tensor; its inner block first holds sim.
// ---------------------------------------------------------------------------------------------
// sim tile 64 x 64 per workgroup; thread (ty = tid/16, tx = tid%16) owns a 4 x 4 block
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
    }
}

__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
  const float mm = fmaxf(m, m2);
  if (mm == -INFINITY) { m = mm; s = 0.f; return; }
  s = s * expf(m - mm) + s2 * expf(m2 - mm);
  m = mm;
}

// one wave per row i < M: log-sum-exp over the N columns of sim
__global__ __launch_bounds__(256) void lg_rowlse_kernel(const LgAssignArgs a) {
  const int lane = threadIdx.x & 63, M = a.M, N = a.N;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)a.B * M) return;
  const int b = (int)(row / M), i = (int)(row - (long)b * M);
  const float* sc = a.scores + (size_t)b * (M + 1) * (N + 1) + (size_t)i * (N + 1);
  float m = -INFINITY, s = 0.f;
  for (int j = lane; j < N; j += 64) lse_merge(m, s, sc[j], 1.f);
  for (int o = 32; o > 0; o >>= 1) lse_merge(m, s, __shfl_xor(m, o), __shfl_xor(s, o));
  if (lane == 0) a.rlse[row] = m + logf(s);
}

// 64 columns x 4 row slices per workgroup.  MODE 0: column log-sum-exp of sim -> clse.
// MODE 1: column max / argmax (lowest row on ties) of the final scores -> m1.
template <int MODE>
__global__ __launch_bounds__(256) void lg_col_kernel(const LgAssignArgs a) {
  __shared__ float sm_m[4][64], sm_s[4][64];
  __shared__ int sm_i[4][64];
  const int b = blockIdx.y, M = a.M, N = a.N;
  const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  const float* sc = a.scores + (size_t)b * (M + 1) * (N + 1);
  float m = -INFINITY, s = 0.f;
  int arg = 0;
  if (j < N) {
    // four rows in flight per thread: the column walk is a chain of dependent, latency-bound loads otherwise
    int i = sl;
    for (; i + 12 < M; i += 16) {
      const float v0 = sc[(size_t)i * (N + 1) + j], v1 = sc[(size_t)(i + 4) * (N + 1) + j];
      const float v2 = sc[(size_t)(i + 8) * (N + 1) + j], v3 = sc[(size_t)(i + 12) * (N + 1) + j];
      if (MODE == 0) {
        const float mm = fmaxf(fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)), m);
        s = s * expf(m - mm) + expf(v0 - mm) + expf(v1 - mm) + expf(v2 - mm) + expf(v3 - mm);
        m = mm;
      } else {
        if (v0 > m) { m = v0; arg = i; }
        if (v1 > m) { m = v1; arg = i + 4; }
        if (v2 > m) { m = v2; arg = i + 8; }
        if (v3 > m) { m = v3; arg = i + 12; }
      }
    }
    for (; i < M; i += 4) {
      const float v = sc[(size_t)i * (N + 1) + j];
      if (MODE == 0) lse_merge(m, s, v, 1.f);
      else if (v > m) { m = v; arg = i; }
    }
  }
  sm_m[sl][c] = m; sm_s[sl][c] = s; sm_i[sl][c] = arg;
  __syncthreads();
  if (sl == 0 && j < N) {
    for (int t = 1; t < 4; ++t) {
      if (MODE == 0) lse_merge(m, s, sm_m[t][c], sm_s[t][c]);
      else if (sm_m[t][c] > m || (sm_m[t][c] == m && sm_i[t][c] < arg)) { m = sm_m[t][c]; arg = sm_i[t][c]; }
    }
    if (MODE == 0) a.clse[(size_t)b * N + j] = m + logf(s);
    else a.m1[(size_t)b * N + j] = arg;
  }
}

__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

// one wave per row i <= M of scores: inner block <- log_softmax rows + log_softmax cols + certainties, last column /
// last row <- logsigmoid(-z); also the row max / argmax of the inner block (lowest column on ties)
__global__ __launch_bounds__(256) void lg_finalize_kernel(const LgAssignArgs a) {
  const int lane = threadIdx.x & 63, M = a.M, N = a.N;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)a.B * (M + 1)) return;
  const int b = (int)(row / (M + 1)), i = (int)(row - (long)b * (M + 1));
  float* sc = a.scores + (size_t)b * (M + 1) * (N + 1) + (size_t)i * (N + 1);
  const float* z1 = a.fz + ((size_t)a.B * M + (size_t)b * N) * a.fs + a.D;     // matchability column of image 1
  if (i == M) {
    for (int j = lane; j < N; j += 64) sc[j] = log_sigmoid(-z1[(size_t)j * a.fs]);
    if (lane == 0) sc[N] = 0.f;
    return;
  }
  const float z0 = a.fz[((size_t)b * M + i) * a.fs + a.D];
  const float base = log_sigmoid(z0) - a.rlse[(size_t)b * M + i];
  const float* cl = a.clse + (size_t)b * N;
  float mx = -INFINITY;
  int arg = 0;
  for (int j = lane; j < N; j += 64) {
    const float v = 2.f * sc[j] + base - cl[j] + log_sigmoid(z1[(size_t)j * a.fs]);
    sc[j] = v;
    if (v > mx) { mx = v; arg = j; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(mx, o);
    const int a2 = __shfl_xor(arg, o);
    if (m2 > mx || (m2 == mx && a2 < arg)) { mx = m2; arg = a2; }
  }
  if (lane == 0) {
    sc[N] = log_sigmoid(-z0);
    a.max0[(size_t)b * M + i] = mx;
    a.m0[(size_t)b * M + i] = arg;
  }
}

// filter_matches (:401-416): thread e < M handles row e, thread e >= M column e - M
__global__ __launch_bounds__(256) void lg_filter_kernel(const LgAssignArgs a) {
  const int b = blockIdx.y, M = a.M, N = a.N;
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int* m0 = a.m0 + (size_t)b * M;
  const int* m1 = a.m1 + (size_t)b * N;
  const float* mx = a.max0 + (size_t)b * M;
  if (e < M) {
    const int j = m0[e];
    const bool mutual = m1[j] == e;
    const float ms = mutual ? expf(mx[e]) : 0.f;
    a.mscores0[(size_t)b * M + e] = ms;
    a.matches0[(size_t)b * M + e] = (mutual && ms > a.th) ? (int64_t)j : (int64_t)-1;
  } else if (e < M + N) {
    const int j = e - M, i = m1[j];
    const bool mutual1 = m0[i] == j;                 // then m1[m0[i]] == i as well: row i is mutual with column j
    const float ms0 = mutual1 ? expf(mx[i]) : 0.f;
    a.mscores1[(size_t)b * N + j] = ms0;
    a.matches1[(size_t)b * N + j] = (mutual1 && ms0 > a.th) ? (int64_t)i : (int64_t)-1;
  }
}

int launch_lg_assign(const LgAssignArgs& a, hipStream_t s) {
  if (a.D < 1 || a.D > 64 || a.M < 1 || a.N < 1) return -1803;
  hipLaunchKernelGGL(lg_sim_kernel, dim3((a.N + 63) / 64, (a.M + 63) / 64, a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(lg_rowlse_kernel, dim3((int)(((long)a.B * a.M + 3) / 4)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(lg_col_kernel<0>, dim3((a.N + 63) / 64, a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(lg_finalize_kernel, dim3((int)(((long)a.B * (a.M + 1) + 3) / 4)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(lg_col_kernel<1>, dim3((a.N + 63) / 64, a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(lg_filter_kernel, dim3((a.M + a.N + 255) / 256, a.B), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace kp2d
