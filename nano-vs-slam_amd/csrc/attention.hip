// Kernels of the SegFormerAttentionModule used by the attention segmentation heads
// (modules/segformer.py:209-220: PreNorm(EfficientSelfAttention) then PreNorm(MixFeedForward), no residuals).
//
//  * channel_layernorm_kernel  custom LayerNorm over channels, eps added to the STD (segformer.py:63-73)
//  * attention_kernel          softmax(q k^T * d^-0.5) v per head (segformer.py:118-131) as a streaming
//                              (flash-style) kernel on the fp32 matrix cores: the S x T score matrix the
//                              reference materialises (92 MB/frame at 240x320, 1.47 GB at 480x640) is never
//                              written; K/V are walked in 64-key LDS chunks with an online softmax.
//  * dwconv3x3_kernel          depthwise 3x3 + bias of MixFeedForward's DsConv2d (segformer.py:45-59)
// The 1x1 convolutions (to_q, to_out, MixFFN) and the 2x2 stride-2 to_kv run through conv3x3.hip with taps = 1.
#include "kp2d_kernels.h"

namespace kp2d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave per pixel, lane = channels lane, lane+64, ... (C <= 64*NV): the channel reduction is a 6-step butterfly
template <int NV>
__global__ __launch_bounds__(256) void channel_layernorm_kernel(const LnArgs a) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwave = (long)gridDim.x * 4;
  const int C = a.C;
  float g[NV], bb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const bool on = lane + 64 * i < C;
    g[i] = on ? a.g[lane + 64 * i] : 0.f;
    bb[i] = on ? a.b[lane + 64 * i] : 0.f;
  }
  const float invC = 1.f / (float)C;
  for (long p = wave; p < a.npix; p += nwave) {
    float v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] = lane + 64 * i < C ? a.x[p * C + lane + 64 * i] : 0.f;
      s += v[i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] = lane + 64 * i < C ? v[i] - mean : 0.f;
      q += v[i] * v[i];
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float stdv = sqrtf(q * invC);          // torch.var(unbiased=False).sqrt()
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < C) a.y[p * C + lane + 64 * i] = v[i] / (stdv + 1e-5f) * g[i] + bb[i];
  }
}

int launch_channel_layernorm(const LnArgs& a, hipStream_t s) {
  if (a.C > 256 || a.C < 1) return -1400;
  long blocks = (a.npix + 3) / 4;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (a.C <= 64) hipLaunchKernelGGL(channel_layernorm_kernel<1>, dim3((int)blocks), dim3(256), 0, s, a);
  else if (a.C <= 128) hipLaunchKernelGGL(channel_layernorm_kernel<2>, dim3((int)blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(channel_layernorm_kernel<4>, dim3((int)blocks), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

// depthwise 3x3, pad 1, bias; NHWC, one thread per (pixel, 4 channels)
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const DwArgs a) {
  const int C = a.C, Q = C >> 2, H = a.H, W = a.W;
  const long total = (long)a.B * H * W * Q;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int q = (int)(e % Q);
  const long pix = e / Q;
  const int x = (int)(pix % W);
  const int y = (int)((pix / W) % H);
  const float4 bias = reinterpret_cast<const float4*>(a.bias)[q];
  float4 acc = bias;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int xx = x + dx;
      if (xx < 0 || xx >= W) continue;
      const float4 v = reinterpret_cast<const float4*>(a.x + (pix + (long)dy * W + dx) * C)[q];
      const float4 w = reinterpret_cast<const float4*>(a.w + ((dy + 1) * 3 + (dx + 1)) * C)[q];
      acc.x = fmaf(v.x, w.x, acc.x); acc.y = fmaf(v.y, w.y, acc.y);
      acc.z = fmaf(v.z, w.z, acc.z); acc.w = fmaf(v.w, w.w, acc.w);
    }
  }
  reinterpret_cast<float4*>(a.y + pix * C)[q] = acc;
}

int launch_dwconv3x3(const DwArgs& a, hipStream_t s) {
  if (a.C & 3) return -1401;
  const long total = (long)a.B * a.H * a.W * (a.C >> 2);
  hipLaunchKernelGGL(dwconv3x3_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Streaming attention, exact fp32 on v_mfma_f32_16x16x4_f32.
//   grid (ceil(S/64), heads, B); 4 waves, each owns 16 queries.  Head dim d <= 64 (padded to 16-channel blocks).
//   S^T tile = K_tile (16 keys x d) . Q^T (d x 16 queries): D[row = key 4g+reg][col = query lane&15], so a
//   lane holds 4 keys of ONE query -> the softmax row reduction is 4 registers + two cross-lane steps.
//   P^T in that same register layout is directly the B operand of O^T += V^T . P^T when the key order of
//   the k-steps is taken as (4g + r): no LDS round trip, no transposition for P.
//   (MFMA k index = lane>>4 = g; step r pairs A[.][g] with B[g][.], so any bijection (g,r) -> channel/key
//   works as long as both operands use the same one.)
// ---------------------------------------------------------------------------------------------
constexpr int AKT = 64;   // keys per LDS chunk

// DB = 16-channel blocks of the head dimension (d <= 16*DB, zero-padded): 1 for the S/N widths, 4 for LARGE_D (d = 64)
template <int DB>
__global__ __launch_bounds__(256) void attention_kernel(const AttnArgs a) {
  constexpr int ADP = 16 * DB + 4;   // padded row (floats): 16-byte aligned, conflict-light for b128/b32 reads
  constexpr int DQ = 4 * DB;         // float4 per padded row
  __shared__ __attribute__((aligned(16))) float Ks[AKT * ADP];
  __shared__ __attribute__((aligned(16))) float Vs[AKT * ADP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const int C = a.C, d = C / a.heads, S = a.S, T = a.T;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const int qi = q0 + j;
  const int qs = a.q_stride ? a.q_stride : C, kvs = a.kv_stride ? a.kv_stride : 2 * C;
  const int vd = (a.v_off ? a.v_off : C) - a.k_off, os = a.out_stride ? a.out_stride : C;
  const float* qp = a.q + ((size_t)b * S + (qi < S ? qi : 0)) * qs + h * d;
  float qf[DB][4];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * db + 4 * g + r;
      qf[db][r] = (qi < S && c < d) ? qp[c] : 0.f;
    }

  f32x4 o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  const float* kvb = a.kv + (size_t)b * T * kvs + a.k_off + h * d;

  for (int kc = 0; kc < T; kc += AKT) {
    __syncthreads();
    for (int e = tid; e < AKT * DQ; e += 256) {
      const int skey = e / DQ, squad = e % DQ;
      float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;
      const int key = kc + skey;
      if (key < T && 4 * squad < d) {
        const float* p = kvb + (size_t)key * kvs + 4 * squad;
        kk = *reinterpret_cast<const float4*>(p);
        vv = *reinterpret_cast<const float4*>(p + vd);
      }
      *reinterpret_cast<float4*>(&Ks[skey * ADP + 4 * squad]) = kk;
      *reinterpret_cast<float4*>(&Vs[skey * ADP + 4 * squad]) = vv;
    }
    __syncthreads();
    f32x4 sc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const float4 kf = *reinterpret_cast<const float4*>(&Ks[(16 * t + j) * ADP + 16 * db + 4 * g]);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[db][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[db][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[db][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[db][3], acc, 0, 0, 0);
      }
      sc[t] = acc;
    }
    float mx = m;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = sc[t][r] * a.scale;
        if (kc + 16 * t + 4 * g + r >= T) v = -INFINITY;
        sc[t][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float alpha = expf(m - mx);       // first chunk: exp(-inf) = 0
    l *= alpha;
#pragma unroll
    for (int db = 0; db < DB; ++db) { o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha; }
    m = mx;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = expf(sc[t][r] - mx);
        l += p;
#pragma unroll
        for (int db = 0; db < DB; ++db)
          o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[(16 * t + 4 * g + r) * ADP + 16 * db + j], p, o[db], 0, 0, 0);
      }
  }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (qi < S) {
    const float inv = 1.f / l;
    float* op = a.out + ((size_t)b * S + qi) * os + h * d;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * db + 4 * g + r < d) op[16 * db + 4 * g + r] = o[db][r] * inv;
  }
}

int launch_attention(const AttnArgs& a, hipStream_t s) {
  const int d = a.C / a.heads;
  if (a.C % a.heads || d > 64 || (d & 3) || (a.C & 3)) return -1402;
  const dim3 grid((a.S + 63) / 64, a.heads, a.B);
  if (d <= 16) hipLaunchKernelGGL(attention_kernel<1>, grid, dim3(256), 0, s, a);
  else if (d <= 32) hipLaunchKernelGGL(attention_kernel<2>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(attention_kernel<4>, grid, dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace kp2d
