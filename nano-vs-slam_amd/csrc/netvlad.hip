// NetVLAD pooling (modules/aggregators/netvlad.py:79-106), restated so the [K][C][S] residual tensor
// (78.6 MB / frame at 240x320) is never formed:
//     xh      = x / max(||x||_C, 1e-12)                          (F.normalize, :83)
//     a[s,k]  = softmax_k( sum_c Wa[k,c] xh[s,c] )               (1x1 conv without bias + softmax, :86-89)
//     V[k,c]  = sum_s a[s,k] xh[s,c]  -  (sum_s a[s,k]) cent[k,c]   (== sum_s a (xh - cent), :94-100)
//     V[k,:] /= max(||V[k,:]||, 1e-12); v = flatten(V) (k-major); v /= max(||v||, 1e-12)   (:102-104)
//
// Two launches, both with a fixed summation order (bit-reproducible, no atomics):
//   pass 1: grid (nsplit, B); each workgroup walks its pixel slab in 64-pixel tiles held in LDS and
//           keeps its K*C partial sums in registers; writes one partial [K*C + K] per (frame, split).
//   pass 2: grid (B); sums the partials in split order, subtracts rowsum * centroid, does both
//           normalisations with wavefront shuffles + one LDS exchange.
// The input is the NHWC output of vlad_head.convlad3, so a pixel's C channels are one contiguous line.
#include <cstdlib>

#include "kp2d_kernels.h"
#include "device_guard.h"

namespace kp2d {

constexpr int VT = 64;  // pixels per LDS tile

int netvlad_nsplit(int S) {
  // Depends on the frame size only: the order of the partial sums (and with it the last bits of the descriptor)
  // must not change with the batch or sub-batch a frame happens to travel in (test_full_size_properties).
  static const int per = getenv("KP2D_VLAD_PX") ? atoi(getenv("KP2D_VLAD_PX")) : 320;   // pixels per workgroup (tuning knob)
  int n = (S + per - 1) / per;
  return n < 1 ? 1 : n;
}

// Workgroups per slab.  1: a workgroup walks its whole slab.  > 1 (few frames per call, where nsplit workgroups per
// frame would leave the chip idle): one workgroup per 64-pixel tile of the slab; the finish pass then adds the
// tile partials of a slab in tile order before it adds the slabs — the same additions in the same order as a
// workgroup walking the slab, so the result does not depend on which mode a frame was processed in.
int netvlad_tiles_per_slab(int S, int B) {
  const int ns = netvlad_nsplit(S);
  if ((long)B * ns >= 256) return 1;
  const int per = (S + ns - 1) / ns;
  return (per + VT - 1) / VT;
}

// NOWN = (cluster, channel) accumulators per thread: K*C <= 256*NOWN
template <int NOWN>
__global__ __launch_bounds__(256) void netvlad_partial_kernel(const VladArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = a.C, K = a.K, S = a.S;
  const int CP = C + 1, KP = K + 1;
  float* s_w = sm;                 // [K][CP]
  float* s_x = s_w + K * CP;       // [VT][CP]
  float* s_a = s_x + VT * CP;      // [VT][KP]
  const int tid = threadIdx.x;
  const int b = blockIdx.y, split = blockIdx.x;
  const int per = (S + a.nsplit - 1) / a.nsplit;
  const int tps = a.tps > 1 ? a.tps : 1;                 // tile mode: blockIdx.x = slab * tps + tile
  const int slab = split / tps;
  const int s_begin = slab * per + (split - slab * tps) * (a.tps > 1 ? VT : 0);
  const int s_end = a.tps > 1 ? min(min(S, slab * per + per), s_begin + VT) : min(S, s_begin + per);

  for (int e = tid; e < K * C; e += 256) s_w[(e / C) * CP + (e % C)] = a.wa[e];

  const int KC_ = K * C;
  const int nown = (KC_ + 255) / 256;          // <= NOWN
  float vacc[NOWN];
  int vk[NOWN], vc[NOWN];
#pragma unroll
  for (int j = 0; j < NOWN; ++j) {
    vacc[j] = 0.f;
    const int e = tid + 256 * j;
    vk[j] = (e < KC_) ? e / C : 0;
    vc[j] = (e < KC_) ? e % C : 0;
  }
  float asum = 0.f;                             // thread t < K owns rowsum(a)[t]

  const int p = tid >> 2, q = tid & 3;          // 4 threads per pixel
  const int cq = C >> 2, kq = K >> 2;
  const float* xb = a.x + (size_t)b * S * C;

  for (int t0 = s_begin; t0 < s_end; t0 += VT) {
    const int np = min(VT, s_end - t0);
    __syncthreads();
    for (int e = tid; e < VT * C; e += 256) {
      const int pp = e / C, cc = e - pp * C;
      s_x[pp * CP + cc] = (pp < np) ? xb[(size_t)(t0 + pp) * C + cc] : 0.f;
    }
    __syncthreads();
    // descriptor-wise L2 normalisation
    float ss = 0.f;
    for (int c = q * cq; c < (q + 1) * cq; ++c) { const float v = s_x[p * CP + c]; ss = fmaf(v, v, ss); }
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    for (int c = q * cq; c < (q + 1) * cq; ++c) s_x[p * CP + c] *= inv;
    __syncthreads();
    // soft assignment: this thread owns clusters [q*kq, (q+1)*kq) of pixel p
    float lg[16];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < kq) {
        const float* wr = &s_w[(q * kq + j) * CP];
        float d = 0.f;
        for (int c = 0; c < C; ++c) d = fmaf(wr[c], s_x[p * CP + c], d);
        lg[j] = d;
        mx = fmaxf(mx, d);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < kq) { lg[j] = expf(lg[j] - mx); se += lg[j]; }
    se += __shfl_xor(se, 1);
    se += __shfl_xor(se, 2);
    const float rs = (p < np) ? 1.f / se : 0.f;   // pixels past the slab contribute nothing
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < kq) s_a[p * KP + q * kq + j] = lg[j] * rs;
    __syncthreads();
    // aggregation
#pragma unroll
    for (int j = 0; j < NOWN; ++j) {
      if (j < nown) {
        float acc = 0.f;                        // the tile's own sum first, then into the slab's (see tiles_per_slab)
        const int k = vk[j], c = vc[j];
        for (int pp = 0; pp < VT; ++pp) acc = fmaf(s_a[pp * KP + k], s_x[pp * CP + c], acc);
        vacc[j] += acc;
      }
    }
    if (tid < K) {
      float at = 0.f;
      for (int pp = 0; pp < VT; ++pp) at += s_a[pp * KP + tid];
      asum += at;
    }
  }
  float* dst = a.part + ((size_t)b * gridDim.x + split) * (KC_ + K);
#pragma unroll
  for (int j = 0; j < NOWN; ++j) {
    const int e = tid + 256 * j;
    if (j < nown && e < KC_) dst[e] = vacc[j];
  }
  if (tid < K) dst[KC_ + tid] = asum;
}

// ---------------------------------------------------------------------------------------------
// Matrix-core version of pass 1 (K in {32, 64}, C <= 64): per 64-pixel tile
//   step 1  logits^T[k][p] = sum_c Wa[k][c] xh[p][c]      one 32x32 (k-tile, p-tile) block per wave
//   softmax over k per pixel (4 threads per pixel, through LDS)
//   step 2  V[k][c] += sum_p a[p][k] xh[p][c]             one 32x32 (k-tile, c-tile) block per wave, the
//                                                         accumulator stays in registers over the whole slab
// Exact fp32 (v_mfma_f32_32x32x2_f32).  In both products lane (i, h) walks the contraction index as
// 32h + s, s = 0..31, so step 1 reads its operands as 8 ds_read_b128 per row (row pitch 68 floats).
// ---------------------------------------------------------------------------------------------
typedef float vf16 __attribute__((ext_vector_type(16)));
constexpr int VP = 68;   // LDS row pitch (floats) of the [row][channel] images
constexpr int AP = 65;   // LDS row pitch of the [pixel][cluster] assignment image

typedef _Float16 vh8 __attribute__((ext_vector_type(8)));
typedef _Float16 vh4 __attribute__((ext_vector_type(4)));
typedef _Float16 vh2 __attribute__((ext_vector_type(2)));
typedef float vf2 __attribute__((ext_vector_type(2)));
constexpr int HP = 72;   // LDS row pitch (halves) of the split-fp16 [row][channel] images (144 B)

// four floats -> fp16 hi halves and the fp16 of the remainders (x = hi + lo to ~2^-22; attention.hip att_split2)
__device__ __forceinline__ void vlad_split4(const float4 v, vh4& hi, vh4& lo) {
  const vf2 a = {v.x, v.y}, b = {v.z, v.w};
  const vh2 ha = __builtin_convertvector(a, vh2), hb = __builtin_convertvector(b, vh2);
  unsigned la, lb;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(ha), "v"(v.x));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(la) : "v"(ha), "v"(v.y));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hb), "v"(v.z));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb) : "v"(hb), "v"(v.w));
  const vh2 l0 = __builtin_bit_cast(vh2, la), l1 = __builtin_bit_cast(vh2, lb);
  hi = vh4{ha[0], ha[1], hb[0], hb[1]};
  lo = vh4{l0[0], l0[1], l1[0], l1[1]};
}

// SPLIT (the f16x3 arithmetic mode): the soft-assignment logits — half of the kernel's matrix work, 32 exact-fp32 MFMAs
// of 64 cycles per 32 x 32 block — as split-fp16 products on v_mfma_f32_32x32x16_f16 (w x = wh xh + wh xl + wl xh: twelve
// MFMAs of 32 cycles), operands |x^| <= 1 and the soft-assign weights.  The AGGREGATION stays exact fp32 in both modes:
// its left operand is the soft assignment itself, whose small entries (1e-6 and below on sharp assignments) lose their
// relative accuracy in fp16 halves and dominate intra-normalised rows of rarely used clusters.
template <int KT, bool SPLIT>
__global__ __launch_bounds__(256) void netvlad_partial_mfma_kernel(const VladArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int KPAD = 32 * KT;
  constexpr int WF = SPLIT ? 0 : KPAD * VP;      // SPLIT keeps its (constant) weight operands in registers
  float* s_w = sm;                   // [KPAD][VP]   soft-assign weights, channels zero-padded to 64
  float* s_x = sm + WF;              // [64][VP]     normalised descriptors of the tile
  float* s_a = s_x + VT * VP;        // [64][AP]     logits, then soft assignments
  _Float16* s_xh = reinterpret_cast<_Float16*>(s_a + VT * AP);      // SPLIT: [64][HP] hi | [64][HP] lo of the same descriptors
  _Float16* s_xl = s_xh + VT * HP;
  const int C = a.C, K = a.K, S = a.S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int b = blockIdx.y, split = blockIdx.x;
  const int per = (S + a.nsplit - 1) / a.nsplit;
  const int tps = a.tps > 1 ? a.tps : 1;                 // tile mode: blockIdx.x = slab * tps + tile
  const int slab = split / tps;
  const int s_begin = slab * per + (split - slab * tps) * (a.tps > 1 ? VT : 0);
  const int s_end = a.tps > 1 ? min(min(S, slab * per + per), s_begin + VT) : min(S, s_begin + per);
  const int CQ = C >> 2;

  if constexpr (!SPLIT) {
    for (int e = tid; e < KPAD * 16; e += 256) {
      const int k = e >> 4, q = e & 15;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < K && q < CQ) v = reinterpret_cast<const float4*>(a.wa + (size_t)k * C)[q];
      *reinterpret_cast<float4*>(&s_w[k * VP + 4 * q]) = v;
    }
  }

  // wave -> blocks: step 1 (k-tile kt1, p-tile pt1), step 2 (k-tile kt2, c-tile ct2)
  const int kt1 = wave % KT, pt1 = wave / KT;          // KT == 2: 4 blocks; KT == 1: waves 0,1 only
  const bool has1 = pt1 < 2;
  const int kt2 = wave % KT, ct2 = wave / KT;
  const bool has2 = ct2 < 2 && ct2 * 32 < C;
  // SPLIT: this wave's weight operand of step 1, all four k-steps, split once: lane (i, h) holds channels
  // 16 s + 8 h .. + 7 of cluster row kt1 * 32 + i
  vh8 wrh[4], wrl[4];
  if constexpr (SPLIT) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      const int k = kt1 * 32 + i, c = 16 * s + 8 * h;
      if (has1 && k < K) {
        if (c < C) v0 = *reinterpret_cast<const float4*>(a.wa + (size_t)k * C + c);
        if (c + 4 < C) v1 = *reinterpret_cast<const float4*>(a.wa + (size_t)k * C + c + 4);
      }
      vh4 h0, l0, h1, l1;
      vlad_split4(v0, h0, l0);
      vlad_split4(v1, h1, l1);
      wrh[s] = vh8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      wrl[s] = vh8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    }
  }
  vf16 vacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) vacc[r] = 0.f;
  float asum = 0.f;
  const int p4 = tid >> 2, q4 = tid & 3;               // 4 threads per pixel
  const int kq = K >> 2;
  const float* xb = a.x + (size_t)b * S * C;

  // A thread owns channels [16 q4, 16 q4 + 16) of pixel p4 of the tile: it reads them straight from global memory
  // (the next tile is fetched into registers while the two products of this one run), normalises in registers
  // and writes the normalised descriptor to LDS once.
  float4 v[4];
  auto fetch = [&](int t0) {
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      const int c = 16 * q4 + 4 * j4;
      v[j4] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t0 + p4 < s_end && c < C) v[j4] = *reinterpret_cast<const float4*>(xb + (size_t)(t0 + p4) * C + c);
    }
  };
  fetch(s_begin);

  for (int t0 = s_begin; t0 < s_end; t0 += VT) {
    const int np = min(VT, s_end - t0);
    __syncthreads();     // the previous tile's step 2 and row sums are done with s_x / s_a
    {  // descriptor-wise L2 normalisation (F.normalize, eps 1e-12)
      float ss = 0.f;
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        ss = fmaf(v[j4].x, v[j4].x, ss); ss = fmaf(v[j4].y, v[j4].y, ss);
        ss = fmaf(v[j4].z, v[j4].z, ss); ss = fmaf(v[j4].w, v[j4].w, ss);
      }
      ss += __shfl_xor(ss, 1);
      ss += __shfl_xor(ss, 2);
      const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {   // channels past C are written as zeros (v stayed 0)
        const float4 xn = make_float4(v[j4].x * inv, v[j4].y * inv, v[j4].z * inv, v[j4].w * inv);
        *reinterpret_cast<float4*>(&s_x[p4 * VP + 16 * q4 + 4 * j4]) = xn;
        if constexpr (SPLIT) {
          vh4 hi, lo;
          vlad_split4(xn, hi, lo);
          *reinterpret_cast<vh4*>(&s_xh[p4 * HP + 16 * q4 + 4 * j4]) = hi;
          *reinterpret_cast<vh4*>(&s_xl[p4 * HP + 16 * q4 + 4 * j4]) = lo;
        }
      }
    }
    if (t0 + VT < s_end) fetch(t0 + VT);
    __syncthreads();
    if (has1) {   // step 1: logits^T block (rows k, cols p)
      vf16 d;
#pragma unroll
      for (int r = 0; r < 16; ++r) d[r] = 0.f;
      if constexpr (SPLIT) {
        // k-step s: lane (i, h) holds channels 16 s + 8 h .. + 7 of its weight row / pixel row
        const int xo = (pt1 * 32 + i) * HP + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const vh8 wh = wrh[s], wl = wrl[s];
          const vh8 xh = *reinterpret_cast<const vh8*>(&s_xh[xo + 16 * s]), xl = *reinterpret_cast<const vh8*>(&s_xl[xo + 16 * s]);
          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, d, 0, 0, 0);
        }
      } else {
      const float* wr = &s_w[(kt1 * 32 + i) * VP + 32 * h];
      const float* xr = &s_x[(pt1 * 32 + i) * VP + 32 * h];
#pragma unroll
      for (int s4 = 0; s4 < 8; ++s4) {
        const float4 wv = *reinterpret_cast<const float4*>(wr + 4 * s4);
        const float4 xv = *reinterpret_cast<const float4*>(xr + 4 * s4);
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, xv.x, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, xv.y, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, xv.z, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, xv.w, d, 0, 0, 0);
      }
      }
      // D[row = k][col = p]: lane holds p = i, rows (r&3) + 8(r>>2) + 4h
#pragma unroll
      for (int r = 0; r < 16; ++r) s_a[(pt1 * 32 + i) * AP + kt1 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = d[r];
    }
    __syncthreads();
    {  // softmax over clusters: this thread owns clusters [q4*kq, (q4+1)*kq) of pixel p4 — read once into registers,
       // written once (the three passes over LDS of the first form were a third of the kernel's vector instructions)
      float* row = &s_a[p4 * AP + q4 * kq];
      constexpr int KQ = 8 * KT;                       // kq <= KPAD / 4
      float lv[KQ];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < KQ; ++j) { lv[j] = j < kq ? row[j] : -INFINITY; mx = fmaxf(mx, lv[j]); }
      mx = fmaxf(mx, __shfl_xor(mx, 1));
      mx = fmaxf(mx, __shfl_xor(mx, 2));
      float se = 0.f;
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        // (the split-fp16 mode takes the hardware exponential, exp(x) = 2^(x log2 e): its logits are split products already)
        lv[j] = SPLIT ? __builtin_amdgcn_exp2f((lv[j] - mx) * 1.44269504088896340736f) : expf(lv[j] - mx);
        se += lv[j];                                   // (clusters past kq: exp(-inf) = 0)
      }
      se += __shfl_xor(se, 1);
      se += __shfl_xor(se, 2);
      const float rs = (p4 < np) ? 1.f / se : 0.f;     // pixels past the slab contribute nothing
#pragma unroll
      for (int j = 0; j < KQ; ++j)
        if (j < kq) row[j] = lv[j] * rs;
    }
    __syncthreads();
    if (has2) {   // step 2: V block (rows k, cols c), contraction over the tile's 64 pixels
      const float* ar = &s_a[(32 * h) * AP + kt2 * 32 + i];
      const float* xr = &s_x[(32 * h) * VP + ct2 * 32 + i];
      vf16 vt;                                   // the tile's own sum first, then into the slab's (see tiles_per_slab)
#pragma unroll
      for (int r = 0; r < 16; ++r) vt[r] = 0.f;
#pragma unroll 8
      for (int s = 0; s < 32; ++s) vt = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[s * AP], xr[s * VP], vt, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) vacc[r] += vt[r];
    }
    if (tid < K) {
      float at = 0.f;
      for (int pp = 0; pp < VT; ++pp) at += s_a[pp * AP + tid];
      asum += at;
    }
  }
  const int KC_ = K * C;
  float* dst = a.part + ((size_t)b * gridDim.x + split) * (KC_ + K);
  if (has2) {
    const int c = ct2 * 32 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = kt2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (c < C && k < K) dst[k * C + c] = vacc[r];
    }
  }
  if (tid < K) dst[KC_ + tid] = asum;
}

// Sum of a frame's partials for element e of [K*C + K], in ONE fixed order whatever produced them: a slab's value
// is the sum of its tile partials in tile order (tile mode; exactly what a slab-walking workgroup accumulates), and
// the slabs go into four interleaved chains (four loads in flight per thread) that are added as (s0 + s1) + (s2 + s3).
__device__ __forceinline__ float vlad_ordered_sum(const VladArgs& a, int b, int e) {
  const size_t ps = (size_t)a.K * a.C + a.K;
  const int tps = a.tps > 1 ? a.tps : 1;
  const float* base = a.part + (size_t)b * a.nsplit * tps * ps;
  auto slab = [&](int sp) {
    const float* q = base + (size_t)sp * tps * ps + e;
    if (tps == 1) return q[0];
    if (tps <= 8) {     // all tile partials in flight at once, then added in tile order
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = i < tps ? q[(size_t)i * ps] : 0.f;
      float v = t[0];
#pragma unroll
      for (int i = 1; i < 8; ++i)
        if (i < tps) v += t[i];
      return v;
    }
    float v = q[0];
    for (int t = 1; t < tps; ++t) v += q[(size_t)t * ps];
    return v;
  };
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int sp = 0;
  for (; sp + 3 < a.nsplit; sp += 4) {
    s0 += slab(sp);
    s1 += slab(sp + 1);
    s2 += slab(sp + 2);
    s3 += slab(sp + 3);
  }
  for (; sp < a.nsplit; ++sp) s0 += slab(sp);
  return (s0 + s1) + (s2 + s3);
}

// Tile mode only (few frames per call): the ordered sums spread over many workgroups — one finish workgroup per
// frame reading nsplit*tps partials was 54 us for a single frame.  Writes sums[b][K*C + K]; the finish pass then
// runs on those as a single "partial" per frame (0 + v is exact).
__global__ __launch_bounds__(256) void netvlad_sum_kernel(const VladArgs a, float* sums) {
  const int e = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  const int n = a.K * a.C + a.K;
  if (e < n) sums[(size_t)b * n + e] = vlad_ordered_sum(a, b, e);
}

constexpr int FIN_T = 1024;   // threads of the finish workgroup: the partial sums are latency-bound reads, so go wide
__global__ __launch_bounds__(FIN_T) void netvlad_finish_kernel(const VladArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = a.C, K = a.K, KC_ = K * C;
  float* s_v = sm;            // [K*C]
  float* s_as = s_v + KC_;    // [K]
  float* s_n = s_as + K;      // [K] row norms, then [FIN_T / 64] wave sums
  const int tid = threadIdx.x, b = blockIdx.x;
  for (int e = tid; e < KC_ + K; e += FIN_T) s_v[e] = vlad_ordered_sum(a, b, e);   // s_as follows s_v contiguously
  __syncthreads();
  for (int e = tid; e < KC_; e += FIN_T) {
    const int k = e / C, c = e - k * C;
    s_v[e] = s_v[e] - s_as[k] * a.cent[k * C + c];
  }
  __syncthreads();
  // intra-normalisation: 4 threads per cluster row
  for (int k = tid >> 2; k < K; k += FIN_T / 4) {
    const int q = tid & 3, cq = C >> 2;
    float ss = 0.f;
    for (int c = q * cq; c < (q + 1) * cq; ++c) { const float v = s_v[k * C + c]; ss = fmaf(v, v, ss); }
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    if (q == 0) s_n[k] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
  }
  __syncthreads();
  float tot = 0.f;
  for (int e = tid; e < KC_; e += FIN_T) {
    const float v = s_v[e] * s_n[e / C];
    s_v[e] = v;
    tot = fmaf(v, v, tot);
  }
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
  __syncthreads();
  if ((tid & 63) == 0) s_n[tid >> 6] = tot;
  __syncthreads();
  float all = 0.f;
#pragma unroll
  for (int w = 0; w < FIN_T / 64; ++w) all += s_n[w];
  const float inv = 1.f / fmaxf(sqrtf(all), 1e-12f);
  for (int e = tid; e < KC_; e += FIN_T) a.out[(size_t)b * KC_ + e] = s_v[e] * inv;
}

// ---------------------------------------------------------------------------------------------
// Config-selectable poolers
// GeM (modules/aggregators/gem.py:21-31): PixelUnshuffle(4) -> clamp(min=1e-6)^p -> global mean -> ^(1/p);
//   output index = c*16 + i*4 + j for input channel c and sub-position (i, j) = (y & 3, x & 3).
// ConvAP (modules/aggregators/convap.py:28-34): [1x1 conv + bias runs in conv3x3.hip] -> AdaptiveAvgPool2d((4,4))
//   -> flatten (c*16 + a*4 + b) -> L2 normalise (eps 1e-12).
// One workgroup per frame; x is the NHWC encoder map [S = Hc*Wc][C].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gem_kernel(const PoolArgs a) {
  const int b = blockIdx.x, C = a.C, Hc = a.Hc, Wc = a.Wc;
  const float p = a.p[0], eps = 1e-6f;
  const float* x = a.x + (size_t)b * Hc * Wc * C;
  const float inv = 1.f / (float)((Hc >> 2) * (Wc >> 2));
  for (int o = threadIdx.x; o < C * 16; o += 256) {
    const int c = o >> 4, i = (o >> 2) & 3, j = o & 3;
    float acc = 0.f;
    for (int y = i; y < Hc; y += 4)
      for (int xx = j; xx < Wc; xx += 4) acc += powf(fmaxf(x[((size_t)y * Wc + xx) * C + c], eps), p);
    a.out[(size_t)b * C * 16 + o] = powf(acc * inv, 1.f / p);
  }
}

__global__ __launch_bounds__(256) void convap_pool_kernel(const PoolArgs a) {
  extern __shared__ float sm[];          // [C*16] pooled values + [4] wave sums
  const int b = blockIdx.x, C = a.C, Hc = a.Hc, Wc = a.Wc;
  const float* x = a.x + (size_t)b * Hc * Wc * C;
  float ss = 0.f;
  for (int o = threadIdx.x; o < C * 16; o += 256) {
    const int c = o >> 4, aa = (o >> 2) & 3, bb = o & 3;
    const int y0 = (aa * Hc) / 4, y1 = ((aa + 1) * Hc + 3) / 4;      // floor / ceil bin edges (AdaptiveAvgPool2d)
    const int x0 = (bb * Wc) / 4, x1 = ((bb + 1) * Wc + 3) / 4;
    float acc = 0.f;
    for (int y = y0; y < y1; ++y)
      for (int xx = x0; xx < x1; ++xx) acc += x[((size_t)y * Wc + xx) * C + c];
    const float v = acc / (float)((y1 - y0) * (x1 - x0));
    sm[o] = v;
    ss = fmaf(v, v, ss);
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((threadIdx.x & 63) == 0) sm[C * 16 + (threadIdx.x >> 6)] = ss;
  __syncthreads();
  const float* w = &sm[C * 16];
  const float inv = 1.f / fmaxf(sqrtf(w[0] + w[1] + w[2] + w[3]), 1e-12f);
  for (int o = threadIdx.x; o < C * 16; o += 256) a.out[(size_t)b * C * 16 + o] = sm[o] * inv;
}

int launch_gem(const PoolArgs& a, hipStream_t s) {
  if ((a.Hc & 3) || (a.Wc & 3)) return -1600;     // PixelUnshuffle(4) needs divisible spatial dims (torch raises too)
  hipLaunchKernelGGL(gem_kernel, dim3(a.B), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

int launch_convap_pool(const PoolArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(convap_pool_kernel, dim3(a.B), dim3(256), (size_t)(a.C * 16 + 4) * sizeof(float), s, a);
  return (int)hipGetLastError();
}

int launch_netvlad(const VladArgs& a, hipStream_t s) {
  if (a.K > 64 || (a.K & 3) || (a.C & 3) || a.K * a.C > 8192 || a.K < 4) return -1100;
  if ((a.K == 32 || a.K == 64) && a.C <= 64) {
    const int kt = a.K / 32;
    static const bool split_on = !(getenv("KP2D_VLAD_SPLIT") && getenv("KP2D_VLAD_SPLIT")[0] == '0');
    const bool split = a.prec == 1 && split_on;
    const dim3 grid(a.nsplit * (a.tps > 1 ? a.tps : 1), a.B);
    if (split) {
      const size_t lds = (size_t)(VT * VP + VT * AP + VT * HP) * sizeof(float);      // (hi | lo half images = HP floats per row)
      if (kt == 2) hipLaunchKernelGGL((netvlad_partial_mfma_kernel<2, true>), grid, dim3(256), lds, s, a);
      else hipLaunchKernelGGL((netvlad_partial_mfma_kernel<1, true>), grid, dim3(256), lds, s, a);
    } else {
      const size_t lds = (size_t)(32 * kt * VP + VT * VP + VT * AP) * sizeof(float);
      if (kt == 2) hipLaunchKernelGGL((netvlad_partial_mfma_kernel<2, false>), grid, dim3(256), lds, s, a);
      else hipLaunchKernelGGL((netvlad_partial_mfma_kernel<1, false>), grid, dim3(256), lds, s, a);
    }
  } else {
    const size_t lds1 = (size_t)(a.K * (a.C + 1) + VT * (a.C + 1) + VT * (a.K + 1)) * sizeof(float);
    if (a.K * a.C <= 4096) {
      hipLaunchKernelGGL(netvlad_partial_kernel<16>, dim3(a.nsplit * (a.tps > 1 ? a.tps : 1), a.B), dim3(256), lds1, s, a);
    } else {   // TINY_F: 64 clusters x 128 channels, 83 KB of LDS
      static PerDeviceOnce lds_once;      // per device: a handle may live on any visible device
      if (int e = lds_opt_in(lds_once, reinterpret_cast<const void*>(&netvlad_partial_kernel<32>))) return e;
      hipLaunchKernelGGL(netvlad_partial_kernel<32>, dim3(a.nsplit * (a.tps > 1 ? a.tps : 1), a.B), dim3(256), lds1, s, a);
    }
  }
  const size_t lds2 = (size_t)(a.K * a.C + a.K + (a.K > FIN_T / 64 ? a.K : FIN_T / 64) + 8) * sizeof(float);
  if (a.tps > 1) {
    const int n = a.K * a.C + a.K;
    float* sums = a.part + (size_t)a.B * a.nsplit * a.tps * n;     // the plan reserves B*n floats behind the partials
    hipLaunchKernelGGL(netvlad_sum_kernel, dim3((n + 255) / 256, a.B), dim3(256), 0, s, a, sums);
    VladArgs f = a;
    f.part = sums; f.nsplit = 1; f.tps = 1;
    hipLaunchKernelGGL(netvlad_finish_kernel, dim3(a.B), dim3(FIN_T), lds2, s, f);
  } else {
    hipLaunchKernelGGL(netvlad_finish_kernel, dim3(a.B), dim3(FIN_T), lds2, s, a);
  }
  return (int)hipGetLastError();
}

}  // namespace kp2d
