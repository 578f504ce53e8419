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
#include <cstdlib>

#include "kp2d_kernels.h"

namespace kp2d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

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

// C = 4 LP with LP a power of two <= 64 (C = 64: 16 lanes per pixel): a lane holds one float4 of its pixel, so every
// load / store instruction of a wave moves 1 KiB of consecutive bytes (64 / LP pixels) and the channel reduction is
// log2(LP) steps inside the lane group — a quarter of the instructions per byte of the wave-per-pixel kernel above
// (2.75 -> 4 TB/s at 64 channels).
template <int LP>
__global__ __launch_bounds__(256) void channel_layernorm_q_kernel(const LnArgs a) {
  constexpr int C = 4 * LP;
  const int sub = threadIdx.x % LP;
  const long gp = ((long)blockIdx.x * 256 + threadIdx.x) / LP;
  const long gstride = (long)gridDim.x * 256 / LP;
  const float4 g = reinterpret_cast<const float4*>(a.g)[sub];
  const float4 b = reinterpret_cast<const float4*>(a.b)[sub];
  const float invC = 1.f / (float)C;
  for (long p = gp; p < a.npix; p += gstride) {    // the LP lanes of a pixel run the loop together
    float4 v = reinterpret_cast<const float4*>(a.x + p * C)[sub];
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int o = 1; o < LP; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s * invC;
    v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
    float q = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
    for (int o = 1; o < LP; o <<= 1) q += __shfl_xor(q, o);
    const float d = sqrtf(q * invC) + 1e-5f;      // torch.var(unbiased=False).sqrt() + eps
    reinterpret_cast<float4*>(a.y + p * C)[sub] =
        make_float4(v.x / d * g.x + b.x, v.y / d * g.y + b.y, v.z / d * g.z + b.z, v.w / d * g.w + b.w);
  }
}

int launch_channel_layernorm(const LnArgs& a, hipStream_t s) {
  if (a.C > 256 || a.C < 1) return -1400;
  if (a.C == 64 || a.C == 128 || a.C == 256 || a.C == 32 || a.C == 16) {
    const int lp = a.C / 4;
    long blocks = (a.npix * lp + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    const dim3 grid((int)blocks), blk(256);
    if (lp == 16) hipLaunchKernelGGL(channel_layernorm_q_kernel<16>, grid, blk, 0, s, a);
    else if (lp == 32) hipLaunchKernelGGL(channel_layernorm_q_kernel<32>, grid, blk, 0, s, a);
    else if (lp == 64) hipLaunchKernelGGL(channel_layernorm_q_kernel<64>, grid, blk, 0, s, a);
    else if (lp == 8) hipLaunchKernelGGL(channel_layernorm_q_kernel<8>, grid, blk, 0, s, a);
    else hipLaunchKernelGGL(channel_layernorm_q_kernel<4>, grid, blk, 0, s, a);
    return (int)hipGetLastError();
  }
  long blocks = (a.npix + 3) / 4;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (a.C <= 64) hipLaunchKernelGGL(channel_layernorm_kernel<1>, dim3((int)blocks), dim3(256), 0, s, a);
  else if (a.C <= 128) hipLaunchKernelGGL(channel_layernorm_kernel<2>, dim3((int)blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(channel_layernorm_kernel<4>, dim3((int)blocks), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

// depthwise 3x3, pad 1, bias; NHWC.  One thread per (4 consecutive pixels of a row, 4 channels): the 3 x 6 input window
// is loaded once for the four outputs (18 sixteen-byte loads instead of 36) and the 9 x C weights sit in LDS.
constexpr int DW_R = 4;   // output rows per thread: 6 input rows feed 4 output rows (2.25 window loads per output
                          // float4 instead of 4.5 — the vertical reuse came out of L2 before, which bounded the kernel)
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_dw[];   // [9][C] weights, [C] bias
  const int C = a.C, Q = C >> 2, H = a.H, W = a.W;
  for (int e = threadIdx.x; e < 10 * Q; e += 256)
    reinterpret_cast<float4*>(s_dw)[e] = e < 9 * Q ? reinterpret_cast<const float4*>(a.w)[e]
                                                    : reinterpret_cast<const float4*>(a.bias)[e - 9 * Q];
  __syncthreads();
  const int WG4 = (W + 3) >> 2;                                   // 4-pixel groups per row
  const int HG = (H + DW_R - 1) / DW_R;                           // row groups per frame
  const long total = (long)a.B * HG * WG4 * Q;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int q = (int)(e % Q);
  const long grp = e / Q;
  const int xg = (int)(grp % WG4);
  const long rg = grp / WG4;                                      // b * HG + row group
  const int b = (int)(rg / HG);
  const int y0 = (int)(rg - (long)b * HG) * DW_R;
  const int x0 = 4 * xg;
  const float4 bias = reinterpret_cast<const float4*>(s_dw + 9 * C)[q];
  float4 acc[DW_R][4];
#pragma unroll
  for (int r = 0; r < DW_R; ++r)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[r][p] = bias;
  // Window loads as buffer loads with 32-bit offsets from the first frame this block touches; a tap outside the
  // image gets an offset past the descriptor, which reads as zeros (the zero padding): no 64-bit address
  // arithmetic and no exec-mask branch per tap.
  const long fb = ((long)blockIdx.x * 256 / Q / WG4) / HG;
  const size_t fbytes = (size_t)H * W * C * sizeof(float);
  const size_t nf = (size_t)(256 * 4 * DW_R) / ((size_t)H * W) + 2;      // frames a 256-thread block can touch
  const size_t fl = (size_t)a.B - (size_t)fb;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x) + (size_t)fb * H * W * C, 0, (int)((fl < nf ? fl : nf) * fbytes), 0x00020000);
  const int fo = (int)((b - fb) * (long)fbytes) + q * 16;
#pragma unroll
  for (int r = -1; r <= DW_R; ++r) {          // input row y0 + r feeds output rows r-1, r, r+1 of the group
    const int yy = y0 + r;
    const bool yok = yy >= 0 && yy < H;
    float4 v[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int xx = x0 - 1 + j;
      const int o = (yok && xx >= 0 && xx < W) ? fo + (yy * W + xx) * C * 4 : 0x7ffffff0;
      v[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, o, 0, 0));
    }
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int orow = r - dy;                 // output row (in the group) that sees this input row at offset dy
      if (orow < 0 || orow >= DW_R) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float4 w = reinterpret_cast<const float4*>(s_dw + ((dy + 1) * 3 + dx) * C)[q];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          acc[orow][p].x = fmaf(v[p + dx].x, w.x, acc[orow][p].x); acc[orow][p].y = fmaf(v[p + dx].y, w.y, acc[orow][p].y);
          acc[orow][p].z = fmaf(v[p + dx].z, w.z, acc[orow][p].z); acc[orow][p].w = fmaf(v[p + dx].w, w.w, acc[orow][p].w);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < DW_R; ++r)
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (y0 + r < H && x0 + p < W)
        reinterpret_cast<float4*>(a.y + (((size_t)b * H + y0 + r) * W + x0 + p) * C)[q] = acc[r][p];
}

int launch_dwconv3x3(const DwArgs& a, hipStream_t s) {
  if ((a.C & 3) || a.C > 1024) return -1401;
  // 32-bit byte offsets inside a descriptor of at most 256 * 4 * DW_R / (H W) + 2 frames
  if (((size_t)(256 * 4 * DW_R) / ((size_t)a.H * a.W) + 2) * a.H * a.W * a.C * sizeof(float) >= 0x7ffffff0u) return -1403;
  const long total = (long)a.B * ((a.H + DW_R - 1) / DW_R) * ((a.W + 3) >> 2) * (a.C >> 2);
  hipLaunchKernelGGL(dwconv3x3_kernel, dim3((int)((total + 255) / 256)), dim3(256), (size_t)10 * a.C * sizeof(float), s, a);
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
  const int C = a.C, d = C / a.heads, S = a.S, TS = a.T;      // TS: rows per sequence in memory
  const int bk = a.kv_bshift ? (b + a.kv_bshift) % a.B : b;
  const int T = a.tcount ? min(TS, a.tcount[bk]) : TS;         // keys that exist (LightGlue on padded keypoint sets)
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
  const float* kvb = a.kv + (size_t)bk * TS * kvs + a.k_off + h * d;

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


// ---------------------------------------------------------------------------------------------
// Split-fp16 streaming attention (prec 1, head dim d <= 16): the same "x = hi + lo, three MFMAs into one fp32
// accumulator" arithmetic as the convolutions (conv3x3.hip), on v_mfma_f32_32x32x16_f16 — 3.5x fewer matrix-core
// cycles than the fp32 16x16x4 kernel above, which is what a 19200-query x 4800-key frame (480x640) is bound by.
//   grid (ceil(S/128), heads, B); wave = 32 queries.  Per 32-key tile:
//   S^T[key][query] = K_tile (32 x 16) . Q^T (16 x 32)                      one MFMA triple, K = d
//   O^T[chan][query] += V^T (16 x 32 keys) . P^T (32 keys x 32 queries)     two MFMA triples (K = 16 keys each)
//   A lane holds 16 keys of ONE query column of S^T, so the online softmax is register-local plus one xor-32 step,
//   and P^T in that layout is the B operand of the second product once V^T is staged with the matching key order:
//   accumulator register r = 8t + j of lane-half h is key 16t + 8(j>>2) + 4h + (j&3) -> slot (t, h, j).
// ---------------------------------------------------------------------------------------------
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void att_split2(float x, float y, h16x2& hi, h16x2& lo) {
  const f32x2_ v = {x, y};
  hi = __builtin_convertvector(v, h16x2);
  // lo = fp16(x - float(hi)): v_fma_mixlo/mixhi_f16 read the fp16 half directly, form the (exact) fp32 difference and
  // round it into one half of the destination — three instructions per two values (a v_fma_mix_f32 per value plus a
  // packing conversion were four)
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(x));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(y));
  lo = __builtin_bit_cast(h16x2, l);
}
__device__ __forceinline__ void att_split8(const float (&x)[8], h16x8& hi, h16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    h16x2 h, l;
    att_split2(x[j], x[j + 1], h, l);
    hi[j] = h[0]; hi[j + 1] = h[1]; lo[j] = l[0]; lo[j + 1] = l[1];
  }
}

constexpr int SKT = 128;   // keys per LDS chunk (four 32-key tiles)
constexpr int SKP = 40;    // K row pitch in halves: 16 hi | 16 lo | 8 pad  (80 B: conflict-free ds_read_b128 of 32 rows)
constexpr int SVP = 96;    // V row pitch in halves: [16 hi | 1, 0 x 15 | 16 lo | 0 x 16 | 32 unused] = 192 B: the four key
                           // rows of a transposed block read (64 B each per 32-lane half) land on disjoint banks
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

// V is staged ROW-major ([key][16 hi | 16 lo], as K) and reaches the matrix cores as the V^T operand through
// ds_read_b64_tr_b16: per group of 16 lanes the instruction reads a block of 4 rows (keys) x 16 columns (channels),
// lane 4q + p supplying the address of row q, columns 4p .. 4p+3, and hands lane c of the group column c of the four
// rows — the transposition is free.  (The first version wrote V^T into LDS with eight 2-byte stores per staged granule:
// 44.6 % of the kernel's LDS-active cycles were bank conflicts, profiles/r2_pmc_cfg4_summary.txt.)
__device__ __forceinline__ h16x4 lds_tr4(const _Float16* p) {
  const fp16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)p);
  return __builtin_bit_cast(h16x4, r);
}

// Online-softmax state of a wave's 32 queries
struct AttAcc {
  f32x16 o;        // O^T accumulator: rows 0..15 channels, row 16 the softmax denominator
  f32x16 negm;     // -m in every register: the C operand of the score product
  float m;         // running reference: the maximum score so far (0 before the first tile)
  bool first;
};

// One 32-key tile: scores of the wave's 32 queries against LDS rows krow.. (this lane's K row, hi at +0 / lo at +16),
// softmax update, P V through the transposed reads at vcur (see the kernels).  kglob = sequence index of the tile's
// first key: only the last tile of a sequence has keys past T.
__device__ __forceinline__ void att_tile(const _Float16* krow, const _Float16* vcur, const h16x8 qh, const h16x8 ql,
                                         int kglob, int T, int h, AttAcc& st) {
  const h16x8 kh = *reinterpret_cast<const h16x8*>(krow);
  const h16x8 kl = *reinterpret_cast<const h16x8*>(krow + 16);
  // The score accumulator starts at -m, the running reference of this query (kept as a 16-register vector that
  // only changes when the reference moves): the products come out as s - m and the common tile — no key beats
  // the reference — goes straight to exp2 with no per-score subtraction.  The kernel is bound by its VALU
  // instruction count (11.6 per MFMA before this, profiles/r1_pmc_cfg4_summary.txt).
  f32x16 sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh, st.negm, 0, 0, 0);
  sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql, sc, 0, 0, 0);
  sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh, sc, 0, 0, 0);
  if (kglob + 32 > T) {                       // (uniform branch)
    const int kbase = kglob + 4 * h;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (kbase + (r & 3) + 8 * (r >> 2) >= T) sc[r] = -INFINITY;
  }
  float mx = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
  for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sc[r]), sc[r + 1]);      // v_max3_f32
  mx = fmaxf(mx, sc[15]);
  {
    // the other 16 keys of this query sit in lane ^ 32: v_permlane32_swap exchanges the wave halves on the VALU
    // (ds_bpermute was an LDS round trip plus an lgkmcnt(0) in the middle of every tile)
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  }
  // first tile: the reference becomes the tile maximum whatever its sign (m and o start at 0); later it only rises
  // (the common tile — no key beats any query's reference — pays one compare: delta itself is formed inside the branch)
  if (st.first || __any(mx > 0.f)) {        // some query of this wave moves its reference: shift and rescale
    const float delta = st.first ? mx : fmaxf(mx, 0.f);
    st.m += delta;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sc[r] -= delta; st.negm[r] = -st.m; }
    if (!st.first) {
      const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
      for (int r = 0; r < 9; ++r) st.o[r] *= alpha;          // rows 0..15 (channels) and row 16 (denominator)
    }
  }
  st.first = false;
  float p[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(sc[r]);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    h16x8 ph, pl;
    const float (&pp)[8] = *reinterpret_cast<const float (*)[8]>(&p[8 * t]);
    att_split8(pp, ph, pl);
    const h16x4 vh0 = lds_tr4(vcur + (16 * t) * SVP), vl0 = lds_tr4(vcur + (16 * t) * SVP + 32);
    const h16x4 vh1 = lds_tr4(vcur + (16 * t + 8) * SVP), vl1 = lds_tr4(vcur + (16 * t + 8) * SVP + 32);
    const h16x8 vh = {vh0[0], vh0[1], vh0[2], vh0[3], vh1[0], vh1[1], vh1[2], vh1[3]};
    const h16x8 vl = {vl0[0], vl0[1], vl0[2], vl0[3], vl1[0], vl1[1], vl1[2], vl1[3]};
    st.o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, st.o, 0, 0, 0);
    st.o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, st.o, 0, 0, 0);
    st.o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, st.o, 0, 0, 0);
  }
}

// Q^T operand of a query: lane (query i, h) holds channels 8h..8h+7, pre-multiplied by scale * log2(e) so that the
// softmax runs on exp2 directly
__device__ __forceinline__ void att_load_q(const AttnArgs& a, const float* qp, bool ok, int h, int d, h16x8& qh, h16x8& ql) {
  const float f = a.scale * 1.44269504088896340736f;
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = (ok && 8 * h + j < d) ? qp[8 * h + j] * f : 0.f;
  att_split8(x, qh, ql);
}

// one staged granule: 4 channels of a key's K and V rows, split and written to the LDS images
__device__ __forceinline__ void att_commit(_Float16* Ks, _Float16* Vs, int key, int q4, const float4 kk, const float4 vv) {
  h16x2 h0, h1, l0, l1;
  att_split2(kk.x, kk.y, h0, l0);
  att_split2(kk.z, kk.w, h1, l1);
  *reinterpret_cast<h16x4*>(&Ks[key * SKP + 4 * q4]) = h16x4{h0[0], h0[1], h1[0], h1[1]};
  *reinterpret_cast<h16x4*>(&Ks[key * SKP + 16 + 4 * q4]) = h16x4{l0[0], l0[1], l1[0], l1[1]};
  att_split2(vv.x, vv.y, h0, l0);
  att_split2(vv.z, vv.w, h1, l1);
  *reinterpret_cast<h16x4*>(&Vs[key * SVP + 4 * q4]) = h16x4{h0[0], h0[1], h1[0], h1[1]};
  *reinterpret_cast<h16x4*>(&Vs[key * SVP + 32 + 4 * q4]) = h16x4{l0[0], l0[1], l1[0], l1[1]};
}

// normalise and store a wave's 32 output rows (o[8] of the h = 0 lanes is the denominator)
__device__ __forceinline__ void att_store(const f32x16& o, float* op, int h, int d) {
  float l = o[8];                 // row 16 of O^T lives in register 8 of the h = 0 lanes (row 20, zero, for h = 1)
  l += __shfl_xor(l, 32);
  if (!op) return;
  const float inv = 1.f / l;
  // accumulator register r < 8 of lane-half h: channel (r & 3) + 8 (r >> 2) + 4h
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int c0 = 8 * g + 4 * h;
    if (c0 + 3 < d) {
      *reinterpret_cast<float4*>(op + c0) = make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (c0 + r < d) op[c0 + r] = o[4 * g + r] * inv;
    }
  }
}

template <int NTHR, int WPS>   // threads per workgroup (32 queries per wave), waves per SIMD asked of the compiler
__global__ __launch_bounds__(NTHR, WPS) void attention_split_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 Ks[SKT * SKP];
  __shared__ __attribute__((aligned(16))) _Float16 Vs[SKT * SVP];                     // [key][16 hi | 1,0.. | 16 lo | 0.. | -]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  // XCD affinity.  Workgroups are dealt round-robin over the 8 XCDs (L mod 8).  With a (tiles, heads, B) grid the
  // query tiles of one (frame, head) spread over all eight L2s, each of which then fetches that pair's K / V — and,
  // as a head only uses 64 B of every 128-B line of the q and kv rows, fetches each line twice as often again
  // (PMC, 480x640 x 32 frames: 1.3 GB fetched per launch against 236 MB algorithmic).  A 1-D grid decoded as
  // frame = xcd + 8 * (j / (tiles * heads))   (j = L / 8), heads in pairs inside a frame (below)
  // keeps every workgroup of a frame on one XCD, so one L2 holds the frame's K / V and every q / kv line is used whole.  Needs B % 8 == 0 (the launcher falls back to (frame, head)
  // pairs per XCD when only heads * B is a multiple of 8, and to the plain 3-D grid otherwise).
  int hd, b, qblk;
  if (gridDim.y == 1 && gridDim.z == 1 && a.heads * a.B > 1) {
    const int tiles = (a.S + NTHR / 2 - 1) / (NTHR / 2);
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    if ((a.B & 7) == 0) {
      const int per = tiles * a.heads;
      b = xcd + 8 * (j / per);
      const int r = j % per;
      if ((a.heads & 1) == 0) {
        // two heads at a time over all query tiles: their q / k / v segments are the two halves of the same 128-B
        // lines, and only two heads' K / V (1.2 MB at 480x640) have to stay in the 4 MB L2 while the tiles stream by
        const int hp = r / (2 * tiles), rr = r % (2 * tiles);
        qblk = rr >> 1; hd = 2 * hp + (rr & 1);
      } else {
        hd = r % a.heads; qblk = r / a.heads;
      }
    } else {
      const int pair = xcd + 8 * (j / tiles);
      qblk = j % tiles; hd = pair % a.heads; b = pair / a.heads;
    }
  } else {
    hd = blockIdx.y; b = blockIdx.z; qblk = blockIdx.x;
  }
  const int C = a.C, d = C / a.heads, S = a.S, TS = a.T;
  const int bk = a.kv_bshift ? (b + a.kv_bshift) % a.B : b;
  const int T = a.tcount ? min(TS, a.tcount[bk]) : TS;
  const int qs = a.q_stride ? a.q_stride : C, kvs = a.kv_stride ? a.kv_stride : 2 * C;
  const int vd = (a.v_off ? a.v_off : C) - a.k_off, os = a.out_stride ? a.out_stride : C;
  const int qi = qblk * (NTHR / 2) + wave * 32 + i;
  h16x8 qh, ql;
  att_load_q(a, a.q + ((size_t)b * S + (qi < S ? qi : 0)) * qs + hd * d, qi < S, h, d, qh, ql);
  AttAcc st;
#pragma unroll
  for (int r = 0; r < 16; ++r) { st.o[r] = 0.f; st.negm[r] = 0.f; }
  st.m = 0.f;
  st.first = true;
  // Rows 16..31 of the V^T operand are padding (head dim <= 16).  Row 16 is set to ones, so that row 16 of O^T
  // accumulates the softmax denominator sum_k (ph + pl) on the matrix cores; the other padding rows are zeros.
  // V^T operand of k-step t (keys 16 t .. 16 t + 15 of the tile), lane-half h, element j = key 16 t + 8 (j >> 2) + 4 h +
  // (j & 3) (the order the score accumulator hands P^T over in): two transposed block reads, keys 16 t + 4 h .. + 3 and
  // 16 t + 8 + 4 h .. + 3.  Lane groups 0 / 2 (lanes 0-15, 32-47) are channels 0-15; groups 1 / 3 are the padding rows
  // and read the 16 constant columns that follow the real ones in every image row ([1, 0 x 15] behind the hi half, zeros
  // behind the lo half, written once here; the staging never touches them): ONE address formula for all 64 lanes, every
  // per-read offset an immediate, EXEC full as the transposed read requires.
  for (int e = tid; e < SKT * 4; e += NTHR) {
    const int row = e >> 2, part = e & 3;      // parts 0, 1: columns 16..31 (behind hi); 2, 3: columns 48..63 (behind lo)
    h16x8 c = {0, 0, 0, 0, 0, 0, 0, 0};
    if (part == 0) c[0] = (_Float16)1.f;
    *reinterpret_cast<h16x8*>(&Vs[row * SVP + 16 + 32 * (part >> 1) + 8 * (part & 1)]) = c;
  }
  const _Float16* const vph = &Vs[(4 * h + ((lane & 15) >> 2)) * SVP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)];
  const float* kvb = a.kv + (size_t)bk * TS * kvs + a.k_off + hd * d;

  for (int kc = 0; kc < T; kc += SKT) {
    __syncthreads();
    // stage: granule e = (key, 4 channels).  K row-major split rows; V transposed with the slot permutation.
    // (Fetching the next chunk's granules into registers during the tiles of the current one measured slower:
    // 3.89 vs 3.60 ms at 480x640 x 32 frames — the other workgroups of the CU already cover this round trip.)
    for (int e = tid; e < SKT * 4; e += NTHR) {
      const int key = e >> 2, q4 = e & 3;
      float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;
      if (kc + key < T && 4 * q4 < d) {
        const float* p = kvb + (size_t)(kc + key) * kvs + 4 * q4;
        kk = *reinterpret_cast<const float4*>(p);
        vv = *reinterpret_cast<const float4*>(p + vd);
      }
      att_commit(Ks, Vs, key, q4, kk, vv);
    }
    __syncthreads();
    const int ntile = min(SKT, T - kc + 31) >> 5;
    const _Float16* vcur = vph;
    for (int t32 = 0; t32 < (SKT >> 5); ++t32) {
      if (t32 >= ntile) break;
      att_tile(&Ks[(t32 * 32 + i) * SKP + 8 * h], vcur, qh, ql, kc + t32 * 32, T, h, st);
      vcur += 32 * SVP;       // next 32-key tile
    }
  }
  att_store(st.o, qi < S ? a.out + ((size_t)b * S + qi) * os + hd * d : nullptr, h, d);
}

// Short sequences (the LightGlue matcher: one image pair is 2 sequences x 4 heads x 1024 keypoints).  The kernel
// above gives a wave 32 queries and ALL keys: at one pair that is 64 workgroups, each wave a serial chain of 32 key
// tiles and 8 barrier-fenced staging rounds with nothing else resident to hide them (23.7 us per launch, 40 % of the
// matcher, profiles/r3_lightglue_kernels.txt).  Here the four waves of a workgroup share 32 queries and split the KEYS:
// wave w owns keys [w Tw, (w+1) Tw), stages them itself 32 at a time into its own quarter of the LDS images (no
// workgroup barrier in the loop: LDS operations of one wave execute in order; the next round's rows are fetched while
// the current tile is multiplied), and the four partial (m, l, O) meet once in LDS at the end.  Grid = 8x the
// workgroups, chain = T / 128 tiles.
__global__ __launch_bounds__(256, 4) void attention_ksplit_kernel(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 Ks[SKT * SKP];
  __shared__ __attribute__((aligned(16))) _Float16 Vs[SKT * SVP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int C = a.C, d = C / a.heads, S = a.S, TS = a.T;
  const int bk = a.kv_bshift ? (b + a.kv_bshift) % a.B : b;
  const int T = a.tcount ? min(TS, a.tcount[bk]) : TS;
  const int qs = a.q_stride ? a.q_stride : C, kvs = a.kv_stride ? a.kv_stride : 2 * C;
  const int vd = (a.v_off ? a.v_off : C) - a.k_off, os = a.out_stride ? a.out_stride : C;
  const int qi = blockIdx.x * 32 + i;
  h16x8 qh, ql;
  att_load_q(a, a.q + ((size_t)b * S + (qi < S ? qi : 0)) * qs + hd * d, qi < S, h, d, qh, ql);
  AttAcc st;
#pragma unroll
  for (int r = 0; r < 16; ++r) { st.o[r] = 0.f; st.negm[r] = 0.f; }
  st.m = 0.f;
  st.first = true;
  _Float16* const Kw = Ks + wave * 32 * SKP;             // this wave's 32 rows of the images
  _Float16* const Vw = Vs + wave * 32 * SVP;
  for (int e = lane; e < 32 * 4; e += 64) {              // the constant columns of the V rows (see the kernel above)
    const int row = e >> 2, pt = e & 3;
    h16x8 c = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pt == 0) c[0] = (_Float16)1.f;
    *reinterpret_cast<h16x8*>(&Vw[row * SVP + 16 + 32 * (pt >> 1) + 8 * (pt & 1)]) = c;
  }
  const _Float16* const vph = &Vw[(4 * h + ((lane & 15) >> 2)) * SVP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)];
  const float* kvb = a.kv + (size_t)bk * TS * kvs + a.k_off + hd * d;
  const int Tw = ((T + 127) >> 7) << 5;                  // keys per wave, a multiple of the 32-key tile
  const int k0 = wave * Tw, k1 = min(T, k0 + Tw);
  // granule (key, 4 channels) of a round: lane -> key lane / 4 (+ 16), channels 4 (lane % 4)
  const int gkey = lane >> 2, gq = lane & 3;
  float4 kk[2], vv[2];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      kk[u] = make_float4(0.f, 0.f, 0.f, 0.f); vv[u] = kk[u];
      const int key = kc + gkey + 16 * u;
      if (key < k1 && 4 * gq < d) {
        const float* p = kvb + (size_t)key * kvs + 4 * gq;
        kk[u] = *reinterpret_cast<const float4*>(p);
        vv[u] = *reinterpret_cast<const float4*>(p + vd);
      }
    }
  };
  if (k0 < k1) fetch(k0);
  for (int kc = k0; kc < k1; kc += 32) {
    att_commit(Kw, Vw, gkey, gq, kk[0], vv[0]);
    att_commit(Kw, Vw, gkey + 16, gq, kk[1], vv[1]);
    if (kc + 32 < k1) fetch(kc + 32);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (ordering only: one wave, in-order LDS)
    att_tile(&Kw[i * SKP + 8 * h], vph, qh, ql, kc, T, h, st);      // (wave ranges are whole tiles: only the sequence's last tile is ragged)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // merge: wave 0 adds the other waves' partial sums, rescaled to the common reference
  const float mw = k0 < k1 ? st.m : -INFINITY;            // (a wave without keys carries no weight)
  // (a wave's partial — m and O rows 0..8 of every lane, pitch 11 floats — goes where its V rows were)
  auto part = [&](int w) { return reinterpret_cast<float*>(Vs + w * 32 * SVP) + lane * 11; };
  if (wave) {
    float* pw = part(wave);
    pw[0] = mw;
#pragma unroll
    for (int r = 0; r < 9; ++r) pw[1 + r] = st.o[r];
  }
  __syncthreads();
  if (wave) return;
  float mm = mw;
#pragma unroll
  for (int w = 1; w < 4; ++w) mm = fmaxf(mm, part(w)[0]);
  {
    const float al = __builtin_amdgcn_exp2f(mw - mm);
#pragma unroll
    for (int r = 0; r < 9; ++r) st.o[r] *= al;
  }
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const float* pw = part(w);
    const float al = __builtin_amdgcn_exp2f(pw[0] - mm);
#pragma unroll
    for (int r = 0; r < 9; ++r) st.o[r] = fmaf(pw[1 + r], al, st.o[r]);
  }
  att_store(st.o, qi < S ? a.out + ((size_t)b * S + qi) * os + hd * d : nullptr, h, d);
}

int launch_attention(const AttnArgs& a, hipStream_t s) {
  const int d = a.C / a.heads;
  if (a.C % a.heads || d > 64 || (d & 3) || (a.C & 3)) return -1402;
  if (a.prec == 1 && d <= 16) {
    // short sequences that leave most of the chip idle in the query-tiled kernel: split the keys over the waves
    static const long ks_max = getenv("KP2D_ATT_KSPLIT") ? atol(getenv("KP2D_ATT_KSPLIT")) : 256;
    if ((long)((a.S + 127) / 128) * a.heads * a.B < ks_max && a.T >= 128) {
      hipLaunchKernelGGL(attention_ksplit_kernel, dim3((a.S + 31) / 32, a.heads, a.B), dim3(256), 0, s, a);
      return (int)hipGetLastError();
    }
    // 256 queries per workgroup halve the K / V staging per query (18 % of the kernel at 128); short sequences
    // keep 128 so that the grid still covers the chip
    static const int big = getenv("KP2D_ATT_Q") ? atoi(getenv("KP2D_ATT_Q")) : 256;
    // (tiles * pairs, 1, 1): XCD-affine pair order (see the kernel); needs pairs % 8 == 0
    const bool affine = (a.heads * a.B) % 8 == 0 && !(getenv("KP2D_ATT_AFFINE") && getenv("KP2D_ATT_AFFINE")[0] == '0');
    if (big == 256 && (long)((a.S + 255) / 256) * a.heads * a.B >= 512) {
      const int tiles = (a.S + 255) / 256;
      hipLaunchKernelGGL((attention_split_kernel<512, 6>), affine ? dim3(tiles * a.heads * a.B) : dim3(tiles, a.heads, a.B), dim3(512), 0, s, a);
    } else {
      const int tiles = (a.S + 127) / 128;
      hipLaunchKernelGGL((attention_split_kernel<256, 4>), affine ? dim3(tiles * a.heads * a.B) : dim3(tiles, a.heads, a.B), dim3(256), 0, s, a);
    }
    return (int)hipGetLastError();
  }
  const dim3 grid((a.S + 63) / 64, a.heads, a.B);
  if (d <= 16) hipLaunchKernelGGL(attention_kernel<1>, grid, dim3(256), 0, s, a);
  else if (d <= 32) hipLaunchKernelGGL(attention_kernel<2>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(attention_kernel<4>, grid, dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace kp2d
