// MixFeedForward's tail as ONE kernel (modules/segformer.py:182-206: self.net[1:] of MixFeedForward inside
// SegFormerAttentionModule, :209-220):   h -> DsConv2d = depthwise 3x3 + bias (net.1.net.0) -> 1x1 128 -> 128 + bias (net.1.net.1)
// -> GELU (net.2) -> 1x1 128 -> 64 + bias (net.3) [-> the MaxPool2d(2, 2) SegmentationFeatHeadLightATT puts behind the first module].
//
// As three launches these layers move 11 tensor-units (a unit = a 64-channel fp32 map: h is two) through HBM for 49 kFLOP per
// pixel: dwconv3x3 0.136 + conv1x1 0.194 + conv1x1 0.092 ms at 480 x 640 x 32 frames (profiles/r5_layers_cfg4.txt), every one of
// them at 3.2-5.2 TB/s — bound by bytes, not by arithmetic.  Fused, h is read once (with a one-pixel halo) and only the 64-channel
// result is written: 3.5 units.
//
// A 512-thread workgroup per 16 x 16 pixel tile (persistent: tiles t0, t0 + G, ...), wave w = pixel rows 2 w, 2 w + 1 (32 pixels):
//   phase A, per 16-channel slice of the hidden width (8 slices = the 8 K-steps of the first product):
//     the slice of the 18 x 18 halo of h -> LDS by buffer_load ... lds (fp32, [pixel][16]; out-of-map pixels = the zero padding),
//     THREE buffers: the copies run two slices ahead (across tile boundaries), behind counted waits — a workgroup is alone on
//     its CU (512 threads, ~145 KB of LDS), so nothing else would cover an HBM round trip (first form: 32-channel slices, one
//     buffer, the copy of slice s + 1 issued after the depthwise pass of slice s: 0.278 ms, every slice waited for its copy);
//     depthwise 3x3 + bias in fp32 FMAs (tap order and arithmetic of dwconv3x3_kernel, attention.hip) -> split -> fp16 hi / lo
//     operand image [pixel][16 k];  D1^T[co][pixel] += W1[co][k-slice] . A^T  on v_mfma_f32_32x32x16_f16 (split-fp16 products
//     as everywhere in this precision mode: wh xl + wl xh + wh xh), weights = A operand, pixels = B operand: a wave holds ALL
//     128 output channels of its 32 pixels (4 accumulator tiles);
//   phase B, in registers: + bias, GELU (conv_common.h gelu_fast, as the 1x1 kernel's epilogue), split — and the result IS the
//     B operand of the next product: accumulator register r of lane (pixel, h) holds channel (r & 3) + 8 (r >> 2) + 4 h of its
//     tile, so registers 8 s .. 8 s + 7 of both lane halves are one 16-channel K-step of net.3 in a fixed permutation, and
//     net.3's weights are laid out in LDS in that same permutation.  No LDS round trip, no barrier between the two 1x1 layers.
//     D2^T[co2][pixel] = W3 . g^T: 64 channels x 32 pixels per wave (2 accumulator tiles), + bias, stored (or 2 x 2 max-pooled
//     across lanes first).
// Tolerance-level parity with the three-launch form (same formulas; the matrix products sum K in another grouping).
#include "conv_common.h"
#include "device_guard.h"

namespace kp2d {

namespace {
constexpr int MF_CH = 128, MF_CO = 64, MF_T = 16, MF_HALO = MF_T + 2, MF_NPX = MF_T * MF_T;
constexpr int MF_SL = 16, MF_NS = MF_CH / MF_SL;         // hidden channels per slice (one K-step of the first product), slices
constexpr int MF_HS = MF_HALO * MF_HALO * MF_SL * 4;     // halo slice, fp32 [pixel][16 channels] (20,736 B)
constexpr int MF_NHS = 3;                                // halo buffers: the copies run two slices ahead of the depthwise pass
constexpr int MF_APITCH = 48;                            // operand rows: 16 k halves = 32 B + 16 B (conflict-light ds_read_b128)
constexpr int MF_AD = MF_NPX * MF_APITCH;                // one plane of the depthwise output's operand image (12,288 B)
constexpr int MF_W1 = MF_CH * MF_APITCH;                 // one plane of a W1 slice (6,144 B)
constexpr int MF_W3PITCH = MF_CH * 2 + 16;               // a W3 row: 128 k halves + 16 B
constexpr int MF_W3 = MF_CO * MF_W3PITCH;                // one plane of W3 (17,408 B)
constexpr int MF_O_HS = 0;
constexpr int MF_O_AD = MF_O_HS + MF_NHS * MF_HS;
constexpr int MF_O_W1 = MF_O_AD + 2 * MF_AD;             // two buffers x (hi | lo)
constexpr int MF_O_W3 = MF_O_W1 + 4 * MF_W1;
constexpr int MF_O_DW = MF_O_W3 + 2 * MF_W3;             // this slice's depthwise weights [9][16] + bias [16]
constexpr int MF_O_B1 = MF_O_DW + 10 * MF_SL * 4;
constexpr int MF_O_B3 = MF_O_B1 + MF_CH * 4;
constexpr int MF_LDS = MF_O_B3 + MF_CO * 4;
static_assert(MF_LDS <= 160 * 1024, "LDS budget");
constexpr int MF_PIECES = (MF_HS + 1023) / 1024;         // 21 LDS-DMA pieces per halo slice (the last one: 16 lanes)
constexpr int MF_PW = 3;                                 // pieces per wave and slice: 8 x 3 = 24 slots, the last three copy pieces 0-2 again
static_assert(MF_PIECES <= 8 * MF_PW && MF_HS - 1024 * (MF_PIECES - 1) == 256, "piece bookkeeping");
typedef float f32x16v __attribute__((ext_vector_type(16)));
}  // namespace

__global__ __launch_bounds__(512, 2) void mff_tail_kernel(const MffTailArgs a, const int ntiles, const int tiles_x, const int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float mf_smem[];
  char* const sm = reinterpret_cast<char*>(mf_smem);
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);       // FP16_OVFL: conversions that overflow clamp to +-65504
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int OOB = 0x7ffffff0;
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));

  // ---- once: net.3's weights in the K permutation phase B hands its operands over in; both biases ----
  // position p of K-step t holds channel 16 t + {0,1,2,3,8,9,10,11 | 4,5,6,7,12,13,14,15}[p]
  {
    const _Float16* src = reinterpret_cast<const _Float16*>(a.w3);      // rows [chunk t][co2]: 16 hi | 16 lo halves
    for (int e = tid; e < MF_CO * 8 * 16; e += 512) {
      const int p = e & 15, t = (e >> 4) & 7, co = e >> 7;
      const int ch = (p & 3) + ((p >> 2) & 1) * 8 + (p >> 3) * 4;
      const _Float16* row = src + ((size_t)t * MF_CO + co) * 32;
      *reinterpret_cast<_Float16*>(sm + MF_O_W3 + co * MF_W3PITCH + (t * 16 + p) * 2) = row[ch];
      *reinterpret_cast<_Float16*>(sm + MF_O_W3 + MF_W3 + co * MF_W3PITCH + (t * 16 + p) * 2) = row[16 + ch];
    }
    for (int e = tid; e < MF_CH; e += 512) reinterpret_cast<float*>(sm + MF_O_B1)[e] = a.sh1[e];
    for (int e = tid; e < MF_CO; e += 512) reinterpret_cast<float*>(sm + MF_O_B3)[e] = a.sh3[e];
  }
  const float sc1 = a.sc1[0], sc3 = a.sc3[0];        // 2^-e of the weight split (bias layers: the scale is uniform)

  // W1 slice s (hidden channels 16 s .. + 15 as K) -> buffer `buf`: row [group co / 64][chunk s][co % 64] of the packed layer — and
  // this slice's depthwise weights [9][16] + bias [16].  Fetched into registers BEFORE a slice's matrix phase and written to LDS
  // after it (a load -> wait -> ds_write in one place would park the wave for an L2 round trip in front of its MFMAs).
  float4 rw1;
  float rdw = 0.f;
  auto fetch_w = [&](int s) {
    const int co = tid >> 2, quad = tid & 3;
    const size_t row = ((size_t)(co >> 6) * 8 + s) * 64 + (co & 63);
    rw1 = reinterpret_cast<const float4*>(a.w1)[row * 4 + quad];
    if (tid < 10 * MF_SL) {
      const int t = tid / MF_SL, c = tid % MF_SL;
      rdw = t < 9 ? a.wdw[t * MF_CH + MF_SL * s + c] : a.bdw[MF_SL * s + c];
    }
  };
  auto commit_w = [&](int buf) {
    const int co = tid >> 2, quad = tid & 3;
    *reinterpret_cast<float4*>(sm + MF_O_W1 + buf * 2 * MF_W1 + (quad >> 1) * MF_W1 + co * MF_APITCH + (quad & 1) * 16) = rw1;
    if (tid < 10 * MF_SL) reinterpret_cast<float*>(sm + MF_O_DW)[tid] = rdw;
  };

  // halo slice by LDS-DMA: piece q covers LDS bytes [1024 q, + 1024) of [halo pixel][64 B]; lane -> (pixel, 16-byte part).  Wave w
  // copies pieces w, w + 8, w + 16 (slots 21-23 = pieces 0-2 a second time: every wave issues exactly three copies per slice,
  // which is what the counted wait below counts)
  int voff[MF_PW];
  int tb = 0, ty0 = 0, tx0 = 0;
  auto enter_tile = [&](int t) {
    const int per = tiles_x * tiles_y;
    tb = t / per;
    const int r = t - tb * per, tyy = r / tiles_x;
    ty0 = tyy * MF_T; tx0 = (r - tyy * tiles_x) * MF_T;
#pragma unroll
    for (int j = 0; j < MF_PW; ++j) {
      int q = wave + 8 * j;
      if (q >= MF_PIECES) q -= MF_PIECES;
      const int byte = 1024 * q + 16 * lane;
      const int hp = byte >> 6, part = (byte >> 4) & 3;
      const int hy = hp / MF_HALO, hx = hp - hy * MF_HALO;
      const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
      const bool ok = byte < MF_HS && gy >= 0 && gy < H && gx >= 0 && gx < W;
      voff[j] = ok ? ((gy * W + gx) * MF_CH + 4 * part) * 4 : OOB;
    }
  };
  auto request_hs = [&](int s, int buf) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.h) + (size_t)tb * H * W * MF_CH, 0, H * W * MF_CH * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < MF_PW; ++j) {
      int q = wave + 8 * j;
      if (q >= MF_PIECES) q -= MF_PIECES;
      if (q < MF_PIECES - 1 || lane < 16)            // (piece 20: the slice's last 256 bytes)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(sm + MF_O_HS + buf * MF_HS + 1024 * q), 16,
                                                 voff[j], s * MF_SL * 4, 0, 0);
    }
  };
  // Barriers without the fence of __syncthreads() (which would drain the LDS-DMA copies in flight): this wave's LDS writes are
  // complete (lgkmcnt(0)); KEEP = vector-memory operations that may stay in flight — the three copies of the newest request.
  auto bar = [&](bool keep3) {
    asm volatile("" ::: "memory");
    if (keep3) __builtin_amdgcn_s_waitcnt(0x0073);   // vmcnt(3) lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x0070);         // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  const int G = gridDim.x;
  int t = blockIdx.x;
  const int n_my = (ntiles - t + G - 1) / G;          // tiles of this workgroup (>= 1: the launcher keeps G <= ntiles)
  const long nsl = (long)n_my * MF_NS;                // its slices, as one sequence g = 0 .. nsl - 1
  // request cursor: two slices ahead
  int rq_s = 0, rq_t = t, rq_buf = 0;
  long rq_g = 0;
  auto request_next = [&]() {                         // copy slice rq_g (if there is one) into buffer rq_g % 3
    if (rq_g < nsl) {
      if (rq_s == 0) enter_tile(rq_t);
      request_hs(rq_s, rq_buf);
    }
    ++rq_g;
    rq_buf = rq_buf + 1 == MF_NHS ? 0 : rq_buf + 1;
    if (++rq_s == MF_NS) { rq_s = 0; rq_t += G; }
  };
  fetch_w(0);
  request_next();                                     // slice 0
  commit_w(0);
  request_next();                                     // slice 1
  long g = 0;
  int hb = 0;                                         // halo buffer of slice g

  for (; t < ntiles; t += G) {
    f32x16v acc1[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[ct][r] = 0.f;
    const int per = tiles_x * tiles_y;
    const int cb = t / per, cr = t - cb * per, cty = cr / tiles_x;
    const int cy0 = cty * MF_T, cx0 = (cr - cty * tiles_x) * MF_T;

#pragma unroll 1
    for (int s = 0; s < MF_NS; ++s, ++g) {
      // halo slice g has landed (only the copies of slice g + 1, if it exists, may still be in flight); W1 slice s and the depthwise
      // weights are in LDS; the operand image is free
      bar(g + 1 < nsl);
      // ---- depthwise 3x3 + bias -> split -> operand image ----
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = tid + 512 * u, px = i >> 2, q = i & 3;
        const int py = px >> 4, pxx = px & 15;
        const float* const dw = reinterpret_cast<const float*>(sm + MF_O_DW);
        float4 acc = *reinterpret_cast<const float4*>(dw + 9 * MF_SL + 4 * q);
        const char* const hp = sm + MF_O_HS + hb * MF_HS + ((py * MF_HALO + pxx) * MF_SL + 4 * q) * 4;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const float4 v = *reinterpret_cast<const float4*>(hp + ((tap / 3) * MF_HALO + tap % 3) * MF_SL * 4);
          const float4 w = *reinterpret_cast<const float4*>(dw + tap * MF_SL + 4 * q);
          acc.x = fmaf(v.x, w.x, acc.x); acc.y = fmaf(v.y, w.y, acc.y); acc.z = fmaf(v.z, w.z, acc.z); acc.w = fmaf(v.w, w.w, acc.w);
        }
        f16x2 h0, l0, h1, l1;
        split2(acc.x, acc.y, h0, l0);
        split2(acc.z, acc.w, h1, l1);
        *reinterpret_cast<h4*>(sm + MF_O_AD + px * MF_APITCH + q * 8) = h4{h0[0], h0[1], h1[0], h1[1]};
        *reinterpret_cast<h4*>(sm + MF_O_AD + MF_AD + px * MF_APITCH + q * 8) = h4{l0[0], l0[1], l1[0], l1[1]};
      }
      // operand image ready; halo buffer hb and the depthwise weights are free.  (The wait keeps the copies of slice g + 1 in flight.)
      bar(g + 1 < nsl);
      // ---- the next slice's weights (registers) and the halo slice after it (LDS-DMA into the buffer just freed... of slice g - 1:
      // buffer (g + 2) % 3) on their way while this slice multiplies.  Weights FIRST: their wait must not include the copies. ----
      const bool more = g + 1 < nsl;
      if (more) fetch_w(s + 1 < MF_NS ? s + 1 : 0);
      request_next();                                 // slice g + 2
      // ---- D1^T += W1[:, slice] . A^T  (one K-step) ----
      {
        const char* const wb = sm + MF_O_W1 + (int)(g & 1) * 2 * MF_W1 + li * MF_APITCH + lh * 16;
        const char* const ab = sm + MF_O_AD + (32 * wave + li) * MF_APITCH + lh * 16;
        const f16x8 bh = *reinterpret_cast<const f16x8*>(ab);
        const f16x8 bl = *reinterpret_cast<const f16x8*>(ab + MF_AD);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const f16x8 wh = *reinterpret_cast<const f16x8*>(wb + ct * 32 * MF_APITCH);
          const f16x8 wl = *reinterpret_cast<const f16x8*>(wb + MF_W1 + ct * 32 * MF_APITCH);
          acc1[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, acc1[ct], 0, 0, 0);
          acc1[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, acc1[ct], 0, 0, 0);
          acc1[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, acc1[ct], 0, 0, 0);
        }
      }
      // (W1 buffer (g + 1) & 1 was last read by slice g - 1's products, which every wave left before this slice's first barrier; the
      // depthwise weights were last read before this slice's second barrier)
      if (more) commit_w((int)((g + 1) & 1));
      hb = hb + 1 == MF_NHS ? 0 : hb + 1;
    }

    // ---- phase B: bias + GELU + split in registers -> the B operands of net.3 ----
    f32x16v acc2[2];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[c2][r] = 0.f;
    const float* const b1 = reinterpret_cast<const float*>(sm + MF_O_B1);
    const char* const w3b = sm + MF_O_W3 + li * MF_W3PITCH + lh * 16;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float g[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int r = 8 * s2 + p;
          const int ch = 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * lh;
          g[p] = gelu_fast(fmaf(acc1[ct][r], sc1, b1[ch]));
        }
        f16x2 hh[4], ll[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) split2(g[2 * p], g[2 * p + 1], hh[p], ll[p]);
        const f16x8 gh = {hh[0][0], hh[0][1], hh[1][0], hh[1][1], hh[2][0], hh[2][1], hh[3][0], hh[3][1]};
        const f16x8 gl = {ll[0][0], ll[0][1], ll[1][0], ll[1][1], ll[2][0], ll[2][1], ll[3][0], ll[3][1]};
        const int kt = 2 * ct + s2;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const f16x8 wh = *reinterpret_cast<const f16x8*>(w3b + c2 * 32 * MF_W3PITCH + kt * 32);
          const f16x8 wl = *reinterpret_cast<const f16x8*>(w3b + MF_W3 + c2 * 32 * MF_W3PITCH + kt * 32);
          acc2[c2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, gh, acc2[c2], 0, 0, 0);
          acc2[c2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, gl, acc2[c2], 0, 0, 0);
          acc2[c2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, gh, acc2[c2], 0, 0, 0);
        }
      }

    // ---- output: lane (pixel li, half lh) holds channels 32 c2 + 8 g4 + 4 lh .. + 3 in registers 4 g4 .. 4 g4 + 3 ----
    const float* const b3 = reinterpret_cast<const float*>(sm + MF_O_B3);
    const int y = cy0 + 2 * wave + (li >> 4), x = cx0 + (li & 15);
    const bool inside = y < H && x < W;
    if (!a.pool) {
      float* const op = a.out + (((size_t)cb * H + y) * W + x) * MF_CO;
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int ch = 32 * c2 + 8 * g4 + 4 * lh;
          float4 v;
          v.x = fmaf(acc2[c2][4 * g4 + 0], sc3, b3[ch + 0]);
          v.y = fmaf(acc2[c2][4 * g4 + 1], sc3, b3[ch + 1]);
          v.z = fmaf(acc2[c2][4 * g4 + 2], sc3, b3[ch + 2]);
          v.w = fmaf(acc2[c2][4 * g4 + 3], sc3, b3[ch + 3]);
          if (inside) *reinterpret_cast<float4*>(op + ch) = v;
        }
    } else {
      // MaxPool2d(2, 2): the 2 x 2 block = lanes li, li ^ 1 (column pair), li ^ 16 (the wave's other row); lane li with an even column
      // in the first row stores.  (H and W are even: a block is inside the map or outside it as a whole.)
      const int Hp = H >> 1, Wp = W >> 1;
      const int yp = (cy0 >> 1) + wave, xp = (cx0 + (li & 15)) >> 1;
      const bool st = (li & 17) == 0 && yp < Hp && xp < Wp;
      float* const op = a.out + (((size_t)cb * Hp + yp) * Wp + xp) * MF_CO;
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int ch = 32 * c2 + 8 * g4 + 4 * lh;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float m = fmaf(acc2[c2][4 * g4 + i], sc3, b3[ch + i]);
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fmaxf(m, __shfl_xor(m, 16));
            v[i] = m;
          }
          if (st) *reinterpret_cast<float4*>(op + ch) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
  }
}

int launch_mff_tail(const MffTailArgs& a, hipStream_t s) {
  if (a.B < 1 || a.H < 1 || a.W < 1) return -1600;
  if (a.pool && ((a.H | a.W) & 1)) return -1601;
  if ((long)a.H * a.W * MF_CH * 4 >= 0x7ffffff0L) return -1602;
  static PerDeviceOnce lds_once;
  if (int e = lds_opt_in(lds_once, reinterpret_cast<const void*>(&mff_tail_kernel))) return e;
  const int cus = device_cu_count();
  const int tiles_x = (a.W + MF_T - 1) / MF_T, tiles_y = (a.H + MF_T - 1) / MF_T;
  const long ntiles = (long)tiles_x * tiles_y * a.B;
  if (ntiles >= (1L << 30)) return -1602;
  const int grid = (int)(ntiles < cus ? ntiles : cus);
  hipLaunchKernelGGL(mff_tail_kernel, dim3(grid), dim3(512), MF_LDS, s, a, (int)ntiles, tiles_x, tiles_y);
  return (int)hipGetLastError();
}

}  // namespace kp2d
