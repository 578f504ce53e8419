// 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on the gfx950 matrix cores, exact fp32.
//
// Replaces every Conv2d(k=3) + BatchNorm2d(eval) + LeakyReLU stack of the reference
// (AnnotatedConvBnReLUModel, modules/base.py:14-46) and the plain biased 3x3 convs of the heads
// (modules/decoders/heads.py:22,72-85; modules/decoders/segmentation.py:108,277-282), with
// MaxPool2d(2,2) (encoders.py:100), PixelShuffle(2) (heads.py:54) and torch.cat (heads.py:99)
// folded into the load / store side instead of being separate passes over HBM.
//
// Mapping (one 256-thread workgroup = 4 waves):
//   * GEMM view: M = output pixels, N = output channels (padded to 32), K = 9 taps x Cin.
//   * Workgroup tile: 16 x 16 output pixels x all N.  Wave w owns rows 4w..4w+3 as two M-tiles of
//     2 rows x 16 cols; inside an M-tile the MFMA row index m maps to the pixel
//     (row = (m>>1)&1, col = 2*(m>>2) + (m&1)), so the four accumulator registers r..r+3 of a lane
//     are one 2x2 pixel block: max-pooling is a register-local max, no LDS, no second kernel.
//   * K is walked in chunks of KC input channels.  Per chunk the 18x18 halo tile of the (possibly
//     concatenated) NHWC input and the [9][N][KC] weight slab are staged into LDS with 16-byte
//     loads; both LDS images are [row][KC+4] so every MFMA operand fetch is one ds_read_b128 and
//     the 2x16 M-tile with a 24-pixel row pitch is bank-conflict free (pitch == 8 mod 16 pixels,
//     pixel stride 20 or 12 floats: see DESIGN.md "LDS images").
//   * v_mfma_f32_32x32x2_f32: lane (i = lane&31, h = lane>>5) supplies A[i][k=h], B[k=h][i]; the
//     lane's KC/2 channels are [h*KC/2, (h+1)*KC/2) so one b128 read feeds 4 MFMAs.
//   * Epilogue on the fp32 accumulator in the reference's order: per-channel affine (BatchNorm
//     scale/shift or bias), activation, then the store mode (NHWC / pooled / both / pixel-shuffled /
//     planar NCHW through an LDS transpose for the API-facing tensors).
//
// Two arithmetic modes share the tiling, the LDS geometry and the epilogue (template parameter PREC):
//   PREC 0  exact fp32: v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fma chain (157 TFLOP/s peak).
//   PREC 1  split fp16 ("f16x3"): every fp32 operand is carried as hi = fp16(x), lo = fp16(x - hi) and
//           x*w is formed as xh*wh + xh*wl + xl*wh on v_mfma_f32_32x32x16_f16 with ONE fp32 accumulator
//           (weights pre-scaled by 2^11 at pack time so wl stays a normal fp16; the epilogue scale carries
//           2^-11).  gfx950 honours fp16 subnormal MFMA operands and the dropped xl*wl term is 2^-22
//           relative, so the result is fp32-grade: measured max error vs fp64 1.7e-6 at K = 576 against
//           1.9e-6 for the exact fp32 chain (tools/probes/mfma_f16_probe.hip, profiles/r1_probe_f16.log).
//           3 MFMAs of 16x the fp32 rate -> 5.3x fewer matrix-core cycles per product.
#include <cstdlib>

#include "conv_common.h"
#include "device_guard.h"

#ifndef KP2D_PITCH_NT1
#define KP2D_PITCH_NT1 20
#endif

namespace kp2d {

constexpr int IN_ROWS = 18;
// LDS row pitch of the input image in pixels.  32-channel tiles (NT = 1) use 20 so that image + weight slab is
// 51.8 KB and THREE workgroups fit a CU (160 registers per thread allow it); the 64-channel tiles use 24.
template <int NT, int WST> struct InPitch { static constexpr int v = (NT == 1 || WST > 1) ? KP2D_PITCH_NT1 : 24; };

// WST = weight stages per K chunk.  1: all taps of the chunk's weight slab sit in LDS at once.  3 (64-channel tiles):
// one tap row (3 taps) at a time, restaged between MFMA segments: 28.8 + 15.4 KB of LDS instead of 80.6 KB and a
// 12-register weight prefetch instead of 36, which is what lets THREE of these workgroups share a CU.
template <int KC, int NT, int TAPS, int PREC, int WST = 1>
__global__ __launch_bounds__(256, ((NT == 1 || WST > 1) && KP2D_PITCH_NT1 < 24) ? 3 : 2) void conv3x3_f32_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int IN_PITCH = InPitch<NT, WST>::v;
  static_assert(TAPS % WST == 0, "weight stages must divide the taps");
  constexpr int STAPS = TAPS / WST;       // taps per weight stage
  constexpr int KCP = KC + 4;
  constexpr int N = NT * 32;
  constexpr int Q = KC / 4;
  constexpr int KH = KC / 2;
  float* s_in = smem;
  float* s_w = smem + IN_ROWS * IN_PITCH * KCP;

  const int tid = threadIdx.x;
  // hwreg(HW_REG_MODE, offset 23, size 1) = FP16_OVFL: fp16 results that overflow clamp to +-65504
  if (PREC == 1) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int i = lane & 31;
  const int h = lane >> 5;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2), so give each
  // XCD a contiguous run of tiles: neighbouring tiles (shared halo rows/columns) then hit the same L2.
  int bid = blockIdx.x;
  if (!KP2D_DBG_ON(64)) {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;     // bijective for any grid size
  }
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int b = bid / a.tiles_y;
  const int y0 = ty * TILE, x0 = tx * TILE;
  const int H = a.H, W = a.W;
  const int n0 = blockIdx.y * N;   // output-channel group handled by this workgroup

  f32x16 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int prow = wave * 4 + ((i >> 1) & 1);
  const int pcol = 2 * (i >> 2) + (i & 1);
  const int a_base = (prow * IN_PITCH + pcol) * KCP + h * KH;
  const int b_base = i * KCP + h * KH;
  const int a_base16 = (prow * IN_PITCH + pcol) * KCP + h * 4;   // split mode: +h*16 B inside the hi block
  const int b_base16 = i * KCP + h * 4;

  const int nchunk = (a.cin + KC - 1) / KC;
  const float* src0 = a.in0.p + (size_t)b * a.in0.bs + a.in0.o;
  const float* src1 = a.in1.p + (size_t)b * a.in1.bs + a.in1.o;
  const int c0 = a.in0.c;
  // Software pipeline over K chunks: the global loads of chunk c+1 are issued into registers before the
  // MFMA loop of chunk c and only written to LDS (with the fp32 -> hi/lo split in PREC 1) after it, so the
  // L2/HBM latency hides under the matrix-core work of this wave and of the co-resident workgroup.
  // A 1x1 convolution reads only the centre 16x16 pixels of the LDS image: stage those (at their halo-offset
  // positions, so the operand reads are the same code) instead of the 18x18 halo tile — 21 % fewer loads,
  // splits and LDS writes, 4 staging iterations instead of 6.
  constexpr int ST_ROWS = TAPS == 1 ? 16 : IN_ROWS;
  constexpr int ST_OFF = TAPS == 1 ? 1 : 0;
  constexpr int IN_G = ST_ROWS * ST_ROWS * Q;
  constexpr int IN_IT = (IN_G + 255) / 256;
  constexpr int W_G = STAPS * N * Q;      // weight granules per stage
  constexpr int W_IT = (W_G + 255) / 256;
  float4 rin[IN_IT], rw[W_IT];

  // Per-thread staging granules, computed ONCE: granule g = tid + 256*it is quad q = tid % Q of halo pixel
  // p = g / Q.  Recomputing the divisions by 18 and the 64-bit addresses for every chunk cost ~600 VALU
  // instructions per chunk per thread (KP2D_DBG=15 "skeleton" runs: 1.0 ms of a 4.2 ms forward).
  const int st_q4 = 4 * (tid % Q);
  // byte offsets from the frame base of each source; OOB marks halo pixels outside the image (and the unused tail
  // granules): a buffer load at that offset is out of range and returns zeros, which IS the zero padding
  constexpr int OOB = 0x7ffffff0;
  int st_off0[IN_IT], st_off1[IN_IT], st_lds[IN_IT];
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int g = tid + 256 * it;
    const int p = g / Q;
    const int sy = p / ST_ROWS;
    const int py = sy + ST_OFF, px = p - sy * ST_ROWS + ST_OFF;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool ok = g < IN_G && gy >= 0 && gy < H && gx >= 0 && gx < W;
    st_off0[it] = ok ? ((int)(gy * a.in0.rs + gx * a.in0.ps) + st_q4) * 4 : OOB;
    st_off1[it] = ok ? ((int)(gy * a.in1.rs + gx * a.in1.ps) + st_q4) * 4 : OOB;
    st_lds[it] = (py * IN_PITCH + px) * KCP;
  }
  const int w_lds = (tid / Q) * KCP + 4 * (tid % Q);     // weight granule it lands at w_lds + it * (256 / Q) * KCP

  // Buffer resources: one per source (the frame's slice, so num_records bounds every legal access) and one for this
  // channel group's packed weights.  MUBUF loads take a 32-bit VGPR offset + an SGPR offset: no 64-bit address
  // arithmetic and no predication per granule.
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(src0), 0, (int)((a.in0.bs - a.in0.o) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(src1), 0, (int)((a.in1.bs - a.in1.o) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.w + (size_t)blockIdx.y * nchunk * TAPS * N * KC), 0, nchunk * TAPS * N * KC * 4, 0x00020000);
  // chunks never straddle the two sources and never run past cin when both are multiples of KC (every S config)
  const bool uniform = ((c0 | a.cin) & (KC - 1)) == 0;

  auto prefetch_w = [&](int ch, int stage) {
#pragma unroll
    for (int it = 0; it < W_IT; ++it)
      rw[it] = KP2D_DBG_ON(4) ? make_float4(0.f, 0.f, 0.f, 0.f)
                              : __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                    rsw, tid * 16, ((ch * WST + stage) * W_G + 256 * it) * 16, 0));
  };
  auto prefetch_in = [&](int ch) {
    if (KP2D_DBG_ON(4)) {
#pragma unroll
      for (int it = 0; it < IN_IT; ++it) rin[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
    if (uniform) {
      if (ch * KC < c0) {
        const int so = ch * KC * 4;
#pragma unroll
        for (int it = 0; it < IN_IT; ++it)
          rin[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs0, st_off0[it], so, 0));
      } else {
        const int so = (ch * KC - c0) * 4;
#pragma unroll
        for (int it = 0; it < IN_IT; ++it)
          rin[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs1, st_off1[it], so, 0));
      }
    } else {
      // per-thread source / tail selection (channel counts that are not multiples of KC: the N configs)
      const int c = ch * KC + st_q4;
      const bool first = c < c0;
      const int so = (first ? ch * KC : ch * KC - c0) * 4;      // per thread: goes into the VGPR offset
      const bool cok = c < a.cin;
#pragma unroll
      for (int it = 0; it < IN_IT; ++it) {
        const int o0 = (cok && first) ? st_off0[it] + so : OOB;
        const int o1 = (cok && !first) ? st_off1[it] + so : OOB;
        const i32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs0, o0, 0, 0);
        const i32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs1, o1, 0, 0);
        rin[it] = __builtin_bit_cast(float4, v0 | v1);      // the other one is all zeros
      }
    }
  };

  auto commit_in = [&]() {
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      if (IN_G % 256 != 0 && tid + 256 * it >= IN_G) continue;
      const float4 v = rin[it];
      if (PREC == 0) {
        *reinterpret_cast<float4*>(&s_in[st_lds[it] + st_q4]) = v;
      } else {
        // pixel row (80 B): [16 x fp16 hi][16 x fp16 lo][pad]; conversions saturate at the fp16 range
        // (MODE.FP16_OVFL, set at kernel entry) instead of producing infinities
        f16x2 h0, h1, l0, l1;
        split2(v.x, v.y, h0, l0);
        split2(v.z, v.w, h1, l1);
        _Float16* row = reinterpret_cast<_Float16*>(&s_in[st_lds[it]]);
        *reinterpret_cast<f16x4*>(row + st_q4) = f16x4{h0[0], h0[1], h1[0], h1[1]};
        *reinterpret_cast<f16x4*>(row + 16 + st_q4) = f16x4{l0[0], l0[1], l1[0], l1[1]};
      }
    }
  };
  auto commit_w = [&]() {
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      if (W_G % 256 != 0 && tid + 256 * it >= W_G) continue;
      *reinterpret_cast<float4*>(&s_w[w_lds + it * ((256 / Q) * KCP)]) = rw[it];
    }
  };

  prefetch_in(0);
  prefetch_w(0, 0);
  for (int ch = 0; ch < nchunk; ++ch) {
#pragma unroll
   for (int stage = 0; stage < WST; ++stage) {
    __syncthreads();          // every wave is done reading the previous stage's LDS image
    if (!KP2D_DBG_ON(2)) {
      if (stage == 0) commit_in();
      commit_w();
    }
    __syncthreads();
    // next loads: the following weight stage of this chunk, or the next chunk's input tile + first weight stage
    if (stage + 1 < WST) prefetch_w(ch, stage + 1);
    else if (ch + 1 < nchunk) { prefetch_in(ch + 1); prefetch_w(ch + 1, 0); }

    if (!KP2D_DBG_ON(8))
#pragma unroll
    for (int tl = 0; tl < STAPS; ++tl) {
      const int tap = stage * STAPS + tl;      // tap in the 3x3 window; tl indexes the staged slab
      const int dy = TAPS == 9 ? tap / 3 : 1, dx = TAPS == 9 ? tap - 3 * (tap / 3) : 1;
      if (PREC == 0) {
        float av[2][KH], bv[NT][KH];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const float* p = &s_in[a_base + ((2 * m + dy) * IN_PITCH + dx) * KCP];
#pragma unroll
          for (int j = 0; j < KH; j += 4) {
            const float4 t = *reinterpret_cast<const float4*>(p + j);
            av[m][j] = t.x; av[m][j + 1] = t.y; av[m][j + 2] = t.z; av[m][j + 3] = t.w;
          }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float* p = &s_w[b_base + (tl * N + n * 32) * KCP];
#pragma unroll
          for (int j = 0; j < KH; j += 4) {
            const float4 t = *reinterpret_cast<const float4*>(p + j);
            bv[n][j] = t.x; bv[n][j + 1] = t.y; bv[n][j + 2] = t.z; bv[n][j + 3] = t.w;
          }
        }
#pragma unroll
        for (int j = 0; j < KH; ++j)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][j], bv[n][j], acc[m][n], 0, 0, 0);
      } else {
        // lane (i,h): k = 8h..8h+7 of the 16-channel chunk; hi block at +0, lo block at +8 floats (32 B)
        f16x8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const float* p = &s_in[a_base16 + ((2 * m + dy) * IN_PITCH + dx) * KCP];
          ah[m] = *reinterpret_cast<const f16x8*>(p);
          al[m] = *reinterpret_cast<const f16x8*>(p + 8);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float* p = &s_w[b_base16 + (tl * N + n * 32) * KCP];
          bh[n] = *reinterpret_cast<const f16x8*>(p);
          bl[n] = *reinterpret_cast<const f16x8*>(p + 8);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
          }
      }
    }
   }
  }

  constexpr int EPI_ROUNDS = (WST > 1 && NT == 2) ? 2 : 1;
  constexpr bool EPI_GELU = TAPS == 1;
  // accumulator layout of the 32x32 MFMA tiles (conv_epilogue.inc): lane (i, h), M-tile m, register r
#define EPI_NM 2
#define EPI_NN NT
#define EPI_R 16
#define EPI_ACC(m, n, r) acc[m][n][r]
#define EPI_CH(n) ((n) * 32 + i)
#define EPI_ROW(m, r) (wave * 4 + 2 * (m) + (((r) >> 1) & 1))
#define EPI_COL(m, r) (2 * h + 4 * ((r) >> 2) + ((r) & 1))
#define EPI_THREADS 256
#include "conv_epilogue.inc"
#undef EPI_THREADS
#undef EPI_NM
#undef EPI_NN
#undef EPI_R
#undef EPI_ACC
#undef EPI_CH
#undef EPI_ROW
#undef EPI_COL
}

template <int KC, int NT, int TAPS, int PREC, int WST = 1>
static int launch_t(const ConvArgs& a, hipStream_t s) {
  static_assert(PREC == 0 || KC == 16, "split-fp16 mode walks K in chunks of 16 (one 32x32x16 MFMA)");
  constexpr int KCP = KC + 4;
  constexpr int IN_PITCH = InPitch<NT, WST>::v;
  size_t lds = (size_t)(IN_ROWS * IN_PITCH * KCP + (TAPS / WST) * NT * 32 * KCP) * sizeof(float);
  const size_t lds_out = (size_t)NT * 32 * 260 * sizeof(float);      // conv_epilogue.inc: planes of 16 x 16 + 4 floats
  if (a.store == ST_NCHW && lds_out > lds) lds = lds_out;
  // [pixel][N] staging tile of the fp32 NHWC epilogue; half the pixel rows per round for the weight-staged 64-channel tile
  const size_t lds_tile = (size_t)16 * 16 * NT * 32 * sizeof(float) / ((WST > 1 && NT == 2) ? 2 : 1);
  if (a.store != ST_NCHW && lds_tile > lds) lds = lds_tile;
  static PerDeviceOnce lds_once;      // per device: a handle may live on any visible device
  if (int e = lds_opt_in(lds_once, reinterpret_cast<const void*>(&conv3x3_f32_kernel<KC, NT, TAPS, PREC, WST>))) return e;
  const int grid = a.tiles_x * a.tiles_y * a.B;
  const int groups = a.npad / (NT * 32);
  if (a.store == ST_NCHW && groups != 1 && a.act == ACT_SOFTMAX_C) return -1002;  // class softmax needs one group
  hipLaunchKernelGGL((conv3x3_f32_kernel<KC, NT, TAPS, PREC, WST>), dim3(grid, groups), dim3(256), lds, s, a);
  return (int)hipGetLastError();
}

int launch_conv3x3(const ConvArgs& a, int kc, hipStream_t s) {
  // npad is 32, or a multiple of 64 handled as npad/64 channel groups (blockIdx.y)
  if (a.npad != 32 && a.npad % 64 != 0) return -1000;
  const bool one = a.npad == 32 || a.ng32;
  conv3x3_note_variant("");
  if (a.prec == 1) {
    if (kc != 16) return -1003;
    if (a.taps == 9) return launch_conv3x3_f16x3(a, s);      // conv3x3_f16.hip
    if (a.taps == 1) return one ? launch_t<16, 1, 1, 1>(a, s) : launch_t<16, 2, 1, 1>(a, s);
    return -1000;
  }
  static const bool wst3f = !(getenv("KP2D_WST") && getenv("KP2D_WST")[0] == '1');
  if (a.taps == 9) {
    if (kc == 16) return one ? launch_t<16, 1, 9, 0>(a, s) : (wst3f ? launch_t<16, 2, 9, 0, 3>(a, s) : launch_t<16, 2, 9, 0>(a, s));
    if (kc == 8) return one ? launch_t<8, 1, 9, 0>(a, s) : launch_t<8, 2, 9, 0>(a, s);
  } else if (a.taps == 1) {
    if (kc == 16) return one ? launch_t<16, 1, 1, 0>(a, s) : launch_t<16, 2, 1, 0>(a, s);
    if (kc == 8) return one ? launch_t<8, 1, 1, 0>(a, s) : launch_t<8, 2, 1, 0>(a, s);
  }
  return -1000;
}

// ---------------------------------------------------------------------------------------------
// backbone.conv1a (encoders.py:20-29): Cin = 3, reads the caller's NCHW frame directly, writes NHWC.
// K = 27 is too shallow for the matrix cores and the layer is HBM-bound (11 FLOP/B): one thread
// per pixel, all output channels in registers, weights broadcast from LDS.
// ---------------------------------------------------------------------------------------------
template <int CO, int CIN>
__global__ __launch_bounds__(256) void conv1a_kernel(const Conv1aArgs a) {
  __shared__ __attribute__((aligned(16))) float4 s_o[256 * 4];
  static_assert(CO == 16, "conv1a: the output staging assumes 16 channels (4 float4 per pixel)");
  const int H = a.H, W = a.W;
  const size_t npix = (size_t)a.B * H * W;
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p < npix) {
    const int x = (int)(p % W);
    const int y = (int)((p / W) % H);
    const int b = (int)(p / ((size_t)W * H));
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    // 27 taps as buffer loads: a tap outside the image gets an offset past the descriptor's range, which the
    // hardware answers with 0 — no lane-dependent branches (the ternary form compiled to 64 exec-mask branches).
    // The descriptor covers this block's frames only (offsets stay 32-bit for any batch size).
    constexpr int NK = 9 * CIN;
    float v[NK];
    {
      const size_t fb = ((size_t)blockIdx.x * 256) / ((size_t)W * H);          // first frame this block touches
      const size_t frames_left = (size_t)a.B - fb;
      const size_t nf = 256 / ((size_t)W * H) + 2;                              // frames a 256-pixel block can touch
      const size_t span = (frames_left < nf ? frames_left : nf) * CIN * (size_t)H * W * sizeof(float);
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(a.x + fb * CIN * (size_t)H * W), 0, (int)span, 0x00020000);
      const int fo = ((b - (int)fb) * CIN * H * W) * 4;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int xx = x + dx - 1;
          const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
          const int o = ok ? fo + (yy * W + xx) * 4 : 0x7ffffff0;
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci)    // the plane stride goes into the scalar offset
            v[ci * 9 + dy * 3 + dx] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, o, ci * H * W * 4, 0));
        }
      }
    }
    // The weight row of a tap is the same for every lane: read it with a uniform index so that it arrives
    // through the scalar cache in SGPRs (s_load_dwordx16) and feeds the FMAs as a scalar operand.  As LDS
    // broadcast reads (4 ds_read_b128 per tap per wave) the weights alone kept the LDS array busy for ~75 us
    // of this kernel.
    const float* wg = a.w;
#pragma unroll 3
    for (int k = 0; k < NK; ++k) {
      const float vk = v[k];
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[c] = fmaf(vk, wg[k * CO + c], acc[c]);
    }
    // BatchNorm affine + LeakyReLU / ReLU / identity as max(v, v*slope)
    const float slope = a.act == ACT_LEAKY ? 0.01f : (a.act == ACT_RELU ? 0.f : 1.f);
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      const float t = fmaf(acc[c], a.scale[c], a.shift[c]);     // uniform index: scalar loads
      acc[c] = fmaxf(t, t * slope);
    }
    // a thread's four float4 go to slots j ^ ((tid >> 1) & 3) of its 64-byte row: straight slots put lanes t, t + 2, t + 4,
    // t + 6 of every 8-lane store group on the same banks (64-byte lane stride = 16 of the 32 store banks): 4-way
    // conflicts, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.667 (profiles/r3_pmc_summary.txt)
#pragma unroll
      for (int c = 0; c < CO; c += 4)
        s_o[threadIdx.x * 4 + ((c / 4) ^ ((threadIdx.x >> 1) & 3))] = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
  }
  // The 256 pixels of this workgroup are one contiguous 16 KiB run of the NHWC output: go through LDS so every
  // store instruction of a wave writes 1 KiB of consecutive bytes (per-thread 64-byte rows cost 1.6x the write
  // traffic: PMC WRITE_SIZE 505 MB vs 315 MB per launch, profiles/r1_traffic.json).
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * 256;
  float4* dst = reinterpret_cast<float4*>(a.out + base * CO);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = j * 256 + threadIdx.x;          // float4 index inside the block's run
    if (base + (e >> 2) < npix) dst[e] = s_o[(e & ~3) | ((e & 3) ^ ((e >> 3) & 3))];      // (undo the slot swizzle of pixel e >> 2)
  }
}

// Any other first-layer width (LARGE_D: 64): same one-pixel-per-thread scheme, 16 output channels at a time,
// each thread writing its own 64-byte rows.  Not tuned: only the 16-wide layer is on a measured path.
__global__ __launch_bounds__(256) void conv1a_wide_kernel(const Conv1aArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];
  const int CO = a.cout, NK = 9 * a.cin;
  float* s_w = s_dyn;                 // [NK][CO]
  float* s_sc = s_dyn + 27 * CO;      // [CO]
  float* s_sh = s_sc + CO;            // [CO]
  for (int t = threadIdx.x; t < NK * CO; t += 256) s_w[t] = a.w[t];
  for (int t = threadIdx.x; t < CO; t += 256) { s_sc[t] = a.scale[t]; s_sh[t] = a.shift[t]; }
  __syncthreads();
  const int H = a.H, W = a.W;
  const size_t npix = (size_t)a.B * H * W;
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  const int x = (int)(p % W);
  const int y = (int)((p / W) % H);
  const int b = (int)(p / ((size_t)W * H));
  float v[27];
#pragma unroll
  for (int ci = 0; ci < 3; ++ci) {
    if (ci >= a.cin) { for (int t = 0; t < 9; ++t) v[ci * 9 + t] = 0.f; continue; }
    const float* plane = a.x + ((size_t)b * a.cin + ci) * H * W;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y + dy - 1;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = x + dx - 1;
        v[ci * 9 + dy * 3 + dx] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? plane[(size_t)yy * W + xx] : 0.f;
      }
    }
  }
  const float slope = a.act == ACT_LEAKY ? 0.01f : (a.act == ACT_RELU ? 0.f : 1.f);
  for (int c0 = 0; c0 < CO; c0 += 16) {
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll 3
    for (int k = 0; k < 27; ++k) {
      if (k >= NK) break;
      const float4* wr = reinterpret_cast<const float4*>(&s_w[k * CO + c0]);
      const float vk = v[k];
#pragma unroll
      for (int c = 0; c < 16; c += 4) {
        const float4 w4 = wr[c / 4];
        acc[c] = fmaf(vk, w4.x, acc[c]);
        acc[c + 1] = fmaf(vk, w4.y, acc[c + 1]);
        acc[c + 2] = fmaf(vk, w4.z, acc[c + 2]);
        acc[c + 3] = fmaf(vk, w4.w, acc[c + 3]);
      }
    }
    float4* dst = reinterpret_cast<float4*>(a.out + p * CO + c0);
#pragma unroll
    for (int c = 0; c < 16; c += 4) {
      float r[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float t = fmaf(acc[c + i], s_sc[c0 + c + i], s_sh[c0 + c + i]);
        r[i] = fmaxf(t, t * slope);
      }
      dst[c / 4] = make_float4(r[0], r[1], r[2], r[3]);
    }
  }
}

int launch_conv1a(const Conv1aArgs& a, hipStream_t s) {
  const size_t npix = (size_t)a.B * a.H * a.W;
  const int grid = (int)((npix + 255) / 256);
  // conv1a_kernel addresses its taps with 32-bit byte offsets inside a descriptor of at most
  // 256 / (H W) + 2 frames
  if ((256 / ((size_t)a.H * a.W) + 2) * 3 * (size_t)a.H * a.W * sizeof(float) >= 0x7ffffff0u) return -1002;
  if (a.cin != 3 && a.cin != 1) return -1001;
  if (a.cout == 16 && a.cin == 3) hipLaunchKernelGGL((conv1a_kernel<16, 3>), dim3(grid), dim3(256), 0, s, a);
  else if (a.cout == 16) hipLaunchKernelGGL((conv1a_kernel<16, 1>), dim3(grid), dim3(256), 0, s, a);
  else if (a.cout % 16 == 0 && a.cout <= 256)
    hipLaunchKernelGGL(conv1a_wide_kernel, dim3(grid), dim3(256), (size_t)29 * a.cout * sizeof(float), s, a);
  else return -1001;
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// conv1a with the frame front-end as its prologue (src/evaluation/visual_odometry.py:77-87 +
// encoders.py:20-29): reads the caller's uint8 HWC frames, and the /255 -> bilinear resize (align_corners=False, no
// antialias) -> (v - 0.5) * 2 step happens while the 18 x 18 x 3 halo tile of a 16 x 16 output tile is staged into
// LDS — the float [B,3,H,W] frame of kp2d_preprocess (12 B per pixel written, then read back) never exists.
// Same arithmetic, in the same order, as preprocess_kernel (post.hip) followed by conv1a_kernel: bit-identical.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1a_u8_kernel(const Conv1aArgs a, const unsigned char* __restrict__ frames, int Hs, int Ws) {
  constexpr int CO = 16, HP = 18, HPP = 19;            // halo rows / LDS row pitch (floats)
  __shared__ float s_in[3 * HP * HPP];
  __shared__ __attribute__((aligned(16))) float4 s_o[256 * 4];
  const int H = a.H, W = a.W;
  const int tiles_x = (W + 15) >> 4, tiles_y = (H + 15) >> 4;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = ty * 16, x0 = tx * 16;
  const unsigned char* img = frames + (size_t)b * Hs * Ws * 3;
  for (int e = threadIdx.x; e < HP * HP; e += 256) {
    const int py = e / HP, px = e - py * HP;
    const int y = y0 - 1 + py, x = x0 - 1 + px;
    float v[3] = {0.f, 0.f, 0.f};                       // the convolution's zero padding (of the NORMALISED frame)
    if (y >= 0 && y < H && x >= 0 && x < W) frame_pixel(img, Hs, Ws, H, W, y, x, v);      // conv_common.h (shared with kp2d_preprocess)
#pragma unroll
    for (int c = 0; c < 3; ++c) s_in[(c * HP + py) * HPP + px] = v[c];
  }
  __syncthreads();
  const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
  const float* wg = a.w;
#pragma unroll 3
  for (int k = 0; k < 27; ++k) {
    const int ci = k / 9, dy = (k % 9) / 3, dx = k % 3;
    const float vk = s_in[(ci * HP + ly + dy) * HPP + lx + dx];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = fmaf(vk, wg[k * CO + c], acc[c]);     // uniform index: scalar operands
  }
  const float slope = a.act == ACT_LEAKY ? 0.01f : (a.act == ACT_RELU ? 0.f : 1.f);
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    const float t = fmaf(acc[c], a.scale[c], a.shift[c]);
    acc[c] = fmaxf(t, t * slope);
  }
#pragma unroll
  for (int c = 0; c < CO; c += 4) s_o[threadIdx.x * 4 + c / 4] = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
  __syncthreads();
  // a tile row (16 pixels x 64 B) is one contiguous KiB of the NHWC output: every wave store covers four of them
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = j * 256 + threadIdx.x;           // float4 index inside the tile: pixel e >> 2, quad e & 3
    const int p = e >> 2, y = y0 + (p >> 4), x = x0 + (p & 15);
    if (y < H && x < W) reinterpret_cast<float4*>(a.out + (((size_t)b * H + y) * W + x) * CO)[e & 3] = s_o[e];
  }
}

int launch_conv1a_u8(const Conv1aArgs& a, const unsigned char* frames, int Hs, int Ws, hipStream_t s) {
  if (a.cout != 16 || a.cin != 3) return -1001;     // the 16-wide RGB first layer (every S / N / F config)
  const int grid = ((a.W + 15) >> 4) * ((a.H + 15) >> 4) * a.B;
  hipLaunchKernelGGL(conv1a_u8_kernel, dim3(grid), dim3(256), 0, s, a, frames, Hs, Ws);
  return (int)hipGetLastError();
}

}  // namespace kp2d
