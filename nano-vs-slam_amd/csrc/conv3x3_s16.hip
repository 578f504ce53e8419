// 3x3 convolution layers with 32 input channels whose INPUT is already split ("S16P" activations): warp-specialised,
// persistent, the operand image copied HBM -> LDS by buffer_load ... lds with no vector instruction in between.
//
// Same operator and arithmetic as conv3x3_f16x3_wsm_kernel (conv3x3_wsm.hip: AnnotatedConvBnReLUModel, modules/base.py:14-46,
// inside BackBone modules/encoders.py:105-129 — conv2a, conv2b, conv3a (32 -> 32) and conv3b (32 -> 64, = skip, + pool)):
// x = xh + xl, w = wh + wl (fp16 halves), fourteen v_mfma_f32_16x16x32_f16 per 16-channel chunk and accumulator tile, the same
// slot order, the same LDS image layout, the same multiplying waves.  Results are bit-identical to the other forms: the
// hi / lo halves a consumer multiplies are those its own staging would have produced from the fp32 activation (conv_common.h
// split2 on the same value, in the PRODUCER's epilogue instead).
//
// Why.  These layers move 4 (Cin + Cout) = 256-384 bytes per pixel for 18.4-36.9 kFLOP: at the matrix rate the chip sustains
// they are bound by HBM (conv2a: 315 MB per 64 frames = 52 us at 6 TB/s against 45 us of matrix time), and the general
// kernels spent their time on the way INTO LDS instead (2.65 vector instructions per MFMA, mfma_busy 0.36, 3.5 TB/s:
// profiles/r4_pmc_summary.txt).  A 32-channel variant of the register-staging warp-specialised form lost too
// (profiles/r4_ab_wsm32.txt): a step was as long as the staging waves' load -> split -> ds_write chain.  Here nothing of
// that chain is left:
//   * S16P activation layout, per frame: [16-channel chunk][row y][plane: hi | lo][x][16 halves] — the bytes of a tile row
//     of one plane are CONTIGUOUS in memory and land as they are in the LDS image ([row][36 px][32 B] planes), so a halo
//     image is 41 LDS-DMA pieces of 1 KiB (64 lanes x 16 B, per-lane source address, out-of-range lanes write the zero
//     padding) issued by four otherwise idle waves: no VGPR staging, no split, no ds_write;
//   * the layer's WHOLE weight pack (2 chunks x 18 / 36 KB) stays in LDS for the launch, so a step moves only its image;
//   * with 32-channel items three image stages fit beside the weights (3 x 41,472 + 36,864 B): the DMA waves run TWO
//     steps ahead of the multiplying waves (83 KB in flight per CU), with 64-channel items two stages, one step ahead
//     (a step is twice as long there);
//   * ST_S16P output (conv2a / 2b / 3a): the epilogue splits in registers and stores [8 hi halves] / [8 lo halves] of a
//     pixel as two 16-byte stores — the same bytes and the same number of stores as the fp32 NHWC form.  For that a lane
//     must hold EIGHT consecutive channels of its pixel: the weight rows of the two N-tiles are interleaved on the way into
//     LDS (N-tile n, row 4 g + i = channel 8 g + 4 n + i).
//
// Barrier contract: both role loops run exactly 1 + nsteps barriers (one before step 0, one after every step); the role
// branch is wave-uniform (readfirstlane of the wave index) and no barrier sits under a condition that differs between
// waves of a role.
#include "conv_common.h"
#include "device_guard.h"
#include <cstdlib>
#include <type_traits>

namespace kp2d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int D_TH = 16, D_TW = 32, D_PITCH = 36, D_ROWS = D_TH + 2, D_COLS = D_TW + 2, D_PXB = 32;
constexpr int D_LO = D_ROWS * D_PITCH * D_PXB;         // byte offset of an image's lo plane (20,736)
constexpr int D_IMG = 2 * D_LO;                        // one input image: hi plane | lo plane (41,472 B)
constexpr int D_NPIECE = (D_IMG + 1023) / 1024;        // 1-KiB LDS-DMA pieces of an image (41: the last one is half a piece)
constexpr int D_PW = (D_NPIECE + 3) / 4;               // pieces per DMA wave and step (11; 44 slots: three duplicates)
constexpr int D_MAXC = 64;                             // output channels (scale / shift vectors in LDS)
// nch = 16-channel chunks of the input: 2 (cin = 32: conv2a .. conv3b) or 4 (cin = 64 -> 32 planar output channels: confBb, convs.8)
__host__ __device__ constexpr int d_wl(int n) { return 9 * n * 32; }                  // byte offset of the wl plane behind the wh plane
__host__ __device__ constexpr int d_wslab(int n) { return 2 * d_wl(n); }              // a chunk's weight slab (18,432 / 36,864 B)
__host__ __device__ constexpr int d_nimg(int n, int nch) { return n * nch == 64 ? 3 : 2; }      // image stages beside the resident weights
__host__ __device__ constexpr int d_wbase(int n, int nch) { return d_nimg(n, nch) * D_IMG; }  // weights behind the images
__host__ __device__ constexpr int d_ss(int n, int nch) { return d_wbase(n, nch) + nch * d_wslab(n); }
__host__ __device__ constexpr int d_lds(int n, int nch) { return d_ss(n, nch) + 2 * D_MAXC * 4; }   // 161,792 / 157,184 / 157,184 B
constexpr int D_THREADS = 768;
static_assert(d_lds(32, 2) <= 160 * 1024 && d_lds(64, 2) <= 160 * 1024 && d_lds(32, 4) <= 160 * 1024, "LDS budget");
static_assert(D_NPIECE == 41 && D_IMG - 1024 * (D_NPIECE - 1) == 512, "the last piece is its first 32 lanes");

__host__ __device__ constexpr int d_slot_tap(int s) { return s == 2 ? 3 : s == 3 ? 4 : s == 4 ? 2 : s; }

struct S16Item { int b, y0, x0; };
}  // namespace

// STORE: ST_S16P (NN = 2), ST_NHWC / ST_NHWC_POOL / ST_NHWC_BOTH / ST_S16P_BOTH (NN = 4), or ST_NCHW (NN = 2, D_NCH = 4: the
// planar API outputs behind a 64-channel S16P tensor).  One channel group: cout <= 16 NN.
// ST_NCHW multiplies UN-transposed (pixels are the A operand, an M-tile is 1 row x 16 columns): a lane's four accumulator
// registers of a tile are four consecutive PIXELS of one channel — a 16-byte store into the channel plane, the four lane
// groups of a tile 64 contiguous bytes, no transposition through LDS.  Each output element is the same sum of the same
// products in the same order as in the transposed tiles (an MFMA's dot products do not know which operand is called A).
template <int STORE, int NN, int D_NCH>
__global__ __launch_bounds__(D_THREADS, 3) void conv3x3_f16x3_s16_kernel(const ConvArgs a, const int nitems) {
  constexpr int D_N = 16 * NN, D_WL = d_wl(D_N), D_WSLAB = d_wslab(D_N), D_NIMG = d_nimg(D_N, D_NCH), D_WB = d_wbase(D_N, D_NCH), D_SS = d_ss(D_N, D_NCH);
  static_assert((STORE == ST_S16P || STORE == ST_NCHW) ? NN == 2 : NN == 4, "32 output channels: S16P or planar; 64: fp32 forms or S16P full + pooled");
  static_assert(D_NCH == (STORE == ST_NCHW ? 4 : 2), "64 input channels: the planar form only");
  constexpr bool S16OUT = STORE == ST_S16P || STORE == ST_S16P_BOTH;
  constexpr bool PLANAR = STORE == ST_NCHW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const sm = reinterpret_cast<char*>(smem);
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);       // FP16_OVFL: conversions that overflow clamp to +-65504
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wave < 8;                    // wave-uniform by construction (see the barrier contract above)
  const int H = a.H, W = a.W;
  constexpr int OOB = 0x7ffffff0;

  // ---- this workgroup's items (16 x 32 pixel tiles): j = i G + off; every XCD owns a contiguous run per round; cheap
  // tiles (half of the waves idle) come last — conv3x3_wsm.hip ----
  const int G = gridDim.x, t0 = blockIdx.x;
  const int off = (t0 & 7) * (G >> 3) + (t0 >> 3);
  const int n_my = (nitems - off + G - 1) / G;       // >= 1: the launcher keeps G <= nitems
  const int nsteps = n_my * D_NCH;
  const int tx_n = a.tiles_x, ty_n = a.tiles_y;
  const int cc = (W - (tx_n - 1) * D_TW <= 16 && tx_n > 1) ? 1 : 0;
  const int rr = (H - (ty_n - 1) * D_TH <= 8 && ty_n > 1) ? 1 : 0;
  const int per_a = (ty_n - rr) * (tx_n - cc), n_a = per_a * a.B;      // per_a >= 1
  const int per_b = cc * (ty_n - rr), n_b = per_b * a.B;
  auto decode = [&](int i) -> S16Item {
    int u = i * G + off;
    S16Item r;
    if (u < n_a) {
      r.b = u / per_a;
      const int q = u - r.b * per_a, ty = q / (tx_n - cc);
      r.y0 = ty * D_TH; r.x0 = (q - ty * (tx_n - cc)) * D_TW;
    } else if (u < n_a + n_b) {
      u -= n_a;
      r.b = u / per_b;
      r.y0 = (u - r.b * per_b) * D_TH; r.x0 = (tx_n - 1) * D_TW;
    } else {
      u -= n_a + n_b;
      r.b = u / tx_n;
      r.y0 = (ty_n - 1) * D_TH; r.x0 = (u - r.b * tx_n) * D_TW;
    }
    return r;
  };

  // ---- once per launch: scale | shift and the layer's weights -> LDS.  Packed rows [chunk][slot][n][16 hi | 16 lo] ->
  // per chunk [wh plane | wl plane], each [slot][n'][32 B].  ST_S16P interleaves the two N-tiles' rows: channel
  // c = 8 g + 4 n + i sits in row n' = 16 n + 4 g + i, so lane group g of the accumulator tiles (n = 0, 1) holds channels
  // 8 g .. 8 g + 7 of its pixel ----
  for (int c = tid; c < D_MAXC; c += D_THREADS) {
    smem[D_SS / 4 + c] = c < a.npad ? a.scale[c] : 0.f;
    smem[D_SS / 4 + D_MAXC + c] = c < a.npad ? a.shift[c] : 0.f;
  }
  for (int gi = tid; gi < D_NCH * 9 * D_N * 4; gi += D_THREADS) {
    const int row = gi >> 2, quad = gi & 3;          // quads 0, 1: the 16 hi halves; 2, 3: the 16 lo halves
    const int n = row % D_N, cs = row / D_N;         // cs = chunk * 9 + slot
    const int np = S16OUT ? (32 * (n >> 5) + 16 * ((n >> 2) & 1) + 4 * ((n >> 3) & 3) + (n & 3)) : n;      // (N-tile PAIRS interleaved)
    const int chunk = cs / 9, slot = cs - 9 * chunk;
    *reinterpret_cast<float4*>(sm + D_WB + chunk * D_WSLAB + (quad >> 1) * D_WL + (slot * D_N + np) * 32 + (quad & 1) * 16) =
        reinterpret_cast<const float4*>(a.w)[gi];
  }

  if (!consumer) {
    // =================================== DMA waves ===================================
    // piece q of an image = LDS bytes [1024 q, 1024 q + 1024) of [hi plane | lo plane]; lane l brings bytes 16 l .. + 15 of
    // it: halo pixel slot (py, px), plane, 16-byte half.  Wave pw owns pieces pw + 4 j (j < 11); 41 pieces over 44 slots:
    // the last slots of waves 1-3 copy pieces 0-2 a second time (same bytes to the same place), so that every wave issues
    // the same straight-line sequence and the counted wait below means the same thing in each.  Piece 40 is half a piece:
    // only its first 32 lanes copy.
    const int pw = wave - 8;
    int t_yx[D_PW], t_ph[D_PW];
#pragma unroll
    for (int j = 0; j < D_PW; ++j) {
      int q = pw + 4 * j;
      if (q >= D_NPIECE) q -= D_NPIECE;
      const int byte = 1024 * q + 16 * lane;
      const int plane = byte >= D_LO ? 1 : 0, pb = byte - plane * D_LO;
      const int sidx = pb >> 5, py = sidx / D_PITCH, px = sidx - py * D_PITCH;
      t_yx[j] = (byte < D_IMG && px < D_COLS) ? (py << 8) | px : -1;      // pitch columns 34, 35 / past the image: zeros
      t_ph[j] = plane * W * 32 + ((pb >> 4) & 1) * 16;
    }
    const long fb = a.in0.bs * 4;                    // frame stride, bytes
    const int cs_bytes = H * 2 * W * 32;             // chunk stride, bytes
    int rq_i = 0, rq_ch = 0, rq_b = 0;
    int voff[D_PW];                                  // byte offset of each piece's 16 bytes inside (frame, chunk), or OOB
    auto enter_item = [&](int i) {
      const S16Item r = decode(i);
      rq_b = r.b;
      const int y0 = r.y0 - 1, x0 = r.x0 - 1;
#pragma unroll
      for (int j = 0; j < D_PW; ++j) {
        const int gy = y0 + (t_yx[j] >> 8), gx = x0 + (t_yx[j] & 255);
        const bool ok = t_yx[j] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        voff[j] = ok ? (gy * 2 * W + gx) * 32 + t_ph[j] : OOB;
      }
    };
    enter_item(0);
    // the step under the cursor -> image `stage`, then advance.  Past the last step the cursor stays on it: the look-ahead
    // requests beyond the end re-read the last step's operands into stages nobody multiplies (every step issues the same
    // number of vector-memory operations — the counted wait — and no load ever leaves the tensor)
    auto request = [&](int stage) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(a.in0.p) + (size_t)rq_b * a.in0.bs, 0, (int)fb, 0x00020000);
      const int so = rq_ch * cs_bytes;
      char* const dst = sm + stage * D_IMG;
#pragma unroll
      for (int j = 0; j < D_PW; ++j) {
        int q = pw + 4 * j;
        if (q >= D_NPIECE) q -= D_NPIECE;
        if (j == D_PW - 1) {
          // wave 0: piece 40, the image's last 512 bytes (lanes 32-63 would write into the next image); waves 1-3: a duplicate
          if (pw != 0 || lane < 32)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + 1024 * q), 16,
                                                     KP2D_DBG_ON(32) ? OOB : voff[j], so, 0, 0);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + 1024 * q), 16,
                                                   KP2D_DBG_ON(32) ? OOB : voff[j], so, 0, 0);
        }
      }
      if (rq_ch + 1 < D_NCH) ++rq_ch;
      else if (rq_i + 1 < n_my) { rq_ch = 0; enter_item(++rq_i); }
      __builtin_amdgcn_sched_barrier(0);
    };
    // vmcnt((LA - 1) D_PW): everything but the newest LA - 1 requests is done, i.e. the step the multiplying waves enter
    // after the barrier has landed (LDS-DMA counts in vmcnt, in issue order)
    constexpr int LA = D_NIMG - 1;                   // steps the requests run ahead
    constexpr int KEEP = (LA - 1) * D_PW;
    constexpr int WAITC = (KEEP & 15) | (7 << 4) | (15 << 8) | ((KEEP >> 4) << 14);
    auto wait_landed = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(WAITC);
      __builtin_amdgcn_sched_barrier(0);
    };
    int stage_in = 0;
#pragma unroll
    for (int k = 0; k < LA; ++k) { request(stage_in); stage_in = stage_in + 1 == D_NIMG ? 0 : stage_in + 1; }
    wait_landed();
    __syncthreads();                                 // barrier 0: stage 0 holds step 0 (and the weights are in LDS)
    for (int s = 0; s < nsteps; ++s) {
      request(stage_in);                             // step s + LA -> the stage step s - 1 was multiplied out of
      stage_in = stage_in + 1 == D_NIMG ? 0 : stage_in + 1;
      wait_landed();                                 // step s + 1 has landed
      // a bare s_barrier: __syncthreads() is fence + barrier, and the fence makes the compiler drain vmcnt while LDS-DMA
      // copies (of step s + 2) are in flight.  This role writes LDS by DMA only — ordered by the counted wait above — and
      // reads none; gfx950 barriers do not drain vector memory (MI355X_MICROARCH.md, "Two waves per SIMD" item 7)
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);              // the look-ahead copies past the end: landed before the LDS is handed back
    return;
  }

  // =================================== multiplying waves (as conv3x3_wsm.hip) ===================================
  // wave (wr, ph) owns tile rows 4 wr .. 4 wr + 3 x columns 16 ph .. 16 ph + 15 = four M-tiles of 2 x 8 pixels, and all NN
  // 16-channel N-tiles; products transposed (weights = A operand, pixels = B): a lane's four accumulator registers of a
  // tile are four consecutive rows of the weight image = channels, of ONE pixel.
  constexpr int MT = 4, CB = 2;
  const int lg = lane >> 4, lp = lane & 15;
  const int ph = wave >> 2, wr = (wave + 2 * ph) & 3;
  // (PLANAR: M-tile m = row 4 wr + m, columns 16 ph .. + 15, lane lp = column; else 2 x 8 pixel M-tiles, pooling quads in 4 lanes)
  const int a0 = PLANAR ? ((wr * 4) * D_PITCH + 16 * ph + lp) * D_PXB + 16 * (lg & 1)
                        : ((wr * 4 + ((lp >> 1) & 1)) * D_PITCH + 16 * ph + 2 * (lp >> 2) + (lp & 1)) * D_PXB + 16 * (lg & 1);
  const int a_dx = a0 + (lg >> 1) * D_PXB;                 // second tap one pixel to the right
  const int a_dy = a0 + (lg >> 1) * D_PITCH * D_PXB;       // second tap one row down
  const int a_s = a0 + (lg >> 1) * D_LO;                   // single tap: k-groups 2, 3 read the lo plane
  const int b_s = D_WB + lp * 32 + 16 * (lg & 1);
  const int b_p = b_s + (lg >> 1) * D_N * 32;              // second tap = next slot
  auto tap_off = [](int t) constexpr { return ((t / 3) * D_PITCH + (t % 3)) * D_PXB; };
  const float slope = a.act == ACT_LEAKY ? 0.01f : (a.act == ACT_RELU ? 0.f : 1.f);
  constexpr int store = STORE;

  f32x4 acc[MT][NN];
  auto clear = [&]() {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto multiply = [&](int ib, int wb) {              // byte offsets of the step's image and of its chunk's weight slab
#pragma unroll
    for (int slot = 0; slot < 9; slot += 2) {
      const int tp = d_slot_tap(slot);
      const bool single = slot == 8;
      const bool dy = slot == 4;
      const int ab = ib + (single ? a_s : (dy ? a_dy : a_dx)) + tap_off(tp);
      const int bb = wb + (single ? b_s : b_p) + slot * D_N * 32;
      f16x8 bh[NN], bl[NN];
#pragma unroll
      for (int n = 0; n < NN; ++n) {
        bh[n] = *reinterpret_cast<const f16x8*>(sm + bb + n * 512);
        bl[n] = *reinterpret_cast<const f16x8*>(sm + bb + n * 512 + D_WL);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int mo = PLANAR ? m * D_PITCH * D_PXB : (2 * (m / CB) * D_PITCH + 8 * (m % CB)) * D_PXB;
        auto mma = [&](const f16x8 w, const f16x8 x, const f32x4 c) -> f32x4 {      // weights x pixels, or (PLANAR) pixels x weights
          return PLANAR ? __builtin_amdgcn_mfma_f32_16x16x32_f16(x, w, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, c, 0, 0, 0);
        };
        if (single) {
          const f16x8 x = *reinterpret_cast<const f16x8*>(sm + ab + mo);
#pragma unroll
          for (int n = 0; n < NN; ++n) {
            acc[m][n] = mma(bl[n], x, acc[m][n]);
            acc[m][n] = mma(bh[n], x, acc[m][n]);
          }
        } else {
          const f16x8 zh = *reinterpret_cast<const f16x8*>(sm + ab + mo);
          const f16x8 zl = *reinterpret_cast<const f16x8*>(sm + ab + mo + D_LO);
#pragma unroll
          for (int n = 0; n < NN; ++n) {
            acc[m][n] = mma(bh[n], zl, acc[m][n]);
            acc[m][n] = mma(bl[n], zh, acc[m][n]);
            acc[m][n] = mma(bh[n], zh, acc[m][n]);
          }
        }
      }
    }
  };

  // ---- epilogue (registers -> memory, no LDS, no barrier).  Stores keep soffset = 0: the gfx950 store hazard of
  // conv3x3_wsm.hip (profiles/r4_wsm_store_hazard.txt; tests/test_host_logic.py reads the code objects for it) ----
  constexpr bool full = store != ST_NHWC_POOL;
  constexpr bool pooled = store == ST_NHWC_POOL || store == ST_NHWC_BOTH;
  const int Hp = H >> 1, Wp = W >> 1;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto affine = [&](const f32x4 v, const f32x4 sc, const f32x4 sh) -> f32x4 {      // BN scale / shift + LeakyReLU / ReLU
    const f32x2 a01 = {v[0], v[1]}, a23 = {v[2], v[3]};
    const f32x2 t01 = __builtin_elementwise_fma(a01, f32x2{sc[0], sc[1]}, f32x2{sh[0], sh[1]});
    const f32x2 t23 = __builtin_elementwise_fma(a23, f32x2{sc[2], sc[3]}, f32x2{sh[2], sh[3]});
    const f32x2 u01 = t01 * slope, u23 = t23 * slope;
    return f32x4{fmaxf(t01[0], u01[0]), fmaxf(t01[1], u01[1]), fmaxf(t23[0], u23[0]), fmaxf(t23[1], u23[1])};
  };
  auto finish = [&](const S16Item& it) {
    const int prow = (lp >> 1) & 1, pcol = 2 * (lp >> 2) + (lp & 1);      // this lane's pixel inside a 2 x 8 M-tile
    if constexpr (PLANAR) {
      // tile (m, n): lane (lp, lg) holds channel 16 n + lp, pixels (row 4 wr + m, columns 16 ph + 4 lg .. + 3).  Plain logits
      // (no activation), every channel into out0 [B, cout, H, W]; W is a multiple of 4, so a lane's four pixels are inside or
      // outside the map together.  (Not for a layer that also writes the dense class map, ConvArgs::ids_out: its channels sit
      // in 32 different lanes here, and the argmax by DPP rotations cost more than the form gains — kp2d_api.cpp s16_planar)
      const int obs = a.cout * H * W;
      const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(a.out0 + (size_t)it.b * obs, 0, obs * 4, 0x00020000);
      const int x = it.x0 + 16 * ph + 4 * lg;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int y = it.y0 + wr * 4 + m;
        const int inv = (y < H && x < W) ? 0 : OOB;
#pragma unroll
        for (int n = 0; n < NN; ++n) {
          const int c = 16 * n + lp;
          const float sc = smem[D_SS / 4 + c], sh = smem[D_SS / 4 + D_MAXC + c];
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[m][n][r], sc, sh);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rs0,
                                                 KP2D_DBG_ON(64) ? OOB : ((((c * H + y) * W + x) * 4) | inv | (c < a.cout ? 0 : OOB)), 0, 0);
        }
      }
    } else if constexpr (S16OUT) {
      // per N-tile pair p: lane group lg holds channels 32 p + 8 lg .. + 7 of its pixel (tiles 2 p, 2 p + 1): chunk
      // 2 p + (lg >> 1), halves 8 (lg & 1) .. + 7.  ST_S16P_BOTH (conv3b): also the 2 x 2 maximum, as a second S16P tensor
      constexpr bool both = store == ST_S16P_BOTH;
      const int obs = H * W * a.cout;                                     // output frame stride, floats (= bytes / 4)
      const int obp = Hp * Wp * a.cout;
      const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(a.out0 + (size_t)it.b * obs, 0, obs * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(a.out1 + (size_t)it.b * obp, 0, both ? obp * 4 : 0, 0x00020000);
      auto split_store = [&](const f32x4 v0, const f32x4 v1, const __amdgpu_buffer_rsrc_t rs, int o, int lo_off, int inv) {
        f16x2 h0, l0, h1, l1, h2, l2, h3, l3;
        split2(v0[0], v0[1], h0, l0);
        split2(v0[2], v0[3], h1, l1);
        split2(v1[0], v1[1], h2, l2);
        split2(v1[2], v1[3], h3, l3);
        const i32x4 hi = {__builtin_bit_cast(int, h0), __builtin_bit_cast(int, h1), __builtin_bit_cast(int, h2), __builtin_bit_cast(int, h3)};
        const i32x4 lo = {__builtin_bit_cast(int, l0), __builtin_bit_cast(int, l1), __builtin_bit_cast(int, l2), __builtin_bit_cast(int, l3)};
        __builtin_amdgcn_raw_buffer_store_b128(hi, rs, KP2D_DBG_ON(64) ? OOB : (o | inv), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(lo, rs, KP2D_DBG_ON(64) ? OOB : ((o + lo_off) | inv), 0, 0);
      };
      auto quad_max = [&](const f32x4 v) -> f32x4 {                       // maximum over lanes 4 q .. 4 q + 3 (conv3x3_wsm.hip)
        f32x4 p;
        asm("s_nop 1\n\t"
            "v_max_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1"
            : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
        return p;
      };
#pragma unroll
      for (int pr = 0; pr < NN / 2; ++pr) {
        const int cl = 32 * pr + 8 * lg;                                  // this lane's first channel
        if (32 * pr >= a.cout) continue;
        const int cinv = cl < a.cout ? 0 : OOB;                           // chunks past cout (a multiple of 16)
        const int cpart = (cl >> 4) * (H * 2 * W * 32) + (lg & 1) * 16;
        const int ppart = (cl >> 4) * (Hp * 2 * Wp * 32) + (lg & 1) * 16;
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(sm + D_SS + cl * 4);
        const f32x4 sc1 = *reinterpret_cast<const f32x4*>(sm + D_SS + (cl + 4) * 4);
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(sm + D_SS + (D_MAXC + cl) * 4);
        const f32x4 sh1 = *reinterpret_cast<const f32x4*>(sm + D_SS + (D_MAXC + cl + 4) * 4);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int y = it.y0 + wr * 4 + 2 * (m / CB) + prow, x = it.x0 + 16 * ph + 8 * (m % CB) + pcol;
          const int inv = ((y < H && x < W) ? 0 : OOB) | cinv;
          const f32x4 v0 = affine(acc[m][2 * pr], sc0, sh0), v1 = affine(acc[m][2 * pr + 1], sc1, sh1);
          split_store(v0, v1, rs0, (y * 2 * W + x) * 32 + cpart, W * 32, inv);
          if constexpr (both) {
            const int yp = y >> 1, xp = x >> 1;
            const int invp = (((lp & 3) == 0 && yp < Hp && xp < Wp) ? 0 : OOB) | cinv;      // lane 4 q stores the quad's maximum
            split_store(quad_max(v0), quad_max(v1), rs1, (yp * 2 * Wp + xp) * 32 + ppart, Wp * 32, invp);
          }
        }
      }
    } else {
      const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
          a.out0 + (size_t)it.b * H * W * a.os0, 0, full ? H * W * a.os0 * 4 : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
          a.out1 + (size_t)it.b * Hp * Wp * a.os1, 0, pooled ? Hp * Wp * a.os1 * 4 : 0, 0x00020000);
      int vo[MT], vp[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int y = it.y0 + wr * 4 + 2 * (m / CB) + prow, x = it.x0 + 16 * ph + 8 * (m % CB) + pcol;
        const int inv = (y < H && x < W) ? 0 : OOB;
        vo[m] = full ? ((y * W + x) * a.os0 + 4 * lg) * 4 | inv : 0;
        const int yp = y >> 1, xp = x >> 1;
        const int invp = ((lp & 3) == 0 && yp < Hp && xp < Wp) ? 0 : OOB;      // lane 4 q stores the quad's maximum
        vp[m] = pooled ? ((yp * Wp + xp) * a.os1 + 4 * lg) * 4 | invp : 0;
      }
#pragma unroll
      for (int n = 0; n < NN; ++n) {
        const int cn = n * 16;                                    // first channel of the N-tile (wave-uniform)
        if (cn >= a.cout) continue;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(sm + D_SS + (cn + 4 * lg) * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(sm + D_SS + (D_MAXC + cn + 4 * lg) * 4);
        const int cinv = cn + 4 * lg < a.cout ? 0 : OOB;
        const int s0 = (a.oo0 + cn) * 4, s1 = (a.oo1 + cn) * 4;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const f32x4 v = affine(acc[m][n], sc, sh);
          if (full) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rs0, KP2D_DBG_ON(64) ? OOB : ((vo[m] + s0) | cinv), 0, 0);
          if (pooled) {
            // 2 x 2 pixel block of a pooled pixel = lanes 4 q .. 4 q + 3: quad maximum in two DPP steps (conv3x3_wsm.hip)
            f32x4 p;
            asm("s_nop 1\n\t"
                "v_max_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_max_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1"
                : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, p), rs1, KP2D_DBG_ON(64) ? OOB : ((vp[m] + s1) | cinv), 0, 0);
          }
        }
      }
    }
  };

  int i = 0, ch = 0, stage = 0;
  S16Item cur = decode(0);
  bool busy = cur.y0 + 4 * wr < H && cur.x0 + 16 * ph < W;      // rows / columns wholly outside the map: nothing to do
  // the two waves of a SIMD (ph 0 / ph 1) finish an item in the same step: the first stores it at the end of that step, the
  // second at the start of the next one (conv3x3_wsm.hip: their store bursts do not share the CU's path to L2)
  bool due = false;
  S16Item prev = cur;
  clear();
  __syncthreads();                                   // barrier 0
  for (int s = 0; s < nsteps; ++s) {
    if (due) { finish(prev); clear(); due = false; }
    if (busy && !KP2D_DBG_ON(8)) multiply(stage * D_IMG, ch * D_WSLAB);
    stage = stage + 1 == D_NIMG ? 0 : stage + 1;
    if (++ch == D_NCH) {
      if (busy && !KP2D_DBG_ON(1)) {
        if (ph == 1) { due = true; prev = cur; }
        else { finish(cur); clear(); }
      }
      ch = 0;
      if (++i < n_my) {
        cur = decode(i);
        busy = cur.y0 + 4 * wr < H && cur.x0 + 16 * ph < W;
      }
    }
    __syncthreads();
  }
  if (due) finish(prev);                             // (the barriers are behind us: no wave waits for this)
}

// ---- launch side -----------------------------------------------------------------------------------------------------
static bool s16_eligible(const ConvArgs& a) {
  if (a.taps != 9 || a.prec != 1 || a.in0.fmt != 1 || a.in1.c != 0 || a.in0.c != a.cin || a.in0.o != 0) return false;
  if (a.store == ST_NCHW) {
    // planar logits behind a 64-channel S16P tensor (confBb, convs.8): one 32-channel group, every channel into out0
    if (a.cin != 64 || a.npad != 32 || a.act != ACT_NONE || a.nsplit != a.cout || a.W < 32 || (a.W & 3) || a.ids_out) return false;
    if (a.in0.bs != (long)a.H * a.W * a.cin || (long)a.H * a.W * a.cin * 4 >= 0x7ffffff0L) return false;
    return true;
  }
  if (a.cin != 32) return false;
  if (a.act > ACT_RELU || a.W < 32 || (a.cout & 15)) return false;
  if (a.store == ST_S16P) { if (a.npad != 32) return false; }
  else if (a.store == ST_NHWC || a.store == ST_NHWC_POOL || a.store == ST_NHWC_BOTH || a.store == ST_S16P_BOTH) { if (a.npad != 64) return false; }
  else return false;
  if ((a.store == ST_NHWC_POOL || a.store == ST_NHWC_BOTH || a.store == ST_S16P_BOTH) && ((a.H | a.W) & 1)) return false;
  if (a.in0.bs != (long)a.H * a.W * a.cin) return false;                      // dense S16P frames
  if ((long)a.H * a.W * a.cin * 4 >= 0x7ffffff0L) return false;
  const long os = (a.store == ST_S16P || a.store == ST_S16P_BOTH) ? a.cout : (a.os0 > a.os1 ? a.os0 : a.os1);
  if ((long)a.H * a.W * os * 4 >= 0x7ffffff0L) return false;
  return true;
}

// workgroups a launch of `nitems` items takes with `lanes` stream lanes side by side (whole rounds, a multiple of 8: contiguous
// runs per XCD), 0: too few items for the form (automatic: at least three rounds, as conv3x3_wsm.hip) — shared by the launcher
// and by the plan, which must know BEFORE it picks the activation layout whether the form will run
static int s16_grid(long nitems, int lanes, int min_items, int grid_opt) {
  const int cus = device_cu_count();
  static const int grid_env = getenv("KP2D_WSM_GRID") ? atoi(getenv("KP2D_WSM_GRID")) : 0;      // (the same knob as conv3x3_wsm.hip)
  int cap = grid_opt > 0 ? grid_opt : (grid_env > 0 ? grid_env : cus / (lanes > 1 ? lanes : 1));
  if (cap > cus) cap = cus;
  cap &= ~7;
  if (cap < 8) return 0;
  const long need = min_items > 0 ? min_items : 2L * cap + 1;
  if (nitems < need || nitems >= (1L << 30)) return 0;
  const long rounds = (nitems + cap - 1) / cap;
  int grid = (int)(((nitems + rounds - 1) / rounds + 7) & ~7L);
  if (grid > cap) grid = cap;
  if (grid > nitems) grid = (int)(nitems & ~7L);
  return grid < 8 ? 0 : grid;
}

bool conv3x3_s16_would_run(int B, int H, int W, int lanes, int min_items, int grid_opt) {
  static const bool off = getenv("KP2D_S16") && getenv("KP2D_S16")[0] == '0';      // (A/B knob)
  if (off || min_items < 0 || W < 32 || ((H | W) & 1)) return false;
  const long nitems = (long)((W + D_TW - 1) / D_TW) * ((H + D_TH - 1) / D_TH) * B;
  return s16_grid(nitems, lanes, min_items, grid_opt) > 0;
}

template <int STORE, int NN, int NCH = 2>
static int s16_launch_one(const ConvArgs& a, int grid, long nitems, hipStream_t s) {
  static PerDeviceOnce lds_once;      // per instantiation and device
  if (int e = lds_opt_in(lds_once, reinterpret_cast<const void*>(&conv3x3_f16x3_s16_kernel<STORE, NN, NCH>))) return e;
  hipLaunchKernelGGL((conv3x3_f16x3_s16_kernel<STORE, NN, NCH>), dim3(grid), dim3(D_THREADS), d_lds(16 * NN, NCH), s, a, (int)nitems);
  return (int)hipGetLastError();
}

// -1000: not a layer of this form; -1006: the plan handed an S16P input to a launch the form cannot take (a plan bug:
// there is no other kernel that reads that layout)
int launch_conv3x3_f16x3_s16(const ConvArgs& a0, hipStream_t s) {
  if (a0.in0.fmt != 1) return -1000;
  if (!s16_eligible(a0)) return -1006;
  ConvArgs a = a0;
  a.tiles_x = (a.W + D_TW - 1) / D_TW;
  a.tiles_y = (a.H + D_TH - 1) / D_TH;
  const long nitems = (long)a.tiles_x * a.tiles_y * a.B;
  const int grid = s16_grid(nitems, a.wsm_lanes, a.s16_min, a.wsm_grid);
  if (grid == 0) return -1006;
  conv3x3_note_variant(a.store == ST_NCHW ? "<s16>planar" : "<s16>");
  switch (a.store) {
    case ST_NCHW: return s16_launch_one<ST_NCHW, 2, 4>(a, grid, nitems, s);
    case ST_S16P: return s16_launch_one<ST_S16P, 2>(a, grid, nitems, s);
    case ST_NHWC: return s16_launch_one<ST_NHWC, 4>(a, grid, nitems, s);
    case ST_NHWC_BOTH: return s16_launch_one<ST_NHWC_BOTH, 4>(a, grid, nitems, s);
    case ST_S16P_BOTH: return s16_launch_one<ST_S16P_BOTH, 4>(a, grid, nitems, s);
    default: return s16_launch_one<ST_NHWC_POOL, 4>(a, grid, nitems, s);
  }
}

// ---- S16P -> planar fp32 (kp2d_set_tap on a layer whose output is kept split): x = float(hi) + float(lo), which is the
// fp32 activation to within its last bit or two (lo is x - hi rounded to 11 bits) ----
__global__ __launch_bounds__(256) void s16p_to_nchw_kernel(const _Float16* __restrict__ in, float* __restrict__ out, int C, int H, int W, int Ct, int c0, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int x = (int)(e % W);
  long r = e / W;
  const int y = (int)(r % H);
  r /= H;
  const int c = (int)(r % C);
  const long b = r / C;
  const int cc = c + c0;
  const _Float16* p = in + (((b * (Ct / 16) + cc / 16) * H + y) * 2 * (long)W + x) * 16 + (cc & 15);
  out[e] = (float)p[0] + (float)p[(long)W * 16];
}
int launch_s16p_to_nchw(const float* in, float* out, int B, int C, int H, int W, int Ct, int c0, hipStream_t s) {
  const long total = (long)B * C * H * W;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(s16p_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const _Float16*>(in), out, C, H, W, Ct, c0, total);
  return (int)hipGetLastError();
}

}  // namespace kp2d
