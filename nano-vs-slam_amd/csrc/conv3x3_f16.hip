// 3x3 / stride 1 / pad 1 convolution in split-fp16 ("f16x3") arithmetic on v_mfma_f32_16x16x32_f16.
//
// Same operator, tiling and epilogue as conv3x3.hip (AnnotatedConvBnReLUModel, modules/base.py:14-46, and the
// biased 3x3 convs of the heads), but the matrix products run on the 16x16x32 MFMA shape (at equal cycles per FLOP
// the chip holds a higher clock on it than on 32x32x16: MI355X_MICROARCH.md "DVFS give-back" (7)) from planar,
// conflict-free LDS images, with 32 accumulator registers per wave.
//
// K = 32 of one MFMA = two taps x 16 channels.  An fp32 operand is x = xh + xl (fp16 halves), a weight w = wh + wl:
//   tap pair (t, t'):  [xl_t | xl_t'] . [wh_t | wh_t']  +  [xh_t | xh_t'] . [wl_t | wl_t']  +  [xh_t | xh_t'] . [wh_t | wh_t']
//   single tap t:      [xh_t | xl_t ] . [wh_t | wh_t ]  +  [xh_t | xl_t ] . [wl_t | wl_t ]      (all four cross terms)
// i.e. 3 MFMAs per two taps + 2 for the ninth: 14 per 16-channel chunk instead of 13.5 (3.7 % padding).
// The packer stores the nine taps of a chunk in slot order {0,1,3,4,2,5,6,7,8} (kp2d_api.cpp pack()): slots (0,1),
// (2,3), (6,7) are taps one pixel apart in x, slots (4,5) = taps (2,5) one row apart, slot 8 is the single.
//
// LDS images are planar and unpadded: input [hi plane | lo plane], each [18 rows][20 px][16 halves = 32 B]; weights
// [wh plane | wl plane], each [slot][n][32 B].  A lane of an operand read is (row p = lane & 15, k-group g = lane >> 4):
// it takes 16 B at (pixel or weight row) * 32 + 16 * (g & 1); the lanes g >= 2 read the second tap (+1 pixel, +1 row or
// the next slot) or, for the single tap, the lo plane.  With a 20-pixel row pitch every ds_read_b128 of either operand is
// bank-conflict free (the 80-byte padded rows of conv3x3.hip were two-way on the pixel operand), and the images are
// 41.5 KB (32-channel tiles) / 59.9 KB (64-channel tiles) per workgroup.
//
// MFMA rows: pixel index p = 4q + r of a 2 x 8 pixel M-tile is (row (r >> 1) & 1, column 2q + (r & 1)), so the four
// accumulator registers of a lane are again one 2x2 pixel block (register-local max-pool).  A wave owns tile rows
// 4w .. 4w+3 (w = wave & 3) as four M-tiles and ONE 32-channel block (two N-tiles): 32 accumulator registers.
// A 64-channel tile is a 512-thread workgroup (NH = 2): waves 4-7 own channels 32-63 of the same 16 x 16 pixels, all
// eight waves stage the one input image (half the iterations each) — the 64-accumulator wave of the 256-thread form
// spilled at the 168-register budget of three workgroups per CU; this one fits 128 (two 512-thread workgroups per CU,
// four waves per SIMD), keeps all nine weight slots resident and so has two barriers per chunk instead of six.
#include "conv_common.h"
#include "device_guard.h"
#include <cstdlib>
#include <type_traits>

namespace kp2d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int F_PXB = 32;
// image row pitch in pixel slots for a tile 16 NP pixels wide: 20 (18 used) / 36 (34 used).  Any pitch = 4 mod 8 keeps
// every ds_read_b128 of the pixel operand conflict-free (row stride = 32 banks mod 64; brute-forced for 20, same residue
// for 36)
__host__ __device__ constexpr int f_pitch(int tw) { return tw == 16 ? 20 : 36; }                    // by tile width (16 or 32 pixels)
__host__ __device__ constexpr int f_lo(int tw, int th) { return (th + 2) * f_pitch(tw) * F_PXB; }   // byte offset of the lo plane
__host__ __device__ constexpr int f_w(int tw, int th) { return 2 * f_lo(tw, th); }                  // weight planes behind the input planes
// tile width: 16 pixels per column block; the 64-channel form with 8-row tiles is 32 wide (below)
__host__ __device__ constexpr int tile_w(int nh, int np, int th) { return (nh == 2 && th == 8) ? 32 : TILE * np; }

__host__ __device__ constexpr int slot_tap(int s) { return s == 2 ? 3 : s == 3 ? 4 : s == 4 ? 2 : s; }
}  // namespace

// NH: 32-channel blocks per workgroup (64-channel tiles: waves 4-7 own channels 32-63 of the same 16 x 16 pixels).
// NP: 16-pixel column blocks per workgroup (NP = 2, NH = 1: waves 4-7 own columns 16-31 of a 16 x 32 pixel tile and the
//     SAME 32 channels: the weight slab — 44 % of a 32-channel tile's staged bytes — is staged once per 512 pixels, and
//     a CU holds 16 waves (two 512-thread workgroups of 59.5 KB) instead of 12 (three 256-thread ones of 41.5 KB)).
// TH: tile height, 16 or 8.  NH = 2, TH = 8: 8 x 32 pixel tiles (a wave owns two rows x 32 columns = the same four M-tiles)
//     for maps whose height leaves the last 16-row tile row at most half full: 120 rows are 15 tile rows of 8 instead of
//     7.5 of 16 (80 -> 75 workgroups per frame at 160 columns, none of them half empty).
//     NH = NP = 1, TH = 8 is the single-frame form: a wave owns two pixel rows (two M-tiles,
//     16 accumulators), so a layer of a 60 x 80 map is 80 workgroups of half the length instead of 40 — a frame's ~25
//     dependent launches are each as long as ONE workgroup's serial chain.
// FLAT32: the 256-thread form on 8 x 32 pixel tiles (a wave owns two rows x 32 columns = four M-tiles, as on 16 x 16) for
//     the planar-output 32-channel layers on maps with a half-empty last 16-row tile row: no ragged row, and every row
//     of a channel plane a workgroup writes is a whole 128-byte line.
// (the body as a device function of (arguments, workgroup index, workgroups, channel group): conv3x3_f16x3_kernel below runs it
// on its own grid, conv3x3_f16x3_multi_kernel runs several independent layers' grids as one launch)
template <int NH, int NP, int TH, bool FLAT32>
__device__ __forceinline__ void conv3x3_f16x3_body(const ConvArgs& a, const int wg_x, const int wg_nx, const int wg_y) {
  static_assert(NH * NP <= 2, "one workgroup is 256 or 512 threads");
  static_assert(TH == 16 || (TH == 8 && NP == 1), "8-row tiles: the 256-thread form (16 wide) and the 64-channel form (32 wide)");
  static_assert(!FLAT32 || (NH == 1 && NP == 1 && TH == 8), "the flat 32-wide form is a 256-thread form");
  constexpr int TW = FLAT32 ? 32 : tile_w(NH, NP, TH);      // tile width in pixels
  constexpr int RW = TH / 4;                        // pixel rows per wave
  constexpr int CB = TW / (8 * NP);                 // 8-pixel column blocks per wave
  constexpr int F_ROWS = TH + 2, MT = (RW / 2) * CB;      // halo rows; M-tiles (2 x 8 pixels) per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const sm = reinterpret_cast<char*>(smem);
  constexpr int NT = NH;                            // 32-channel blocks per workgroup (conv_epilogue.inc)
  constexpr int N = NH * 32, NN = 2;
  constexpr int THREADS = 256 * NH * NP;
  constexpr int WL = 9 * N * 32;                    // byte offset of the wl plane behind the wh plane
  constexpr int KC = 16, Q = 4;
  constexpr int F_PITCH = f_pitch(TW), F_LO = f_lo(TW, TH), F_W = f_w(TW, TH);
  auto tap_off = [](int t) constexpr { return ((t / 3) * F_PITCH + (t % 3)) * F_PXB; };
  // NP = 2: the weight slab of a chunk (18 KB) is copied global -> LDS by buffer_load ... lds (LDS-DMA, no registers, no
  // ds_write) into one of TWO slabs, requested a chunk ahead: the 512-thread form has no registers left for a weight
  // prefetch (it spilled 9-16 at its 128-register budget), and 59.5 + 18 KB still lets two workgroups share a CU.
  constexpr bool WDMA = NP == 2;

  const int tid = threadIdx.x;
  // hwreg(HW_REG_MODE, offset 23, size 1) = FP16_OVFL: fp16 conversions that overflow clamp to +-65504
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
  const int lane = tid & 63;
  const int half = tid >> 8;                        // second half of a 512-thread workgroup: ...
  const int nh = NP == 1 ? half : 0;                // ... the second 32-channel block of the same pixels (NH = 2)
  const int ph = NP == 1 ? 0 : half;                // ... or the second 16-pixel column block of the same channels (NP = 2)
  // pixel rows MT wave .. MT wave + MT - 1 (MT = 4: two row pairs x two 8-pixel column halves).  The second half's waves take the row groups rotated by two, so that the
  // two waves of a workgroup that share a SIMD (w and w + 4: waves go to the SIMDs cyclically) own different rows: in
  // the ragged last tile row of a map (below) the waves that still have work are then spread over all four SIMDs.
  const int wave = __builtin_amdgcn_readfirstlane(((tid >> 6) + 2 * half) & 3);
  // XCD-aware tile order (conv3x3.hip): every XCD gets a contiguous run of tiles
  int bid = wg_x;
  {
    const int nblk = wg_nx, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int b = bid / a.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW;
  const int H = a.H, W = a.W;
  const int n0 = wg_y * N;
  // Ragged last tile row (map height not a multiple of 16: 120 -> 8 valid rows, 60 -> 12, 30 -> 14): a wave whose
  // four rows lie wholly below the map multiplies and stores nothing (it still stages and joins every barrier).
  // H = 120 / 60 otherwise spend 6.25 % of their matrix work on padding rows.
  // (and, in a 32-pixel-wide tile that hangs over the right edge, the column block wholly beyond it)
  const bool busy = y0 + RW * wave < H && x0 + 16 * ph < W;

  f32x4 acc[MT][NN];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // operand read addresses (bytes from the start of LDS); M-tile, tap, N-tile and slot offsets are immediates
  const int lg = lane >> 4, lp = lane & 15;     // k-group and operand row of this lane
  const int a0 = ((wave * RW + ((lp >> 1) & 1)) * F_PITCH + 16 * ph + 2 * (lp >> 2) + (lp & 1)) * F_PXB + 16 * (lg & 1);
  const int a_dx = a0 + (lg >> 1) * F_PXB;               // second tap one pixel to the right
  const int a_dy = a0 + (lg >> 1) * F_PITCH * F_PXB;     // second tap one row down
  const int a_s = a0 + (lg >> 1) * F_LO;                 // single tap: k-groups 2, 3 read the lo plane
  const int b_s = F_W + (nh * 32 + lp) * 32 + 16 * (lg & 1);
  const int b_p = b_s + (lg >> 1) * N * 32;              // second tap = next slot

  const int nchunk = (a.cin + KC - 1) / KC;
  const float* src0 = a.in0.p + (size_t)b * a.in0.bs + a.in0.o;
  const float* src1 = a.in1.p + (size_t)b * a.in1.bs + a.in1.o;
  const int c0 = a.in0.c;

  // staging: as conv3x3.hip (granule = 4 channels of one halo pixel, MUBUF loads whose out-of-range offset returns the
  // zero padding, next chunk prefetched into registers), except that the granules walk the 18 x 20 LDS image itself
  // (columns 18, 19 are never loaded): granule gi = tid + 256 it lands at byte 8 gi of the hi plane, so the LDS side
  // needs no per-iteration register, and the global side keeps ONE pixel index per iteration — the byte offset into
  // either source is pix * pixel-stride (both are dense NHWC views: row stride = W * pixel stride, checked at launch)
  // NP = 2 walks only the 34 used columns of a row (2448 granules = 5 per thread; the 36-slot rows would need 6) and
  // keeps the granule's image slot beside its pixel index: st = slot << 20 | pixel (0xfffff = zero padding; the
  // launcher keeps H * W below 2^20 for this variant)
  constexpr int IN_COLS = TW == 16 ? F_PITCH : TW + 2;
  constexpr int IN_G = F_ROWS * IN_COLS * Q;
  constexpr int IN_IT = (IN_G + THREADS - 1) / THREADS;
  constexpr int PIX_NONE = 0xfffff;
  constexpr int W_G = 9 * N * Q;                    // weight granules of a chunk
  constexpr int W_IT = (W_G + THREADS - 1) / THREADS;
  float4 rin[IN_IT], rw[WDMA ? 1 : W_IT];
  const int st_q4 = 4 * (tid % Q);
  constexpr int OOB = 0x7ffffff0;
  int st_pix[IN_IT];
#pragma unroll
  for (int it = 0; it < IN_IT; ++it) {
    const int gi = tid + THREADS * it;
    const int hp = gi / Q;
    const int py = hp / IN_COLS, px = hp - py * IN_COLS;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool ok = gi < IN_G && px < TW + 2 && gy >= 0 && gy < H && gx >= 0 && gx < W;
    if (TW == 16) st_pix[it] = ok ? gy * W + gx : -1;
    else st_pix[it] = ((py * F_PITCH + px) << 20) | (ok ? gy * W + gx : PIX_NONE);
  }
  const int ps0 = (int)a.in0.ps * 4, ps1 = (int)a.in1.ps * 4;      // pixel strides in bytes
  // weight granule gi = tid + THREADS it of a chunk: 16-byte quad q = gi & 3 of row gi >> 2 (row = slot * N + n);
  // quads 0, 1 are the hi halves, 2, 3 the lo halves of the packed [16 hi | 16 lo] row
  const int w_lds = F_W + (tid >> 2) * 32 + (tid & 1) * 16 + ((tid >> 1) & 1) * WL;

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(src0), 0, (int)((a.in0.bs - a.in0.o) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(src1), 0, (int)((a.in1.bs - a.in1.o) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.w + (size_t)wg_y * nchunk * 9 * N * KC), 0, nchunk * 9 * N * KC * 4, 0x00020000);
  const bool uniform = ((c0 | a.cin) & (KC - 1)) == 0;

  auto prefetch_in = [&](int ch) {
    if (KP2D_DBG_ON(4) || KP2D_DBG_ON(32)) {     // timing ablations (-DKP2D_ABLATE builds only; conv_common.h)
#pragma unroll
      for (int it = 0; it < IN_IT; ++it) rin[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
    if (uniform) {
      // chunks never straddle the two sources and never run past cin (every S config)
      const bool first = ch * KC < c0;
      const int so = (first ? ch * KC : ch * KC - c0) * 4 + st_q4 * 4;
      const int ps = first ? ps0 : ps1;
#pragma unroll
      for (int it = 0; it < IN_IT; ++it) {
        int pix = st_pix[it];
        asm volatile("" : "+v"(pix));      // keep ONE register per granule: the per-source products must not be hoisted
        if (TW != 16) pix = (pix & PIX_NONE) == PIX_NONE ? -1 : (pix & PIX_NONE);
        const int off = pix < 0 ? OOB : pix * ps + so;
        rin[it] = __builtin_bit_cast(float4, first ? __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0)
                                                   : __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0));
      }
    } else {
      // per-thread source / tail selection (channel counts that are not multiples of 16: the N configs)
      const int c = ch * KC + st_q4;
      const bool first = c < c0;
      const int so = (first ? c : c - c0) * 4;
      const bool cok = c < a.cin;
#pragma unroll
      for (int it = 0; it < IN_IT; ++it) {
        int pix = st_pix[it];
        asm volatile("" : "+v"(pix));
        if (TW != 16) pix = (pix & PIX_NONE) == PIX_NONE ? -1 : (pix & PIX_NONE);
        const bool pok = cok && pix >= 0;
        const int o0 = (pok && first) ? pix * ps0 + so : OOB;
        const int o1 = (pok && !first) ? pix * ps1 + so : OOB;
        const i32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs0, o0, 0, 0);
        const i32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs1, o1, 0, 0);
        rin[it] = __builtin_bit_cast(float4, v0 | v1);      // the other one is all zeros
      }
    }
  };
  auto commit_in = [&]() {
#pragma unroll
    for (int it = 0; it < IN_IT; ++it) {
      if (it == IN_IT - 1 && IN_G % THREADS != 0 && tid + THREADS * it >= IN_G) continue;
      const float4 v = rin[it];
      f16x2 h0, h1, l0, l1;
      split2(v.x, v.y, h0, l0);
      split2(v.z, v.w, h1, l1);
      // image byte of the granule: linear in the granule index (NP = 1), or slot * 32 + 8 * (granule & 3)
      const int lb = TW == 16 ? tid * 8 + it * THREADS * 8 : (int)((unsigned)st_pix[it] >> 20) * F_PXB + (tid & 3) * 8;
      *reinterpret_cast<f16x4*>(sm + lb) = f16x4{h0[0], h0[1], h1[0], h1[1]};
      *reinterpret_cast<f16x4*>(sm + F_LO + lb) = f16x4{l0[0], l0[1], l1[0], l1[1]};
    }
  };

  auto prefetch_w = [&](int ch) {
    if constexpr (WDMA) {
      // piece p = wave + 8 j (18 pieces of 1 KiB per chunk): LDS bytes [1024 p, 1024 p + 1024) of slab ch & 1 =
      // [wh plane | wl plane], each [slot][n][32 B]; a lane's 16 bytes come from the packed [16 hi | 16 lo] row
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int pc = wv + 8 * j;
        if (pc < 2 * WL / 1024) {
          const int o = 1024 * pc + 16 * lane;
          const int plane = o >= WL ? 1 : 0, o2 = o - plane * WL;
          const int voff = (o2 >> 5) * 64 + plane * 32 + ((o2 >> 4) & 1) * 16;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(sm + F_W + (ch & 1) * 2 * WL + 1024 * pc),
                                                   16, voff, ch * W_G * 16, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < W_IT; ++it)
        rw[it] = (KP2D_DBG_ON(4) || KP2D_DBG_ON(16)) ? make_float4(0.f, 0.f, 0.f, 0.f) : __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsw, tid * 16, (ch * W_G + THREADS * it) * 16, 0));
    }
  };
  prefetch_in(0);
  prefetch_w(0);

  for (int ch = 0; ch < nchunk; ++ch) {
    __syncthreads();          // every wave is done reading the previous chunk's LDS images
    if (!KP2D_DBG_ON(2)) commit_in();
    if constexpr (!WDMA) {
#pragma unroll
      for (int it = 0; it < W_IT; ++it) {
        if (KP2D_DBG_ON(2)) break;
        if (it == W_IT - 1 && W_G % THREADS != 0 && tid + THREADS * it >= W_G) continue;
        *reinterpret_cast<float4*>(sm + w_lds + it * (THREADS / 4) * 32) = rw[it];
      }
    } else {
      __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): this wave's pieces of the chunk's weight slab have landed
    }
    __syncthreads();
    // (WDMA: slab (ch + 1) & 1 was last read in the matrix phase of chunk ch - 1, which every wave left before the
    // first barrier of this chunk)
    if (ch + 1 < nchunk) { prefetch_in(ch + 1); prefetch_w(ch + 1); }
    const int wbo = WDMA ? (ch & 1) * 2 * WL : 0;      // weight slab of this chunk

    if (!KP2D_DBG_ON(8) && busy)
#pragma unroll
    for (int slot = 0; slot < 9; slot += 2) {
      // (timing emulation of Winograd F(2,3): bit 128 drops 5 of a chunk's 14 MFMA groups — results are wrong)
      if (KP2D_DBG_ON(128) && slot >= 6) break;
      const int t = slot_tap(slot);
      const bool single = slot == 8;
      const bool dy = slot == 4;
      const int ab = (single ? a_s : (dy ? a_dy : a_dx)) + tap_off(t);
      const int bb = (single ? b_s : b_p) + slot * N * 32 + wbo;
      f16x8 bh[NN], bl[NN];
#pragma unroll
      for (int n = 0; n < NN; ++n) {
        bh[n] = *reinterpret_cast<const f16x8*>(sm + bb + n * 512);
        bl[n] = *reinterpret_cast<const f16x8*>(sm + bb + n * 512 + WL);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int mo = (2 * (m / CB) * F_PITCH + 8 * (m % CB)) * F_PXB;
        if (single) {
          // k-groups 0, 1 carry xh, groups 2, 3 xl of the same tap: both weight halves see both operand halves
          const f16x8 x = *reinterpret_cast<const f16x8*>(sm + ab + mo);
#pragma unroll
          for (int n = 0; n < NN; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, bl[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, bh[n], acc[m][n], 0, 0, 0);
          }
        } else {
          const f16x8 zh = *reinterpret_cast<const f16x8*>(sm + ab + mo);
          const f16x8 zl = *reinterpret_cast<const f16x8*>(sm + ab + mo + F_LO);
#pragma unroll
          for (int n = 0; n < NN; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zl, bh[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zh, bl[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zh, bh[n], acc[m][n], 0, 0, 0);
          }
        }
      }
    }
  }

  constexpr int EPI_ROUNDS = 1;
  constexpr bool EPI_GELU = false;
  // accumulator layout of the 16x16 MFMA tiles (conv_epilogue.inc): lane (lp, lg), M-tile m, register r
#define EPI_NM MT
#define EPI_NN NN
#define EPI_R 4
#define EPI_ACC(m, n, r) acc[m][n][r]
#define EPI_CH(n) (nh * 32 + (n) * 16 + lp)
#define EPI_ROW(m, r) (wave * RW + 2 * ((m) / CB) + (((r) >> 1) & 1))
#define EPI_COL(m, r) (16 * ph + 8 * ((m) % CB) + 2 * lg + ((r) & 1))
#define EPI_TW TW
#define EPI_TH TH
#define EPI_MVALID(m) busy
#define EPI_THREADS THREADS
#include "conv_epilogue.inc"
#undef EPI_THREADS
#undef EPI_NM
#undef EPI_NN
#undef EPI_R
#undef EPI_ACC
#undef EPI_CH
#undef EPI_ROW
#undef EPI_COL
#undef EPI_MVALID
#undef EPI_TW
#undef EPI_TH
}

template <int NH, int NP, int TH, bool FLAT32 = false>
__global__ __launch_bounds__(256 * NH * NP, NH * NP == 1 ? 3 : 4) void conv3x3_f16x3_kernel(const ConvArgs a) {
  conv3x3_f16x3_body<NH, NP, TH, FLAT32>(a, blockIdx.x, gridDim.x, blockIdx.y);
}

// Up to four INDEPENDENT layers as one launch (blockIdx.z = layer; a workgroup beyond its layer's grid leaves at once).  A single
// frame's forward is a chain of ~25 dependent launches, each as long as ONE workgroup's serial chain (8-10 us whatever it
// computes): layers of different heads that wait for the same predecessor cost one such latency together instead of one each
// (kp2d_api.cpp groups them on small grids).  Same code per layer as its own launch: bit-identical.
struct ConvMultiArgs { ConvArgs a[4]; int n; };
__global__ __launch_bounds__(256, 3) void conv3x3_f16x3_multi_kernel(const ConvMultiArgs m) {
  const ConvArgs& a = m.a[blockIdx.z];
  const int nx = a.tiles_x * a.tiles_y * a.B;
  if ((int)blockIdx.x >= nx || (int)blockIdx.y >= a.npad / 32) return;
  conv3x3_f16x3_body<1, 1, 8, false>(a, blockIdx.x, nx, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------------------------
// Single-chunk layers (conv1b: 16 -> 32 channels at full resolution, max-pooled output): warp-specialised and persistent.
// With one chunk a workgroup of the kernel above is load -> commit -> multiply -> store with nothing of its own to overlap,
// and the layer ran at (memory time + matrix time): 0.171 ms for 0.086 + 0.090.  (A persistent loop in which every wave
// does both jobs lost, r3_ab_persistent.txt: the compiler merges a wave's waits over the loop's predecessors and the first
// loads of a tile wait for the previous tile's stores.)  Here the jobs belong to DIFFERENT waves, each with its own
// counters: waves 8-11 only stage (global -> registers -> split -> LDS image of tile i + 1, the loads of tile i + 2 already
// in flight), waves 0-7 only multiply tile i out of the other image and store its pooled result; one barrier per tile; the
// nine weight slots stay in LDS for the whole launch.  One 768-thread workgroup per CU walks tiles g, g + G, ...
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int WS_TH = 16, WS_TW = 32, WS_PITCH = 36, WS_ROWS = WS_TH + 2, WS_COLS = WS_TW + 2;
constexpr int WS_LO = WS_ROWS * WS_PITCH * F_PXB;      // byte offset of an image's lo plane (20,736)
constexpr int WS_IMG = 2 * WS_LO;                      // one input image (41,472 B)
constexpr int WS_N = 32, WS_WL = 9 * WS_N * 32;        // channels; byte offset of the wl plane behind the wh plane
constexpr int WS_W = 2 * WS_IMG;                       // weight planes behind the two images
constexpr int WS_LDS = WS_W + 2 * WS_WL;               // 101,376 B
constexpr int WS_BLK = 2048;                           // S16P output: a wave's 16 pooled pixels x 32 channels x (hi, lo) on their way out
constexpr int WS_G = WS_ROWS * WS_COLS * 4;            // 16-byte granules of a halo tile (2448)
constexpr int WS_IT = (WS_G + 255) / 256;              // per staging thread (10)
// trips of BOTH role loops of a workgroup that walks tiles t0, t0 + G, ... two per trip (the barrier contract below)
constexpr int WS_BARRIERS_PER_TRIP = 2;
__host__ __device__ constexpr int ws_trips(int ntiles, int t0, int G) { return (ntiles - t0 + 2 * G - 1) / (2 * G); }
}  // namespace

// S16OUT: the pooled output leaves as an S16P tensor (kp2d_kernels.h; its consumer is conv3x3_s16.hip).  Same products, same
// pooling; the pooled value of a lane (one channel of one pooled pixel per accumulator tile) is split there and the halves
// go through a 2-KB wave-private LDS block laid out in the OUTPUT's order, so a wave's 16 pooled pixels x 32 channels leave
// as TWO 16-byte stores per lane — four contiguous 256-byte row segments each — instead of eight 4-byte stores.
// (First form of this epilogue: transposed products, a DPP quad maximum and the two N-tiles' weight rows interleaved so that a
// lane held 8 channels of a pixel — 240 vector instructions per tile and stores from a quarter of the lanes: 0.150 ms against
// 0.134 for the fp32 output, the layer became bound by vector issue.)
// STEM: the staging waves do not read backbone.conv1a's output, they COMPUTE it — conv1a (3 -> 16 channels, 27 taps, BatchNorm,
// LeakyReLU: modules/encoders.py:20-29, 108) for the tile's 18 x 34 halo pixels straight from the RGB planes, on the matrix
// cores, split and written into conv1b's operand image.  conv1a's launch and the 315 MB (64 frames) it writes and conv1b reads
// back — the largest tensor of the forward after `skip` — disappear.  Two VALU forms of this fusion lost in round 3
// (profiles/r3_ab_warp_specialised.txt: 529 kFLOP of fp32 FMAs per tile are more cycles of the CU's whole vector ALU than
// conv1b's matrix work).  Here a staging wave owns 153 consecutive halo pixels (a quarter of the 18 x 34, five halo rows): it keeps
// their RGB window (7 rows x 40 columns) as fp16 hi / lo planes in a wave-private LDS block laid out [row][x][c0 c1 c2 0] — 8 bytes
// per pixel, so a lane's B operand of v_mfma_f32_16x16x32_f16 (K = 8 taps x 4 channels: two taps per k-group) is two aligned
// ds_read_b64 — and runs 10 M-tiles of 16 halo pixels x 16 channels, six MFMAs each (8 taps + the ninth, times the three split
// terms), straight-line: no branch between the M-tiles, two accumulators per M-tile, so that one M-tile's epilogue lies under
// the next one's products (first form: a branch per M-tile and one accumulation chain — the staging wave's serial chain was
// 11k cycles per tile and the fused layer took 0.218 ms against 0.097 + 0.137 for the two launches).
// No barrier beyond the tile's own: the window is wave-private and a wave's LDS instructions execute in order.
// Arithmetic: split-fp16 products like every other layer of this precision mode (x = xh + xl, w 2^e = wh + wl, fp32
// accumulate: ~1e-7 relative to the exact fp32 FMA chain of round 4's conv1a_kernel).  conv1a's own launch in this mode
// (conv1a_mfma_kernel below: small grids, sub-batches, uint8 frames) issues the SAME products in the same order, so fused and
// unfused first layers agree bit for bit and a forward's results do not depend on which one a grid size picks.
struct StemArgs {
  const float* x;                     // [B,3,H,W] frames
  const float* w;                     // conv1a weights [27][16] (k = ci 9 + dy 3 + dx)
  const float* scale; const float* shift;   // BatchNorm folded, [16]
  const float* wsc;                   // device pointer to 2^e: the weights are split as w 2^e (lo halves stay normal); part of the weight blob
  int act;
};
namespace {
constexpr int ST_XR = 7, ST_XC = 40;                   // window rows (a 5-row band + 2) and columns (x0 - 4 .. x0 + 35)
constexpr int ST_XPL = ST_XR * ST_XC * 8;              // one plane of a wave's window (2,240 B)
constexpr int ST_NPX = WS_ROWS * WS_COLS / 4;          // halo pixels per staging wave: 612 / 4 = 153 consecutive ones (five halo rows)
constexpr int ST_MT = (ST_NPX + 15) / 16;              // M-tiles of 16 halo pixels per wave (10)
static_assert(ST_NPX * 4 == WS_ROWS * WS_COLS, "the halo splits evenly over the four staging waves");
}  // namespace

template <bool S16OUT, bool STEM>
__global__ __launch_bounds__(768, 3) void conv3x3_f16x3_ws_kernel(const ConvArgs a, const StemArgs st, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const sm = reinterpret_cast<char*>(smem);
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);       // FP16_OVFL: conversions that overflow clamp to +-65504
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wave8 < 8;                   // waves 0-7 multiply (two per SIMD), waves 8-11 stage
  const int H = a.H, W = a.W, per_frame = a.tiles_x * a.tiles_y;
  constexpr int OOB = 0x7ffffff0;

  // weights: the chunk's [slot][n][16 hi | 16 lo] rows -> [wh plane | wl plane] (as the kernel above), once
  for (int gi = tid; gi < 9 * WS_N * 4; gi += 768)
    *reinterpret_cast<float4*>(sm + WS_W + (gi >> 2) * 32 + (gi & 1) * 16 + ((gi >> 1) & 1) * WS_WL) =
        reinterpret_cast<const float4*>(a.w)[gi];

  // ---- producer state: granule gi = ptid + 256 it of a halo tile = (halo pixel gi / 4, channels 4 (gi % 4) ..) ----
  const int ptid = tid - 512;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in0.p + a.in0.o), 0, (int)(((long)a.B * a.in0.bs - a.in0.o) * 4), 0x00020000);
  const int ps = (int)a.in0.ps * 4;                   // pixel stride, bytes
  // TWO register sets: a CU has one workgroup, so the bytes it keeps in flight are what its staging waves hold — with one
  // set (41 KB per CU) the layer could not pass ~2.8 TB/s whatever else overlapped
  float4 rin[2][WS_IT];
  auto request = [&](int t, auto set_c) {             // issue the loads of tile t (past the last tile: zeros, same count)
    constexpr int RS = decltype(set_c)::value;
    const bool live = t < ntiles;
    const int b = t / per_frame, r = t - b * per_frame;
    const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
    const int y0 = ty * WS_TH - 1, x0 = tx * WS_TW - 1;
    const int fb = b * (int)a.in0.bs * 4;
#pragma unroll
    for (int it = 0; it < WS_IT; ++it) {
      const int gi = ptid + 256 * it, hp = gi >> 2;
      const int py = hp / WS_COLS, px = hp - py * WS_COLS;
      const int gy = y0 + py, gx = x0 + px;
      const bool ok = live && gi < WS_G && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const int off = ok ? fb + (gy * W + gx) * ps + (gi & 3) * 16 : OOB;
      rin[RS][it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
  };
  auto commit = [&](int buf, auto set_c) {            // registers -> split -> image buf
    constexpr int RS = decltype(set_c)::value;
#pragma unroll
    for (int it = 0; it < WS_IT; ++it) {
      const int gi = ptid + 256 * it, hp = gi >> 2;
      if (gi >= WS_G) continue;
      const int py = hp / WS_COLS, px = hp - py * WS_COLS;
      const int lb = buf * WS_IMG + (py * WS_PITCH + px) * F_PXB + (gi & 3) * 8;
      const float4 v = rin[RS][it];
      f16x2 h0, h1, l0, l1;
      split2(v.x, v.y, h0, l0);
      split2(v.z, v.w, h1, l1);
      *reinterpret_cast<f16x4*>(sm + lb) = f16x4{h0[0], h0[1], h1[0], h1[1]};
      *reinterpret_cast<f16x4*>(sm + WS_LO + lb) = f16x4{l0[0], l0[1], l1[0], l1[1]};
    }
  };

  // ---- STEM producer state (see the kernel comment) ----
  const int pw = wave8 - 8;                                                        // staging wave 0..3
  const int px0 = ST_NPX * (pw & 3);                                               // halo pixels [px0, px0 + 153) of the 18 x 34
  const int hr0 = px0 / WS_COLS;                                                   // their first halo row (window row 0 = hr0 - 1)
  char* const xw = sm + WS_LDS + (S16OUT ? 8 * WS_BLK : 0) + (pw & 3) * 2 * ST_XPL;   // this wave's window: hi plane | lo plane
  // window loads of ONE tile in flight (requested while the tile before it is computed): slot e = (window row, column quad);
  // a lane brings all three planes of its slot(s) — slots 0-63 and, lanes 0-5, slots 64-69 — so that a pixel's (c0, c1, c2, 0)
  // halves leave as one ds_write_b64 per plane
  float4 rx[STEM ? 6 : 1];
  int tset = 0;                                                                    // tile the registers carry
  f16x8 swh[2], swl[2];                                                            // A operands: taps (2 lg, 2 lg + 1) and tap 8
  float ssc[4], ssh[4];
  int sbase[STEM ? ST_MT : 1], simg[STEM ? ST_MT : 1], sprc[STEM ? ST_MT : 1];
  int so0 = 0, so1 = 0;
  const int slg = lane >> 4, slp = lane & 15;
  if constexpr (STEM) {
    if (!consumer) {
      for (int e = lane; e < 2 * ST_XPL / 16; e += 64) *reinterpret_cast<float4*>(xw + 16 * e) = make_float4(0.f, 0.f, 0.f, 0.f);
      // weights of output channel slp: k-group slg holds taps 2 slg, 2 slg + 1 (MFMA 0) / tap 8 in k-group 0 (MFMA 1), 4 halves per
      // tap = channels (0, 1, 2, padding)
      const float wscale = st.wsc[0];                 // a power of two: 1 / wscale is exact
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) {
          const int tap = j == 0 ? 2 * slg + (hh >> 2) : 8, ci = hh & 3;
          const bool on = ci < 3 && (j == 0 || (slg == 0 && hh < 4));
          const float wv = on ? st.w[(ci * 9 + tap) * 16 + slp] * wscale : 0.f;
          const _Float16 h = (_Float16)wv;
          swh[j][hh] = h;
          swl[j][hh] = (_Float16)(wv - (float)h);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) { ssc[i] = st.scale[4 * slg + i] * (1.f / wscale); ssh[i] = st.shift[4 * slg + i]; }
      auto toff = [](int t) { return ((t / 3) * ST_XC + (t % 3)) * 8; };
      so0 = toff(2 * slg);
      so1 = toff(2 * slg + 1);
#pragma unroll
      for (int mt = 0; mt < ST_MT; ++mt) {
        const int q = 16 * mt + slp;
        const int hp = px0 + (q < ST_NPX ? q : ST_NPX - 1);      // (lanes past the wave's share read its last pixel and write nothing)
        const int hr = hp / WS_COLS, hc = hp - hr * WS_COLS;
        sbase[mt] = ((hr - hr0) * ST_XC + hc + 2) * 8;           // window pixel of tap (0, 0): halo column hc = map column x0 - 1 + hc
        // (lanes past the share write into pitch column 34 of their row — never read — so that the stores need no branch)
        simg[mt] = (hr * WS_PITCH + (q < ST_NPX ? hc : WS_COLS)) * F_PXB + slg * 8;
        sprc[mt] = (hr << 8) | hc;
      }
    }
  }
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(STEM ? st.x : a.in0.p), 0, STEM ? (int)((long)a.B * 3 * H * W * 4) : 0, 0x00020000);
  // loads of tile t's window: slot e = lane + 64 u (u = 0, 1; 70 slots) -> window row e / 10, columns 4 (e % 10) .. + 3, three planes
  auto stem_request = [&](int t) {
    tset = t;
    const bool live = t < ntiles;
    const int b = t / per_frame, r = t - b * per_frame;
    const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
    const int gy0 = ty * WS_TH - 2 + hr0, gx0 = tx * WS_TW - 4;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = lane + 64 * u, wr = e / 10, q = e - 10 * wr;
      const int gy = gy0 + wr, gx = gx0 + 4 * q;
      const bool ok = live && wr < ST_XR && gy >= 0 && gy < H && gx >= 0 && gx < W;      // (W is a multiple of 8: whole float4s)
      const int off = (ok && !KP2D_DBG_ON(32)) ? ((b * 3 * H + gy) * W + gx) * 4 : OOB;      // (timing ablations: conv_common.h)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        rx[STEM ? 3 * u + c : 0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsx, off, c * H * W * 4, 0));
    }
  };
  auto stem_commit = [&](int buf) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    // window: fp32 -> hi / lo halves at [row][x][c0 c1 c2 0]
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = lane + 64 * u, wr = e / 10, q = e - 10 * wr;
      if (wr >= ST_XR || KP2D_DBG_ON(512)) continue;
      const float4 v0 = rx[STEM ? 3 * u : 0], v1 = rx[STEM ? 3 * u + 1 : 0], v2 = rx[STEM ? 3 * u + 2 : 0];
      const float c0[4] = {v0.x, v0.y, v0.z, v0.w}, c1[4] = {v1.x, v1.y, v1.z, v1.w}, c2[4] = {v2.x, v2.y, v2.z, v2.w};
      char* const d = xw + (wr * ST_XC + 4 * q) * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f16x2 ha, la, hb, lb;
        split2(c0[i], c1[i], ha, la);
        split2(c2[i], 0.f, hb, lb);
        *reinterpret_cast<h4*>(d + 8 * i) = h4{ha[0], ha[1], hb[0], hb[1]};
        *reinterpret_cast<h4*>(d + 8 * i + ST_XPL) = h4{la[0], la[1], lb[0], lb[1]};
      }
    }
    asm volatile("" ::: "memory");                     // (halves written as fp16 lvalues, read back as 8-byte vectors below)
    const int t = tset;
    const int b = t / per_frame, r = t - b * per_frame;
    const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
    const int y0 = ty * WS_TH - 1, x0 = tx * WS_TW - 1;      // map position of halo pixel (0, 0)
    (void)b;
    const float slope1 = st.act == ACT_LEAKY ? 0.01f : (st.act == ACT_RELU ? 0.f : 1.f);
    // a tile whose whole halo lies inside the map needs no zero-padding test (wave-uniform: most tiles of a big map)
    const bool interior = y0 >= 0 && y0 + WS_ROWS <= H && x0 >= 0 && x0 + WS_COLS <= W;
    char* const img = sm + buf * WS_IMG;
    // M-tiles in PAIRS: the two accumulation chains are interleaved (a dependent MFMA waits for its predecessor's result; the
    // other M-tile's MFMA fills that slot), the operands of the next pair are read before this pair's products, and the two
    // epilogues' vector instructions follow the twelve MFMAs as one block
    h4 na0[2], na1[2], nb0[2], nb1[2], na8[2], nb8[2];
    auto fetch = [&](int mt, int u) {
      const char* const ph = xw + sbase[mt];
      na0[u] = *reinterpret_cast<const h4*>(ph + so0); na1[u] = *reinterpret_cast<const h4*>(ph + so1);
      nb0[u] = *reinterpret_cast<const h4*>(ph + so0 + ST_XPL); nb1[u] = *reinterpret_cast<const h4*>(ph + so1 + ST_XPL);
      na8[u] = *reinterpret_cast<const h4*>(ph + (2 * ST_XC + 2) * 8); nb8[u] = *reinterpret_cast<const h4*>(ph + (2 * ST_XC + 2) * 8 + ST_XPL);
    };
    static_assert(ST_MT % 2 == 0, "M-tiles are processed in pairs");
    fetch(0, 0);
    fetch(1, 1);
#pragma unroll
    for (int mt = 0; mt < ST_MT; mt += 2) {
      if (KP2D_DBG_ON(2048)) break;                    // (ablation: no conv1a products, epilogue or image writes at all)
      f16x8 xh[2], xl[2], yh[2], yl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        xh[u] = f16x8{na0[u][0], na0[u][1], na0[u][2], na0[u][3], na1[u][0], na1[u][1], na1[u][2], na1[u][3]};
        xl[u] = f16x8{nb0[u][0], nb0[u][1], nb0[u][2], nb0[u][3], nb1[u][0], nb1[u][1], nb1[u][2], nb1[u][3]};
        yh[u] = f16x8{na8[u][0], na8[u][1], na8[u][2], na8[u][3], 0, 0, 0, 0};
        yl[u] = f16x8{nb8[u][0], nb8[u][1], nb8[u][2], nb8[u][3], 0, 0, 0, 0};
      }
      if (mt + 2 < ST_MT) { fetch(mt + 2, 0); fetch(mt + 3, 1); }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      // (small terms first)
      if (!KP2D_DBG_ON(128)) {
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(swh[1], yl[u], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(swl[1], yh[u], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(swh[0], xl[u], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(swl[0], xh[u], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(swh[1], yh[u], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(swh[0], xh[u], acc[u], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (KP2D_DBG_ON(1024)) continue;               // (ablation: no epilogue / image writes)
        // lane (slp, slg): halo pixel px0 + 16 (mt + u) + slp, channels 4 slg .. 4 slg + 3
        const int y = y0 + (sprc[mt + u] >> 8), x = x0 + (sprc[mt + u] & 255);
        const bool in = interior || (y >= 0 && y < H && x >= 0 && x < W);      // outside the map: conv1b's zero padding
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float tt = fmaf(acc[u][i], ssc[i], ssh[i]);
          v[i] = in ? fmaxf(tt, tt * slope1) : 0.f;
        }
        f16x2 h0, l0, h1, l1;
        split2(v[0], v[1], h0, l0);
        split2(v[2], v[3], h1, l1);
        *reinterpret_cast<f16x4*>(img + simg[mt + u]) = f16x4{h0[0], h0[1], h1[0], h1[1]};
        *reinterpret_cast<f16x4*>(img + WS_LO + simg[mt + u]) = f16x4{l0[0], l0[1], l1[0], l1[1]};
      }
    }
    asm volatile("" ::: "memory");
  };

  // ---- consumer state: wave (w, ph) owns tile rows 4 w .. 4 w + 3 x columns 16 ph .. + 15 = four M-tiles (2 x 8 pixels),
  //      two N-tiles (one lone multiplying wave per SIMD cannot hide its own LDS latency: 0.191 ms against 0.178 for the
  //      kernel above; two per SIMD as there) ----
  constexpr int MT = 4, NN = 2, CB = 2;
  const int lg = lane >> 4, lp = lane & 15;
  const int wr = wave8 & 3, ph = (wave8 >> 2) & 1;
  const int a0 = ((wr * 4 + ((lp >> 1) & 1)) * WS_PITCH + 16 * ph + 2 * (lp >> 2) + (lp & 1)) * F_PXB + 16 * (lg & 1);
  const int a_dx = a0 + (lg >> 1) * F_PXB;
  const int a_dy = a0 + (lg >> 1) * WS_PITCH * F_PXB;
  const int a_s = a0 + (lg >> 1) * WS_LO;
  const int b_s = WS_W + lp * 32 + 16 * (lg & 1);
  const int b_p = b_s + (lg >> 1) * WS_N * 32;
  auto tap_off = [](int t) constexpr { return ((t / 3) * WS_PITCH + (t % 3)) * F_PXB; };
  const float slope = a.act == ACT_LEAKY ? 0.01f : (a.act == ACT_RELU ? 0.f : 1.f);
  const int Hp = H >> 1, Wp = W >> 1;
  float sc[NN], sh[NN];
#pragma unroll
  for (int n = 0; n < NN; ++n) { sc[n] = a.scale[n * 16 + lp]; sh[n] = a.shift[n * 16 + lp]; }

  auto multiply = [&](int t, int buf) {
    const int b = t / per_frame, r = t - b * per_frame;
    const int ty = r / a.tiles_x, tx = r - ty * a.tiles_x;
    const int y0 = ty * WS_TH, x0 = tx * WS_TW;
    if (y0 + 4 * wr >= H || x0 + 16 * ph >= W) return;      // rows / columns wholly outside the map (ragged tiles)
    f32x4 acc[MT][NN];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ib = buf * WS_IMG;
    if (!KP2D_DBG_ON(8))                               // (timing ablations: conv_common.h)
#pragma unroll
    for (int slot = 0; slot < 9; slot += 2) {
      const int tp = slot_tap(slot);
      const bool single = slot == 8;
      const bool dy = slot == 4;
      const int ab = ib + (single ? a_s : (dy ? a_dy : a_dx)) + tap_off(tp);
      const int bb = (single ? b_s : b_p) + slot * WS_N * 32;
      f16x8 bh[NN], bl[NN];
#pragma unroll
      for (int n = 0; n < NN; ++n) {
        bh[n] = *reinterpret_cast<const f16x8*>(sm + bb + n * 512);
        bl[n] = *reinterpret_cast<const f16x8*>(sm + bb + n * 512 + WS_WL);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int mo = (2 * (m / CB) * WS_PITCH + 8 * (m % CB)) * F_PXB;
        if (single) {
          const f16x8 x = *reinterpret_cast<const f16x8*>(sm + ab + mo);
#pragma unroll
          for (int n = 0; n < NN; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, bl[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, bh[n], acc[m][n], 0, 0, 0);
          }
        } else {
          const f16x8 zh = *reinterpret_cast<const f16x8*>(sm + ab + mo);
          const f16x8 zl = *reinterpret_cast<const f16x8*>(sm + ab + mo + WS_LO);
#pragma unroll
          for (int n = 0; n < NN; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zl, bh[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zh, bl[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zh, bh[n], acc[m][n], 0, 0, 0);
          }
        }
      }
    }
    if (KP2D_DBG_ON(1)) return;
    if constexpr (S16OUT) {
      // lane (lp, lg), tile (m, n): channel 16 n + lp of the pooled pixel (row m / 2, column 4 (m % 2) + lg) of this wave's 2 x 8
      // pooled pixels.  LeakyReLU / ReLU is monotonic, so it is applied once, to the maximum (the same bits as the maximum of
      // the four activations).  Block layout = the order the bytes leave in: piece e = ((n 2 + plane) 2 + row) 16 + 2 column
      // + (lp >> 3) of 16 bytes, halves lp & 7 inside it.
      char* const blk = sm + WS_LDS + wave8 * WS_BLK;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float v[NN];
#pragma unroll
        for (int n = 0; n < NN; ++n) {
          const float t0 = fmaf(acc[m][n][0], sc[n], sh[n]), t1 = fmaf(acc[m][n][1], sc[n], sh[n]);
          const float t2 = fmaf(acc[m][n][2], sc[n], sh[n]), t3 = fmaf(acc[m][n][3], sc[n], sh[n]);
          const float tm = fmaxf(fmaxf(t0, t1), fmaxf(t2, t3));
          v[n] = fmaxf(tm, tm * slope);
        }
        f16x2 hi, lo;
        split2(v[0], v[1], hi, lo);                    // (element n of hi / lo = the halves of N-tile n's value)
        const int e0 = ((m / CB) * 16 + (4 * (m % CB) + lg) * 2 + (lp >> 3)) * 16 + (lp & 7) * 2;
#pragma unroll
        for (int n = 0; n < NN; ++n) {
          *reinterpret_cast<_Float16*>(blk + e0 + (n * 2 + 0) * 512) = hi[n];
          *reinterpret_cast<_Float16*>(blk + e0 + (n * 2 + 1) * 512) = lo[n];
        }
      }
      // (the halves were written as fp16 lvalues and are read back as 16-byte vectors: a compiler fence keeps the reads behind the
      // writes — the LDS itself executes a wave's instructions in order — and a second one keeps the next tile's writes behind these reads)
      asm volatile("" ::: "memory");
      // lane L, chunk n: piece e = L + 64 n -> half L & 1, column (L >> 1) & 7, row (L >> 4) & 1, plane L >> 5
      const int obs = Hp * Wp * a.cout;                // output frame stride, floats (S16P: the same bytes as fp32 NHWC)
      const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(a.out1 + (size_t)b * obs, 0, obs * 4, 0x00020000);
      const int yp = ((y0 + wr * 4) >> 1) + ((lane >> 4) & 1), xp = ((x0 + 16 * ph) >> 1) + ((lane >> 1) & 7);
      const int inv = (yp < Hp && xp < Wp) ? 0 : OOB;
      const int o = ((yp * 2 + (lane >> 5)) * Wp + xp) * 32 + (lane & 1) * 16;
#pragma unroll
      for (int n = 0; n < NN; ++n) {
        const i32x4 pc = *reinterpret_cast<const i32x4*>(blk + (lane + 64 * n) * 16);
        // (soffset = 0: the gfx950 store hazard, conv3x3_wsm.hip)
        __builtin_amdgcn_raw_buffer_store_b128(pc, rso, (o + n * (Hp * 2 * Wp * 32)) | inv | (16 * n < a.cout ? 0 : OOB), 0, 0);
      }
      asm volatile("" ::: "memory");
      return;
    }
    // pooled epilogue (conv_epilogue.inc, ST_NHWC_POOL): the four registers of a lane are one 2 x 2 pixel block
    unsigned* o1 = reinterpret_cast<unsigned*>(a.out1) + (size_t)b * Hp * Wp * a.os1;
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      const int co = n * 16 + lp;
      const bool cok = co < a.cout;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float tt = fmaf(acc[m][n][r], sc[n], sh[n]);
          v[r] = fmaxf(tt, tt * slope);
        }
        const int yp = (y0 + wr * 4 + 2 * (m / CB)) >> 1, xp = (x0 + 16 * ph + 8 * (m % CB) + 2 * lg) >> 1;
        const float vmax = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        if (cok && yp < Hp && xp < Wp) o1[(yp * Wp + xp) * a.os1 + a.oo1 + co] = __float_as_uint(vmax);
      }
    }
  };

  // Tile i of this workgroup is t0 + i G.  Staging waves: set 0 carries the odd tiles (and tile 0), set 1 the even ones;
  // in iteration i they commit tile i + 1 into image (i + 1) & 1 and request tile i + 3 into the set just freed.  Every
  // request / commit runs unconditionally (tiles past the end load and commit zeros nobody reads), so that each path has
  // the same number of loads in flight and the waits in front of a commit stay partial.
  // The two jobs are two separate loops: in a shared loop the staging registers would be live through the multiply code
  // as far as the compiler can tell (57 spilled registers).
  //
  // BARRIER CONTRACT.  s_barrier counts waves, not source lines: every wave of the workgroup must execute the same NUMBER
  // of barriers, from whichever loop.  That holds because
  //   (1) the role branch is wave-uniform — `consumer` comes from readfirstlane(tid >> 6), all 64 lanes of a wave agree —
  //       so no barrier is ever reached by part of a wave;
  //   (2) both loops run ws_trips(...) trips — ONE expression, evaluated once, before the branch — and each trip executes
  //       exactly WS_BARRIERS_PER_TRIP barriers, after the one barrier in front of the loops (1 + 2 trips in all);
  //   (3) no barrier sits under any other condition (the second multiply of a trip is conditional, its barrier is not).
  // tests/test_gpu_parity.py::test_warp_specialised_conv1b_trip_count_edges runs odd tile counts, fewer tiles than
  // workgroups and 2 G + 1 tiles against the general kernel.
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  const int G = gridDim.x, t0 = blockIdx.x;
  const int trips = ws_trips(ntiles, t0, G);         // the launcher keeps G <= ntiles, so t0 < ntiles and trips >= 1
  if (consumer) {
    __syncthreads();
    for (int k = 0; k < trips; ++k) {
      const int t = t0 + 2 * k * G;
      multiply(t, 0);
      __syncthreads();
      if (t + G < ntiles) multiply(t + G, 1);
      __syncthreads();
      static_assert(WS_BARRIERS_PER_TRIP == 2, "multiplying loop: two barriers per trip");
    }
  } else if constexpr (STEM) {
    stem_request(t0);
    stem_commit(0);
    stem_request(t0 + G);
    __syncthreads();
    for (int k = 0; k < trips; ++k) {
      const int t = t0 + 2 * k * G;
      stem_commit(1);                  // tile t + G, requested a tile ago
      stem_request(t + 2 * G);
      __syncthreads();
      stem_commit(0);                  // tile t + 2 G
      stem_request(t + 3 * G);
      __syncthreads();
      static_assert(WS_BARRIERS_PER_TRIP == 2, "staging loop (stem): two barriers per trip");
    }
  } else {
    request(t0, S0{});
    commit(0, S0{});
    request(t0 + G, S0{});
    request(t0 + 2 * G, S1{});
    __syncthreads();
    for (int k = 0; k < trips; ++k) {
      const int t = t0 + 2 * k * G;
      commit(1, S0{});
      request(t + 3 * G, S0{});
      __syncthreads();
      commit(0, S1{});
      request(t + 4 * G, S1{});
      __syncthreads();
      static_assert(WS_BARRIERS_PER_TRIP == 2, "staging loop: two barriers per trip");
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backbone.conv1a as its own launch in the split-fp16 arithmetic — the SAME products, in the same MFMA sequence with the same
// operand layout, as the STEM staging code above: an output value depends on its pixel's 27 inputs and the weights only, so
// this kernel and the fused form agree bit for bit, and a forward's results do not depend on which of the two a grid size,
// a sub-batch split or a lane count picks (tests/test_gpu_parity.py::test_full_size_properties).  Small grids, sub-batches
// below the warp-specialised form's tile count, and uint8 frames (U8: /255, the bilinear resize and .sub(0.5).mul(2) of
// kp2d_preprocess computed while the window is filled — conv3x3.hip conv1a_u8_kernel's arithmetic) take this kernel.
// A wave = one row x 64 pixels (four M-tiles); its 3 x 72 window [row][x][c0 c1 c2 0] hi / lo lives in its own LDS block.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int C1_CW = 72, C1_PL = 3 * C1_CW * 8;       // window columns (x0 - 4 .. x0 + 67); one plane (1,728 B)
}
template <bool U8>
__global__ __launch_bounds__(256) void conv1a_mfma_kernel(const Conv1aArgs a, const StemArgs st, const unsigned char* __restrict__ frames,
                                                          const int Hs, const int Ws) {
  __shared__ __attribute__((aligned(16))) char s_win[4 * 2 * C1_PL];
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);       // FP16_OVFL (as every split kernel)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slg = lane >> 4, slp = lane & 15;
  const int H = a.H, W = a.W;
  const int tiles_x = (W + 63) >> 6, tiles_y = (H + 3) >> 2;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y = ty * 4 + wave, x0 = tx * 64;
  char* const xw = s_win + wave * 2 * C1_PL;
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  constexpr int OOB = 0x7ffffff0;
  // weights / BatchNorm: exactly the STEM set-up
  f16x8 swh[2], swl[2];
  float ssc[4], ssh[4];
  const float wscale = st.wsc[0];                     // a power of two: 1 / wscale is exact
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int hh = 0; hh < 8; ++hh) {
      const int tap = j == 0 ? 2 * slg + (hh >> 2) : 8, ci = hh & 3;
      const bool on = ci < 3 && (j == 0 || (slg == 0 && hh < 4));
      const float wv = on ? st.w[(ci * 9 + tap) * 16 + slp] * wscale : 0.f;
      const _Float16 h = (_Float16)wv;
      swh[j][hh] = h;
      swl[j][hh] = (_Float16)(wv - (float)h);
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) { ssc[i] = st.scale[4 * slg + i] * (1.f / wscale); ssh[i] = st.shift[4 * slg + i]; }
  // window: slot e = (row e / 18, column quad e % 18), three planes
  if (lane < 54) {
    const int wr = lane / 18, q = lane - 18 * wr;
    const int gy = y - 1 + wr, gx = x0 - 4 + 4 * q;
    float c0[4], c1[4], c2[4];
    if constexpr (!U8) {
      const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(st.x), 0, (int)((long)a.B * 3 * H * W * 4), 0x00020000);
      const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;      // (W is a multiple of 8: whole float4s)
      const int off = ok ? ((b * 3 * H + gy) * W + gx) * 4 : OOB;
      const float4 v0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0));
      const float4 v1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsx, off, H * W * 4, 0));
      const float4 v2 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 2 * H * W * 4, 0));
      c0[0] = v0.x; c0[1] = v0.y; c0[2] = v0.z; c0[3] = v0.w;
      c1[0] = v1.x; c1[1] = v1.y; c1[2] = v1.z; c1[3] = v1.w;
      c2[0] = v2.x; c2[1] = v2.y; c2[2] = v2.z; c2[3] = v2.w;
    } else {
      const unsigned char* img = frames + (size_t)b * Hs * Ws * 3;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int xx = gx + i;
        float v[3] = {0.f, 0.f, 0.f};                     // the convolution's zero padding (of the NORMALISED frame)
        if (gy >= 0 && gy < H && xx >= 0 && xx < W) frame_pixel(img, Hs, Ws, H, W, gy, xx, v);      // conv_common.h (shared with kp2d_preprocess)
        c0[i] = v[0]; c1[i] = v[1]; c2[i] = v[2];
      }
    }
    char* const d = xw + (wr * C1_CW + 4 * q) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f16x2 ha, la, hb, lb;
      split2(c0[i], c1[i], ha, la);
      split2(c2[i], 0.f, hb, lb);
      *reinterpret_cast<h4*>(d + 8 * i) = h4{ha[0], ha[1], hb[0], hb[1]};
      *reinterpret_cast<h4*>(d + 8 * i + C1_PL) = h4{la[0], la[1], lb[0], lb[1]};
    }
  }
  asm volatile("" ::: "memory");                     // (wave-private window: in-order LDS; the compiler must keep the order too)
  auto toff = [](int t) { return ((t / 3) * C1_CW + (t % 3)) * 8; };
  const int so0 = toff(2 * slg), so1 = toff(2 * slg + 1);
  const float slope1 = st.act == ACT_LEAKY ? 0.01f : (st.act == ACT_RELU ? 0.f : 1.f);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const char* const ph = xw + (16 * mt + slp + 3) * 8;      // window pixel of tap (0, 0): column x - 1 = window column x - x0 + 3
    const h4 a0 = *reinterpret_cast<const h4*>(ph + so0), a1 = *reinterpret_cast<const h4*>(ph + so1);
    const h4 b0 = *reinterpret_cast<const h4*>(ph + so0 + C1_PL), b1 = *reinterpret_cast<const h4*>(ph + so1 + C1_PL);
    const h4 a8 = *reinterpret_cast<const h4*>(ph + toff(8)), b8 = *reinterpret_cast<const h4*>(ph + toff(8) + C1_PL);
    const f16x8 xh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const f16x8 xl = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    const f16x8 yh = {a8[0], a8[1], a8[2], a8[3], 0, 0, 0, 0};
    const f16x8 yl = {b8[0], b8[1], b8[2], b8[3], 0, 0, 0, 0};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};                // (the STEM sequence, term for term)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(swh[1], yl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(swl[1], yh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(swh[0], xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(swl[0], xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(swh[1], yh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(swh[0], xh, acc, 0, 0, 0);
    const int x = x0 + 16 * mt + slp;
    float4 v;
    { const float tt = fmaf(acc[0], ssc[0], ssh[0]); v.x = fmaxf(tt, tt * slope1); }
    { const float tt = fmaf(acc[1], ssc[1], ssh[1]); v.y = fmaxf(tt, tt * slope1); }
    { const float tt = fmaf(acc[2], ssc[2], ssh[2]); v.z = fmaxf(tt, tt * slope1); }
    { const float tt = fmaf(acc[3], ssc[3], ssh[3]); v.w = fmaxf(tt, tt * slope1); }
    if (y < H && x < W) reinterpret_cast<float4*>(a.out + (((size_t)b * H + y) * W + x) * 16)[slg] = v;
  }
}

int launch_conv1a_mfma(const Conv1aArgs& a, const float* wscale_dev, const unsigned char* frames, int Hs, int Ws, hipStream_t s) {
  if (a.cout != 16 || a.cin != 3 || (a.W & 7)) return -1001;
  if ((long)a.B * 3 * a.H * a.W * 4 >= 0x7ffffff0L) return -1002;
  const StemArgs st{a.x, a.w, a.scale, a.shift, wscale_dev, a.act};
  const int grid = ((a.W + 63) >> 6) * ((a.H + 3) >> 2) * a.B;
  if (frames) hipLaunchKernelGGL(conv1a_mfma_kernel<true>, dim3(grid), dim3(256), 0, s, a, st, frames, Hs, Ws);
  else hipLaunchKernelGGL(conv1a_mfma_kernel<false>, dim3(grid), dim3(256), 0, s, a, st, frames, Hs, Ws);
  return (int)hipGetLastError();
}

template <bool S16OUT, bool STEM>
static int launch_ws_t(const ConvArgs& a0, const StemArgs& st, hipStream_t s) {
  ConvArgs a = a0;
  a.tiles_x = (a.W + WS_TW - 1) / WS_TW;
  a.tiles_y = (a.H + WS_TH - 1) / WS_TH;
  const long ntiles = (long)a.tiles_x * a.tiles_y * a.B;
  static PerDeviceOnce lds_once;
  if (int e = lds_opt_in(lds_once, reinterpret_cast<const void*>(&conv3x3_f16x3_ws_kernel<S16OUT, STEM>))) return e;
  const int cus = device_cu_count();
  static const int grid_env = getenv("KP2D_WSM_GRID") ? atoi(getenv("KP2D_WSM_GRID")) : 0;      // (A/B knob: most workgroups of a persistent launch)
  const int cap = grid_env > 0 && grid_env < cus ? grid_env : cus;
  const int grid = (int)(ntiles < cap ? ntiles : cap);
  hipLaunchKernelGGL((conv3x3_f16x3_ws_kernel<S16OUT, STEM>), dim3(grid), dim3(768),
                     WS_LDS + (S16OUT ? 8 * WS_BLK : 0) + (STEM ? 4 * 2 * ST_XPL : 0), s, a, st, (int)ntiles);
  return (int)hipGetLastError();
}
template <bool S16OUT>
static int launch_ws(const ConvArgs& a, hipStream_t s) {
  if (a.stem_x) {      // conv1a computed by the staging waves (kp2d_api.cpp hands its arguments over instead of launching it)
    const StemArgs st{a.stem_x, a.stem_w, a.stem_scale, a.stem_shift, a.stem_wscale, a.stem_act};
    return launch_ws_t<S16OUT, true>(a, st, s);
  }
  return launch_ws_t<S16OUT, false>(a, StemArgs{}, s);
}

// the map-side conditions of the warp-specialised conv1b form (the layer-side ones — 16 -> 32 channels, pooled — are the
// caller's): the plan asks before it decides to keep conv1b's output split (ST_S16P_POOL has no other producer)
static const bool ws_on = !(getenv("KP2D_WS") && getenv("KP2D_WS")[0] == '0');
bool conv3x3_ws_would_run(int B, int H, int W, int ws_min) {
  return ws_on && !(H & 1) && !(W & 1) && W >= 32 && (long)((W + 31) / 32) * ((H + 15) / 16) * B >= (ws_min > 0 ? ws_min : 1024) &&
         (long)B * H * W * 16 * 4 < 0x7ffffff0L;
}

template <int NH, int NP, int TH = TILE, bool FLAT32 = false>
static int launch_f(const ConvArgs& a0, hipStream_t s) {
  constexpr int N = NH * 32, TW = FLAT32 ? 32 : tile_w(NH, NP, TH);
  ConvArgs a = a0;
  a.tiles_x = (a.W + TW - 1) / TW;
  a.tiles_y = (a.H + TH - 1) / TH;
  size_t lds = (size_t)f_w(TW, TH) + (size_t)(NP == 2 ? 2 : 1) * 2 * 9 * N * 32;     // NP = 2: two weight slabs (LDS-DMA)
  const size_t lds_out = (size_t)N * (TH * TW + 4) * sizeof(float);
  if (a.store == ST_NCHW && lds_out > lds) lds = lds_out;
  const size_t lds_tile = (size_t)TH * TW * N * sizeof(float);
  if (a.store != ST_NCHW && lds_tile > lds) lds = lds_tile;
#ifdef KP2D_ABLATE
  // Winograd F(2,3) emulation (KP2D_DBG bit 128): its transformed input image and its 12 weight slots need 35 KB more
  if ((a.dbg & 128) && NH == 2) lds += 35 * 1024;
#endif
  static PerDeviceOnce lds_once;      // per device: a handle may live on any visible device
  if (int e = lds_opt_in(lds_once, reinterpret_cast<const void*>(&conv3x3_f16x3_kernel<NH, NP, TH, FLAT32>))) return e;
  const int grid = a.tiles_x * a.tiles_y * a.B;
  const int groups = a.npad / N;
  if (a.store == ST_NCHW && groups != 1 && a.act == ACT_SOFTMAX_C) return -1002;  // class softmax needs one group
  hipLaunchKernelGGL((conv3x3_f16x3_kernel<NH, NP, TH, FLAT32>), dim3(grid, groups), dim3(256 * NH * NP), lds, s, a);
  return (int)hipGetLastError();
}

namespace { thread_local const char* g_variant = ""; }

static size_t f_lds_bytes(const ConvArgs& a, int N, int TW, int TH, int slabs) {
  size_t lds = (size_t)f_w(TW, TH) + (size_t)slabs * 2 * 9 * N * 32;
  const size_t lds_out = (size_t)N * (TH * TW + 4) * sizeof(float);
  if (a.store == ST_NCHW && lds_out > lds) lds = lds_out;
  const size_t lds_tile = (size_t)TH * TW * N * sizeof(float);
  if (a.store != ST_NCHW && lds_tile > lds) lds = lds_tile;
  return lds;
}

// n <= 4 independent layers, every one a layer the single-frame form <1,1,8> would take (32-channel groups, a grid below 256
// workgroups); -1000: not all of them are — the caller then launches them one by one
int launch_conv3x3_f16x3_multi(const ConvArgs* list, int n, hipStream_t s) {
  if (n < 2 || n > 4) return -1000;
  static const bool multi_on = !(getenv("KP2D_MULTI") && getenv("KP2D_MULTI")[0] == '0');      // (A/B knob)
  if (!multi_on) return -1000;
  ConvMultiArgs m{};
  m.n = n;
  int gx = 0, gy = 0;
  size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    ConvArgs a = list[i];
    if (a.taps != 9 || a.prec != 1 || a.in0.fmt != 0 || a.store == ST_S16P || a.store == ST_S16P_POOL) return -1000;
    if (a.in0.rs != (long)a.W * a.in0.ps || (a.in1.c > 0 && a.in1.rs != (long)a.W * a.in1.ps)) return -1000;
    if ((long)a.H * a.W * (a.in0.ps > a.in1.ps ? a.in0.ps : a.in1.ps) * 4 >= 0x7ffffff0L) return -1000;
    if (!(a.npad == 32 || a.ng32)) return -1000;                       // 32-channel groups (kp2d_api.cpp: small grids)
    a.tiles_x = (a.W + 15) / 16;
    a.tiles_y = (a.H + 7) / 8;
    const int nx = a.tiles_x * a.tiles_y * a.B, ny = a.npad / 32;
    if ((long)((a.W + 15) / 16) * ((a.H + 15) / 16) * a.B * ny >= 256) return -1000;      // (the condition of the <1,1,8> form)
    if (a.store == ST_NCHW && ny != 1 && a.act == ACT_SOFTMAX_C) return -1000;
    gx = nx > gx ? nx : gx;
    gy = ny > gy ? ny : gy;
    const size_t l = f_lds_bytes(a, 32, 16, 8, 1);
    lds = l > lds ? l : lds;
    m.a[i] = a;
  }
  static PerDeviceOnce lds_once;
  if (int e = lds_opt_in(lds_once, reinterpret_cast<const void*>(&conv3x3_f16x3_multi_kernel))) return e;
  g_variant = "<1,1,8>x";
  hipLaunchKernelGGL(conv3x3_f16x3_multi_kernel, dim3(gx, gy, n), dim3(256), lds, s, m);
  return (int)hipGetLastError();
}
const char* conv3x3_last_variant() { return g_variant; }
void conv3x3_note_variant(const char* v) { g_variant = v; }

int launch_conv3x3_f16x3(const ConvArgs& a, hipStream_t s) {
  if (a.taps != 9 || a.prec != 1) return -1000;
  // S16P tensors (kp2d_kernels.h): read by conv3x3_s16.hip (32 input channels) and by conv3x3_wsm.hip's IN16 form, written by
  // both and by the conv1b form below.  No other kernel takes the layout: -1006 is a plan bug
  if (a.in0.fmt == 1 && a.in1.c == 0 && (a.cin == 32 || a.store == ST_NCHW)) return launch_conv3x3_f16x3_s16(a, s);
  if (a.wsm_force) {
    const int e = launch_conv3x3_f16x3_wsm(a, s, 64);
    return e == -1000 ? -1006 : e;
  }
  if (a.in0.fmt == 1 || a.in1.fmt == 1 || a.store == ST_S16P || a.store == ST_S16P_BOTH || a.store == ST_S16P_SHUFFLE || a.store == ST_MIX16) return -1006;
  if (a.store == ST_S16P_POOL) {
    if (!(a.cin == 16 && a.in0.c == 16 && a.in1.c == 0 && a.npad == 32 && a.cout == 32 && a.act <= ACT_RELU &&
          a.in0.rs == (long)a.W * a.in0.ps && a.in0.ps == 16 && a.in0.o == 0 && conv3x3_ws_would_run(a.B, a.H, a.W, a.ws_min))) return -1006;
    g_variant = a.stem_x ? "<ws>stem+s16" : "<ws>s16";
    return launch_ws<true>(a, s);
  }
  // the staging addresses a source pixel as (y * W + x) * pixel stride
  if (a.in0.rs != (long)a.W * a.in0.ps || (a.in1.c > 0 && a.in1.rs != (long)a.W * a.in1.ps)) return -1004;
  if ((long)a.H * a.W * (a.in0.ps > a.in1.ps ? a.in0.ps : a.in1.ps) * 4 >= 0x7ffffff0L) return -1002;
  if (a.npad != 32 && a.npad % 64 != 0) return -1000;
  const bool one = a.npad == 32 || a.ng32;
  if (!one) {
    // multi-chunk layers with 64-channel groups on grids that fill the chip: warp-specialised persistent form
    // (conv3x3_wsm.hip; policy and overrides at its launcher)
    {
      const int e = launch_conv3x3_f16x3_wsm(a, s, 64);
      if (e != -1000) return e;      // (the launcher noted "<wsm>" or, for the transposed walk, "<wsm>t")
    }
    // map heights that leave the last 16-row tile row at most half full (120 = 7.5 x 16): 8 x 32 tiles, no ragged row
    static const bool flat_on = !(getenv("KP2D_FLAT") && getenv("KP2D_FLAT")[0] == '0');
    const int rag = a.H & 15;
    if (flat_on && rag >= 1 && rag <= 8 && a.W >= 32 && (long)a.H * a.W < (1L << 20)) { g_variant = "<2,1,8>"; return launch_f<2, 1, 8>(a, s); }
    g_variant = "<2,1,16>";
    return launch_f<2, 1>(a, s);
  }
  // 32-channel layers on grids that fill the chip anyway: 16 x 32 pixel tiles (one weight slab per 512 pixels, 16 waves
  // per CU).  Small grids keep the 16 x 16 tiles (twice the workgroups, half as long: single frames).  KP2D_WIDE=0: never.
  // single-chunk, max-pooled, 32 channels (conv1b) on grids that fill the chip several times: warp-specialised persistent form
  if (ws_on && a.cin == 16 && a.in0.c == 16 && a.in1.c == 0 && a.npad == 32 && a.store == ST_NHWC_POOL && a.act <= ACT_RELU &&
      !(a.H & 1) && !(a.W & 1) && a.W >= 32 && (long)((a.W + 31) / 32) * ((a.H + 15) / 16) * a.B >= (a.ws_min > 0 ? a.ws_min : 1024) &&
      (long)a.B * a.in0.bs * 4 < 0x7ffffff0L)
  { g_variant = a.stem_x ? "<ws>stem" : "<ws>"; return launch_ws<false>(a, s); }
  // 32-channel layers on grids that fill the chip several times: the warp-specialised persistent form with 32-channel items
  if (a.npad == 32 && !a.ng32) {
    const int e = launch_conv3x3_f16x3_wsm(a, s, 32);
    if (e != -1000) return e;
  }
  static const bool wide_on = !(getenv("KP2D_WIDE") && getenv("KP2D_WIDE")[0] == '0');
  const long wide_tiles = (long)((a.W + 31) / 32) * a.tiles_y * a.B * (a.npad / 32);
  // (planar API outputs keep the 16-pixel tiles: measured 0.148 -> 0.151 ms on desc_head.confBb with the wide ones)
  if (wide_on && a.store != ST_NCHW && a.W >= 32 && wide_tiles >= 1024 && (long)a.H * a.W < (1L << 20)) { g_variant = "<1,2,16>"; return launch_f<1, 2>(a, s); }
  // planar outputs on maps with a half-empty last 16-row tile row (confBb / the class map at 120 rows): 8 x 32 tiles
  {
    static const bool flat_on = !(getenv("KP2D_FLAT") && getenv("KP2D_FLAT")[0] == '0');
    const int rag = a.H & 15;
    if (flat_on && a.store == ST_NCHW && rag >= 1 && rag <= 8 && a.W >= 32 && !(a.W & 3) && (long)a.H * a.W < (1L << 20) &&
        (long)((a.W + 31) / 32) * ((a.H + 7) / 8) * a.B * (a.npad / 32) >= 512)
    { g_variant = "<1,1,8,flat32>"; return launch_f<1, 1, 8, true>(a, s); }
  }
  // single frames (the grid of 16 x 16 tiles would leave most CUs idle): 8-row tiles, twice the workgroups, half as long
  static const bool short_on = !(getenv("KP2D_SHORT") && getenv("KP2D_SHORT")[0] == '0');
  if (short_on && (long)a.tiles_x * a.tiles_y * a.B * (a.npad / 32) < 256) { g_variant = "<1,1,8>"; return launch_f<1, 1, 8>(a, s); }
  g_variant = "<1,1,16>";
  return launch_f<1, 1>(a, s);
}

}  // namespace kp2d
