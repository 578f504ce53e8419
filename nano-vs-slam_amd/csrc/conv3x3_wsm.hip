// Multi-chunk 3x3 convolution layers with 64-channel groups in split-fp16 arithmetic: warp-specialised, persistent.
//
// Same operator and arithmetic as conv3x3_f16x3_kernel<2, 1, *> (conv3x3_f16.hip: AnnotatedConvBnReLUModel,
// modules/base.py:14-46, inside BackBone modules/encoders.py:105-129, the heads modules/decoders/heads.py:38-104 and
// SegmentationHead modules/decoders/segmentation.py:126-157): x = xh + xl, w = wh + wl (fp16 halves), fourteen
// v_mfma_f32_16x16x32_f16 per 16-channel chunk and accumulator tile, the same slot order, the same LDS image layout.
// What differs is WHO does what, and when.  There a 512-thread workgroup is load -> split -> commit -> barrier ->
// multiply -> barrier per chunk, then an epilogue through LDS, and the phases of one workgroup overlap only with those of
// the ONE other workgroup its CU holds: measured (profiles/r3_ablation_conv.txt) matrix time and everything else add up.
// Here one 768-thread workgroup per CU walks work items (tile, 64-channel group) j0, j0 + G, ... as ONE flat sequence of
// steps (item, chunk), with two roles on different waves, each with its own vmcnt:
//   * waves 8-11 STAGE: the 18 x 34 halo image of step s + 1 goes registers -> hi / lo fp16 planes -> LDS stage (s + 1) & 1
//     while the loads of steps s + 2 and s + 3 are in flight in two register sets, and the step's 36 KB weight slab is
//     copied global -> LDS by buffer_load ... lds (no registers); nothing here ever waits for a multiply;
//   * waves 0-7 MULTIPLY step s out of stage s & 1 — a wave owns 4 rows x 16 columns x ALL 64 channels of a 16 x 32 pixel
//     tile (64 accumulator registers: half the weight-operand reads per MFMA of the 32-channel waves, and one weight slab
//     per 512 pixels instead of per 256) — and, after an item's last chunk, store it straight from the accumulators.
//     The products are TRANSPOSED (weights are the A operand, pixels the B operand), so a lane's four accumulator
//     registers are four consecutive channels of ONE pixel: a 16-byte NHWC store, no transposition through LDS and no
//     barrier in the epilogue; the pooled output is a quad max over four lanes (DPP).
// One barrier per step; the first loads of the next item are in flight two steps before its first multiply, so an item's
// load latency and its epilogue hide behind the neighbouring items' matrix work.
//
// Barrier contract: BOTH role loops run exactly 1 + wsm_padded_steps(nsteps) barriers (one before step 0, one after
// every step, steps padded to an even count); the role branch is wave-uniform (readfirstlane of the wave index), every wave of the
// workgroup executes every barrier, and no barrier sits under a condition that differs between waves of a role.
#include "conv_common.h"
#include "device_guard.h"
#include <cstdlib>
#include <type_traits>

namespace kp2d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int M_TH = 16, M_TW = 32, M_PITCH = 36, M_ROWS = M_TH + 2, M_COLS = M_TW + 2, M_PXB = 32;
constexpr int M_LO = M_ROWS * M_PITCH * M_PXB;         // byte offset of an image's lo plane (20,736)
constexpr int M_IMG = 2 * M_LO;                        // one input image: hi plane | lo plane (41,472 B)
// N = channels per work item: 64 (four 16-channel N-tiles per multiplying wave) or 32 (two; the 32-channel layers)
__host__ __device__ constexpr int m_wl(int n) { return 9 * n * 32; }                  // byte offset of the wl plane behind the wh plane (18,432 at 64)
__host__ __device__ constexpr int m_wslab(int n) { return 2 * m_wl(n); }              // a chunk's weight slab (36,864 B)
__host__ __device__ constexpr int m_stage(int n) { return M_IMG + m_wslab(n); }       // one stage: image | weights (78,336 B)
__host__ __device__ constexpr int m_ss(int n) { return 2 * m_stage(n); }              // scale | shift vectors behind the two stages (156,672)
constexpr int M_MAXN = 320;                            // channels of a layer (scale / shift in LDS; 320 = the five heads' first layers as one)
__host__ __device__ constexpr int m_lds(int n) { return m_ss(n) + 2 * M_MAXN * 4; }   // 159,232 B / 122,368 B
constexpr int M_G = M_ROWS * M_COLS * 4;               // 16-byte granules of a halo image (2448)
constexpr int M_PT = 256;                              // staging threads
constexpr int M_IT = (M_G + M_PT - 1) / M_PT;          // granules per staging thread (10)
__host__ __device__ constexpr int m_pieces(int n) { return m_wslab(n) / 1024; }       // 1-KiB LDS-DMA pieces of a weight slab (36 / 18)
constexpr int M_THREADS = 768;
// IN16 (S16P inputs): a halo image as 1-KiB LDS-DMA pieces, as conv3x3_s16.hip
constexpr int M_NPIECE = (M_IMG + 1023) / 1024;        // 41: the last one is half a piece
constexpr int M_IPW = (M_NPIECE + 3) / 4;              // image pieces per staging wave and step (11; 44 slots: three duplicates)
static_assert(M_NPIECE == 41 && M_IMG - 1024 * (M_NPIECE - 1) == 512, "the last image piece is its first 32 lanes");
static_assert(m_wslab(32) % 1024 == 0 && m_wslab(64) % 1024 == 0, "weight slabs are whole 1-KiB pieces");
static_assert(m_lds(64) <= 160 * 1024, "LDS budget");

__host__ __device__ constexpr int m_slot_tap(int s) { return s == 2 ? 3 : s == 3 ? 4 : s == 4 ? 2 : s; }
// barriers each role executes for `nsteps` steps (the ONE definition both loops are written against)
__host__ __device__ constexpr int wsm_padded_steps(int nsteps) { return (nsteps + 1) & ~1; }

struct WsmItem { int b, y0, x0, g; };
}  // namespace

// STORE: the layer's store mode (kp2d_kernels.h::Store) as a template parameter — the pooled path's registers and DPP code
// exist only in the two instantiations that pool
// IN16: the layer's input(s) are S16P tensors (kp2d_kernels.h) — the staging waves copy the halo image HBM -> LDS by LDS-DMA
// like the weights, one step ahead (both stages hold image + weights: there is no third image stage to run further ahead, and
// no register set either), and issue nothing else.
// S16P outputs (ST_S16P, ST_S16P_SHUFFLE, the upper groups of ST_MIX16): the weight rows of an item's N-tile PAIRS are
// interleaved on their way into LDS (row 32 p + 16 q + 4 g + i <- channel 32 p + 8 g + 4 q + i), so lane group g of the
// accumulator tiles (2 p, 2 p + 1) holds channels 32 p + 8 g .. + 7 of its pixel = 16 bytes of a plane; the epilogue
// splits in registers and stores [8 hi halves] / [8 lo halves] — the same bytes and store count as the fp32 NHWC form.
template <int STORE, int NN, bool IN16>
__global__ __launch_bounds__(M_THREADS, 3) void conv3x3_f16x3_wsm_kernel(const ConvArgs a, const int nitems, const int ntiles) {
  constexpr int M_N = 16 * NN, M_WL = m_wl(M_N), M_WSLAB = m_wslab(M_N), M_STAGE = m_stage(M_N), M_SS = m_ss(M_N);
  constexpr int M_PIECES = m_pieces(M_N), M_PW = (M_PIECES + 3) / 4;      // pieces per staging wave: 9, or 5 / 4
  constexpr bool S16OUT = STORE == ST_S16P || STORE == ST_S16P_SHUFFLE || STORE == ST_MIX16;
  static_assert(!(S16OUT || IN16) || NN == 4, "S16P tensors: 64-channel items");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const sm = reinterpret_cast<char*>(smem);
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);       // FP16_OVFL: conversions that overflow clamp to +-65504
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wave < 8;                    // wave-uniform by construction (see the barrier contract above)
  // Tile space.  A tile is 16 rows x 32 columns of (H, W); with a.wsm_tr those are the map's COLUMNS x ROWS — the tile walks
  // the map transposed (tile row = map column), the weight slab has its taps transposed to match (kp2d_api.cpp w16t), and
  // only the two places that form global addresses know: the staging waves' pixel index and the epilogue's (y, x).
  // Why: a 30 x 40 map is 2 x 2 tiles with 8 of the second tile column's 32 columns inside (24 quarter-SIMD units of matrix
  // work per frame for 18.75 of pixels); as 40 x 30 it is 3 x 1 tiles, the last with 8 of 16 rows = half of the waves, none
  // of them multiplying padding (20 units).  The launcher picks the cheaper walk per layer (wsm_walk_cost).
  const bool tr = a.wsm_tr != 0;
  const int H = tr ? a.W : a.H, W = tr ? a.H : a.W;
  const int Wg = a.W;                                // pixels per row of the map in memory
  const int groups = a.npad / M_N;
  const int nchunk = a.cin >> 4;
  const int c0 = a.in0.c;
  constexpr int OOB = 0x7ffffff0;

  // ---- this workgroup's items: j = i G + off; every XCD (workgroups t0 = x mod 8) owns a contiguous run per round ----
  const int G = gridDim.x, t0 = blockIdx.x;
  const int off = (t0 & 7) * (G >> 3) + (t0 >> 3);
  const int n_my = (nitems - off + G - 1) / G;      // >= 1: the launcher keeps G <= nitems
  const int nsteps = n_my * nchunk;
  const int nsteps_p = wsm_padded_steps(nsteps);
  // item j -> (tile u = j / groups, group j % groups).  Tile order: the cheap tiles come LAST, so that the static schedule
  // (workgroup w walks items w, w + G, ...) ends on them everywhere instead of handing some workgroups only full tiles
  // and others only half ones (with G = 96 and three tile columns, the third half empty, a third of the workgroups had
  // half the work of the rest).  A tile is cheap when half of its waves idle: the last tile column when at most 16 of
  // its 32 pixel columns exist, the last tile row when at most 8 of its 16 rows exist.  Order: [tiles in neither],
  // [last column, rows above the last row], [last row].
  const int tx_n = a.tiles_x, ty_n = a.tiles_y;
  const int cc = (W - (tx_n - 1) * M_TW <= 16 && tx_n > 1) ? 1 : 0;
  const int rr = (H - (ty_n - 1) * M_TH <= 8 && ty_n > 1) ? 1 : 0;
  const int per_a = (ty_n - rr) * (tx_n - cc), n_a = per_a * a.B;      // per_a >= 1
  const int per_b = cc * (ty_n - rr), n_b = per_b * a.B;
  auto decode = [&](int i) -> WsmItem {
    const int j = i * G + off;
    int u = j / groups;
    WsmItem r;
    r.g = j - u * groups;
    if (u < n_a) {
      r.b = u / per_a;
      const int q = u - r.b * per_a, ty = q / (tx_n - cc);
      r.y0 = ty * M_TH; r.x0 = (q - ty * (tx_n - cc)) * M_TW;
    } else if (u < n_a + n_b) {
      u -= n_a;
      r.b = u / per_b;
      r.y0 = (u - r.b * per_b) * M_TH; r.x0 = (tx_n - 1) * M_TW;
    } else {
      u -= n_a + n_b;
      r.b = u / tx_n;
      r.y0 = (ty_n - 1) * M_TH; r.x0 = (u - r.b * tx_n) * M_TW;
    }
    return r;
  };

#ifdef KP2D_ABLATE
  // (experiment: start the workgroups out of step — every CU otherwise reaches its items' store bursts together)
  if (KP2D_DBG_ON(1024)) for (int i = 0; i < (int)(blockIdx.x & 7); ++i) __builtin_amdgcn_s_sleep(33);
  if (KP2D_DBG_ON(2048)) for (int i = 0; i < (int)(blockIdx.x & 3); ++i) __builtin_amdgcn_s_sleep(100);
  // (blockIdx.x & 7 is the XCD: the two skews above shift whole XCDs against each other.  Within an XCD — whose L2 and
  // fabric port take the 32 CUs' item stores — neighbours are blockIdx.x >> 3:)
  if (KP2D_DBG_ON(8192)) for (int i = 0; i < 2 * (int)((blockIdx.x >> 3) & 1); ++i) __builtin_amdgcn_s_sleep(115);      // half of the CUs ~7 us late
  if (KP2D_DBG_ON(16384)) for (int i = 0; i < (int)((blockIdx.x >> 3) & 3); ++i) __builtin_amdgcn_s_sleep(58);           // quarters, ~1.75 us apart
#endif
  // per-channel scale | shift of the whole layer -> LDS (the multiplying waves read 4 channels per ds_read_b128)
  for (int c = tid; c < a.npad; c += M_THREADS) {
    smem[M_SS / 4 + c] = a.scale[c];
    smem[M_SS / 4 + M_MAXN + c] = a.shift[c];
  }

  if (!consumer) {
    // =================================== staging waves ===================================
    // The staging waves share their SIMDs with the multiplying waves, whose MFMAs leave about half of the vector issue
    // slots: every instruction here is issue time taken from, or waited for by, the matrix stream (first form of this
    // kernel: ~870 instructions per step and s_setprio 1 — the two roles ran one after the other, profiles/r4_wsm_*).
    // So everything that does not change from step to step lives in per-thread tables, the per-item part (pixel indices
    // of the halo granules) is computed once per item, and a load is four VALU instructions.
    if (KP2D_DBG_ON(256)) __builtin_amdgcn_s_setprio(1);
    const int ptid = tid - 512, pw = wave - 8;
    // weight slab of a step -> stage: piece p = pw + 4 j covers LDS bytes [1024 p, 1024 p + 1024) of [wh plane | wl plane],
    // each [slot][n][32 B]; a lane's 16 bytes come from the packed [16 hi | 16 lo] row: lane l reads row 32 p' + l / 2
    // (p' = piece inside its plane) — for S16P outputs row 32 p' + perm(l / 2), the N-tile pairs interleaved (above) —
    // half (l & 1) of the plane's 32 bytes: the lane part is one register for all pieces, the piece part is scalar
    const int w_row = lane >> 1;
    const int w_lane_f = w_row * 64 + (lane & 1) * 16;
    const int w_lane_p = (8 * ((w_row >> 2) & 3) + 4 * (w_row >> 4) + (w_row & 3)) * 64 + (lane & 1) * 16;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w), 0, groups * nchunk * M_WSLAB, 0x00020000);
    auto group_of = [&](int i) { const int j = i * G + off; return groups == 1 ? 0 : j % groups; };
    auto copy_slab = [&](int stage, int g, int ch) {
      const int sbase = (g * nchunk + ch) * M_WSLAB;
      const int w_lane = (STORE == ST_MIX16 ? g * M_N >= a.nsplit : S16OUT) ? w_lane_p : w_lane_f;
#pragma unroll
      for (int j = 0; j < M_PW; ++j) {
        // (18 pieces over four waves: waves 2, 3 copy pieces 0, 1 a second time in their fifth slot — the same bytes to
        // the same place — so that every wave issues the same straight-line sequence)
        const int pc = M_PIECES % 4 == 0 ? pw + 4 * j : (pw + 4 * j) % M_PIECES;
        const int plane = pc >= M_WL / 1024 ? 1 : 0, pp = pc - plane * (M_WL / 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(sm + stage * M_STAGE + M_IMG + 1024 * pc),
                                                 16, KP2D_DBG_ON(16) ? OOB : w_lane, sbase + 2048 * pp + 32 * plane, 0, 0);
      }
    };
    if constexpr (IN16) {
      // ---- S16P inputs.  Image piece q = LDS bytes [1024 q, 1024 q + 1024) of [hi plane | lo plane]; lane l brings bytes
      // 16 l .. + 15 of it: halo pixel slot (py, px), plane, 16-byte half — its source address is the lane's own, lanes
      // on the zero padding (or on pitch columns 34, 35) point out of range and write zeros.  Wave pw owns pieces
      // pw + 4 j (j < 11): 41 pieces over 44 slots, the last slots of waves 1-3 copy pieces 0-2 a second time; piece 40
      // is half a piece (its first 32 lanes).  Per step and wave: 9 weight + 11 image copies, nothing else ----
      int t_yx[M_IPW], t_ph[M_IPW];
#pragma unroll
      for (int j = 0; j < M_IPW; ++j) {
        int q = pw + 4 * j;
        if (q >= M_NPIECE) q -= M_NPIECE;
        const int byte = 1024 * q + 16 * lane;
        const int plane = byte >= M_LO ? 1 : 0, pb = byte - plane * M_LO;
        const int sidx = pb >> 5, py = sidx / M_PITCH, px = sidx - py * M_PITCH;
        t_yx[j] = (byte < M_IMG && px < M_COLS) ? (py << 8) | px : -1;
        t_ph[j] = plane * W * 32 + ((pb >> 4) & 1) * 16;
      }
      const int cs_bytes = H * 2 * W * 32;           // chunk stride of an S16P tensor, bytes
      int rq_i = 0, rq_ch = 0, rq_b = 0, rq_g = group_of(0);
      int voff[M_IPW];                               // byte offset of each piece's 16 bytes inside (frame, chunk), or OOB
      auto enter_item = [&](int i) {
        const WsmItem r = decode(i);
        rq_b = r.b; rq_g = r.g;
        const int y0 = r.y0 - 1, x0 = r.x0 - 1;
#pragma unroll
        for (int j = 0; j < M_IPW; ++j) {
          const int gy = y0 + (t_yx[j] >> 8), gx = x0 + (t_yx[j] & 255);
          const bool ok = t_yx[j] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
          voff[j] = ok ? (gy * 2 * W + gx) * 32 + t_ph[j] : OOB;
        }
      };
      enter_item(0);
      // the step under the cursor -> `stage` (weights, then image), then advance.  Past the last step the cursor stays on
      // it: the look-ahead beyond the end re-reads the last step's operands into a stage nobody multiplies
      auto request = [&](int stage) {
        copy_slab(stage, rq_g, rq_ch);
        const bool first = rq_ch * 16 < c0;
        const ConvSrc& src = first ? a.in0 : a.in1;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(src.p) + (size_t)rq_b * src.bs, 0, (int)(src.bs * 4), 0x00020000);
        const int so = ((src.o >> 4) + (first ? rq_ch : rq_ch - (c0 >> 4))) * cs_bytes;
        char* const dst = sm + stage * M_STAGE;
#pragma unroll
        for (int j = 0; j < M_IPW; ++j) {
          int q = pw + 4 * j;
          if (q >= M_NPIECE) q -= M_NPIECE;
          if (j == M_IPW - 1) {
            // wave 0: piece 40, the image's last 512 bytes (lanes 32-63 would write into the weights); waves 1-3: a duplicate
            if (pw != 0 || lane < 32)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + 1024 * q), 16,
                                                       KP2D_DBG_ON(32) ? OOB : voff[j], so, 0, 0);
          } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + 1024 * q), 16,
                                                     KP2D_DBG_ON(32) ? OOB : voff[j], so, 0, 0);
          }
        }
        if (rq_ch + 1 < nchunk) ++rq_ch;
        else if (rq_i + 1 < n_my) { rq_ch = 0; enter_item(++rq_i); }
        __builtin_amdgcn_sched_barrier(0);
      };
      auto wait_landed = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0): LDS-DMA counts in vmcnt
        __builtin_amdgcn_sched_barrier(0);
      };
      request(0);                                      // step 0
      wait_landed();
      __syncthreads();                                 // barrier 0: stage 0 holds step 0 (and scale | shift are in LDS)
      for (int s = 0; s < nsteps_p; ++s) {
        request((s + 1) & 1);                          // step s + 1 -> the stage step s - 1 was multiplied out of
        wait_landed();
        // a bare s_barrier: __syncthreads() is fence + barrier, and this role neither reads LDS nor writes it other than by
        // the copies the counted wait above has seen land (conv3x3_s16.hip)
        __builtin_amdgcn_s_barrier();
      }
      return;
    }
    // granule gi = ptid + 256 it of a halo image = (halo pixel hp = gi / 4 = (py, px), channels 4 (gi % 4) ...)
    int g_yx[M_IT], g_lds[M_IT];
#pragma unroll
    for (int it = 0; it < M_IT; ++it) {
      const int gi = ptid + M_PT * it, hp = gi >> 2;
      const int py = hp / M_COLS, px = hp - py * M_COLS;
      g_yx[it] = gi < M_G ? (py << 8) | px : -1;             // past the image: never loaded, never committed
      g_lds[it] = (py * M_PITCH + px) * M_PXB + (ptid & 3) * 8;
    }
    const int q16 = (ptid & 3) * 16;
    const int ps0 = (int)a.in0.ps * 4, ps1 = (int)a.in1.ps * 4;
    float4 rin[2][M_IT];

    // cursors over the flat step sequence: one for the input requests, one for the weight copies.  Past the last step
    // they stay on it: the few look-ahead requests / copies beyond the end re-read the last step's operands into stages
    // nobody multiplies (every step issues the same number of vector-memory operations — the counted wait below — and
    // no load ever leaves the tensors)
    int rq_i = 0, rq_ch = 0, dm_i = 0, dm_ch = 0, dm_g = 0;
    int rq_b = 0;
    int pix[M_IT];                                   // pixel index of each granule in the request cursor's item, < 0: zero padding
    auto enter_item = [&](int i) {                   // request cursor enters item i
      const WsmItem r = decode(i);
      rq_b = r.b;
      const int y0 = r.y0 - 1, x0 = r.x0 - 1;
#pragma unroll
      for (int it = 0; it < M_IT; ++it) {
        const int gy = y0 + (g_yx[it] >> 8), gx = x0 + (g_yx[it] & 255);
        const bool ok = g_yx[it] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        pix[it] = ok ? (tr ? gx * Wg + gy : gy * Wg + gx) : -1;
      }
    };
    enter_item(0);
    dm_g = group_of(0);
    auto request = [&](auto set_c) {                 // loads of the step under the request cursor, then advance it
      constexpr int RS = decltype(set_c)::value;
      const bool first = rq_ch * 16 < c0;
      const ConvSrc& src = first ? a.in0 : a.in1;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(src.p + (size_t)rq_b * src.bs + src.o), 0, (int)((src.bs - src.o) * 4), 0x00020000);
      const int ps = first ? ps0 : ps1;
      const int so = (first ? rq_ch * 16 : rq_ch * 16 - c0) * 4 + q16;
#pragma unroll
      for (int it = 0; it < M_IT; ++it) {
        // a negative pixel index ORs the offset up to >= 0x7ffffff0: out of range, the load returns the zero padding
        int o = (pix[it] * ps + so) | ((pix[it] >> 31) & OOB);
        if (KP2D_DBG_ON(32)) o = OOB;                // (timing ablations: conv_common.h)
        rin[RS][it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0));
      }
      if (rq_ch + 1 < nchunk) ++rq_ch;
      else if (rq_i + 1 < n_my) { rq_ch = 0; enter_item(++rq_i); }
    };
    auto commit = [&](int stage, auto set_c) {       // registers -> hi / lo halves -> image of `stage`
      constexpr int RS = decltype(set_c)::value;
#pragma unroll
      for (int it = 0; it < M_IT; ++it) {
        if ((it == M_IT - 1 && g_yx[it] < 0) || KP2D_DBG_ON(2)) continue;      // (only the last granule can lie past the image)
        const int lb = stage * M_STAGE + g_lds[it];
        const float4 v = rin[RS][it];
        f16x2 h0, h1, l0, l1;
        split2(v.x, v.y, h0, l0);
        split2(v.z, v.w, h1, l1);
        *reinterpret_cast<f16x4*>(sm + lb) = f16x4{h0[0], h0[1], h1[0], h1[1]};
        *reinterpret_cast<f16x4*>(sm + M_LO + lb) = f16x4{l0[0], l0[1], l1[0], l1[1]};
      }
    };
    auto copy_w = [&](int stage) {                   // the step under the copy cursor, then advance it
      copy_slab(stage, dm_g, dm_ch);
      if (dm_ch + 1 < nchunk) ++dm_ch;
      else if (dm_i + 1 < n_my) { dm_ch = 0; dm_g = group_of(++dm_i); }
      __builtin_amdgcn_sched_barrier(0);             // the counted wait (WAIT_W) needs the copies OLDER than the next request
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // vmcnt(M_IT): everything but the newest M_IT vector-memory operations — the one request issued after a weight copy —
    // is done, i.e. this wave's pieces of the slab have landed (loads, stores and LDS-DMA count in issue order)
    constexpr int WAIT_W = (M_IT & 15) | (7 << 4) | (15 << 8) | ((M_IT >> 4) << 14);
    auto wait_w = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      if (KP2D_DBG_ON(512)) __builtin_amdgcn_s_waitcnt(0x0f70);      // (diagnostic: drain everything)
      else __builtin_amdgcn_s_waitcnt(WAIT_W);
      __builtin_amdgcn_sched_barrier(0);
    };

    // step k's image travels in register set k & 1 and lands in stage k & 1
    request(S0{});                                   // step 0
    request(S1{});                                   // step 1
    copy_w(0);                                       // weights of step 0
    commit(0, S0{});
    request(S0{});                                   // step 2
    wait_w();
    __syncthreads();                                 // barrier 0: stage 0 holds step 0
    for (int s = 0; s < nsteps_p; s += 2) {
      // during step s (even): stage 1 <- step s + 1; request step s + 3
      copy_w(1);
      commit(1, S1{});
      request(S1{});
      wait_w();
      __syncthreads();
      // during step s + 1: stage 0 <- step s + 2; request step s + 4
      copy_w(0);
      commit(0, S0{});
      request(S0{});
      wait_w();
      __syncthreads();
    }
    return;
  }

  // =================================== multiplying waves ===================================
  // wave (wr, ph) owns tile rows 4 wr .. 4 wr + 3 x columns 16 ph .. 16 ph + 15 = four M-tiles of 2 x 8 pixels, and all
  // four 16-channel N-tiles.  The second column half's waves take the row groups rotated by two, so the two waves of a
  // SIMD (w and w + 4) differ in rows AND columns: in ragged tiles the waves that still work spread over all four SIMDs.
  constexpr int MT = 4, CB = 2;
  const int lg = lane >> 4, lp = lane & 15;
  const int ph = wave >> 2, wr = (wave + 2 * ph) & 3;
  const int a0 = ((wr * 4 + ((lp >> 1) & 1)) * M_PITCH + 16 * ph + 2 * (lp >> 2) + (lp & 1)) * M_PXB + 16 * (lg & 1);
  const int a_dx = a0 + (lg >> 1) * M_PXB;                 // second tap one pixel to the right
  const int a_dy = a0 + (lg >> 1) * M_PITCH * M_PXB;       // second tap one row down
  const int a_s = a0 + (lg >> 1) * M_LO;                   // single tap: k-groups 2, 3 read the lo plane
  const int b_s = M_IMG + lp * 32 + 16 * (lg & 1);
  const int b_p = b_s + (lg >> 1) * M_N * 32;              // second tap = next slot
  auto tap_off = [](int t) constexpr { return ((t / 3) * M_PITCH + (t % 3)) * M_PXB; };
  const float slope = a.act == ACT_LEAKY ? 0.01f : (a.act == ACT_RELU ? 0.f : 1.f);
  constexpr int store = STORE;

  f32x4 acc[MT][NN];
  auto clear = [&]() {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto multiply = [&](int sb) {                      // sb: byte offset of the step's stage
#pragma unroll
    for (int slot = 0; slot < 9; slot += 2) {
      if (KP2D_DBG_ON(4) && slot >= 6) continue;      // (timing ablation: 9 of the 14 MFMAs per tile and chunk — what Winograd F(2,3) along rows would leave)
      const int tp = m_slot_tap(slot);
      const bool single = slot == 8;
      const bool dy = slot == 4;
      const int ab = sb + (single ? a_s : (dy ? a_dy : a_dx)) + tap_off(tp);
      const int bb = sb + (single ? b_s : b_p) + slot * M_N * 32;
      f16x8 bh[NN], bl[NN];
#pragma unroll
      for (int n = 0; n < NN; ++n) {
        bh[n] = *reinterpret_cast<const f16x8*>(sm + bb + n * 512);
        bl[n] = *reinterpret_cast<const f16x8*>(sm + bb + n * 512 + M_WL);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int mo = (2 * (m / CB) * M_PITCH + 8 * (m % CB)) * M_PXB;
        if (single) {
          const f16x8 x = *reinterpret_cast<const f16x8*>(sm + ab + mo);
#pragma unroll
          for (int n = 0; n < NN; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[n], x, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[n], x, acc[m][n], 0, 0, 0);
          }
        } else {
          const f16x8 zh = *reinterpret_cast<const f16x8*>(sm + ab + mo);
          const f16x8 zl = *reinterpret_cast<const f16x8*>(sm + ab + mo + M_LO);
#pragma unroll
          for (int n = 0; n < NN; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[n], zl, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[n], zh, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[n], zh, acc[m][n], 0, 0, 0);
          }
        }
      }
    }
  };
  // ---- epilogue.  Accumulator tile (m, n): lane (lp, lg) holds pixel lp of M-tile m, channels 16 n + 4 lg .. + 3 of the
  // item's group.  A lane's byte offset per M-tile (pixel part, padding lanes ORed out of range) is computed once per
  // item; the channel / sub-pixel part of an N-tile is wave-uniform: one scalar added per store.
  // (NOT in the store's soffset operand: with an SGPR there hipcc pads no wait state between a 16-byte buffer store and
  // the next VALU write of its data registers — LLVM's rule says that form has no hazard — and on gfx950 the first
  // dword of lanes 12-15 of every row of 16 was then, now and again, the NEXT tile's value: profiles/r4_wsm_store_hazard.txt)
  // (Measured and rejected, profiles/r4_wsm_split_epilogue.txt: an item's stores spread over its last chunk and the next
  // item's first — N-tile pairs finished one after the other, the finished pair stored in pieces between the other
  // pair's slot groups.  With separate bodies for plain chunks the register allocator no longer kept the MFMAs in place
  // and spilled 40-80 accumulator registers; as ONE body with always-running pieces the matrix phase became issue-bound
  // (2.36 ms for the layers that take 1.96 ms this way), with branch-skipped pieces 2.12 ms.) ----
  constexpr bool full = store != ST_NHWC_POOL;
  constexpr bool pooled = store == ST_NHWC_POOL || store == ST_NHWC_BOTH;
  constexpr int up = (store == ST_SHUFFLE || store == ST_S16P_SHUFFLE) ? 2 : 1;
  const int HH = a.H * up, WW = a.W * up, Hp = a.H >> 1, Wp = a.W >> 1;      // the outputs, as they lie in memory
  const int cq = a.cout >> 2;
  // S16P output of an item: per N-tile pair p the lane group lg holds channels cn + 8 lg .. + 7 (cn = 64 g + 32 p) of its
  // pixel — chunk (cn + 8 lg) / 16, halves 8 (lg & 1) .. + 7 — as two accumulator tiles; split2 as a consumer's staging
  // would have done it, two 16-byte stores (hi plane, lo plane) per M-tile.  ST_S16P_SHUFFLE: a pair is one sub-pixel
  // (cout / 4 is a multiple of 32).  ST_MIX16: the S16P tensor is out1 and starts at channel nsplit.
  auto finish16 = [&](const WsmItem& it) {
    // (lane-derived values of this epilogue are recomputed per item from an opaque copy of the lane index: hoisted out of the
    // step loop they cost the registers the multiply phase needs — ST_MIX16, with both epilogues, spilled eight of them)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int lg = lane_o >> 4, lp = lane_o & 15;
    float* const op = store == ST_MIX16 ? a.out1 : a.out0;
    const int Ct = store == ST_MIX16 ? a.os1 : a.os0;                      // channels of the S16P tensor
    const int cbase = store == ST_MIX16 ? a.nsplit - a.oo1 : -a.oo0;       // layer channel of the tensor's channel 0
    const int obs = HH * WW * Ct;                                         // frame stride, floats (= bytes / 4)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(op + (size_t)it.b * obs, 0, obs * 4, 0x00020000);
    const int prow = (lp >> 1) & 1, pcol = 2 * (lp >> 2) + (lp & 1);      // this lane's pixel inside a 2 x 8 M-tile
    int vo[MT], vi[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int y = it.y0 + wr * 4 + 2 * (m / CB) + prow, x = it.x0 + 16 * ph + 8 * (m % CB) + pcol;
      vo[m] = (up * y * 2 * WW + up * x) * 32;
      vi[m] = (y < H && x < W) ? 0 : OOB;
    }
#pragma unroll
    for (int pr = 0; pr < NN / 2; ++pr) {
      const int cn = it.g * M_N + pr * 32;                                // first channel of the pair (wave-uniform)
      if (cn >= a.cout) continue;
      const int cl = cn + 8 * lg;                                         // this lane's first channel
      const f32x4 sc0 = *reinterpret_cast<const f32x4*>(sm + M_SS + cl * 4);
      const f32x4 sc1 = *reinterpret_cast<const f32x4*>(sm + M_SS + (cl + 4) * 4);
      const f32x4 sh0 = *reinterpret_cast<const f32x4*>(sm + M_SS + (M_MAXN + cl) * 4);
      const f32x4 sh1 = *reinterpret_cast<const f32x4*>(sm + M_SS + (M_MAXN + cl + 4) * 4);
      const int cinv = cl < a.cout ? 0 : OOB;                             // chunks past cout (a multiple of 16)
      int cpart;
      if (store == ST_S16P_SHUFFLE) {
        const int sub = cn / cq, oc = cl - sub * cq - cbase;
        cpart = (oc >> 4) * (HH * 2 * WW * 32) + ((sub >> 1) * 2 * WW + (sub & 1)) * 32 + (lg & 1) * 16;
      } else {
        cpart = ((cl - cbase) >> 4) * (HH * 2 * WW * 32) + (lg & 1) * 16;
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        auto affine = [&](const f32x4 v, const f32x4 sc, const f32x4 sh) -> f32x4 {
          const f32x2 a01 = {v[0], v[1]}, a23 = {v[2], v[3]};
          const f32x2 t01 = __builtin_elementwise_fma(a01, f32x2{sc[0], sc[1]}, f32x2{sh[0], sh[1]});
          const f32x2 t23 = __builtin_elementwise_fma(a23, f32x2{sc[2], sc[3]}, f32x2{sh[2], sh[3]});
          const f32x2 u01 = t01 * slope, u23 = t23 * slope;
          return f32x4{fmaxf(t01[0], u01[0]), fmaxf(t01[1], u01[1]), fmaxf(t23[0], u23[0]), fmaxf(t23[1], u23[1])};
        };
        const f32x4 v0 = affine(acc[m][2 * pr], sc0, sh0), v1 = affine(acc[m][2 * pr + 1], sc1, sh1);
        f16x2 h0, l0, h1, l1, h2, l2, h3, l3;
        split2(v0[0], v0[1], h0, l0);
        split2(v0[2], v0[3], h1, l1);
        split2(v1[0], v1[1], h2, l2);
        split2(v1[2], v1[3], h3, l3);
        const i32x4 hi = {__builtin_bit_cast(int, h0), __builtin_bit_cast(int, h1), __builtin_bit_cast(int, h2), __builtin_bit_cast(int, h3)};
        const i32x4 lo = {__builtin_bit_cast(int, l0), __builtin_bit_cast(int, l1), __builtin_bit_cast(int, l2), __builtin_bit_cast(int, l3)};
        const int o = vo[m] + cpart, inv = vi[m] | cinv;
        __builtin_amdgcn_raw_buffer_store_b128(hi, rs, KP2D_DBG_ON(64) ? OOB : (o | inv), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(lo, rs, KP2D_DBG_ON(64) ? OOB : ((o + WW * 32) | inv), 0, 0);
      }
    }
  };
  auto finish = [&](const WsmItem& it) {
    if constexpr (S16OUT) {
      if (store != ST_MIX16 || it.g * M_N >= a.nsplit) { finish16(it); return; }
    }
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
        a.out0 + (size_t)it.b * HH * WW * a.os0, 0, full ? HH * WW * a.os0 * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        a.out1 + (size_t)it.b * Hp * Wp * a.os1, 0, pooled ? Hp * Wp * a.os1 * 4 : 0, 0x00020000);
    const int prow = (lp >> 1) & 1, pcol = 2 * (lp >> 2) + (lp & 1);      // this lane's pixel inside a 2 x 8 M-tile
    int vo[MT], vp[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int ty = it.y0 + wr * 4 + 2 * (m / CB) + prow, tx = it.x0 + 16 * ph + 8 * (m % CB) + pcol;      // tile space
      const int inv = (ty < H && tx < W) ? 0 : OOB;
      const int y = tr ? tx : ty, x = tr ? ty : tx;
      vo[m] = full ? ((up * y * WW + up * x) * a.os0 + 4 * lg) * 4 | inv : 0;
      const int yp = y >> 1, xp = x >> 1;
      const int invp = ((lp & 3) == 0 && yp < Hp && xp < Wp) ? 0 : OOB;      // lane 4 q stores the quad's maximum
      vp[m] = pooled ? ((yp * Wp + xp) * a.os1 + 4 * lg) * 4 | invp : 0;
    }
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      const int cn = it.g * M_N + n * 16;                       // first channel of the N-tile (wave-uniform)
      if (cn >= a.cout) continue;                               // a padded N-tile (cout = 48: the fourth)
      const f32x4 sc = *reinterpret_cast<const f32x4*>(sm + M_SS + (cn + 4 * lg) * 4);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(sm + M_SS + (M_MAXN + cn + 4 * lg) * 4);
      const int cinv = cn + 4 * lg < a.cout ? 0 : OOB;          // channels past cout (a multiple of 4)
      int s0;                                                   // scalar byte offset: channel (and sub-pixel of PixelShuffle)
      if (store == ST_SHUFFLE) {
        const int sub = cn / cq;
        s0 = (((sub >> 1) * WW + (sub & 1)) * a.os0 + a.oo0 + cn - sub * cq) * 4;
      } else {
        s0 = (a.oo0 + cn) * 4;
      }
      const int s1 = (a.oo1 + cn) * 4;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        // scale / shift and the slope product two values per instruction (v_pk_fma_f32 / v_pk_mul_f32: same fused
        // arithmetic as fmaf / *, half the issue slots — no MFMA of this wave or of its SIMD partner is in flight here)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 a01 = {acc[m][n][0], acc[m][n][1]}, a23 = {acc[m][n][2], acc[m][n][3]};
        const f32x2 t01 = __builtin_elementwise_fma(a01, f32x2{sc[0], sc[1]}, f32x2{sh[0], sh[1]});
        const f32x2 t23 = __builtin_elementwise_fma(a23, f32x2{sc[2], sc[3]}, f32x2{sh[2], sh[3]});
        const f32x2 u01 = t01 * slope, u23 = t23 * slope;
        const f32x4 v = {fmaxf(t01[0], u01[0]), fmaxf(t01[1], u01[1]), fmaxf(t23[0], u23[0]), fmaxf(t23[1], u23[1])};
        if (full) {
#ifdef KP2D_ABLATE
          if (KP2D_DBG_ON(4096)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rs0, (vo[m] + s0) | cinv, 0, 2);      // (experiment: nt)
          else
#endif
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rs0, KP2D_DBG_ON(64) ? OOB : ((vo[m] + s0) | cinv), 0, 0);
        }
        if (pooled) {
          // the 2 x 2 pixel block of a pooled pixel = lanes 4 q .. 4 q + 3: maximum over the quad in two DPP steps per
          // value.  v_max_f32_dpp by hand (the builtin form costs a v_mov_dpp + two v_max per step); the s_nop covers
          // "VALU write -> DPP read" for the inputs, the four independent instructions in between cover it for stage two
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
  };

  int i = 0, ch = 0;
  WsmItem cur = decode(0);
  bool busy = cur.y0 + 4 * wr < H && cur.x0 + 16 * ph < W;      // rows / columns wholly outside the map: nothing to do
  // The two waves of a SIMD (ph 0 / ph 1) finish an item in the same step.  The first stores it at the END of that step, the
  // second at the START of the next one — while the first is already multiplying — so the two store bursts do not compete for
  // the CU's one path to L2 in the same microsecond (+0.3-0.5 % end to end in six of six alternating runs; the other
  // arrangements tried — priorities, stage hand-over by LDS counters instead of barriers — gained nothing or lost:
  // profiles/r4_ab_wsm_epilogue_phase.txt).
  bool due = false;
  WsmItem prev = cur;
  clear();
  __syncthreads();                                   // barrier 0
  for (int s = 0; s < nsteps_p; ++s) {
    if (s < nsteps) {
      if (due) { finish(prev); clear(); due = false; }
      if (busy && !KP2D_DBG_ON(8)) multiply((s & 1) * M_STAGE);
      if (++ch == nchunk) {
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
    }
    __syncthreads();
  }
  if (due) finish(prev);                             // (the barriers are behind us: no wave waits for this)
}

// true when the layer can run as the kernel above (launch_conv3x3_f16x3 falls back to the general kernel otherwise)
static bool wsm_eligible(const ConvArgs& a, int N) {
  if (a.taps != 9 || a.prec != 1 || a.ng32 || a.npad % N != 0 || a.npad > M_MAXN) return false;
  const bool s16out = a.store == ST_S16P || a.store == ST_S16P_SHUFFLE || a.store == ST_MIX16;
  if (a.store != ST_NHWC && a.store != ST_SHUFFLE && a.store != ST_NHWC_BOTH && a.store != ST_NHWC_POOL && !s16out) return false;
  if (a.act > ACT_RELU) return false;
  if (((a.in0.c | a.cin) & 15) != 0 || a.cin < 32) return false;      // whole 16-channel chunks, never straddling the sources
  if (a.cout & 3) return false;
  if (a.store == ST_SHUFFLE && ((a.cout >> 2) & 15)) return false;      // a 16-channel N-tile is one sub-pixel
  if (a.W < 32) return false;
  const long ps = a.in0.ps > a.in1.ps ? a.in0.ps : a.in1.ps;
  if ((long)a.H * a.W * ps * 4 >= 0x7ffffff0L) return false;
  if (a.in0.fmt == 1) {
    // S16P sources: whole chunks of dense tensors (a view is a run of chunks: ps = the tensor's channels)
    if (N != 64 || (a.in1.c > 0 && a.in1.fmt != 1) || ((a.in0.o | a.in1.o) & 15)) return false;
    if (a.in0.bs != (long)a.H * a.W * a.in0.ps || (a.in1.c > 0 && a.in1.bs != (long)a.H * a.W * a.in1.ps)) return false;
  } else {
    if (a.in1.c > 0 && a.in1.fmt == 1) return false;
    if (a.in0.rs != (long)a.W * a.in0.ps || (a.in1.c > 0 && a.in1.rs != (long)a.W * a.in1.ps)) return false;
  }
  if (s16out) {
    // whole chunks out, 64-channel items, a pair of N-tiles = 32 channels of one sub-pixel
    if (N != 64 || (a.cout & 15)) return false;
    if (a.store == ST_S16P_SHUFFLE && (((a.cout >> 2) & 31) || (a.os0 & 15) || (a.oo0 & 15))) return false;
    if (a.store == ST_S16P && ((a.os0 & 15) || (a.oo0 & 15))) return false;
    if (a.store == ST_MIX16 && ((a.nsplit & 63) || a.nsplit <= 0 || a.nsplit >= a.cout || (a.os1 & 15) || (a.oo1 & 15) || (a.os0 & 3))) return false;
  }
  const long up = (a.store == ST_SHUFFLE || a.store == ST_S16P_SHUFFLE) ? 4 : 1;
  if ((long)a.H * a.W * up * a.os0 * 4 >= 0x7ffffff0L) return false;
  if (a.store == ST_MIX16 && (long)a.H * a.W * a.os1 * 4 >= 0x7ffffff0L) return false;
  return true;
}

// Matrix time of one map walked as Ht x Wt in tile space, in units of one wave's four M-tiles: a multiplying wave (wr, ph)
// works when its 4 rows x 16 columns touch the map, SIMD s holds waves s and s + 4 = (wr s, ph 0) and (wr (s + 2) & 3, ph 1),
// and a step lasts as long as its busiest SIMD.  (What the model leaves out — staging, the barrier — is the same per step, and
// the cheaper walk never has more steps.)
static int wsm_walk_cost(int Ht, int Wt) {
  int cost = 0;
  for (int y0 = 0; y0 < Ht; y0 += M_TH)
    for (int x0 = 0; x0 < Wt; x0 += M_TW) {
      int worst = 0;
      for (int sd = 0; sd < 4; ++sd) {
        const int b0 = (y0 + 4 * sd < Ht) ? 1 : 0;                                         // ph 0: its 16 columns start at x0
        const int b1 = (y0 + 4 * ((sd + 2) & 3) < Ht && x0 + 16 < Wt) ? 1 : 0;
        worst = b0 + b1 > worst ? b0 + b1 : worst;
      }
      cost += worst;
    }
  return cost;
}

template <int STORE, int NN, bool IN16 = false>
static int wsm_launch_one(const ConvArgs& a, int grid, long nitems, long ntiles, hipStream_t s) {
  static PerDeviceOnce lds_once;      // per instantiation and device
  if (int e = lds_opt_in(lds_once, reinterpret_cast<const void*>(&conv3x3_f16x3_wsm_kernel<STORE, NN, IN16>))) return e;
  hipLaunchKernelGGL((conv3x3_f16x3_wsm_kernel<STORE, NN, IN16>), dim3(grid), dim3(M_THREADS), m_lds(16 * NN), s, a, (int)nitems, (int)ntiles);
  return (int)hipGetLastError();
}
template <int NN>
static int wsm_launch(const ConvArgs& a, int grid, long nitems, long ntiles, hipStream_t s) {
  if constexpr (NN == 4) {
    if (a.in0.fmt == 1) {             // S16P inputs (the plan's big-grid layout, kp2d_api.cpp)
      switch (a.store) {
        case ST_NHWC: return wsm_launch_one<ST_NHWC, 4, true>(a, grid, nitems, ntiles, s);
        case ST_NHWC_POOL: return wsm_launch_one<ST_NHWC_POOL, 4, true>(a, grid, nitems, ntiles, s);
        case ST_S16P: return wsm_launch_one<ST_S16P, 4, true>(a, grid, nitems, ntiles, s);
        case ST_S16P_SHUFFLE: return wsm_launch_one<ST_S16P_SHUFFLE, 4, true>(a, grid, nitems, ntiles, s);
        case ST_MIX16: return wsm_launch_one<ST_MIX16, 4, true>(a, grid, nitems, ntiles, s);
        default: return -1006;
      }
    }
    if (a.store == ST_S16P_SHUFFLE) return wsm_launch_one<ST_S16P_SHUFFLE, 4>(a, grid, nitems, ntiles, s);
    if (a.store == ST_S16P || a.store == ST_MIX16) return -1006;
  }
  switch (a.store) {
    case ST_NHWC: return wsm_launch_one<ST_NHWC, NN>(a, grid, nitems, ntiles, s);
    case ST_SHUFFLE: return wsm_launch_one<ST_SHUFFLE, NN>(a, grid, nitems, ntiles, s);
    case ST_NHWC_BOTH: return wsm_launch_one<ST_NHWC_BOTH, NN>(a, grid, nitems, ntiles, s);
    case ST_NHWC_POOL: return wsm_launch_one<ST_NHWC_POOL, NN>(a, grid, nitems, ntiles, s);
    default: return -1006;
  }
}

// Policy.  A workgroup of this form fills its CU's LDS, so launches of two stream lanes can only run side by side on
// DISJOINT CUs: with L lanes a launch takes at most cap = CUs / L workgroups (kp2d_api.cpp passes L; a profiling forward
// runs one lane and takes the whole chip).  The form is used when a launch has more than TWO rounds of work items for that
// grid — at one item per workgroup nothing is left to overlap and the general kernels are as fast or faster (30 x 40 maps:
// 0.031 / 0.047 ms general against 0.032 / 0.052 ms, profiles/r4_layers_wsm_vs_general.txt) — and the grid is sized for whole rounds:
// rounds = ceil(items / cap), grid = ceil(items / rounds) rounded up to a multiple of 8 (192 items on a cap of 128 run
// as 2 rounds on 96 workgroups, not 1.5 rounds on 128: the first automatic policy lost 5 % at 32 frames and 20 % at
// 120 x 160 frames that way, profiles/r4_sweep_first_policy.jsonl).  Overrides: ConvArgs::wsm_min / wsm_grid
// (kp2d_set_option), else KP2D_WSM (0 = never, N = least items) and KP2D_WSM_GRID.
// n_item = 64: layers packed in 64-channel groups; 32: the 32-channel layers (npad = 32; automatic use only with KP2D_WSM32=1).
int launch_conv3x3_f16x3_wsm(const ConvArgs& a0, hipStream_t s, int n_item) {
  static const long min_env = getenv("KP2D_WSM") ? atol(getenv("KP2D_WSM")) : -1;      // -1: automatic
  static const int grid_env = getenv("KP2D_WSM_GRID") ? atoi(getenv("KP2D_WSM_GRID")) : 0;
  // 32-channel items are correct (bit-identical, tested through kp2d_set_option) but not faster: conv2a / 2b / 3a 0.088-0.094 ms
  // against 0.087-0.089 ms on the wide LDS-DMA tiles, -0.8 % end to end (profiles/r4_ab_wsm32.txt) — automatic use is off
  static const bool n32_on = getenv("KP2D_WSM32") && getenv("KP2D_WSM32")[0] == '1';
  const bool forced = a0.wsm_force != 0;      // S16P in or out: the plan already asked conv3x3_wsm_would_run; no other kernel takes the layout
  if (!forced && (a0.wsm_min < 0 || (a0.wsm_min == 0 && min_env == 0))) return -1000;
  if (n_item != 64 && n_item != 32) return forced ? -1006 : -1000;
  if (!wsm_eligible(a0, n_item)) return forced ? -1006 : -1000;
  ConvArgs a = a0;
  // Transposed walk (tile rows = map columns; needs the layer's transposed-tap pack).  OFF unless asked for: it sums the nine
  // taps in another order than every other form, and the engine's results are bit-identical whatever the batch size, lane
  // count or tile form — a walk chosen by grid size would break that.  ConvArgs::wsm_tr in (kp2d_set_option
  // "wsm_transposed", else KP2D_WSM_TR): 0 never, 1 always, 2 where the matrix-time model says it is cheaper; out: the decision.
  static const int tr_env = getenv("KP2D_WSM_TR") ? atoi(getenv("KP2D_WSM_TR")) : 0;
  const int tr_mode = a0.wsm_tr != 0 ? a0.wsm_tr : tr_env;
  a.wsm_tr = 0;
  const bool s16_any = a.in0.fmt == 1 || a.store == ST_S16P || a.store == ST_S16P_SHUFFLE || a.store == ST_MIX16;      // S16P rows are map rows
  if (n_item == 64 && a.w_tr && tr_mode > 0 && !s16_any && a.H >= 16 && (tr_mode == 1 || wsm_walk_cost(a.W, a.H) < wsm_walk_cost(a.H, a.W))) {
    a.wsm_tr = 1;
    a.w = a.w_tr;
  }
  const int Ht = a.wsm_tr ? a.W : a.H, Wt = a.wsm_tr ? a.H : a.W;
  a.tiles_x = (Wt + M_TW - 1) / M_TW;
  a.tiles_y = (Ht + M_TH - 1) / M_TH;
  const long ntiles = (long)a.tiles_x * a.tiles_y * a.B;
  const long nitems = ntiles * (a.npad / n_item);
  const int cus = device_cu_count();
  const int lanes = a.wsm_lanes > 1 ? a.wsm_lanes : 1;
  int cap = a.wsm_grid > 0 ? a.wsm_grid : (grid_env > 0 ? grid_env : cus / lanes);
  if (cap > cus) cap = cus;
  cap &= ~7;                                                   // a multiple of 8: contiguous runs per XCD
  const bool automatic = a.wsm_min == 0 && min_env < 0;
  // automatic: at least three rounds of work per workgroup (the form's start-up — two steps of loads before the first
  // product — and its drain are paid per launch: at 32 frames, 192 items per lane, it lost 7 % end to end) and, for the
  // 64-channel items, at least four chunks (conv3b, two chunks and two stores per item, is slower in this form: 0.165
  // against 0.157 ms)
  const long min_items = a.wsm_min > 0 ? a.wsm_min : (min_env > 0 ? min_env : 2 * cap + 1);
  if (cap < 8 || nitems >= (1L << 30)) return forced ? -1006 : -1000;
  if (!forced) {
    if (nitems < min_items) return -1000;
    if (automatic && n_item == 64 && a.cin < 64) return -1000;
    if (automatic && n_item == 32 && !n32_on) return -1000;
  }
  const long rounds = (nitems + cap - 1) / cap;
  int grid = (int)(((nitems + rounds - 1) / rounds + 7) & ~7L);
  if (grid > cap) grid = cap;
  if (grid > nitems) grid = (int)(nitems & ~7L);
  if (grid < 8) return forced ? -1006 : -1000;
  conv3x3_note_variant(n_item == 32 ? "<wsm32>" : (a.wsm_tr ? "<wsm>t" : (a.in0.fmt == 1 ? (s16_any && a.store != ST_NHWC && a.store != ST_NHWC_POOL ? "<wsm>s16io" : "<wsm>s16in") : (s16_any ? "<wsm>s16out" : "<wsm>"))));      // (what the engine's profile records)
  return n_item == 64 ? wsm_launch<4>(a, grid, nitems, ntiles, s) : wsm_launch<2>(a, grid, nitems, ntiles, s);
}

// the automatic policy above for a 64-channel-group layer (the plan fixes the S16P layout of the big-grid forward on the
// answer for its smallest such layer, backbone.conv4a)
// full_rounds: the automatic policy asks for more than this many full rounds of work items (2: the register-staging form;
// 1: the S16P-input form, whose start-up is one LDS-DMA round trip instead of two steps of loads — 32 frames of 240 x 320,
// 384 items on 256 workgroups: 22.8k -> 23.1k frames/s with the layout on, profiles/r5_ab_s16_all.txt)
bool conv3x3_wsm_would_run(int B, int H, int W, int groups, int lanes, int wsm_min, int grid_opt, int full_rounds) {
  static const long min_env = getenv("KP2D_WSM") ? atol(getenv("KP2D_WSM")) : -1;
  static const int grid_env = getenv("KP2D_WSM_GRID") ? atoi(getenv("KP2D_WSM_GRID")) : 0;
  if (wsm_min < 0 || (wsm_min == 0 && min_env == 0) || W < 32) return false;
  const int cus = device_cu_count();
  int cap = grid_opt > 0 ? grid_opt : (grid_env > 0 ? grid_env : cus / (lanes > 1 ? lanes : 1));
  if (cap > cus) cap = cus;
  cap &= ~7;
  const long nitems = (long)((W + M_TW - 1) / M_TW) * ((H + M_TH - 1) / M_TH) * B * groups;
  const long min_items = wsm_min > 0 ? wsm_min : (min_env > 0 ? min_env : full_rounds * cap + 1);
  return cap >= 8 && nitems >= min_items && nitems < (1L << 30);
}

}  // namespace kp2d
