// 3x3 convolution on PRE-SPLIT ("S16") activations: the fast path of the f16x3 mode.
//
// Same GEMM view, tile (16x16 pixels x 64 channels per 256-thread workgroup) and epilogue as conv3x3.hip,
// but nothing on the load side touches a VGPR or the VALU:
//   * producers already wrote hi = fp16(x), lo = fp16(x - hi) per 16-channel block (conv_common.h), and the
//     packer wrote the weights in the same [16 hi | 16 lo] 64-byte rows, so both operands are copied
//     global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction);
//   * halo pixels outside the image are never written by the DMA (lanes masked off) and both input buffers
//     are zeroed once per workgroup, which is the convolution's zero padding;
//   * LDS rows are 64 B with no padding; bank conflicts are removed by an XOR swizzle of the 16-byte quad
//     index with ((row >> 2) & 3), applied on the SOURCE address of the DMA (input) or at pack time (weights)
//     and on the ds_read_b128 address — the LDS destination stays lane-linear as the DMA requires;
//   * K is walked in stages of (one 16-channel chunk, one kernel row dy = 3 taps).  Input tiles (20.25 KiB)
//     and weight slabs (12 KiB) are double-buffered: the DMA of stage s+1 is issued right after the barrier
//     that opens stage s and lands while the 36 MFMAs of stage s run.  66 KiB of LDS -> two workgroups per CU.
// Arithmetic: x*w = xh*wh + xh*wl + xl*wh on v_mfma_f32_32x32x16_f16, one fp32 accumulator (conv3x3.hip PREC 1).
#include "conv_common.h"

namespace kp2d {

constexpr int S_ROWS = 18;                       // 16 + halo
constexpr int S_SLOTS = S_ROWS * S_ROWS * 4;     // 16-byte quads in one input tile (1296)
constexpr int S_IN_INSTR = (S_SLOTS + 63) / 64;  // DMA wave-instructions per input tile (21)
constexpr int S_IN_BYTES = S_IN_INSTR * 1024;

#define KP2D_LDS(ptr) ((__attribute__((address_space(3))) void*)(ptr))
#define KP2D_GLB(ptr) ((const __attribute__((address_space(1))) void*)(ptr))

template <int NT>
__global__ __launch_bounds__(256, 2) void conv3x3_s16_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int N = NT * 32;
  constexpr int W_BYTES = 3 * N * 64;            // one stage: 3 taps x N rows x 64 B
  constexpr int W_INSTR = W_BYTES / 1024;
  char* lds = reinterpret_cast<char*>(smem);
  char* s_in = lds;                              // [2][S_IN_BYTES]
  char* s_w = lds + 2 * S_IN_BYTES;              // [2][W_BYTES]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int b = bid / a.tiles_y;
  const int y0 = ty * TILE, x0 = tx * TILE;
  const int H = a.H, W = a.W;
  const int n0 = blockIdx.y * N;
  const int nchunk = a.cin >> 4;
  const int nstage = 3 * nchunk;

  // zero both input buffers once: out-of-image halo slots are never written afterwards
  for (int e = tid; e < 2 * S_IN_BYTES / 16; e += 256)
    reinterpret_cast<float4*>(s_in)[e] = make_float4(0.f, 0.f, 0.f, 0.f);

  const float* src0 = a.in0.p + (size_t)b * a.in0.bs + a.in0.o;
  const float* src1 = a.in1.p + (size_t)b * a.in1.bs + a.in1.o;
  const int c0_blocks = a.in0.c >> 4;
  const float* wbase = a.w + (size_t)blockIdx.y * nchunk * 9 * N * 16;

  auto issue_in = [&](int c) {
    const bool first = c < c0_blocks;
    const float* src = first ? src0 + (c << 4) : src1 + ((c - c0_blocks) << 4);
    const long rs = first ? a.in0.rs : a.in1.rs, ps = first ? a.in0.ps : a.in1.ps;
    char* dst = s_in + (c & 1) * S_IN_BYTES;
    for (int k = wave; k < S_IN_INSTR; k += 4) {
      const int slot = k * 64 + lane;
      const int p = slot >> 2;
      const int q = (slot & 3) ^ ((p >> 2) & 3);        // source-side swizzle
      const int py = p / S_ROWS, px = p - py * S_ROWS;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      if (slot < S_SLOTS && gy >= 0 && gy < H && gx >= 0 && gx < W)
        __builtin_amdgcn_global_load_lds(KP2D_GLB(src + gy * rs + gx * ps + (q << 2)), KP2D_LDS(dst + k * 1024), 16, 0, 0);
    }
  };
  auto issue_w = [&](int s) {
    const float* slab = wbase + (size_t)s * (W_BYTES / 4);
    char* dst = s_w + (s & 1) * W_BYTES;
    for (int k = wave; k < W_INSTR; k += 4)
      __builtin_amdgcn_global_load_lds(KP2D_GLB(slab + (k * 64 + lane) * 4), KP2D_LDS(dst + k * 1024), 16, 0, 0);
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int prow = wave * 4 + ((i >> 1) & 1);
  const int pcol = 2 * (i >> 2) + (i & 1);
  const int pbase = prow * S_ROWS + pcol;
  // weight row r = dx*N + n*32 + i: (r >> 2) & 3 == (i >> 2) & 3 because N and 32 are multiples of 16
  const int wq = (h ^ ((i >> 2) & 3)) << 4;      // byte offset of this lane's hi quad inside a weight row

  __syncthreads();            // zero fill complete before any DMA lands
  issue_in(0);
  issue_w(0);

  for (int s = 0; s < nstage; ++s) {
    const int c = s / 3, dy = s - 3 * c;
    __syncthreads();          // vmcnt(0) + barrier: stage s has landed, stage s-1 is fully consumed
    if (s + 1 < nstage) issue_w(s + 1);
    if (dy == 0 && c + 1 < nchunk) issue_in(c + 1);

    const char* in = s_in + (c & 1) * S_IN_BYTES;
    const char* wrow = s_w + (s & 1) * W_BYTES + i * 64;
    const int pdy = pbase + dy * S_ROWS;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      f16x8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int p = pdy + 2 * m * S_ROWS + dx;
        const int off = p * 64 + ((h ^ ((p >> 2) & 3)) << 4);
        ah[m] = *reinterpret_cast<const f16x8*>(in + off);
        al[m] = *reinterpret_cast<const f16x8*>(in + (off ^ 32));
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const char* p = wrow + (dx * N + n * 32) * 64;
        bh[n] = *reinterpret_cast<const f16x8*>(p + wq);
        bl[n] = *reinterpret_cast<const f16x8*>(p + (wq ^ 32));
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
  constexpr int EPI_ROUNDS = 1;
  constexpr bool EPI_GELU = false;
#include "conv_epilogue.inc"
}

template <int NT>
static int launch_s16_t(const ConvArgs& a, hipStream_t s) {
  size_t lds = 2 * S_IN_BYTES + 2 * 3 * NT * 32 * 64;
  const size_t lds_out = (size_t)NT * 32 * 257 * sizeof(float);
  if (a.store == ST_NCHW && lds_out > lds) lds = lds_out;
  const size_t lds_tile = (size_t)16 * 16 * NT * 32 * sizeof(float);   // [pixel][N] staging of the fp32 NHWC epilogue
  if (a.store != ST_NCHW && lds_tile > lds) lds = lds_tile;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_s16_kernel<NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const int grid = a.tiles_x * a.tiles_y * a.B;
  const int groups = a.npad / (NT * 32);
  if (a.store == ST_NCHW && groups != 1 && a.act == ACT_SOFTMAX_C) return -1002;
  hipLaunchKernelGGL((conv3x3_s16_kernel<NT>), dim3(grid, groups), dim3(256), lds, s, a);
  return (int)hipGetLastError();
}

// Requirements: taps == 9, both sources S16 with channel counts / offsets multiples of 16, cin % 16 == 0.
int launch_conv3x3_s16(const ConvArgs& a, hipStream_t s) {
  if (a.taps != 9 || (a.cin & 15) || (a.in0.c & 15) || (a.in0.o & 15) || (a.in1.c & 15) || (a.in1.o & 15)) return -1010;
  if (a.npad != 32 && a.npad % 64 != 0) return -1000;
  return a.npad == 32 ? launch_s16_t<1>(a, s) : launch_s16_t<2>(a, s);
}

}  // namespace kp2d
