// 3x3 / stride 1 / pad 1 convolution with one to four output channels, written straight to the API's planar outputs:
// score_head.convDb (64 -> 1, sigmoid), loc_head.convDb (64 -> 2, tanh), V3's score_loc_head.convDb (64 -> 3) and the
// depth outputs (heads.py:22,28-35; kp2dtiny.py:574-575, :927-935, :956).
//
// On the matrix cores these layers compute a 32-channel tile for 1-3 real channels (9.6 / 18.6 TFLOP/s,
// profiles/r2_layers.txt).  They are HBM-bound dot products: 4 * Cin bytes read per pixel for 18 * Cin * Cout FLOP.
// A workgroup owns a 4 x 16 pixel tile: it stages the 6 x 18 halo of up to 64 input channels (fp32; 68-float pixel
// pitch and 1280-float row pitch: brute-forced over the ds_read_b128 lane groups, every operand read is bank-conflict
// free), and WAVE c multiplies 16-channel chunk c for all 64 pixels (lane = pixel), so the weights of a wave are uniform
// and reach the FMAs as scalar operands (s_load), as in conv1a.  The four partial sums of a pixel meet in LDS and are added in chunk order — one fixed order whatever the
// batch or grid size, so results do not depend on either.  Exact fp32 arithmetic in both precision modes.
// (First version: 16 x 16 tiles, a thread walked all chunks of its pixel: at one frame its 20 workgroups were 17-19 us
// serial chains, slower than the matrix-core kernel it replaced.)
#include "conv_common.h"

namespace kp2d {

namespace {
constexpr int HD_TY = 4, HD_TX = 16, HD_HY = HD_TY + 2, HD_HX = HD_TX + 2;      // tile and halo, rows x columns
constexpr int HD_SC = 64, HD_PITCH = HD_SC + 4, HD_ROW = 1280;                  // channels staged at a time; LDS pitches (floats)
constexpr int HD_G = HD_HY * HD_HX * (HD_SC / 4);     // float4 granules of a staged super-chunk (1728)
constexpr int HD_IT = (HD_G + 255) / 256;
}  // namespace

// a.w: [chunk][tap][4][16] floats (kp2d_api.cpp pack(): ConvPack::wd_off), a.scale / a.shift: [cout]
// LDS (dynamic): s_in [HD_HY * HD_ROW] floats (30,720 B), then s_part [4 * CO * 64]
template <int CO>
__device__ __forceinline__ void head3x3_body(const ConvArgs& a, float* const s_in) {
  float* const s_part = s_in + HD_HY * HD_ROW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware tile order (as the conv kernels): workgroups are dealt round-robin over the 8 XCDs, each with its own L2;
  // in launch order the tiles that share halo rows sat on eight different L2s and every one of them fetched the halo from
  // HBM — PMC read 127 MB per launch against 78.6 MB algorithmic = the 6 x 18 / 4 x 16 halo ratio (profiles/r3_traffic.json).
  // Every XCD now owns a contiguous run of tiles, so a tile's neighbours hit in its L2.
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int b = bid / a.tiles_y;
  const int y0 = ty * HD_TY, x0 = tx * HD_TX;
  const int H = a.H, W = a.W;
  const int nchunk = (a.cin + 15) >> 4;

  const float* src = a.in0.p + (size_t)b * a.in0.bs + a.in0.o;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(src), 0, (int)((a.in0.bs - a.in0.o) * 4), 0x00020000);
  constexpr int OOB = 0x7ffffff0;
  const int ps = (int)a.in0.ps * 4;
  const int q4 = 4 * (tid & 15);                    // first channel (within the super-chunk) of this thread's granules

  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
  const int ly = lane >> 4, lx = lane & 15;
  const float* __restrict__ wg = a.w;

  for (int sc = 0; sc * HD_SC < a.cin; ++sc) {
    if (sc) __syncthreads();
    // stage the halo of channels [64 sc, 64 sc + 64): granule = 4 channels of one halo pixel
    const bool cok = sc * HD_SC + q4 < a.cin;
#pragma unroll
    for (int it = 0; it < HD_IT; ++it) {
      const int g = tid + 256 * it, hp = g >> 4;
      if (it == HD_IT - 1 && g >= HD_G) continue;
      const int py = hp / HD_HX, px = hp - py * HD_HX;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      const bool ok = cok && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const int off = ok ? (gy * W + gx) * ps + (sc * HD_SC + q4) * 4 : OOB;
      *reinterpret_cast<float4*>(&s_in[py * HD_ROW + px * HD_PITCH + q4]) =
          __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
    __syncthreads();
    const int ch = sc * 4 + wave;                   // this wave's 16-channel chunk
    if (ch < nchunk) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float* sp = &s_in[(ly + tap / 3) * HD_ROW + (lx + tap % 3) * HD_PITCH + wave * 16];
        const float* wt = wg + ((size_t)ch * 9 + tap) * 4 * 16;      // wave-uniform: scalar loads
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 x = *reinterpret_cast<const float4*>(sp + 4 * q);
#pragma unroll
          for (int c = 0; c < CO; ++c) {
            acc[c] = fmaf(x.x, wt[c * 16 + 4 * q + 0], acc[c]);
            acc[c] = fmaf(x.y, wt[c * 16 + 4 * q + 1], acc[c]);
            acc[c] = fmaf(x.z, wt[c * 16 + 4 * q + 2], acc[c]);
            acc[c] = fmaf(x.w, wt[c * 16 + 4 * q + 3], acc[c]);
          }
        }
      }
    }
  }
  // the four chunk-lanes of a pixel, added in chunk order by wave 0
#pragma unroll
  for (int c = 0; c < CO; ++c) s_part[(wave * CO + c) * 64 + lane] = acc[c];
  __syncthreads();
  if (wave != 0) return;
  const int y = y0 + ly, x = x0 + lx;
  if (y < H && x < W) {
    const size_t plane = (size_t)H * W;
    const int ns = a.nsplit;
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      const float sum = ((s_part[(0 * CO + c) * 64 + lane] + s_part[(1 * CO + c) * 64 + lane]) + s_part[(2 * CO + c) * 64 + lane]) +
                        s_part[(3 * CO + c) * 64 + lane];
      float v = fmaf(sum, a.scale[c], a.shift[c]);
      if (a.act != ACT_NONE) v = act_apply(v, a.act, c);
      float* dst = (c < ns) ? a.out0 + ((size_t)b * ns + c) * plane
                            : a.out1 + ((size_t)b * (a.cout - ns) + (c - ns)) * plane;
      dst[(size_t)y * W + x] = v;
    }
  }
}

template <int CO>
__global__ __launch_bounds__(256) void head3x3_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float hd_smem[];
  head3x3_body<CO>(a, hd_smem);
}

// KP2DTinyV2's score head (64 -> 1, sigmoid) and location head (64 -> 2, tanh) on the same grid in ONE launch
// (blockIdx.y picks the head): at a single frame a launch costs >= 4.3 us whatever it computes.
__global__ __launch_bounds__(256) void head3x3_pair_kernel(const ConvArgs a0, const ConvArgs a1) {
  extern __shared__ __attribute__((aligned(16))) float hd_smem[];
  if (blockIdx.y == 0) head3x3_body<1>(a0, hd_smem);
  else head3x3_body<2>(a1, hd_smem);
}

static int head3x3_check(const ConvArgs& a0) {
  if (a0.taps != 9 || a0.cout < 1 || a0.cout > 4 || a0.store != ST_NCHW || a0.in1.c != 0 || a0.act == ACT_SOFTMAX_C) return -1000;
  if (a0.in0.rs != (long)a0.W * a0.in0.ps) return -1004;
  if ((long)a0.H * a0.W * a0.in0.ps * 4 >= 0x7ffffff0L) return -1002;
  return 0;
}
static size_t head3x3_lds(int co) { return (size_t)(HD_HY * HD_ROW + 4 * co * 64) * sizeof(float); }

int launch_head3x3(const ConvArgs& a0, hipStream_t s) {
  if (int e = head3x3_check(a0)) return e;
  ConvArgs a = a0;
  a.tiles_x = (a.W + HD_TX - 1) / HD_TX;
  a.tiles_y = (a.H + HD_TY - 1) / HD_TY;
  const dim3 grid(a.tiles_x * a.tiles_y * a.B);
  const size_t lds = head3x3_lds(a.cout);
  switch (a.cout) {
    case 1: hipLaunchKernelGGL((head3x3_kernel<1>), grid, dim3(256), lds, s, a); break;
    case 2: hipLaunchKernelGGL((head3x3_kernel<2>), grid, dim3(256), lds, s, a); break;
    case 3: hipLaunchKernelGGL((head3x3_kernel<3>), grid, dim3(256), lds, s, a); break;
    default: hipLaunchKernelGGL((head3x3_kernel<4>), grid, dim3(256), lds, s, a); break;
  }
  return (int)hipGetLastError();
}

// a0: one output channel, a1: two; same map size and batch
int launch_head3x3_pair(const ConvArgs& a0, const ConvArgs& a1, hipStream_t s) {
  if (int e = head3x3_check(a0)) return e;
  if (int e = head3x3_check(a1)) return e;
  if (a0.cout != 1 || a1.cout != 2 || a0.H != a1.H || a0.W != a1.W || a0.B != a1.B) return -1005;
  ConvArgs x = a0, y = a1;
  x.tiles_x = y.tiles_x = (x.W + HD_TX - 1) / HD_TX;
  x.tiles_y = y.tiles_y = (x.H + HD_TY - 1) / HD_TY;
  hipLaunchKernelGGL(head3x3_pair_kernel, dim3(x.tiles_x * x.tiles_y * x.B, 2), dim3(256), head3x3_lds(2), s, x, y);
  return (int)hipGetLastError();
}

}  // namespace kp2d
