// 3x3 / stride 1 / pad 1 convolution with one to four output channels, written straight to the API's planar outputs:
// score_head.convDb (64 -> 1, sigmoid), loc_head.convDb (64 -> 2, tanh), V3's score_loc_head.convDb (64 -> 3) and the
// depth outputs (heads.py:22,28-35; kp2dtiny.py:574-575, :927-935, :956).
//
// On the matrix cores these layers compute a 32-channel tile for 1-3 real channels (9.6 / 18.6 TFLOP/s,
// profiles/r2_layers.txt).  They are HBM-bound dot products: 4 * Cin bytes read per pixel for 18 * Cin * Cout FLOP.
// Here a workgroup stages the 18 x 18 halo tile of a 16 x 16 pixel tile in 16-channel chunks (fp32, 80-byte pixel
// pitch), a thread owns one pixel, and the weights reach the FMAs as scalar operands (uniform index -> s_load), as
// in conv1a.  Exact fp32 arithmetic in both precision modes.
#include "conv_common.h"

namespace kp2d {

namespace {
constexpr int HD_KC = 16, HD_HP = 18, HD_PITCH = HD_KC + 4;
constexpr int HD_G = HD_HP * HD_HP * (HD_KC / 4);          // float4 granules of a chunk's halo tile
constexpr int HD_IT = (HD_G + 255) / 256;
}  // namespace

// a.w: [chunk][tap][4][16] floats (kp2d_api.cpp pack(): ConvPack::wd_off), a.scale / a.shift: [cout]
template <int CO>
__global__ __launch_bounds__(256) void head3x3_kernel(const ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float s_in[HD_HP * HD_HP * HD_PITCH];
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int b = bid / a.tiles_y;
  const int y0 = ty * TILE, x0 = tx * TILE;
  const int H = a.H, W = a.W;
  const int nchunk = (a.cin + HD_KC - 1) / HD_KC;

  const float* src = a.in0.p + (size_t)b * a.in0.bs + a.in0.o;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(src), 0, (int)((a.in0.bs - a.in0.o) * 4), 0x00020000);
  constexpr int OOB = 0x7ffffff0;
  const int ps = (int)a.in0.ps * 4;
  const int q4 = 4 * (tid & 3);
  int st_off[HD_IT];          // byte offset of this thread's granule in the source (chunk 0), OOB for the zero padding
#pragma unroll
  for (int it = 0; it < HD_IT; ++it) {
    const int g = tid + 256 * it, hp = g >> 2;
    const int py = hp / HD_HP, px = hp - py * HD_HP;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool ok = g < HD_G && gy >= 0 && gy < H && gx >= 0 && gx < W;
    st_off[it] = ok ? (gy * W + gx) * ps + q4 * 4 : OOB;
  }
  float4 r[HD_IT];
  auto prefetch = [&](int ch) {
    const bool cok = ch * HD_KC + q4 < a.cin;       // channel tail of a last, partial chunk
#pragma unroll
    for (int it = 0; it < HD_IT; ++it) {
      const int off = (cok && st_off[it] != OOB) ? st_off[it] + ch * HD_KC * 4 : OOB;
      r[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
  };

  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
  const int ly = tid >> 4, lx = tid & 15;
  const float* __restrict__ wg = a.w;

  prefetch(0);
  for (int ch = 0; ch < nchunk; ++ch) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < HD_IT; ++it) {
      const int g = tid + 256 * it;
      if (it == HD_IT - 1 && g >= HD_G) continue;
      *reinterpret_cast<float4*>(&s_in[(g >> 2) * HD_PITCH + q4]) = r[it];
    }
    __syncthreads();
    if (ch + 1 < nchunk) prefetch(ch + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float* sp = &s_in[((ly + tap / 3) * HD_HP + lx + tap % 3) * HD_PITCH];
      const float* wt = wg + ((size_t)ch * 9 + tap) * 4 * HD_KC;      // uniform: scalar loads
#pragma unroll
      for (int q = 0; q < HD_KC / 4; ++q) {
        const float4 x = *reinterpret_cast<const float4*>(sp + 4 * q);
#pragma unroll
        for (int c = 0; c < CO; ++c) {
          acc[c] = fmaf(x.x, wt[c * HD_KC + 4 * q + 0], acc[c]);
          acc[c] = fmaf(x.y, wt[c * HD_KC + 4 * q + 1], acc[c]);
          acc[c] = fmaf(x.z, wt[c * HD_KC + 4 * q + 2], acc[c]);
          acc[c] = fmaf(x.w, wt[c * HD_KC + 4 * q + 3], acc[c]);
        }
      }
    }
  }

  const int y = y0 + ly, x = x0 + lx;
  if (y < H && x < W) {
    const size_t plane = (size_t)H * W;
    const int ns = a.nsplit;
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      float v = fmaf(acc[c], a.scale[c], a.shift[c]);
      if (a.act == ACT_SOFTMAX_C) continue;   // (not a head activation; launch_head3x3 refuses it)
      if (a.act != ACT_NONE) v = act_apply(v, a.act, c);
      float* dst = (c < ns) ? a.out0 + ((size_t)b * ns + c) * plane
                            : a.out1 + ((size_t)b * (a.cout - ns) + (c - ns)) * plane;
      dst[(size_t)y * W + x] = v;
    }
  }
}

int launch_head3x3(const ConvArgs& a, hipStream_t s) {
  if (a.taps != 9 || a.cout < 1 || a.cout > 4 || a.store != ST_NCHW || a.in1.c != 0 || a.act == ACT_SOFTMAX_C) return -1000;
  if (a.in0.rs != (long)a.W * a.in0.ps) return -1004;
  if ((long)a.H * a.W * a.in0.ps * 4 >= 0x7ffffff0L) return -1002;
  const dim3 grid(a.tiles_x * a.tiles_y * a.B);
  switch (a.cout) {
    case 1: hipLaunchKernelGGL((head3x3_kernel<1>), grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((head3x3_kernel<2>), grid, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((head3x3_kernel<3>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((head3x3_kernel<4>), grid, dim3(256), 0, s, a); break;
  }
  return (int)hipGetLastError();
}

}  // namespace kp2d
