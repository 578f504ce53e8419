// Post-processing and keypoint selection for the kp2dtiny path.
//
//  * post_kernel       KP2DTinyV2/V3.post_processing (models/kp2dtiny.py:593-625 / :959-993):
//                      border mask (:520-528), coordinate decode + clamp (:597-614, grid from
//                      utils/image.py:44-75), normalize_coord (:642-647), grid_sample(bilinear,
//                      align_corners=True, zeros) + division by the L2 norm without eps (:627-631).
//  * seg_argmax_kernel sample_seg (:633-640 / :1001-1008): argmax over classes, int64, H/2 x W/2.
//  * topk_kernel       the callers' selectors: score > thr, then the k best,
//                      ordered (score desc, flat index asc) — the order of torch.topk in
//                      gluefactory/models/extractors/kp2dtiny.py:40; K1/K2 use the same SET.
// Every float op that feeds an index decision is written with explicit round-to-nearest
// intrinsics so hipcc's default fp-contraction cannot fuse it differently from the reference.
#include <cstdlib>

#include "conv_common.h"
#include "device_guard.h"

namespace kp2d {

// Workgroup = 64 cells x 4 channel quarters: wave w samples channels [w*C/4, (w+1)*C/4) of the same 64 cells (each
// wave-load still walks one feature plane), the squared norms meet in LDS.  8 gathers x 4 corners per thread instead
// of 32 x 4 keeps four times as many loads in flight per cell.
template <int C>
__device__ __forceinline__ void post_body(const PostArgs& a, const int bx, const int b) {
  __shared__ float s_ss[4][64];
  constexpr int CW = C / 4;
  const int Hc = a.Hc, Wc = a.Wc;
  const int ncell = Hc * Wc;
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  // lanes past the last cell of a partial block recompute the last cell and store nothing: one control path, one barrier
  const bool live = bx * 64 + lane < ncell;
  const int cell_id = live ? bx * 64 + lane : ncell - 1;
  const int yc = cell_id / Wc, xc = cell_id - yc * Wc;
  const size_t so = (size_t)b * ncell + cell_id;

  const bool border = (yc == 0) | (yc == Hc - 1) | (xc == 0) | (xc == Wc - 1);
  if (part == 0 && live) a.score_out[so] = __fmul_rn(a.score_in[so], border ? 0.f : 1.f);

  const float step = (float)(a.cell - 1) * 0.5f;
  const float gain = a.cross_ratio * step;
  const float sx = a.shift[((size_t)b * 2 + 0) * ncell + cell_id];
  const float sy = a.shift[((size_t)b * 2 + 1) * ncell + cell_id];
  float cx = __fadd_rn(__fadd_rn(__fmul_rn((float)xc, (float)a.cell), step), __fmul_rn(sx, gain));
  float cy = __fadd_rn(__fadd_rn(__fmul_rn((float)yc, (float)a.cell), step), __fmul_rn(sy, gain));
  cx = fminf(fmaxf(cx, 0.f), (float)(a.W - 1));
  cy = fminf(fmaxf(cy, 0.f), (float)(a.H - 1));
  if (part == 0 && live) {
    a.coord[((size_t)b * 2 + 0) * ncell + cell_id] = cx;
    a.coord[((size_t)b * 2 + 1) * ncell + cell_id] = cy;
  }
  if (a.desc == nullptr) return;      // uniform: no barrier is reached on this path

  // normalize_coord with the IMAGE size, then grid_sample's un-normalisation with the FEATURE size
  const float gx = __fsub_rn(__fdiv_rn(cx, (float)(a.W - 1) * 0.5f), 1.f);
  const float gy = __fsub_rn(__fdiv_rn(cy, (float)(a.H - 1) * 0.5f), 1.f);
  const int Hf = a.Hf, Wf = a.Wf;
  const float ix = __fmul_rn(__fmul_rn(__fadd_rn(gx, 1.f), 0.5f), (float)(Wf - 1));
  const float iy = __fmul_rn(__fmul_rn(__fadd_rn(gy, 1.f), 0.5f), (float)(Hf - 1));
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = __fsub_rn(ix, fx0), wx0 = __fsub_rn(__fadd_rn(fx0, 1.f), ix);
  const float wy1 = __fsub_rn(iy, fy0), wy0 = __fsub_rn(__fadd_rn(fy0, 1.f), iy);
  const bool vx0 = x0 >= 0 && x0 < Wf, vx1 = x1 >= 0 && x1 < Wf;
  const bool vy0 = y0 >= 0 && y0 < Hf, vy1 = y1 >= 0 && y1 < Hf;
  const float w00 = (vx0 && vy0) ? __fmul_rn(wx0, wy0) : 0.f;
  const float w01 = (vx1 && vy0) ? __fmul_rn(wx1, wy0) : 0.f;
  const float w10 = (vx0 && vy1) ? __fmul_rn(wx0, wy1) : 0.f;
  const float w11 = (vx1 && vy1) ? __fmul_rn(wx1, wy1) : 0.f;
  const int cx0 = min(max(x0, 0), Wf - 1), cx1 = min(max(x1, 0), Wf - 1);
  const int cy0 = min(max(y0, 0), Hf - 1), cy1 = min(max(y1, 0), Hf - 1);
  const size_t plane = (size_t)Hf * Wf;
  const float* f = a.feat + ((size_t)b * C + (size_t)part * CW) * plane;
  const size_t o00 = (size_t)cy0 * Wf + cx0, o01 = (size_t)cy0 * Wf + cx1;
  const size_t o10 = (size_t)cy1 * Wf + cx0, o11 = (size_t)cy1 * Wf + cx1;
  float d[CW];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const float* fp = f + c * plane;
    float v = __fmul_rn(fp[o00], w00);
    v = __fadd_rn(v, __fmul_rn(fp[o01], w01));
    v = __fadd_rn(v, __fmul_rn(fp[o10], w10));
    v = __fadd_rn(v, __fmul_rn(fp[o11], w11));
    d[c] = v;
    ss = fmaf(v, v, ss);
  }
  s_ss[part][lane] = ss;
  __syncthreads();
  // channel-ascending summation order of the quarter sums (the per-quarter sums are themselves channel-ascending)
  const float nrm = sqrtf(((s_ss[0][lane] + s_ss[1][lane]) + s_ss[2][lane]) + s_ss[3][lane]);   // no eps: kp2dtiny.py:629-630
  if (live) {
#pragma unroll
    for (int c = 0; c < CW; ++c) a.desc[((size_t)b * C + part * CW + c) * ncell + cell_id] = __fdiv_rn(d[c], nrm);
  }
}

template <int C>
__global__ __launch_bounds__(256) void post_kernel(const PostArgs a) { post_body<C>(a, blockIdx.x, blockIdx.y); }

int launch_post(const PostArgs& a, hipStream_t s) {
  const int ncell = a.Hc * a.Wc;
  dim3 grid((ncell + 63) / 64, a.B);
  switch (a.C) {
    case 32: hipLaunchKernelGGL(post_kernel<32>, grid, dim3(256), 0, s, a); break;
    case 64: hipLaunchKernelGGL(post_kernel<64>, grid, dim3(256), 0, s, a); break;
    case 128: hipLaunchKernelGGL(post_kernel<128>, grid, dim3(256), 0, s, a); break;
    default: return -1200;
  }
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void seg_argmax_kernel(const ArgmaxArgs a) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (p >= (size_t)a.HW) return;
  const float* sp = a.seg + (size_t)b * a.C * a.HW + p;
  float best = sp[0];
  int bi = 0;
  for (int c = 1; c < a.C; ++c) {
    const float v = sp[(size_t)c * a.HW];
    if (v > best) { best = v; bi = c; }   // first maximum wins, as torch.argmax on CPU
  }
  a.ids[(size_t)b * a.HW + p] = (int64_t)bi;
}

// four consecutive pixels per thread (HW % 4 == 0): 16-byte loads per class plane, two 16-byte stores of ids
__device__ __forceinline__ void seg_argmax4_body(const ArgmaxArgs& a, const int bx, const int b) {
  const size_t q = (size_t)bx * 256 + threadIdx.x;       // quad of pixels
  if (q * 4 >= (size_t)a.HW) return;
  const float4* sp = reinterpret_cast<const float4*>(a.seg + (size_t)b * a.C * a.HW) + q;
  const size_t plane4 = (size_t)a.HW >> 2;
  float4 best = sp[0];
  int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
#pragma unroll 4
  for (int c = 1; c < a.C; ++c) {
    const float4 v = sp[(size_t)c * plane4];
    if (v.x > best.x) { best.x = v.x; b0 = c; }   // first maximum wins, as torch.argmax on CPU
    if (v.y > best.y) { best.y = v.y; b1 = c; }
    if (v.z > best.z) { best.z = v.z; b2 = c; }
    if (v.w > best.w) { best.w = v.w; b3 = c; }
  }
  longlong2* o = reinterpret_cast<longlong2*>(a.ids + (size_t)b * a.HW + q * 4);
  o[0] = make_longlong2(b0, b1);
  o[1] = make_longlong2(b2, b3);
}

__global__ __launch_bounds__(256) void seg_argmax4_kernel(const ArgmaxArgs a) { seg_argmax4_body(a, blockIdx.x, blockIdx.y); }

// post_processing's two independent halves in ONE launch: blocks [0, npost) are post_kernel's, the rest the dense class
// argmax.  At a single frame every launch costs >= 4.3 us whatever it does (profiles/r3_ab_deep_prefetch.txt).
template <int C>
__global__ __launch_bounds__(256) void post_seg_kernel(const PostArgs a, const ArgmaxArgs g, const int npost) {
  if ((int)blockIdx.x < npost) post_body<C>(a, blockIdx.x, blockIdx.y);
  else seg_argmax4_body(g, blockIdx.x - npost, blockIdx.y);
}

int launch_post_seg(const PostArgs& a, const ArgmaxArgs& g, hipStream_t s) {
  const bool quad = (g.HW & 3) == 0 && ((uintptr_t)g.seg & 15) == 0 && ((uintptr_t)g.ids & 15) == 0;
  if (!quad || g.B != a.B || (a.C != 32 && a.C != 64 && a.C != 128)) {
    if (int e = launch_post(a, s)) return e;
    return launch_seg_argmax(g, s);
  }
  const int npost = (a.Hc * a.Wc + 63) / 64, nseg = (g.HW / 4 + 255) / 256;
  const dim3 grid(npost + nseg, a.B);
  switch (a.C) {
    case 32: hipLaunchKernelGGL(post_seg_kernel<32>, grid, dim3(256), 0, s, a, g, npost); break;
    case 64: hipLaunchKernelGGL(post_seg_kernel<64>, grid, dim3(256), 0, s, a, g, npost); break;
    default: hipLaunchKernelGGL(post_seg_kernel<128>, grid, dim3(256), 0, s, a, g, npost); break;
  }
  return (int)hipGetLastError();
}

// sample_seg with sample_segmentation=True (models/kp2dtiny.py:634-639): nearest-neighbour grid_sample of the
// class map at every cell's predicted coordinate (align_corners=True, zeros outside), then argmax.
__global__ __launch_bounds__(256) void seg_sample_argmax_kernel(const SegSampleArgs a) {
  const int cell = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y, ncell = a.Hc * a.Wc;
  if (cell >= ncell) return;
  const float cx = a.coord[((size_t)b * 2 + 0) * ncell + cell], cy = a.coord[((size_t)b * 2 + 1) * ncell + cell];
  const float gx = __fsub_rn(__fdiv_rn(cx, (float)(a.W - 1) * 0.5f), 1.f);
  const float gy = __fsub_rn(__fdiv_rn(cy, (float)(a.H - 1) * 0.5f), 1.f);
  const float ix = __fmul_rn(__fmul_rn(__fadd_rn(gx, 1.f), 0.5f), (float)(a.Ws - 1));
  const float iy = __fmul_rn(__fmul_rn(__fadd_rn(gy, 1.f), 0.5f), (float)(a.Hs - 1));
  const int x = (int)nearbyintf(ix), y = (int)nearbyintf(iy);       // round half to even, as torch's nearest mode
  const bool ok = x >= 0 && x < a.Ws && y >= 0 && y < a.Hs;
  const size_t plane = (size_t)a.Hs * a.Ws;
  const float* sp = a.seg + (size_t)b * a.C * plane + (ok ? (size_t)y * a.Ws + x : 0);
  float best = ok ? sp[0] : 0.f;
  int bi = 0;
  for (int c = 1; c < a.C; ++c) {
    const float v = ok ? sp[c * plane] : 0.f;
    if (v > best) { best = v; bi = c; }
  }
  a.ids[(size_t)b * ncell + cell] = (int64_t)bi;
}

int launch_seg_sample_argmax(const SegSampleArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(seg_sample_argmax_kernel, dim3((a.Hc * a.Wc + 255) / 256, a.B), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

int launch_seg_argmax(const ArgmaxArgs& a, hipStream_t s) {
  if ((a.HW & 3) == 0 && ((uintptr_t)a.seg & 15) == 0 && ((uintptr_t)a.ids & 15) == 0)
    hipLaunchKernelGGL(seg_argmax4_kernel, dim3((a.HW / 4 + 255) / 256, a.B), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(seg_argmax_kernel, dim3((a.HW + 255) / 256, a.B), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// top-k: one workgroup per frame.  Keys are 64-bit (order-preserving score bits << 32 | ~index), so
// they are unique: an 8-pass MSB radix select finds the count-th largest key exactly, the survivors
// are compacted and sorted descending -> (score desc, index asc), deterministic.
//   k <= TOPK_LDS_MAX: survivors live in LDS (8 B per key, up to 128 KiB of the CU's 160) and are bitonic-sorted there
//                      (256 threads up to 4096 keys, 1024 beyond);
//   larger k ("no cap": every cell above the threshold, frontend.py:122 / visual_odometry.py:112 with top_k <= 0, or
//   a cap above 16384 on a big frame): survivors are compacted into the caller's idx row and sorted IN PLACE in
//   global memory, keys rebuilt from score[idx] — no scratch buffer in the ABI, any k up to n.
// ---------------------------------------------------------------------------------------------
constexpr int TOPK_SMALL_MAX = 256;      // 256-thread workgroups up to here; beyond, 1024 threads hold one key each in the sort
constexpr int TOPK_LDS_MAX = 16384;      // keys that fit the LDS path

__device__ __forceinline__ unsigned long long topk_key(float s, int idx, float thr) {
  if (!(s > thr)) return 0ull;
  unsigned u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)idx);
}

// Shared front half: count the candidates and, when there are more than k of them, find the k-th largest key.
// hist: LDS [260] (256 bins + rank, ncand, fill, done); s_prefix: LDS.  Returns the selection threshold
// (1: every candidate, ~0: nothing) and the number of selected keys through `count_out`.
// kreg / cached: the thread's keys (element tid + u NTHR) already in registers — the frame is then read from memory
// once, all loads in flight together, instead of once per pass in a loop whose round trips run one after another
constexpr int TOPK_KPT = 8;
template <int NTHR>
__device__ __forceinline__ unsigned long long topk_threshold(const float* sc, int n, int k, float thr, unsigned* hist,
                                                             unsigned long long& s_prefix, unsigned& count_out,
                                                             const unsigned long long (&kreg)[TOPK_KPT], bool cached) {
  const int tid = threadIdx.x;
  unsigned& s_rank = hist[256];
  unsigned& s_ncand = hist[257];
  unsigned& s_done = hist[259];
  if (tid == 0) s_ncand = 0;
  __syncthreads();
  unsigned local = 0;
  if (cached) {
#pragma unroll
    for (int u = 0; u < TOPK_KPT; ++u) local += kreg[u] != 0ull ? 1u : 0u;
  } else {
    for (int e = tid; e < n; e += NTHR) local += (sc[e] > thr) ? 1u : 0u;
  }
  atomicAdd(&s_ncand, local);
  __syncthreads();
  const unsigned ncand = s_ncand;
  const unsigned count = min((unsigned)k, ncand);
  count_out = count;
  if (ncand <= (unsigned)k) return 1ull;      // every candidate is selected (candidate keys are non-zero): no rank to find
  if (count == 0) return ~0ull;
  if (tid == 0) { s_prefix = 0ull; s_rank = count; s_done = 0u; }
  __syncthreads();
  for (int pass = 7; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned long long prefix = s_prefix;
    const int shift = pass * 8;
    const unsigned long long himask = (pass == 7) ? 0ull : (~0ull << (shift + 8));
    if (cached) {
#pragma unroll
      for (int u = 0; u < TOPK_KPT; ++u) {
        const unsigned long long key = kreg[u];
        if (key != 0ull && (key & himask) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
      }
    } else {
      for (int e = tid; e < n; e += NTHR) {
        const unsigned long long key = topk_key(sc[e], e, thr);
        if (key != 0ull && (key & himask) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
      }
    }
    __syncthreads();
    if (tid < 64) {
      // wave 0 locates the bin holding the rank-th largest key: lane l owns bins 4l..4l+3, a suffix sum over lanes
      // gives the number of keys in higher bins (the serial 256-step scan this replaces was ~10 % of the kernel)
      const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
      const unsigned mine = h0 + h1 + h2 + h3;
      unsigned above = mine;                       // inclusive suffix sum over lanes >= tid
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_down(above, o);
        if (tid + o < 64) above += v;
      }
      above -= mine;                               // keys in bins of higher lanes
      const unsigned rank = s_rank;
      if (above < rank && rank <= above + mine) {  // exactly one lane: the rank-th key is in one of its 4 bins
        unsigned cum = above;
        int bin = 4 * tid + 3;
        unsigned hb = h3;
        if (cum + h3 < rank) { cum += h3; bin = 4 * tid + 2; hb = h2;
          if (cum + h2 < rank) { cum += h2; bin = 4 * tid + 1; hb = h1;
            if (cum + h1 < rank) { cum += h1; bin = 4 * tid; hb = h0; } } }
        s_rank = rank - cum;
        s_prefix = prefix | ((unsigned long long)bin << shift);
        // every key of the chosen bin is selected: the remaining digits cannot change the set (keys are unique,
        // so this is reached at the latest when the bin holds one key) -> stop refining
        s_done = (rank - cum == hb) ? 1u : 0u;
      }
    }
    __syncthreads();
    if (s_done) break;
  }
  return s_prefix;   // the count-th largest key
}

template <int NTHR>
__global__ __launch_bounds__(NTHR) void topk_kernel(const TopkArgs a, const int kpow) {
  // all LDS in the dynamic region (16-byte aligned base): [kpow] keys | prefix | (pad) | hist[256] | 4 counters | [NTHR] keys
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_keys[];
  const int tid = threadIdx.x, b = blockIdx.x, n = a.n;
  const float* sc = a.score + (size_t)b * n;
  unsigned long long& s_prefix = s_keys[kpow];
  unsigned* hist = reinterpret_cast<unsigned*>(&s_keys[kpow + 2]);
  unsigned& s_fill = hist[258];
  unsigned long long* s_keys2 = &s_keys[kpow + 2 + 130];      // [NTHR], the register sort's second exchange buffer

  // kpow >= 2 always (launch_topk): the sort's first compare reads s_keys[0] and s_keys[1], both inside the key region
  for (int e = tid; e < kpow; e += NTHR) s_keys[e] = 0ull;
  const bool cached = n <= TOPK_KPT * NTHR;
  unsigned long long kreg[TOPK_KPT];
#pragma unroll
  for (int u = 0; u < TOPK_KPT; ++u) {
    const int e = tid + u * NTHR;
    kreg[u] = (cached && e < n) ? topk_key(sc[e], e, a.thr) : 0ull;
  }
  unsigned count;
  const unsigned long long thresh = topk_threshold<NTHR>(sc, n, a.k, a.thr, hist, s_prefix, count, kreg, cached);
  if (tid == 0) { a.count[b] = (int)count; s_fill = 0; }
  __syncthreads();
  if (cached) {
#pragma unroll
    for (int u = 0; u < TOPK_KPT; ++u) {
      const unsigned long long key = kreg[u];
      if (key != 0ull && key >= thresh) {
        const unsigned slot = atomicAdd(&s_fill, 1u);
        if (slot < (unsigned)kpow) s_keys[slot] = key;
      }
    }
  } else {
    for (int e = tid; e < n; e += NTHR) {
      const unsigned long long key = topk_key(sc[e], e, a.thr);
      if (key != 0ull && key >= thresh) {
        const unsigned slot = atomicAdd(&s_fill, 1u);
        if (slot < (unsigned)kpow) s_keys[slot] = key;
      }
    }
  }
  __syncthreads();
  // bitonic sort, descending, of the smallest power-of-two prefix that holds the selected keys (the rest are zeros)
  int ksort = 2;
  while (ksort < (int)count) ksort <<= 1;
  if (ksort <= NTHR) {
    // One key per thread, in a register.  Compare-exchanges whose partner is in the same wave (stride < 64: 45 of the
    // 55 steps at 1024 keys) are two lane shuffles; only the wider ones go through LDS, ping-ponging between the key
    // array and a second buffer so that each costs ONE barrier.  (A barrier per step, 1024 threads: 14 us of the
    // kernel's 25 at one frame.)
    unsigned long long* buf[2] = {s_keys, s_keys2};
    int pp = 0;
    unsigned long long x = tid < ksort ? s_keys[tid] : 0ull;
    for (int size = 2; size <= ksort; size <<= 1) {
      const bool desc = (tid & size) == 0;
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        unsigned long long y;
        if (stride >= 64) {
          if (tid < ksort) buf[pp][tid] = x;   // (its readers of two exchanges ago have all passed the last barrier)
          __syncthreads();
          y = tid < ksort ? buf[pp][tid ^ stride] : 0ull;
          pp ^= 1;
        } else {
          y = __shfl_xor(x, stride);
        }
        const bool want_max = ((tid & stride) == 0) == desc;
        x = want_max ? (x > y ? x : y) : (x < y ? x : y);
      }
    }
    __syncthreads();
    if (tid < ksort) s_keys[tid] = x;
    __syncthreads();
  } else {
    for (int size = 2; size <= ksort; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < (ksort >> 1); t += NTHR) {
          const int lo = 2 * t - (t & (stride - 1));
          const int hi = lo + stride;
          const bool desc = ((lo & size) == 0);
          const unsigned long long x = s_keys[lo], y = s_keys[hi];
          if ((x < y) == desc) { s_keys[lo] = y; s_keys[hi] = x; }
        }
        __syncthreads();
      }
    }
  }
  for (int e = tid; e < a.k; e += NTHR) {
    const bool ok = (unsigned)e < count;
    const unsigned long long key = ok ? s_keys[e] : 0ull;
    const int idx = ok ? (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)) : -1;
    a.idx[(size_t)b * a.k + e] = idx;
    if (a.val) a.val[(size_t)b * a.k + e] = ok ? sc[idx] : 0.f;
  }
}

// Large k: the selected cells are compacted into the frame's idx row (arrival order) and sorted there.  The network is
// the one-direction ("flip") form of the bitonic sort — every compare-exchange puts the larger key at the lower
// position — so positions past `count` behave as minimal sentinels without existing: a pair whose upper position is
// >= count is simply skipped, and any count (not only powers of two) sorts in place.
__global__ __launch_bounds__(1024) void topk_global_kernel(const TopkArgs a) {
  __shared__ unsigned long long s_prefix;
  __shared__ unsigned hist[260];
  constexpr int NTHR = 1024;
  const int tid = threadIdx.x, b = blockIdx.x, n = a.n;
  const float* sc = a.score + (size_t)b * n;
  int32_t* idx = a.idx + (size_t)b * a.k;
  unsigned& s_fill = hist[258];
  unsigned count;
  const unsigned long long no_keys[TOPK_KPT] = {};
  const unsigned long long thresh = topk_threshold<NTHR>(sc, n, a.k, a.thr, hist, s_prefix, count, no_keys, false);
  if (tid == 0) { a.count[b] = (int)count; s_fill = 0; }
  __syncthreads();
  for (int e = tid; e < n; e += NTHR) {
    const unsigned long long key = topk_key(sc[e], e, a.thr);
    if (key != 0ull && key >= thresh) {
      const unsigned slot = atomicAdd(&s_fill, 1u);
      if (slot < count) idx[slot] = e;
    }
  }
  __syncthreads();       // global stores of this workgroup are visible to its other waves after the barrier
  unsigned npow = 2;
  while (npow < count) npow <<= 1;
  for (unsigned size = 2; size <= npow; size <<= 1) {
    // flip step: position i of the lower half of each block meets its mirror in the upper half
    for (unsigned t = tid; t < (npow >> 1); t += NTHR) {
      const unsigned blk = t / (size >> 1), off = t % (size >> 1);
      const unsigned lo = blk * size + off, hi = blk * size + size - 1 - off;
      if (hi < count) {
        const int il = idx[lo], ih = idx[hi];
        if (topk_key(sc[il], il, -INFINITY) < topk_key(sc[ih], ih, -INFINITY)) { idx[lo] = ih; idx[hi] = il; }
      }
    }
    __syncthreads();
    for (unsigned stride = size >> 2; stride > 0; stride >>= 1) {
      for (unsigned t = tid; t < (npow >> 1); t += NTHR) {
        const unsigned lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        if (hi < count) {
          const int il = idx[lo], ih = idx[hi];
          if (topk_key(sc[il], il, -INFINITY) < topk_key(sc[ih], ih, -INFINITY)) { idx[lo] = ih; idx[hi] = il; }
        }
      }
      __syncthreads();
    }
  }
  for (int e = tid; e < a.k; e += NTHR) {
    const bool ok = (unsigned)e < count;
    if (!ok) idx[e] = -1;
    if (a.val) a.val[(size_t)b * a.k + e] = ok ? sc[idx[e]] : 0.f;
  }
}

int launch_topk(const TopkArgs& a, hipStream_t s) {
  if (a.k < 1 || a.n < 1) return -1300;
  const int keff = a.k < a.n ? a.k : a.n;      // more than n keys can never be selected
  if (keff > TOPK_LDS_MAX) {
    hipLaunchKernelGGL(topk_global_kernel, dim3(a.B), dim3(1024), 0, s, a);
    return (int)hipGetLastError();
  }
  int kpow = 2;                                 // >= 2: the sort network always touches two key slots
  while (kpow < keff) kpow <<= 1;
  const size_t lds = (size_t)kpow * 8 + 16 + 260 * 4 + 1024 * 8;
  static const int small_max = getenv("KP2D_TOPK_SMALL") ? atoi(getenv("KP2D_TOPK_SMALL")) : TOPK_SMALL_MAX;
  if (keff <= small_max) {
    hipLaunchKernelGGL(topk_kernel<256>, dim3(a.B), dim3(256), lds, s, a, kpow);
  } else {
    static PerDeviceOnce lds_once;      // per device: a handle may live on any visible device
    if (int e = lds_opt_in(lds_once, reinterpret_cast<const void*>(&topk_kernel<1024>))) return e;
    hipLaunchKernelGGL(topk_kernel<1024>, dim3(a.B), dim3(1024), lds, s, a, kpow);
  }
  return (int)hipGetLastError();
}

// thread = (channel c, four keypoints j, j + K4, j + 2 K4, j + 3 K4 of a frame; K4 = ceil(k / 4)): consecutive lanes
// are consecutive channels of one keypoint (coalesced stores), the four gathers of a thread are in flight together.
// (One wave per keypoint — 64 000 waves of one 4-byte load each at 64 frames x 1000 keypoints — took 30 us for 9 MB.)
__global__ __launch_bounds__(256) void gather_kernel(const GatherArgs a) {
  const int b = blockIdx.y, C = a.C, k = a.k;
  const int K4 = (k + 3) >> 2;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= K4 * C) return;
  const int j0 = e / C, c = e - j0 * C;
  const int32_t* ip = a.idx + (size_t)b * k;
  const float* dp = a.desc + ((size_t)b * C + c) * a.n;
  int idx[4];
  float v[4], p[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = j0 + u * K4;
    idx[u] = j < k ? ip[j] : -1;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    v[u] = idx[u] >= 0 ? dp[idx[u]] : 0.f;
    p[u] = (c < 2 && idx[u] >= 0) ? a.coord[((size_t)b * 2 + c) * a.n + idx[u]] : 0.f;      // lanes c = 0, 1 also carry x, y
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = j0 + u * K4;
    if (j < k) {
      a.dsel[((size_t)b * k + j) * C + c] = v[u];
      if (c < 2) a.pts[((size_t)b * k + j) * 2 + c] = p[u];
    }
  }
}

// Dense selections (k a sizeable part of the n cells: the top-1000 of a 30 x 40 grid): every (keypoint, channel) gather
// above touches its own 128-byte line of the planar descriptor map — 32 lines per keypoint, 256 MB of L2 traffic for 8 MB
// of results at 64 frames x 1000 keypoints (29 us).  Here a workgroup copies CPG whole channel planes of its frame into
// LDS with coalesced reads (the map is read once: 154 KB per frame) and gathers from there.
template <int CPG>
__global__ __launch_bounds__(512) void gather_lds_kernel(const GatherArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_d[];      // [CPG][n]
  const int b = blockIdx.y, g = blockIdx.x, C = a.C, k = a.k, n = a.n, tid = threadIdx.x;
  const float* dp = a.desc + ((size_t)b * C + (size_t)g * CPG) * n;
  const int tot = CPG * n;
  if ((n & 3) == 0) {
    for (int e = tid; e < tot >> 2; e += 512) reinterpret_cast<float4*>(s_d)[e] = reinterpret_cast<const float4*>(dp)[e];
  } else {
    for (int e = tid; e < tot; e += 512) s_d[e] = dp[e];
  }
  __syncthreads();
  const int32_t* ip = a.idx + (size_t)b * k;
  for (int j = tid; j < k; j += 512) {
    const int idx = ip[j];
    float v[CPG];
#pragma unroll
    for (int c = 0; c < CPG; ++c) v[c] = idx >= 0 ? s_d[c * n + idx] : 0.f;
    float* o = a.dsel + ((size_t)b * k + j) * C + g * CPG;
    if (CPG % 4 == 0 && (C & 3) == 0) {
#pragma unroll
      for (int c = 0; c < CPG; c += 4) *reinterpret_cast<float4*>(o + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
    } else {
#pragma unroll
      for (int c = 0; c < CPG; ++c) o[c] = v[c];
    }
    if (g == 0) {
      a.pts[((size_t)b * k + j) * 2] = idx >= 0 ? a.coord[((size_t)b * 2) * n + idx] : 0.f;
      a.pts[((size_t)b * k + j) * 2 + 1] = idx >= 0 ? a.coord[((size_t)b * 2 + 1) * n + idx] : 0.f;
    }
  }
}

int launch_gather(const GatherArgs& a, hipStream_t s) {
  if (a.C < 2) return -1310;      // (x, y ride on channels 0 and 1)
  static const bool lds_on = !(getenv("KP2D_GATHER_LDS") && getenv("KP2D_GATHER_LDS")[0] == '0');
  // LDS form: selections of at least an eighth of the cells, planes that fit 64 KB in groups of 8 / 4 / 2 channels, and
  // enough frames to fill the chip with (frame, channel group) workgroups (one frame: 0.268 vs 0.265 ms with the direct form)
  if (lds_on && (long)a.k * 8 >= a.n && (a.C & 7) == 0 && (long)a.B * (a.C / 8) >= 128) {
    const dim3 block(512);
    if ((long)a.n * 8 * 4 <= 65536) {
      hipLaunchKernelGGL(gather_lds_kernel<8>, dim3(a.C / 8, a.B), block, (size_t)a.n * 8 * 4, s, a);
      return (int)hipGetLastError();
    }
    // Four planes per workgroup keep the 16-byte stores of a keypoint's row; past 64 KB that takes the opt-in to the
    // CU's whole LDS (4800 cells at 240 x 320: 76.8 KB, two workgroups per CU).  Two planes per workgroup wrote 8 bytes
    // per keypoint from sixteen workgroups per frame: 30 us at 64 frames.
    if ((long)a.n * 4 * 4 <= 80 * 1024) {
      static PerDeviceOnce lds_once;
      if (int e = lds_opt_in(lds_once, reinterpret_cast<const void*>(&gather_lds_kernel<4>))) return e;
      hipLaunchKernelGGL(gather_lds_kernel<4>, dim3(a.C / 4, a.B), block, (size_t)a.n * 4 * 4, s, a);
      return (int)hipGetLastError();
    }
    if ((long)a.n * 2 * 4 <= 65536) {
      hipLaunchKernelGGL(gather_lds_kernel<2>, dim3(a.C / 2, a.B), block, (size_t)a.n * 2 * 4, s, a);
      return (int)hipGetLastError();
    }
  }
  const int K4 = (a.k + 3) >> 2;
  hipLaunchKernelGGL(gather_kernel, dim3((K4 * a.C + 255) / 256, a.B), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

// Frame front-end (src/evaluation/visual_odometry.py:77-87): uint8 HWC frame -> /255 -> bilinear resize
// (kornia.geometry.transform.resize = F.interpolate(bilinear, align_corners=False, no antialias)) -> (v - 0.5) * 2,
// written as the planar [B,3,H,W] tensor the network reads.  One pass, no intermediate float image.
__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char* src, float* dst, int Hs, int Ws, int H, int W) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (p >= H * W) return;
  const int y = p / W, x = p - y * W;
  const unsigned char* img = src + (size_t)b * Hs * Ws * 3;
  float v[3];
  frame_pixel(img, Hs, Ws, H, W, y, x, v);             // conv_common.h: /255, bilinear resize, .sub(0.5).mul(2)
#pragma unroll
  for (int c = 0; c < 3; ++c) dst[((size_t)b * 3 + c) * H * W + p] = v[c];
}

int launch_preprocess(const unsigned char* src, float* dst, int B, int Hs, int Ws, int H, int W, hipStream_t s) {
  hipLaunchKernelGGL(preprocess_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, s, src, dst, Hs, Ws, H, W);
  return (int)hipGetLastError();
}

// NHWC (with channel stride / offset) -> planar NCHW; used for API-facing copies of internal tensors.
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* in, float* out, int C, int HW, int istride,
                                                           int ioff) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int p = p0 + r, c = c0 + tx;
    tile[r][tx] = (p < HW && c < C) ? in[((size_t)b * HW + p) * istride + ioff + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, p = p0 + tx;
    if (c < C && p < HW) out[((size_t)b * C + c) * HW + p] = tile[tx][r];
  }
}

// L2Norm over channels (modules/base.py:5-11: F.normalize(p=2, dim=1), eps 1e-12), in place on an NHWC map.
// One wave per pixel, lane = channels lane, lane+64, ... (C <= 256).
__global__ __launch_bounds__(256) void l2norm_channels_kernel(float* x, long npix, int C) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwave = (long)gridDim.x * 4;
  for (long p = wave; p < npix; p += nwave) {
    float v[4];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = lane + 64 * i < C ? x[p * C + lane + 64 * i] : 0.f;
      ss = fmaf(v[i], v[i], ss);
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (lane + 64 * i < C) x[p * C + lane + 64 * i] = v[i] * inv;
  }
}

int launch_l2norm_channels(float* x, long npix, int C, hipStream_t s) {
  if (C < 1 || C > 256) return -1700;
  long blocks = (npix + 3) / 4;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(l2norm_channels_kernel, dim3((int)blocks), dim3(256), 0, s, x, npix, C);
  return (int)hipGetLastError();
}

int launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, int istride, int ioff, hipStream_t s) {
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((HW + 63) / 64, (C + 63) / 64, B), dim3(256), 0, s, in, out, C, HW,
                     istride, ioff);
  return (int)hipGetLastError();
}

}  // namespace kp2d
