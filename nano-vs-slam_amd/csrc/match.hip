// Brute-force descriptor matching on the device — the step right after the kp2dtiny path in the VO pipeline
// (SURVEY.md §8f rank 1).  Replaces BfFeatureMatcher.match = cv2.BFMatcher(NORM_L2).knnMatch(des1, des2, k=2)
// followed by goodMatchesOneToOne (src/visual_odometry/feature_matcher.py:89-98, :179-209):
//   knn2_kernel     for every query descriptor the nearest and second-nearest train descriptor,
//                   distance = sqrt(sum (a-b)^2) in fp32 (OpenCV's L2 norm), lowest index wins ties
//   assign_kernel   ratio test  d1 <= ratio * d2, then one-to-one: each train index keeps the query with the
//                   smallest distance (first query wins ties) via a 64-bit atomicMin on (distance bits, query)
// Batched over B frame pairs with per-pair descriptor counts, so the per-frame D2H copy + CPU matcher of the
// reference disappears.  HBM-light (descriptors are tiny); bound by VALU: n0*n1*C*3 ops per pair.
#include "kp2d_kernels.h"

namespace kp2d {

constexpr int MQ = 64;    // queries per workgroup (4 threads each)
constexpr int MT = 128;   // train descriptors per LDS tile

template <int C>
__global__ __launch_bounds__(256) void knn2_kernel(const MatchArgs a) {
  __shared__ __attribute__((aligned(16))) float s_t[MT * C];
  const int b = blockIdx.y;
  const int n0 = a.n0[b], n1 = a.n1[b];
  const int q = blockIdx.x * MQ + (threadIdx.x >> 2);
  const int part = threadIdx.x & 3;
  if (blockIdx.x * MQ >= n0) return;     // whole workgroup out of range (uniform)
  float qv[C];
  const float* qp = a.d0 + ((size_t)b * a.max0 + (q < n0 ? q : 0)) * C;
#pragma unroll
  for (int c = 0; c < C; ++c) qv[c] = qp[c];
  float best = INFINITY, second = INFINITY;
  int bi = -1;
  for (int t0 = 0; t0 < n1; t0 += MT) {
    const int nt = min(MT, n1 - t0);
    __syncthreads();
    for (int e = threadIdx.x; e < MT * C / 4; e += 256) {
      const int row = e / (C / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nt) v = reinterpret_cast<const float4*>(a.d1 + ((size_t)b * a.max1 + t0) * C)[e];
      reinterpret_cast<float4*>(s_t)[e] = v;
    }
    __syncthreads();
    // this thread scans train rows part, part+4, ... (ascending index inside a thread)
    for (int r = part; r < nt; r += 4) {
      const float* tp = &s_t[r * C];
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) { const float d = qv[c] - tp[c]; acc = fmaf(d, d, acc); }
      const float dist = sqrtf(acc);
      if (dist < best) { second = best; best = dist; bi = t0 + r; }
      else if (dist < second) second = dist;
    }
  }
  // merge the 4 partial (best, second) lists of a query; ties -> lowest train index
#pragma unroll
  for (int o = 1; o < 4; o <<= 1) {
    const float ob = __shfl_xor(best, o), os = __shfl_xor(second, o);
    const int oi = __shfl_xor(bi, o);
    const bool take = (ob < best) || (ob == best && oi >= 0 && (bi < 0 || oi < bi));
    const float nb = take ? ob : best;
    const float ns = take ? fminf(best, os) : fminf(second, ob);
    bi = take ? oi : bi;
    best = nb; second = ns;
  }
  if (part == 0 && q < n0) {
    a.nn_idx[(size_t)b * a.max0 + q] = bi;
    a.nn_dist[(size_t)b * a.max0 + q] = best;
    a.nn_dist2[(size_t)b * a.max0 + q] = second;
  }
}

__global__ __launch_bounds__(256) void match_init_kernel(unsigned long long* best, long n) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < n) best[e] = ~0ull;
}

__global__ __launch_bounds__(256) void match_assign_kernel(const MatchArgs a) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= a.n0[b]) return;
  const size_t o = (size_t)b * a.max0 + q;
  const int t = a.nn_idx[o];
  const float d1 = a.nn_dist[o], d2 = a.nn_dist2[o];
  if (t < 0 || d1 > a.ratio * d2) return;          // feature_matcher.py:190 (needs a second neighbour: d2 = inf passes)
  const unsigned long long key = ((unsigned long long)__float_as_uint(d1) << 32) | (unsigned)q;
  atomicMin(&a.train_best[(size_t)b * a.max1 + t], key);
}

__global__ __launch_bounds__(256) void match_emit_kernel(const MatchArgs a) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.max1) return;
  const size_t o = (size_t)b * a.max1 + t;
  const unsigned long long key = t < a.n1[b] ? a.train_best[o] : ~0ull;
  const bool ok = key != ~0ull;
  a.match_q[o] = ok ? (int)(unsigned)(key & 0xffffffffull) : -1;
  a.match_d[o] = ok ? __uint_as_float((unsigned)(key >> 32)) : 0.f;
}

int launch_match(const MatchArgs& a, hipStream_t s) {
  if (a.C != 32 && a.C != 64) return -1500;
  dim3 g0((a.max0 + MQ - 1) / MQ, a.B);
  if (a.C == 32) hipLaunchKernelGGL(knn2_kernel<32>, g0, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(knn2_kernel<64>, g0, dim3(256), 0, s, a);
  const long n = (long)a.B * a.max1;
  hipLaunchKernelGGL(match_init_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, s, a.train_best, n);
  hipLaunchKernelGGL(match_assign_kernel, dim3((a.max0 + 255) / 256, a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(match_emit_kernel, dim3((a.max1 + 255) / 256, a.B), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace kp2d
