// Brute-force descriptor matching on the device — the step right after the kp2dtiny path in the VO pipeline
// (SURVEY.md §8f rank 1).  Replaces
//   * BfFeatureMatcher.match = cv2.BFMatcher(NORM_L2).knnMatch(des1, des2, k=2) followed by goodMatchesOneToOne
//     (src/visual_odometry/feature_matcher.py:89-98, :179-209) — the VO loop's matcher (visual_odometry.py:270-284);
//   * the same per semantic class (visual_odometry.py:347-380 match_semantic: one BF match per class id) as ONE launch
//     with per-row class ids: a query only sees train rows of its own class;
//   * cv2.BFMatcher(NORM_L2, crossCheck=False).match = plain nearest neighbour (src/evaluation/descriptor.py:132-134:
//     nn_idx) and crossCheck=True = mutual nearest neighbours (descriptor.py:221-222).
// Kernels:
//   knn2_kernel     for every query descriptor the nearest and second-nearest train descriptor,
//                   distance = sqrt(sum (a-b)^2) in fp32 (OpenCV's L2 norm), lowest index wins ties
//   assign / emit   ratio test  d1 <= ratio * d2, then one-to-one: each train index keeps the query with the
//                   smallest distance (first query wins ties) via a 64-bit atomicMin on (distance bits, query)
//   mutual_kernel   train row t keeps query q = its nearest query iff q's nearest train row is t
//   pairs_kernel    compacts the matched rows of a pair into (x0, y0, x1, y1) / (q, t) / distance lists (train order)
// Batched over B frame pairs with per-pair descriptor counts, so the per-frame D2H copy + CPU matcher of the
// reference disappears.  HBM-light (descriptors are tiny); bound by VALU: n0*n1*C*3 ops per pair.
#include <cstdlib>

#include "kp2d_kernels.h"

namespace kp2d {

constexpr int MQ = 64;    // queries per workgroup (4 threads each)
// train descriptors per LDS tile; rows are padded by 4 floats: the four rows a wave reads together (one per `part`) then
// sit on different banks (unpadded 128- / 512-byte rows: 2- / 4-way conflicts, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
// 0.49 at C = 32 and 0.74 at C = 128, profiles/r4_match_pmc_summary.txt)
__host__ __device__ constexpr int mt_rows(int C) { return C >= 128 ? 64 : 128; }

// d0 / n0 / c0: queries, d1 / n1 / c1: train rows (the reverse pass of the mutual mode swaps them).
// MASKED: rows carry class ids; a (query, train) pair of different classes does not exist for the search.
// Train rows of a query are split over gridDim.z slices (single pairs of thousands of rows would otherwise be a
// handful of long workgroups on a 256-CU chip); slice z writes its (best, second, index) to partial arrays that
// knn2_merge_kernel combines — for gridDim.z == 1 the results go straight out.
template <int C, bool MASKED>
__global__ __launch_bounds__(256) void knn2_kernel(const float* __restrict__ d0, const float* __restrict__ d1,
                                                   const int32_t* __restrict__ n0p, const int32_t* __restrict__ n1p,
                                                   const int32_t* __restrict__ c0, const int32_t* __restrict__ c1,
                                                   int max0, int max1, int32_t* __restrict__ nn_idx,
                                                   float* __restrict__ nn_dist, float* __restrict__ nn_dist2,
                                                   unsigned long long* __restrict__ init_best, long init_n) {
  constexpr int MT = mt_rows(C), CP = C + 4;
  __shared__ __attribute__((aligned(16))) float s_t[MT * CP];
  __shared__ int s_c[MT];
  // (the one-to-one table of the assign step that follows this launch starts at "no query": saves a launch per call)
  if (init_best) {
    const long wg = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (long e = wg * 256 + threadIdx.x; e < init_n; e += (long)gridDim.x * gridDim.y * gridDim.z * 256) init_best[e] = ~0ull;
  }
  const int b = blockIdx.y;
  const int n0 = n0p[b], n1 = n1p[b];
  const int q = blockIdx.x * MQ + (threadIdx.x >> 2);
  const int part = threadIdx.x & 3;
  if (blockIdx.x * MQ >= n0) return;     // whole workgroup out of range (uniform)
  const int nz = gridDim.z, z = blockIdx.z;
  // this slice's train rows [t_lo, t_hi): whole LDS tiles per slice
  const int tiles = (n1 + MT - 1) / MT, per = (tiles + nz - 1) / nz;
  const int t_lo = min(n1, z * per * MT), t_hi = min(n1, (z + 1) * per * MT);
  float qv[C];
  const float* qp = d0 + ((size_t)b * max0 + (q < n0 ? q : 0)) * C;
#pragma unroll
  for (int c = 0; c < C; c += 4) {
    const float4 v = *reinterpret_cast<const float4*>(qp + c);
    qv[c] = v.x; qv[c + 1] = v.y; qv[c + 2] = v.z; qv[c + 3] = v.w;
  }
  const int qc = MASKED ? c0[(size_t)b * max0 + (q < n0 ? q : 0)] : 0;
  float best = INFINITY, second = INFINITY;
  int bi = -1;
  for (int t0 = t_lo; t0 < t_hi; t0 += MT) {
    const int nt = min(MT, t_hi - t0);
    __syncthreads();
    for (int e = threadIdx.x; e < MT * C / 4; e += 256) {
      const int row = e / (C / 4), c4 = e - row * (C / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nt) v = reinterpret_cast<const float4*>(d1 + ((size_t)b * max1 + t0) * C)[e];
      *reinterpret_cast<float4*>(&s_t[row * CP + 4 * c4]) = v;
    }
    if (MASKED && threadIdx.x < MT) s_c[threadIdx.x] = threadIdx.x < nt ? c1[(size_t)b * max1 + t0 + threadIdx.x] : -1;
    __syncthreads();
    // this thread scans train rows part, part+4, ... (ascending index inside a thread)
    for (int r = part; r < nt; r += 4) {
      const float* tp = &s_t[r * CP];
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) { const float d = qv[c] - tp[c]; acc = fmaf(d, d, acc); }
      float dist = sqrtf(acc);
      if (MASKED && s_c[r] != qc) dist = INFINITY;      // (inf < anything is false: the row is never a neighbour)
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
    const size_t o = ((size_t)z * gridDim.y + b) * max0 + q;      // z = 0 of a one-slice launch: the outputs themselves
    nn_idx[o] = bi;
    nn_dist[o] = best;
    if (nn_dist2) nn_dist2[o] = second;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same search with the distances on the matrix cores (round 4).  The VALU kernel above spends ~70 instructions per
// (query, train) pair; here a 32 x 32 block of dot products is 3 C / 16 MFMAs (split-fp16 operands: q = qh + ql, t = th + tl,
// q.t = qh.th + qh.tl + ql.th on v_mfma_f32_32x32x16_f16, fp32 accumulation: error ~1e-7 |q||t|) and a pair costs nine VALU
// instructions (key = |t|^2 - 2 q.t, then a running best / second WITH indices).  The approximate keys only RANK: every
// query's final answer comes from an exact pass — the candidates (best and second of each of the query's two lanes, four
// per query) are re-evaluated with the arithmetic of the VALU kernel, sqrt(sum_c fma(d, d, .)) in channel order, and the
// nearest / second-nearest are taken by (exact distance, lower index).  So nn_dist / nn_dist2 are bit-identical to the
// VALU kernel's, and nn_idx differs from it only if three train rows of one lane's half lie within ~1e-6 of each other
// without being identical (identical rows have identical keys and keep their index order).
// Precondition of the ranking, checked in the kernel: every row's norm in [0.5, 2^15].  A workgroup that sees a row outside
// it (descriptors scaled by 1e5 or 1e-6, an all-zero row) falls back to an exact scan of its slice — any input gets the
// VALU kernel's answer, the matrix cores only ever make the in-range case fast.
// A workgroup = four waves x 32 queries against the same LDS tiles of train rows.
// ---------------------------------------------------------------------------------------------------------------------
typedef _Float16 mh8 __attribute__((ext_vector_type(8)));
typedef _Float16 mh4 __attribute__((ext_vector_type(4)));
typedef _Float16 mh2 __attribute__((ext_vector_type(2)));
typedef float mf16 __attribute__((ext_vector_type(16)));
typedef float mf2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void m_split4(const float4 v, mh4& hi, mh4& lo) {      // (as netvlad.hip vlad_split4)
  const mf2 a = {v.x, v.y}, b = {v.z, v.w};
  const mh2 ha = __builtin_convertvector(a, mh2), hb = __builtin_convertvector(b, mh2);
  unsigned la, lb;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(ha), "v"(v.x));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(la) : "v"(ha), "v"(v.y));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hb), "v"(v.z));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb) : "v"(hb), "v"(v.w));
  const mh2 l0 = __builtin_bit_cast(mh2, la), l1 = __builtin_bit_cast(mh2, lb);
  hi = mh4{ha[0], ha[1], hb[0], hb[1]};
  lo = mh4{l0[0], l0[1], l1[0], l1[1]};
}

constexpr int MM_Q = 128;                                    // queries per workgroup
__host__ __device__ constexpr int mm_rows(int C) { return C >= 128 ? 64 : 128; }      // train rows per LDS fill

template <int C, bool MASKED>
__global__ __launch_bounds__(256) void knn2_mfma_kernel(const float* __restrict__ d0, const float* __restrict__ d1,
                                                        const int32_t* __restrict__ n0p, const int32_t* __restrict__ n1p,
                                                        const int32_t* __restrict__ c0, const int32_t* __restrict__ c1,
                                                        int max0, int max1, int32_t* __restrict__ nn_idx,
                                                        float* __restrict__ nn_dist, float* __restrict__ nn_dist2,
                                                        unsigned long long* __restrict__ init_best, long init_n) {
  constexpr int MT = mm_rows(C), HP = C + 8, KS = C / 16;    // LDS row pitch in halves (16 bytes of padding), k-steps
  __shared__ __attribute__((aligned(16))) _Float16 s_h[MT * HP];
  __shared__ __attribute__((aligned(16))) _Float16 s_l[MT * HP];
  __shared__ __attribute__((aligned(16))) float s_n[MT];
  __shared__ int s_c[MT];
  __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);       // FP16_OVFL: conversions past the fp16 range clamp (ranking only)
  if (init_best) {
    const long wg = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (long e = wg * 256 + threadIdx.x; e < init_n; e += (long)gridDim.x * gridDim.y * gridDim.z * 256) init_best[e] = ~0ull;
  }
  const int b = blockIdx.y;
  const int n0 = n0p[b], n1 = n1p[b];
  if (blockIdx.x * MM_Q >= n0) return;               // whole workgroup out of range (uniform)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int q = blockIdx.x * MM_Q + wave * 32 + j;   // this lane's query (both lane halves hold it)
  const int nz = gridDim.z, z = blockIdx.z;
  const int tiles = (n1 + MT - 1) / MT, per = (tiles + nz - 1) / nz;
  const int t_lo = min(n1, z * per * MT), t_hi = min(n1, (z + 1) * per * MT);
  const float* qp = d0 + ((size_t)b * max0 + (q < n0 ? q : 0)) * C;
  // B operand: lane (j, h) holds channels 16 s + 8 h .. + 7 of its query, hi and lo halves, for every k-step
  mh8 qh[KS], ql[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float4 v0 = *reinterpret_cast<const float4*>(qp + 16 * s + 8 * h), v1 = *reinterpret_cast<const float4*>(qp + 16 * s + 8 * h + 4);
    mh4 h0, l0, h1, l1;
    m_split4(v0, h0, l0);
    m_split4(v1, h1, l1);
    qh[s] = mh8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    ql[s] = mh8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
  }
  const int qc = MASKED ? c0[(size_t)b * max0 + (q < n0 ? q : 0)] : 0;
  // Range guard.  The split-fp16 keys rank well only while a row's hi halves are normal fp16 numbers with their lo halves
  // above the subnormal floor, and below the clamp of FP16_OVFL: rows with |row| outside [0.5, 2^15] (unit-norm descriptors
  // sit in the middle of it) would rank on saturated or vanished keys and the exact pass would then re-score the wrong
  // candidates.  A workgroup that meets such a row — among its queries or in its slice of the train rows — discards its
  // ranking and scans its slice with the exact arithmetic instead (slow, correct; below).
  auto norm_ok = [](float n2) { return n2 >= 0.25f && n2 <= 1073741824.f; };
  bool bad = false;
  {
    float qn = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 v0 = *reinterpret_cast<const float4*>(qp + 16 * s + 8 * h), v1 = *reinterpret_cast<const float4*>(qp + 16 * s + 8 * h + 4);
      qn += (v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w) + (v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w);
    }
    qn += __shfl_xor(qn, 32);
    bad = q < n0 && !norm_ok(qn);
  }
  float best = INFINITY, second = INFINITY;          // approximate keys |t|^2 - 2 q.t of this lane's half of the train rows
  int bi = -1, si = -1;
  constexpr int LPR = C / 4;                          // lanes per train row in the staging loop (a float4 each)
  for (int t0 = t_lo; t0 < t_hi; t0 += MT) {
    const int nt = min(MT, t_hi - t0);
    __syncthreads();
    for (int e = tid; e < MT * LPR; e += 256) {
      const int row = e / LPR, c4 = e - row * LPR;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nt) v = reinterpret_cast<const float4*>(d1 + ((size_t)b * max1 + t0) * C)[e];
      mh4 hi, lo;
      m_split4(v, hi, lo);
      *reinterpret_cast<mh4*>(&s_h[row * HP + 4 * c4]) = hi;
      *reinterpret_cast<mh4*>(&s_l[row * HP + 4 * c4]) = lo;
      float nn = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
      for (int o = 1; o < LPR; o <<= 1) nn += __shfl_xor(nn, o);
      bad |= row < nt && !norm_ok(nn);
      if (c4 == 0) {
        s_n[row] = row < nt ? nn : INFINITY;          // rows past the range can never be a neighbour
        if (MASKED) s_c[row] = row < nt ? c1[(size_t)b * max1 + t0 + row] : -1;
      }
    }
    __syncthreads();
    for (int tt = 0; tt < MT; tt += 32) {
      if (tt >= nt) break;
      mf16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const _Float16* th = &s_h[(tt + j) * HP + 8 * h];
      const _Float16* tl = &s_l[(tt + j) * HP + 8 * h];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const mh8 ah = *reinterpret_cast<const mh8*>(th + 16 * s), al = *reinterpret_cast<const mh8*>(tl + 16 * s);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[s], acc, 0, 0, 0);
      }
      // accumulator register r of lane (j, h): train row tt + (r & 3) + 8 (r >> 2) + 4 h, ascending in r
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 tn = *reinterpret_cast<const float4*>(&s_n[tt + 8 * g + 4 * h]);
        const float tnv[4] = {tn.x, tn.y, tn.z, tn.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = tt + 8 * g + 4 * h + u;
          float key = fmaf(-2.f, acc[4 * g + u], tnv[u]);
          if (MASKED && s_c[row] != qc) key = INFINITY;
          const bool lt1 = key < best, lt2 = key < second;
          si = lt1 ? bi : (lt2 ? t0 + row : si);
          second = lt1 ? best : (lt2 ? key : second);
          bi = lt1 ? t0 + row : bi;
          best = lt1 ? key : best;
        }
      }
    }
  }
  // ---- exact pass: this lane's two candidates with the VALU kernel's arithmetic ----
  auto exact = [&](int t) -> float {
    if (t < 0) return INFINITY;
    const float* tp = d1 + ((size_t)b * max1 + t) * C;
    float acc = 0.f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) { const float d = qp[c] - tp[c]; acc = fmaf(d, d, acc); }
    return sqrtf(acc);
  };
  if (!(best < INFINITY)) bi = -1;                    // (masked-out / empty ranges leave +inf keys with stale indices)
  if (!(second < INFINITY)) si = -1;
  float e1 = exact(bi), e2 = exact(si);
  int i1 = bi, i2 = si;
  auto before = [](float da, int ia, float db, int ib) { return ib < 0 || (ia >= 0 && (da < db || (da == db && ia < ib))); };
  if (!before(e1, i1, e2, i2)) { const float te = e1; e1 = e2; e2 = te; const int ti = i1; i1 = i2; i2 = ti; }
  if (__syncthreads_or(bad ? 1 : 0)) {
    // a row outside the range guard: this lane's half of the workgroup's train slice, every row with the exact arithmetic
    const int mid = t_lo + ((t_hi - t_lo + 1) >> 1);
    const int s0 = h == 0 ? t_lo : mid, s1 = h == 0 ? mid : t_hi;
    e1 = e2 = INFINITY;
    i1 = i2 = -1;
    for (int t = s0; t < s1; ++t) {
      if (MASKED && c1[(size_t)b * max1 + t] != qc) continue;
      const float d = exact(t);
      if (before(d, t, e1, i1)) { e2 = e1; i2 = i1; e1 = d; i1 = t; }
      else if (before(d, t, e2, i2)) { e2 = d; i2 = t; }
    }
  }
  // merge with the other half of the train rows (lane ^ 32): two sorted pairs -> the two smallest by (distance, index)
  const float f1 = __shfl_xor(e1, 32), f2 = __shfl_xor(e2, 32);
  const int k1 = __shfl_xor(i1, 32), k2 = __shfl_xor(i2, 32);
  float rb, rs; int ri;
  if (before(e1, i1, f1, k1)) {
    rb = e1; ri = i1;
    rs = before(e2, i2, f1, k1) ? e2 : f1;
  } else {
    rb = f1; ri = k1;
    rs = before(f2, k2, e1, i1) ? f2 : e1;
  }
  if (h == 0 && q < n0) {
    const size_t o = ((size_t)z * gridDim.y + b) * max0 + q;
    nn_idx[o] = ri;
    nn_dist[o] = ri >= 0 ? rb : INFINITY;
    if (nn_dist2) nn_dist2[o] = rs;
  }
}

// combine the nz partial (best, second, index) triples of every query: slices hold ascending train ranges, so on equal
// distances the earlier slice (lower index) wins
__global__ __launch_bounds__(256) void knn2_merge_kernel(const int32_t* __restrict__ p_idx, const float* __restrict__ p_d,
                                                         const float* __restrict__ p_d2, int nz, int B, int max0,
                                                         const int32_t* __restrict__ n0p, int32_t* __restrict__ nn_idx,
                                                         float* __restrict__ nn_dist, float* __restrict__ nn_dist2,
                                                         unsigned long long* __restrict__ train_best, int max1, float ratio,
                                                         int masked) {
  const int b = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n0p[b]) return;
  float best = INFINITY, second = INFINITY;
  int bi = -1;
  for (int z = 0; z < nz; ++z) {
    const size_t o = ((size_t)z * B + b) * max0 + q;
    const float ob = p_d[o], os = p_d2[o];
    const int oi = p_idx[o];
    const bool take = ob < best;
    const float ns = take ? fminf(best, os) : fminf(second, ob);
    if (take) { best = ob; bi = oi; }
    second = ns;
  }
  const size_t o = (size_t)b * max0 + q;
  nn_idx[o] = bi;
  nn_dist[o] = best;
  if (nn_dist2) nn_dist2[o] = second;
  // the assign step of the ratio / one-to-one matcher, fused (same rule as match_assign_kernel)
  if (train_best && bi >= 0 && !(best > ratio * second) && !(masked && !(second < INFINITY)))
    atomicMin(&train_best[(size_t)b * max1 + bi], ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)q);
}

__global__ __launch_bounds__(256) void match_assign_kernel(const MatchArgs a) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= a.n0[b]) return;
  const size_t o = (size_t)b * a.max0 + q;
  const int t = a.nn_idx[o];
  const float d1 = a.nn_dist[o], d2 = a.nn_dist2[o];
  if (t < 0 || d1 > a.ratio * d2) return;          // feature_matcher.py:190 (needs a second neighbour: d2 = inf passes)
  // class-masked matching: a class with a single train row has no second neighbour — the reference's per-class
  // knnMatch(k=2) then yields one-element lists, goodMatchesOneToOne cannot unpack them and match_semantic skips the
  // whole class (visual_odometry.py:365-376): no match for such a query
  if (a.cls0 && !(d2 < INFINITY)) return;
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

// cv2.BFMatcher(crossCheck=True): (q, t) is a match iff t is q's nearest train row and q is t's nearest query
__global__ __launch_bounds__(256) void match_mutual_kernel(const MatchArgs a) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.max1) return;
  const size_t o = (size_t)b * a.max1 + t;
  int q = t < a.n1[b] ? a.rnn_idx[o] : -1;
  if (q >= 0 && a.nn_idx[(size_t)b * a.max0 + q] != t) q = -1;
  a.match_q[o] = q;
  a.match_d[o] = q >= 0 ? a.nn_dist[(size_t)b * a.max0 + q] : 0.f;
}

// matched rows of every pair, compacted in train order: one workgroup per pair walks the train rows 256 at a time
__global__ __launch_bounds__(256) void match_pairs_kernel(const PairsArgs a) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (int t0 = 0; t0 < a.max1; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const int q = t < a.max1 ? a.match_q[(size_t)b * a.max1 + t] : -1;
    const unsigned long long m = __ballot(q >= 0);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    if (q >= 0) {
      const size_t o = (size_t)b * a.max1 + off + before;
      if (a.idx) { a.idx[2 * o] = q; a.idx[2 * o + 1] = t; }
      if (a.dist) a.dist[o] = a.match_d[(size_t)b * a.max1 + t];
      if (a.pairs) {
        const float2 p0 = reinterpret_cast<const float2*>(a.pts0)[(size_t)b * a.max0 + q];
        const float2 p1 = reinterpret_cast<const float2*>(a.pts1)[(size_t)b * a.max1 + t];
        reinterpret_cast<float4*>(a.pairs)[o] = make_float4(p0.x, p0.y, p1.x, p1.y);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.count[b] = s_base;
}

template <int C>
static void launch_knn2(const float* d0, const float* d1, const int32_t* n0, const int32_t* n1, const int32_t* c0,
                        const int32_t* c1, int B, int max0, int max1, int nz, bool mfma, int32_t* idx, float* dist, float* dist2,
                        unsigned long long* init_best, long init_n, hipStream_t s) {
  if (mfma) {
    dim3 g((max0 + MM_Q - 1) / MM_Q, B, nz);
    if (c0) hipLaunchKernelGGL((knn2_mfma_kernel<C, true>), g, dim3(256), 0, s, d0, d1, n0, n1, c0, c1, max0, max1, idx, dist, dist2, init_best, init_n);
    else hipLaunchKernelGGL((knn2_mfma_kernel<C, false>), g, dim3(256), 0, s, d0, d1, n0, n1, c0, c1, max0, max1, idx, dist, dist2, init_best, init_n);
    return;
  }
  dim3 g((max0 + MQ - 1) / MQ, B, nz);
  if (c0) hipLaunchKernelGGL((knn2_kernel<C, true>), g, dim3(256), 0, s, d0, d1, n0, n1, c0, c1, max0, max1, idx, dist, dist2, init_best, init_n);
  else hipLaunchKernelGGL((knn2_kernel<C, false>), g, dim3(256), 0, s, d0, d1, n0, n1, c0, c1, max0, max1, idx, dist, dist2, init_best, init_n);
}

// one direction of the search; slices of the train range when the grid would leave most of the chip idle.
// assign: this is the forward search of the ratio / one-to-one matcher — the launch also resets the one-to-one table and,
// when the search is sliced, the merge launch applies the ratio test and claims train rows (*assigned = true)
static int knn2(const MatchArgs& a, bool reverse, int32_t* idx, float* dist, float* dist2, bool assign, bool* assigned, hipStream_t s) {
  const float* d0 = reverse ? a.d1 : a.d0;
  const float* d1 = reverse ? a.d0 : a.d1;
  const int32_t* n0 = reverse ? a.n1 : a.n0;
  const int32_t* n1 = reverse ? a.n0 : a.n1;
  const int32_t* c0 = reverse ? a.cls1 : a.cls0;
  const int32_t* c1 = reverse ? a.cls0 : a.cls1;
  const int max0 = reverse ? a.max1 : a.max0, max1 = reverse ? a.max0 : a.max1;
  // matrix-core form from 256 train rows (below that a query meets one or two LDS tiles and the exact pass dominates);
  // KP2D_MATCH_MFMA=0: always the VALU form (A/B)
  static const bool mfma_on = !(getenv("KP2D_MATCH_MFMA") && getenv("KP2D_MATCH_MFMA")[0] == '0');
  const bool mfma = mfma_on && max1 >= 256;
  const int qpw = mfma ? MM_Q : MQ, rows = mfma ? mm_rows(a.C) : mt_rows(a.C);
  const long wgs = (long)((max0 + qpw - 1) / qpw) * a.B;
  int nz = 1;
  if (a.part_idx && wgs < 512) {
    const int tiles = (max1 + rows - 1) / rows;
    nz = (int)((1024 + wgs - 1) / wgs);
    if (nz > tiles) nz = tiles;
    if (nz > a.part_slices) nz = a.part_slices;
    if (nz < 1) nz = 1;
  }
  int32_t* o_idx = nz > 1 ? a.part_idx : idx;
  float* o_d = nz > 1 ? a.part_d : dist;
  float* o_d2 = nz > 1 ? a.part_d2 : dist2;
  unsigned long long* ib = assign ? a.train_best : nullptr;
  const long in = (long)a.B * a.max1;
  switch (a.C) {
    case 32: launch_knn2<32>(d0, d1, n0, n1, c0, c1, a.B, max0, max1, nz, mfma, o_idx, o_d, o_d2, ib, in, s); break;
    case 64: launch_knn2<64>(d0, d1, n0, n1, c0, c1, a.B, max0, max1, nz, mfma, o_idx, o_d, o_d2, ib, in, s); break;
    case 128: launch_knn2<128>(d0, d1, n0, n1, c0, c1, a.B, max0, max1, nz, mfma, o_idx, o_d, o_d2, ib, in, s); break;
    default: return -1500;
  }
  if (assigned) *assigned = false;
  if (nz > 1) {
    hipLaunchKernelGGL(knn2_merge_kernel, dim3((max0 + 255) / 256, a.B), dim3(256), 0, s, a.part_idx, a.part_d, a.part_d2, nz,
                       a.B, max0, n0, idx, dist, dist2, ib, a.max1, a.ratio, a.cls0 ? 1 : 0);
    if (assigned) *assigned = assign;
  }
  return 0;
}

int launch_match(const MatchArgs& a, hipStream_t s) {
  if (a.C != 32 && a.C != 64 && a.C != 128) return -1500;
  if ((a.cls0 == nullptr) != (a.cls1 == nullptr)) return -1501;
  bool assigned = false;
  if (int e = knn2(a, false, a.nn_idx, a.nn_dist, a.nn_dist2, !a.mutual, &assigned, s)) return e;
  if (a.mutual) {
    if (!a.rnn_idx || !a.rnn_dist) return -1502;
    if (int e = knn2(a, true, a.rnn_idx, a.rnn_dist, nullptr, false, nullptr, s)) return e;
    hipLaunchKernelGGL(match_mutual_kernel, dim3((a.max1 + 255) / 256, a.B), dim3(256), 0, s, a);
    return (int)hipGetLastError();
  }
  // three launches per call: search (+ table reset), [merge +] assign, emit
  if (!assigned) hipLaunchKernelGGL(match_assign_kernel, dim3((a.max0 + 255) / 256, a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(match_emit_kernel, dim3((a.max1 + 255) / 256, a.B), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

// ---- the top_k_matches cap of the VO loop (visual_odometry.py:260-266, :272-283): keys -> launch_topk (post.hip) -> gather ----
// key of source row r: the value to rank by, larger = better (BF: minus the match distance; LightGlue: the matching score),
// -inf for rows without a match (kp2d_select_topk's `score > thr` with thr = -inf drops exactly those)
__global__ __launch_bounds__(256) void match_topk_keys_kernel(const TopkPairsArgs a) {
  const int b = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
  if (r >= a.n) return;
  const size_t o = (size_t)b * a.n + r;
  const bool on = a.mode == 0 ? a.match_q[o] >= 0 : a.matches0[o] >= 0;
  const float v = a.val[o];
  a.keys[o] = on ? (a.mode == 0 ? -v : v) : -INFINITY;
}
// selected source row sel[i] -> (row in set 0, row in set 1), the pair's coordinates and its value
__global__ __launch_bounds__(256) void match_topk_gather_kernel(const TopkPairsArgs a) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.kcap) return;
  const size_t oo = (size_t)b * a.kcap + i;
  const int r = i < a.count[b] ? a.sel[oo] : -1;
  int r0 = -1, r1 = -1;
  float v = 0.f;
  if (r >= 0) {
    const size_t o = (size_t)b * a.n + r;
    if (a.mode == 0) { r0 = a.match_q[o]; r1 = r; } else { r0 = r; r1 = (int)a.matches0[o]; }
    v = a.val[o];
  }
  if (a.idx) { a.idx[2 * oo] = r0; a.idx[2 * oo + 1] = r1; }
  if (a.out_val) a.out_val[oo] = v;
  if (a.pairs) {
    float4 pr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r >= 0) {
      const float2 p0 = reinterpret_cast<const float2*>(a.pts0)[(size_t)b * a.max0 + r0];
      const float2 p1 = reinterpret_cast<const float2*>(a.pts1)[(size_t)b * a.max1 + r1];
      pr = make_float4(p0.x, p0.y, p1.x, p1.y);
    }
    reinterpret_cast<float4*>(a.pairs)[oo] = pr;
  }
}
int launch_match_topk_pairs(const TopkPairsArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(match_topk_keys_kernel, dim3((a.n + 255) / 256, a.B), dim3(256), 0, s, a);
  TopkArgs t{a.keys, a.B, a.n, a.kcap, -INFINITY, a.sel, nullptr, a.count};
  if (int e = launch_topk(t, s)) return e;
  hipLaunchKernelGGL(match_topk_gather_kernel, dim3((a.kcap + 255) / 256, a.B), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

int launch_match_pairs(const PairsArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(match_pairs_kernel, dim3(a.B), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace kp2d
