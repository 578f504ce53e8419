// Host side of include/kp2d_lightglue.h: weight description / packing and the launch sequence of one
// LightGlue.forward (lightglue/lightglue.py:484-614).  No CPU compute path.
#include "../../include/kp2d_lightglue.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "device_guard.h"
#include "kp2d_kernels.h"

using namespace kp2d;

namespace {

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  set_last_error(buf);
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(KP2D_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v, size_t a = ALIGN) { return (v + a - 1) / a * a; }

struct Spec {
  std::string key;
  std::vector<int64_t> shape;
  size_t numel() const { size_t n = 1; for (auto s : shape) n *= (size_t)s; return n; }
};

// one packed linear layer: W^T [K][nout] then bias [nout] (nout padded to a multiple of 32)
// mf: the same matrix as split-fp16 matrix-core operand tiles (mfma_image), K * nout * 4 bytes, when K % 32 == 0
struct Lin { size_t w = 0, b = 0, mf = 0; int K = 0, nout = 0; };
struct Ffn { Lin l0, l3; size_t g = 0, be = 0; };
struct Layer { Lin qkv, out_proj, qkv_x, to_out; Ffn fs, fc; };

}  // namespace

struct kp2d_lg {
  kp2d_lg_config cfg{};
  std::vector<Spec> specs;
  std::map<std::string, int> index;
  std::map<std::string, std::vector<float>> host;
  std::vector<Layer> layers;
  Lin input_proj, final;
  size_t wr = 0, blob_floats = 0;
  float* blob = nullptr;
  bool finalized = false;
};

namespace {

int pad32(int n) { return (n + 31) / 32 * 32; }

void add_spec(kp2d_lg* m, const std::string& k, std::vector<int64_t> shape) {
  m->index[k] = (int)m->specs.size();
  m->specs.push_back(Spec{k, std::move(shape)});
}
void add_linear(kp2d_lg* m, const std::string& p, int out, int in) {
  add_spec(m, p + ".weight", {out, in});
  add_spec(m, p + ".bias", {out});
}
void add_ffn(kp2d_lg* m, const std::string& p, int d) {
  add_linear(m, p + ".0", 2 * d, 2 * d);
  add_spec(m, p + ".1.weight", {2 * d});
  add_spec(m, p + ".1.bias", {2 * d});
  add_linear(m, p + ".3", d, 2 * d);
}

// registration order of LightGlue.__init__ (lightglue.py:444-470)
void describe(kp2d_lg* m) {
  const int d = m->cfg.descriptor_dim, din = m->cfg.input_dim, n = m->cfg.n_layers, hd = d / m->cfg.num_heads;
  if (din != d) add_linear(m, "input_proj", d, din);
  add_spec(m, "posenc.Wr.weight", {hd / 2, 2});
  for (int i = 0; i < n; ++i) {
    const std::string s = "transformers." + std::to_string(i) + ".self_attn";
    add_linear(m, s + ".Wqkv", 3 * d, d);
    add_linear(m, s + ".out_proj", d, d);
    add_ffn(m, s + ".ffn", d);
    const std::string c = "transformers." + std::to_string(i) + ".cross_attn";
    add_linear(m, c + ".to_qk", d, d);
    add_linear(m, c + ".to_v", d, d);
    add_linear(m, c + ".to_out", d, d);
    add_ffn(m, c + ".ffn", d);
  }
  for (int i = 0; i < n; ++i) {
    add_linear(m, "log_assignment." + std::to_string(i) + ".matchability", 1, d);
    add_linear(m, "log_assignment." + std::to_string(i) + ".final_proj", d, d);
  }
  for (int i = 0; i + 1 < n; ++i) add_linear(m, "token_confidence." + std::to_string(i) + ".token.0", 1, d);

  size_t off = 0;
  auto take = [&](size_t floats) { size_t o = off; off = align_up(off + floats, ALIGN / 4); return o; };
  auto lin = [&](int K, int nout) {
    Lin l; l.K = K; l.nout = pad32(nout); l.w = take((size_t)K * l.nout); l.b = take(l.nout);
    if (K % 32 == 0) l.mf = take((size_t)K * l.nout);
    return l;
  };
  auto ffn = [&]() { Ffn f; f.l0 = lin(2 * d, 2 * d); f.g = take(2 * d); f.be = take(2 * d); f.l3 = lin(2 * d, d); return f; };
  if (din != d) m->input_proj = lin(din, d);
  m->wr = take(hd);
  m->layers.resize(n);
  for (auto& L : m->layers) {
    L.qkv = lin(d, 3 * d); L.out_proj = lin(d, d); L.fs = ffn();
    L.qkv_x = lin(d, 2 * d); L.to_out = lin(d, d); L.fc = ffn();
  }
  m->final = lin(d, d + 1);
  m->blob_floats = off;
}

const std::vector<float>& H(const kp2d_lg* m, const std::string& k) { return m->host.at(k); }

// W [nout][K] (torch Linear layout) -> W^T [K][nout_pad] at column offset c0, rows optionally permuted / scaled
void put_linear(std::vector<float>& blob, const Lin& l, const std::vector<float>& w, const std::vector<float>* b,
                int rows, int c0, float scale, const std::vector<int>* perm = nullptr) {
  for (int r = 0; r < rows; ++r) {
    const int c = c0 + (perm ? (*perm)[r] : r);
    for (int k = 0; k < l.K; ++k) blob[l.w + (size_t)k * l.nout + c] = w[(size_t)r * l.K + k] * scale;
    if (b) blob[l.b + c] = (*b)[r] * scale;
  }
}

// W^T [K][nout] (already in the blob) -> the A operands of v_mfma_f32_16x16x32_f16 for Y^T = W X^T, one 2-KB block
// per (16-feature tile t, 32-wide k step s): [hi | lo][lane][8 halves].  Lane l = (feature 16t + l%16, group g = l/16);
// its element j multiplies input feature 32s + 4g + j (j < 4) or 32s + 16 + 4g + (j - 4): the order in which the lanes
// of the PREVIOUS product's accumulator tiles (feature 16t' + 4g + r of row l%16) hold a row — a product's output is
// the next product's B operand without leaving its registers (lightglue.hip lg_tail_kernel).
void mfma_image(std::vector<float>& blob, const Lin& l) {
  if (!l.mf) return;
  _Float16* img = reinterpret_cast<_Float16*>(blob.data() + l.mf);
  const int KS = l.K / 32;
  for (int t = 0; t < l.nout / 16; ++t)
    for (int s = 0; s < KS; ++s)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int n = 16 * t + (lane & 15), g = lane >> 4;
          const int k = 32 * s + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
          const float w = blob[l.w + (size_t)k * l.nout + n];
          const _Float16 hi = (_Float16)w;
          const _Float16 lo = (_Float16)(w - (float)hi);
          _Float16* blk = img + (size_t)(t * KS + s) * 1024;
          blk[lane * 8 + j] = hi;
          blk[512 + lane * 8 + j] = lo;
        }
}

int pack(kp2d_lg* m, std::vector<float>& blob) {
  for (const auto& s : m->specs)
    if (!m->host.count(s.key)) return fail(KP2D_ERR_WEIGHT, "missing tensor '%s'", s.key.c_str());
  blob.assign(m->blob_floats, 0.f);
  const int d = m->cfg.descriptor_dim, n = m->cfg.n_layers, h = m->cfg.num_heads, hd = d / h;
  if (m->cfg.input_dim != d) put_linear(blob, m->input_proj, H(m, "input_proj.weight"), &H(m, "input_proj.bias"), d, 0, 1.f);
  { const auto& wr = H(m, "posenc.Wr.weight"); std::copy(wr.begin(), wr.end(), blob.begin() + m->wr); }
  // Wqkv rows: reference channel h*(3*hd) + dd*3 + t (unflatten(-1, (heads, -1, 3)), lightglue.py:254-255) -> t*d + h*hd + dd
  std::vector<int> perm(3 * d);
  for (int hh = 0; hh < h; ++hh)
    for (int dd = 0; dd < hd; ++dd)
      for (int t = 0; t < 3; ++t) perm[hh * 3 * hd + dd * 3 + t] = t * d + hh * hd + dd;
  auto put_ffn = [&](const Ffn& f, const std::string& p) {
    put_linear(blob, f.l0, H(m, p + ".0.weight"), &H(m, p + ".0.bias"), 2 * d, 0, 1.f);
    std::copy(H(m, p + ".1.weight").begin(), H(m, p + ".1.weight").end(), blob.begin() + f.g);
    std::copy(H(m, p + ".1.bias").begin(), H(m, p + ".1.bias").end(), blob.begin() + f.be);
    put_linear(blob, f.l3, H(m, p + ".3.weight"), &H(m, p + ".3.bias"), d, 0, 1.f);
  };
  for (int i = 0; i < n; ++i) {
    const Layer& L = m->layers[i];
    const std::string s = "transformers." + std::to_string(i) + ".self_attn";
    put_linear(blob, L.qkv, H(m, s + ".Wqkv.weight"), &H(m, s + ".Wqkv.bias"), 3 * d, 0, 1.f, &perm);
    put_linear(blob, L.out_proj, H(m, s + ".out_proj.weight"), &H(m, s + ".out_proj.bias"), d, 0, 1.f);
    put_ffn(L.fs, s + ".ffn");
    const std::string c = "transformers." + std::to_string(i) + ".cross_attn";
    put_linear(blob, L.qkv_x, H(m, c + ".to_qk.weight"), &H(m, c + ".to_qk.bias"), d, 0, 1.f);
    put_linear(blob, L.qkv_x, H(m, c + ".to_v.weight"), &H(m, c + ".to_v.bias"), d, d, 1.f);
    put_linear(blob, L.to_out, H(m, c + ".to_out.weight"), &H(m, c + ".to_out.bias"), d, 0, 1.f);
    put_ffn(L.fc, c + ".ffn");
  }
  // MatchAssignment of the LAST layer (lightglue.py:572): final_proj / d^0.25 (:389-390) and the matchability logit
  const std::string a = "log_assignment." + std::to_string(n - 1);
  put_linear(blob, m->final, H(m, a + ".final_proj.weight"), &H(m, a + ".final_proj.bias"), d, 0, 1.f / std::pow((float)d, 0.25f));
  put_linear(blob, m->final, H(m, a + ".matchability.weight"), &H(m, a + ".matchability.bias"), 1, d, 1.f);
  for (const Layer& L : m->layers)
    for (const Lin* l : {&L.qkv, &L.out_proj, &L.qkv_x, &L.to_out, &L.fs.l0, &L.fs.l3, &L.fc.l0, &L.fc.l3}) mfma_image(blob, *l);
  mfma_image(blob, m->final);
  return KP2D_OK;
}

struct Ws {
  size_t x, t3, ctx, msg, hb, cs, fz, rp_m, rp_s, cp_m, cp_s, rmax, rarg, cmax, carg, cnt, total;
};
Ws layout(const kp2d_lg* m, int B, int M, int N) {
  const size_t R = (size_t)B * (M + N), d = m->cfg.descriptor_dim, hd = d / m->cfg.num_heads;
  Ws w{};
  size_t off = 0;
  auto take = [&](size_t floats) { size_t o = off; off = align_up(off + floats * 4); return o; };
  w.x = take(R * d); w.t3 = take(R * 3 * d); w.ctx = take(R * d); w.msg = take(R * d); w.hb = take(R * 2 * d);
  w.cs = take(R * hd); w.fz = take(R * (d + 32));
  const size_t rp = (size_t)B * ((N + 63) / 64) * M, cp = (size_t)B * ((M + 63) / 64) * N;      // per-tile partials
  w.rp_m = take(rp); w.rp_s = take(rp); w.cp_m = take(cp); w.cp_s = take(cp);
  w.rmax = take(rp); w.rarg = take(rp); w.cmax = take(cp); w.carg = take(cp);
  w.cnt = take((size_t)2 * B);      // [n0 | n1]: key counts of the 2B sequences (kp2d_lg_forward_counts)
  w.total = off;
  return w;
}

}  // namespace

extern "C" {

int kp2d_lg_create(const kp2d_lg_config* cfg, kp2d_lg** out) {
  if (!cfg || !out) return fail(KP2D_ERR_ARG, "null argument");
  if (cfg->struct_size != (int32_t)sizeof(kp2d_lg_config)) return fail(KP2D_ERR_ARG, "kp2d_lg_config.struct_size mismatch");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(KP2D_ERR_HIP, "no HIP device visible: this library has no CPU path");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(KP2D_ERR_ARG, "device %d out of range (%d visible)", cfg->device, ndev);
  const int d = cfg->descriptor_dim;
  if (d % 32 || d < 32 || d > 64) return fail(KP2D_ERR_UNSUPPORTED, "descriptor_dim=%d (32 and 64 are built)", d);
  if (cfg->num_heads < 1 || d % cfg->num_heads || (d / cfg->num_heads) % 4 || d / cfg->num_heads > 16)
    return fail(KP2D_ERR_UNSUPPORTED, "num_heads=%d: head dim must be a multiple of 4, <= 16", cfg->num_heads);
  if (cfg->input_dim < 1 || cfg->input_dim > 128) return fail(KP2D_ERR_UNSUPPORTED, "input_dim=%d (<= 128)", cfg->input_dim);
  if (cfg->n_layers < 1 || cfg->n_layers > 32) return fail(KP2D_ERR_ARG, "n_layers=%d", cfg->n_layers);
  auto* m = new kp2d_lg();
  m->cfg = *cfg;
  describe(m);
  *out = m;
  return KP2D_OK;
}

void kp2d_lg_destroy(kp2d_lg* m) {
  if (!m) return;
  if (m->blob) (void)hipFree(m->blob);
  delete m;
}

int kp2d_lg_num_weights(const kp2d_lg* m) { return m ? (int)m->specs.size() : 0; }

int kp2d_lg_weight_info(const kp2d_lg* m, int index, const char** key, int64_t shape[4], int* ndim) {
  if (!m || index < 0 || index >= (int)m->specs.size() || !key || !shape || !ndim) return fail(KP2D_ERR_ARG, "bad argument");
  const Spec& s = m->specs[index];
  *key = s.key.c_str();
  *ndim = (int)s.shape.size();
  for (int i = 0; i < 4; ++i) shape[i] = i < *ndim ? s.shape[i] : 1;
  return KP2D_OK;
}

int kp2d_lg_set_weight(kp2d_lg* m, const char* key, const float* host, const int64_t* shape, int ndim) {
  if (!m || !key || !host || (!shape && ndim > 0)) return fail(KP2D_ERR_ARG, "null argument");
  auto it = m->index.find(key);
  if (it == m->index.end()) return fail(KP2D_ERR_WEIGHT, "unexpected key '%s'", key);
  const Spec& s = m->specs[it->second];
  bool same = (int)s.shape.size() == ndim;
  for (int i = 0; same && i < ndim; ++i) same = s.shape[i] == shape[i];
  if (!same) return fail(KP2D_ERR_WEIGHT, "shape mismatch for '%s'", key);
  m->host[key].assign(host, host + s.numel());
  m->finalized = false;
  return KP2D_OK;
}

int kp2d_lg_finalize_weights(kp2d_lg* m) {
  if (!m) return fail(KP2D_ERR_ARG, "null model");
  std::vector<float> blob;
  int rc = pack(m, blob);
  if (rc != KP2D_OK) return rc;
  kp2d::DeviceGuard guard(m->cfg.device);
  if (!m->blob) HIP_TRY(hipMalloc((void**)&m->blob, m->blob_floats * sizeof(float)));
  HIP_TRY(hipMemcpy(m->blob, blob.data(), m->blob_floats * sizeof(float), hipMemcpyHostToDevice));
  m->finalized = true;
  return KP2D_OK;
}

size_t kp2d_lg_workspace_bytes(const kp2d_lg* m, int B, int M, int N) {
  if (!m || B < 1 || M < 1 || N < 1) return 0;
  return layout(m, B, M, N).total;
}

int kp2d_lg_forward(kp2d_lg* m, const float* kpts0, const float* kpts1, const float* desc0, const float* desc1,
                    const float* size0, const float* size1, int B, int M, int N, float filter_threshold,
                    float* log_assignment, int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1,
                    float* ref_desc0, float* ref_desc1, void* workspace, size_t workspace_bytes, void* stream) {
  return kp2d_lg_forward_counts(m, kpts0, kpts1, desc0, desc1, size0, size1, nullptr, nullptr, B, M, N, filter_threshold,
                                log_assignment, matches0, matches1, mscores0, mscores1, ref_desc0, ref_desc1, workspace,
                                workspace_bytes, stream);
}

int kp2d_lg_forward_counts(kp2d_lg* m, const float* kpts0, const float* kpts1, const float* desc0, const float* desc1,
                           const float* size0, const float* size1, const int32_t* n0, const int32_t* n1, int B, int M, int N,
                           float filter_threshold, float* log_assignment, int64_t* matches0, int64_t* matches1,
                           float* mscores0, float* mscores1, float* ref_desc0, float* ref_desc1, void* workspace,
                           size_t workspace_bytes, void* stream) {
  if ((n0 == nullptr) != (n1 == nullptr)) return fail(KP2D_ERR_ARG, "keypoint counts must be given for both sets or neither");
  if (n0 && (!size0 || !size1)) return fail(KP2D_ERR_ARG, "padded keypoint sets need the image sizes (the default derives them from every row)");
  if (!m || !kpts0 || !kpts1 || !desc0 || !desc1 || !log_assignment || !workspace) return fail(KP2D_ERR_ARG, "null argument");
  if (!matches0 || !matches1 || !mscores0 || !mscores1) return fail(KP2D_ERR_ARG, "null match outputs");
  if (!m->finalized) return fail(KP2D_ERR_STATE, "weights not finalised (kp2d_lg_finalize_weights)");
  if (B < 1 || M < 1 || N < 1) return fail(KP2D_ERR_ARG, "B, M, N must be >= 1");
  if ((long)B * (M + N) > (1l << 24)) return fail(KP2D_ERR_ARG, "too many keypoints in one call");
  if ((uintptr_t)workspace % ALIGN) return fail(KP2D_ERR_WORKSPACE, "workspace must be %zu-byte aligned", ALIGN);
  const Ws w = layout(m, B, M, N);
  if (workspace_bytes < w.total) return fail(KP2D_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, w.total);
  kp2d::DeviceGuard guard(m->cfg.device);
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)workspace;
  auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
  const int d = m->cfg.descriptor_dim, din = m->cfg.input_dim, heads = m->cfg.num_heads, hd = d / heads;
  const int R = B * (M + N), R0 = B * M;
  float *X = F(w.x), *T3 = F(w.t3), *CTX = F(w.ctx), *MSG = F(w.msg), *HB = F(w.hb), *CS = F(w.cs), *FZ = F(w.fz);
  // padded keypoint sets: [n0 | n1] as the key counts of the 2B sequences the attention launches walk
  int32_t* CNT = n0 ? reinterpret_cast<int32_t*>(base + w.cnt) : nullptr;
  if (n0) {
    HIP_TRY(hipMemcpyAsync(CNT, n0, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(CNT + B, n1, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
  }
  const float* blob = m->blob;
  int e;
#define LG_CHECK(call, what)                                                                                    \
  do {                                                                                                          \
    e = (call);                                                                                                 \
    if (e) return fail(e < 0 ? KP2D_ERR_UNSUPPORTED : KP2D_ERR_HIP, "%s: launch failed (%d)", what, e);         \
  } while (0)

  {
    LgPosArgs a{kpts0, kpts1, size0, size1, blob + m->wr, CS, B, M, N, hd};
    LG_CHECK(launch_lg_posenc(a, st), "posenc");
  }
  auto linear = [&](const Lin& l, const float* x0, int k0, int xs0, const float* x1, int k1, int xs1, float* out, int os,
                    int rows, int nvalid, int epi) {
    LgLinArgs a{};
    a.x0 = x0; a.x1 = x1; a.k0 = k0; a.k1 = k1; a.xs0 = xs0; a.xs1 = xs1;
    a.w = blob + l.w; a.bias = blob + l.b; a.out = out; a.os = os; a.oo = 0;
    a.rows = rows; a.nout = l.nout; a.nvalid = nvalid; a.epi = epi;
    return a;
  };
  if (din != d) {
    LgLinArgs a = linear(m->input_proj, desc0, din, din, nullptr, 0, 0, X, d, R0, d, LG_EPI_NONE);
    LG_CHECK(launch_lg_linear(a, st), "input_proj");
    a = linear(m->input_proj, desc1, din, din, nullptr, 0, 0, X + (size_t)R0 * d, d, B * N, d, LG_EPI_NONE);
    LG_CHECK(launch_lg_linear(a, st), "input_proj");
  } else {
    HIP_TRY(hipMemcpyAsync(X, desc0, (size_t)R0 * d * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(X + (size_t)R0 * d, desc1, (size_t)B * N * d * 4, hipMemcpyDeviceToDevice, st));
  }
  auto ffn = [&](const Ffn& f, const char* what) -> int {
    // x + ffn(cat[x, message]) (lightglue.py:261 / :325-326): Linear -> LayerNorm -> GELU, then Linear + residual in place
    LgLinArgs a = linear(f.l0, X, d, d, MSG, d, d, HB, 2 * d, R, 2 * d, LG_EPI_LNGELU);
    a.ln_g = blob + f.g; a.ln_b = blob + f.be;
    LG_CHECK(launch_lg_linear(a, st), what);
    a = linear(f.l3, HB, 2 * d, 2 * d, nullptr, 0, 0, X, d, R, d, LG_EPI_RESID);
    a.res = X; a.rs = d;
    LG_CHECK(launch_lg_linear(a, st), what);
    return KP2D_OK;
  };
  // D = 32 (configs S, A): out_proj / to_out + ffn + residual as ONE row-local kernel (lightglue.hip lg_tail_kernel)
  static const bool fuse_tail = !(getenv("KP2D_LG_FUSE") && getenv("KP2D_LG_FUSE")[0] == '0');
  // `next`: the token-wise projection that follows the block (the cross block's [to_qk | to_v], the next layer's Wqkv
  // with rotary, the final projection) runs in the tail's launch on the rows it has just updated: 8 launches fewer
  // per forward than one lg_linear per projection (the matcher is a chain of ~35 dependent launches of ~10 us)
  static const bool fuse_next = !(getenv("KP2D_LG_FUSE_NEXT") && getenv("KP2D_LG_FUSE_NEXT")[0] == '0');
  auto tail = [&](const Lin& proj, const Ffn& f, const char* what, const Lin* next, float* nout, int nos, int nvalid,
                  bool rotary) -> int {
    LgTailArgs t{};
    t.x = X; t.ctx = CTX; t.io = blob + proj.mf; t.bo = blob + proj.b; t.i1 = blob + f.l0.mf; t.b1 = blob + f.l0.b;
    t.ln_g = blob + f.g; t.ln_b = blob + f.be; t.i2 = blob + f.l3.mf; t.b2 = blob + f.l3.b; t.rows = R; t.D = d;
    if (next) {
      t.in = blob + next->mf; t.bn = blob + next->b; t.on = nout; t.nn = next->nout; t.nos = nos; t.nvalid = nvalid;
      if (rotary) { t.cs = CS; t.hd = hd; t.rot_cols = 2 * d; }
    }
    LG_CHECK(launch_lg_tail(t, st), what);
    return KP2D_OK;
  };
  const bool fused = fuse_tail && fuse_next && d == 32;
  const float scale = 1.f / std::sqrt((float)hd);
  for (int i = 0; i < m->cfg.n_layers; ++i) {
    const Layer& L = m->layers[i];
    // ---- SelfBlock (lightglue.py:247-261), both images in one launch per token-wise layer ----
    {
      LgLinArgs a = linear(L.qkv, X, d, d, nullptr, 0, 0, T3, 3 * d, R, 3 * d, LG_EPI_ROTARY);
      a.cs = CS; a.hd = hd; a.rot_cols = 2 * d;
      if (fused && i == 0) {           // the tail kernel's projection-only mode: same arithmetic as every later Wqkv
        LgTailArgs t{};
        t.x = X; t.rows = R; t.D = d;
        t.in = blob + L.qkv.mf; t.bn = blob + L.qkv.b; t.on = T3; t.nn = L.qkv.nout; t.nos = 3 * d; t.nvalid = 3 * d;
        t.cs = CS; t.hd = hd; t.rot_cols = 2 * d;
        LG_CHECK(launch_lg_tail(t, st), "self_attn.Wqkv");
      } else if (!fused) {             // (fused, i > 0: done by the previous tail)
        LG_CHECK(launch_lg_linear(a, st), "self_attn.Wqkv");
      }
      if (M == N) {   // both images as one batch of 2B sequences
        AttnArgs t{T3, T3, CTX, 2 * B, M, M, d, heads, scale};
        t.q_stride = 3 * d; t.kv_stride = 3 * d; t.k_off = d; t.v_off = 2 * d; t.out_stride = d; t.prec = 1;
        t.tcount = CNT;
        LG_CHECK(launch_attention(t, st), "self_attn.inner_attn");
      } else {
        for (int set = 0; set < 2; ++set) {
          const size_t r0 = set ? (size_t)R0 : 0;
          const int n = set ? N : M;
          AttnArgs t{T3 + r0 * 3 * d, T3 + r0 * 3 * d, CTX + r0 * d, B, n, n, d, heads, scale};
          t.q_stride = 3 * d; t.kv_stride = 3 * d; t.k_off = d; t.v_off = 2 * d; t.out_stride = d;
          t.prec = 1;      // split-fp16 MFMA (fp32-grade, attention.hip): 3x faster than the fp32 16x16x4 kernel at head dim 8
          t.tcount = CNT ? CNT + (set ? B : 0) : nullptr;
          LG_CHECK(launch_attention(t, st), "self_attn.inner_attn");
        }
      }
      if (fuse_tail && d == 32) {
        int rc = tail(L.out_proj, L.fs, "self_attn tail", fused ? &L.qkv_x : nullptr, T3, 2 * d, 2 * d, false);
        if (rc != KP2D_OK) return rc;
      } else {
        a = linear(L.out_proj, CTX, d, d, nullptr, 0, 0, MSG, d, R, d, LG_EPI_NONE);
        LG_CHECK(launch_lg_linear(a, st), "self_attn.out_proj");
        int rc = ffn(L.fs, "self_attn.ffn");
        if (rc != KP2D_OK) return rc;
      }
    }
    // ---- CrossBlock (lightglue.py:303-327): [to_qk | to_v] in one layer, attention both ways ----
    {
      LgLinArgs a = linear(L.qkv_x, X, d, d, nullptr, 0, 0, T3, 2 * d, R, 2 * d, LG_EPI_NONE);
      if (!fused) LG_CHECK(launch_lg_linear(a, st), "cross_attn.to_qk/to_v");
      if (M == N) {   // both directions in one launch: sequence b attends to sequence (b + B) mod 2B
        AttnArgs t{T3, T3, CTX, 2 * B, M, M, d, heads, scale};
        t.q_stride = 2 * d; t.kv_stride = 2 * d; t.k_off = 0; t.v_off = d; t.out_stride = d; t.prec = 1;
        t.kv_bshift = B;
        t.tcount = CNT;
        LG_CHECK(launch_attention(t, st), "cross_attn");
      } else {
        for (int set = 0; set < 2; ++set) {
          const size_t rq = set ? (size_t)R0 : 0, rk = set ? 0 : (size_t)R0;
          AttnArgs t{T3 + rq * 2 * d, T3 + rk * 2 * d, CTX + rq * d, B, set ? N : M, set ? M : N, d, heads, scale};
          t.q_stride = 2 * d; t.kv_stride = 2 * d; t.k_off = 0; t.v_off = d; t.out_stride = d;
          t.prec = 1;
          t.tcount = CNT ? CNT + (set ? 0 : B) : nullptr;      // the keys are the OTHER set's rows
          LG_CHECK(launch_attention(t, st), "cross_attn");
        }
      }
      if (fuse_tail && d == 32) {
        const bool last = i + 1 == m->cfg.n_layers;
        const Lin* nx = !fused ? nullptr : (last ? &m->final : &m->layers[i + 1].qkv);
        int rc = last ? tail(L.to_out, L.fc, "cross_attn tail", nx, FZ, d + 32, d + 1, false)
                      : tail(L.to_out, L.fc, "cross_attn tail", nx, T3, 3 * d, 3 * d, true);
        if (rc != KP2D_OK) return rc;
      } else {
        a = linear(L.to_out, CTX, d, d, nullptr, 0, 0, MSG, d, R, d, LG_EPI_NONE);
        LG_CHECK(launch_lg_linear(a, st), "cross_attn.to_out");
        int rc = ffn(L.fc, "cross_attn.ffn");
        if (rc != KP2D_OK) return rc;
      }
    }
  }
  {
    LgLinArgs a = linear(m->final, X, d, d, nullptr, 0, 0, FZ, d + 32, R, d + 1, LG_EPI_NONE);
    if (!fused || m->cfg.n_layers == 0) LG_CHECK(launch_lg_linear(a, st), "log_assignment.final_proj");
    LgAssignArgs g{};
    g.fz = FZ; g.fs = d + 32; g.D = d; g.B = B; g.M = M; g.N = N; g.scores = log_assignment;
    g.rp_m = F(w.rp_m); g.rp_s = F(w.rp_s); g.cp_m = F(w.cp_m); g.cp_s = F(w.cp_s);
    g.rmax = F(w.rmax); g.cmax = F(w.cmax);
    g.rarg = reinterpret_cast<int*>(base + w.rarg); g.carg = reinterpret_cast<int*>(base + w.carg);
    g.th = filter_threshold;
    g.cnt0 = n0; g.cnt1 = n1;
    g.matches0 = matches0; g.matches1 = matches1; g.mscores0 = mscores0; g.mscores1 = mscores1;
    LG_CHECK(launch_lg_assign(g, st), "log_assignment");
  }
  if (ref_desc0 && ref_desc1 == ref_desc0 + (size_t)R0 * d) {      // one buffer behind both outputs: one copy
    HIP_TRY(hipMemcpyAsync(ref_desc0, X, (size_t)R * d * 4, hipMemcpyDeviceToDevice, st));
    return KP2D_OK;
  }
  if (ref_desc0) HIP_TRY(hipMemcpyAsync(ref_desc0, X, (size_t)R0 * d * 4, hipMemcpyDeviceToDevice, st));
  if (ref_desc1) HIP_TRY(hipMemcpyAsync(ref_desc1, X + (size_t)R0 * d, (size_t)B * N * d * 4, hipMemcpyDeviceToDevice, st));
#undef LG_CHECK
  return KP2D_OK;
}

}  // extern "C"
