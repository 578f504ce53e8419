#include <cstddef>
// Host side of the C ABI declared in include/kp2d.h: model description, weight packing, the per-call
// launch plan and the measurement hooks.  All arithmetic happens in the HIP kernels of this directory;
// there is no CPU compute path here (a missing device or library is an error, never a fallback).
#include "../../include/kp2d.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "device_guard.h"
#include "kp2d_kernels.h"

using namespace kp2d;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(KP2D_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v, size_t a = ALIGN) { return (v + a - 1) / a * a; }

struct WeightSpec {
  std::string key;
  std::vector<int64_t> shape;
  size_t numel() const {
    size_t n = 1;
    for (auto s : shape) n *= (size_t)s;
    return n;
  }
};

// one packed 3x3 convolution
struct ConvPack {
  std::string name;     // state-dict prefix, e.g. "backbone.conv2a"
  bool bn = false;      // AnnotatedConvBnReLUModel (conv.weight + bn.*) vs plain Conv2d (weight + bias)
  bool shuffle = false; // rows permuted for the PixelShuffle-folding store
  bool bias = true;     // plain conv only: has a .bias tensor
  bool tconv = false;   // TransposedConvUpsampleModel (base.py:80-117) restated as a pixel-shuffled 3x3 conv (add_tconv)
  int kind = 0;         // 0: 3x3 [co][ci][3][3]   1: 1x1 [co][ci][1][1]   2: 2x2 stride 2 [co][ci][2][2] as 1x1 over 4*ci
  std::vector<std::pair<std::string, int>> parts;   // merged CBRs over one input (name, cout): rows = the parts' rows in order
  int taps = 9;
  int cin = 0, cout = 0, npad = 0, kc = 16;   // cin = GEMM K per tap (4*ci for kind 2)
  size_t w_off = 0, sc_off = 0, sh_off = 0;   // float offsets into the blob
  size_t w16_off = 0, sc16_off = 0;           // split-fp16 pack: [hi16|lo16] half rows of w * 2^e, scale * 2^-e (pack())
  size_t w16n_off = 0;                        // the same rows in 32-channel groups (npad >= 64): small-grid launches
  size_t w16t_off = 0;                        // the 64-channel-group rows with the taps transposed (3x3, npad >= 64): transposed tiles of conv3x3_wsm.hip
  size_t wd_off = 0;                          // head layers (3x3, <= 4 output channels): fp32 [chunk][tap][4][16] for head3x3.hip
  bool head() const { return kind == 0 && cout <= 4 && !shuffle && !tconv && parts.empty(); }
  size_t wd_floats() const { return (size_t)((cin + 15) / 16) * 9 * 4 * 16; }
  size_t w_floats() const { return (size_t)((cin + kc - 1) / kc) * taps * npad * kc; }
  size_t w16_floats() const { return (size_t)((cin + 15) / 16) * taps * npad * 16; }
};

struct VecPack { size_t off = 0; int n = 0; };   // small per-channel vectors (LayerNorm g/b, depthwise w/b)

struct ProfRec {
  std::string layer, kernel;
  double flops = 0, bytes = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};

// first-fit arena over the caller's workspace
struct Arena {
  struct Blk { size_t off, size; };
  std::vector<Blk> free_;
  size_t cap = 0, high = 0;
  void reset(size_t capacity) { cap = capacity; free_.assign(1, Blk{0, capacity}); high = 0; }
  size_t alloc(size_t bytes) {
    bytes = align_up(bytes);
    for (size_t i = 0; i < free_.size(); ++i) {
      if (free_[i].size >= bytes) {
        const size_t off = free_[i].off;
        free_[i].off += bytes;
        free_[i].size -= bytes;
        if (free_[i].size == 0) free_.erase(free_.begin() + i);
        high = std::max(high, off + bytes);
        return off;
      }
    }
    return (size_t)-1;
  }
  void release(size_t off, size_t bytes) {
    bytes = align_up(bytes);
    free_.push_back(Blk{off, bytes});
    std::sort(free_.begin(), free_.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
    for (size_t i = 0; i + 1 < free_.size();) {
      if (free_[i].off + free_[i].size == free_[i + 1].off) {
        free_[i].size += free_[i + 1].size;
        free_.erase(free_.begin() + i + 1);
      } else {
        ++i;
      }
    }
  }
};

// NHWC activation living in the workspace
struct Act {
  size_t off = 0, bytes = 0;
  int C = 0, H = 0, W = 0;
  int PS = 0, CO = 0;  // channel-slice view of a wider tensor: pixel stride (0 = C) and first channel; bytes = 0 (not owned)
  int fmt = 0;         // 1: an S16P tensor (kp2d_kernels.h: the fp16 halves of the split, planar rows; same bytes); a view is a run of whole chunks
};

}  // namespace

struct kp2d_model {
  kp2d_config cfg{};
  int c1, c2, c3, c4, c5, d1;
  std::vector<WeightSpec> specs;
  std::map<std::string, int> spec_index;
  std::map<std::string, std::vector<float>> host;
  std::vector<ConvPack> convs;
  std::map<std::string, int> conv_index;
  size_t conv1a_w = 0, conv1a_sc = 0, conv1a_sh = 0;
  size_t vlad_wa = 0, vlad_cent = 0;
  std::map<std::string, VecPack> vecs;
  size_t blob_floats = 0;
  float* blob = nullptr;
  bool finalized = false;
  int chunk_frames = 0;
  int ws_min = 0;         // kp2d_set_option("ws_min_tiles"): least tiles of a launch for the warp-specialised conv1b form (0 = 1024)
  int wsm_grid = 0;       // kp2d_set_option("wsm_grid"): most workgroups per launch of that form (0 = KP2D_WSM_GRID or one per CU)
  int wsm_tr = 0;         // kp2d_set_option("wsm_transposed")
  int wsm_min = 0;        // kp2d_set_option("wsm_min_items"): 0 = automatic (KP2D_WSM, else one item per workgroup), < 0 = never (conv3x3_wsm.hip)
  bool mff_fused = !(getenv("KP2D_MFF") && getenv("KP2D_MFF")[0] == '0');   // kp2d_set_option("mff_fused"): MixFeedForward's tail as one launch (mff_tail.hip)
  // kp2d_set_option("stem_fusion") / KP2D_STEM: the first layer — 1 (default): split-fp16 products, computed inside conv1b's launch on
  // big grids (conv3x3_f16.hip STEM) and by conv1a_mfma_kernel otherwise (the same bits); 2: the same arithmetic, never fused; 0: round 4's
  // exact-fp32 FMA kernels (conv1a_kernel / conv1a_u8_kernel)
  int stem_fusion = getenv("KP2D_STEM") ? std::max(0, std::min(2, atoi(getenv("KP2D_STEM")))) : 1;
  size_t conv1a_ws = 0;       // blob offset of 2^e, the scale conv1a's weights are split at (pack(); in the blob, so that it travels with an RCCL weight broadcast)
  bool multi_launch = true;   // kp2d_set_option("multi_launch"): independent layers of a level as one launch on small grids
  bool s16_all = true;        // kp2d_set_option("s16_all"): big grids keep every tensor the warp-specialised 3x3 layers read as S16P (build())
  int s16_min = 0;        // kp2d_set_option("s16_min_items"): conv3x3_s16.hip — 0 = automatic (three rounds of tiles per workgroup), N = from N tiles, < 0 = never
  int precision = KP2D_PREC_F16X3;
  std::map<uint64_t, size_t> plan_cache;
  int lanes = 2;          // independent sub-batches run concurrently on this many HIP streams (KP2D_LANES); +3 %
  int lanes_default = 2;  // what kp2d_set_option("lanes", 0) restores
  std::vector<hipStream_t> lane_streams;
  std::vector<hipEvent_t> lane_events;
  hipEvent_t fork_event = nullptr;
  // single frames (the level schedule of build()): NetVLAD's launches on a side stream beside the segmentation head's chain
  hipStream_t side_stream = nullptr;
  hipEvent_t side_fork = nullptr, side_join = nullptr;
  bool side_overlap = true;   // kp2d_set_option("side_overlap")
  bool profiling = false;
  int64_t* seg_ids_dst = nullptr;   // kp2d_set_seg_ids: class ids [B,1,H2,W2] written by the forward's last segmentation layer
  size_t seg_ids_cap = 0;
  std::string tap_name;   // kp2d_set_tap: one intermediate activation copied out (planar) during forward
  float* tap_dst = nullptr;
  size_t tap_cap = 0;
  bool head_dot = !(getenv("KP2D_HEAD_DOT") && getenv("KP2D_HEAD_DOT")[0] == '0');   // KP2D_HEAD_DOT=0: heads on the matrix-core kernels (A/B)
  bool small_grid_ng32 = !(getenv("KP2D_NG32") && getenv("KP2D_NG32")[0] == '0');   // KP2D_NG32=0: always 64-channel groups
  std::vector<ProfRec> prof;
  size_t prof_used = 0;
  hipStream_t prof_stream = nullptr;
};

namespace {

// ------------------------------------------------------------------------------------------------
// model description (state-dict layout: SURVEY.md App. C; constructors kp2dtiny.py:347-449 / :732-803)
// ------------------------------------------------------------------------------------------------
void add_spec(kp2d_model* m, const std::string& key, std::vector<int64_t> shape) {
  m->spec_index[key] = (int)m->specs.size();
  m->specs.push_back(WeightSpec{key, std::move(shape)});
}

void add_cbr(kp2d_model* m, const std::string& p, int ci, int co, bool shuffle = false) {
  add_spec(m, p + ".conv.weight", {co, ci, 3, 3});
  add_spec(m, p + ".bn.weight", {co});
  add_spec(m, p + ".bn.bias", {co});
  add_spec(m, p + ".bn.running_mean", {co});
  add_spec(m, p + ".bn.running_var", {co});
  ConvPack c;
  c.name = p; c.bn = true; c.shuffle = shuffle; c.cin = ci; c.cout = co;
  m->conv_index[p] = (int)m->convs.size();
  m->convs.push_back(c);
}

void add_conv(kp2d_model* m, const std::string& p, int ci, int co, bool shuffle = false) {
  add_spec(m, p + ".weight", {co, ci, 3, 3});
  add_spec(m, p + ".bias", {co});
  ConvPack c;
  c.name = p; c.bn = false; c.shuffle = shuffle; c.cin = ci; c.cout = co;
  m->conv_index[p] = (int)m->convs.size();
  m->convs.push_back(c);
}

// TransposedConvUpsampleModel(c) (base.py:80-117): ConvTranspose2d(c, c/4, k3, s2, p1, output_padding 1, no bias)
// -> BatchNorm2d(c/4) -> (Leaky)ReLU.  Output pixel (2y+a, 2x+b) only sees inputs (y..y+1, x..x+1):
//   a = 0: in[y] * w[ky=1];   a = 1: in[y] * w[ky=2] + in[y+1] * w[ky=0]      (same along x)
// so it IS a 3x3 convolution c -> 4*(c/4) with the dy = -1 / dx = -1 taps zero, followed by PixelShuffle(2)
// (virtual channel 4*co + 2a + b), and runs through the pixel-shuffle store of the conv kernel unchanged.
void add_tconv(kp2d_model* m, const std::string& p, int c) {
  add_spec(m, p + ".transposed_conv.weight", {c, c / 4, 3, 3});
  add_spec(m, p + ".bn.weight", {c / 4});
  add_spec(m, p + ".bn.bias", {c / 4});
  add_spec(m, p + ".bn.running_mean", {c / 4});
  add_spec(m, p + ".bn.running_var", {c / 4});
  ConvPack k;
  k.name = p; k.bn = true; k.shuffle = true; k.tconv = true; k.cin = c; k.cout = c;
  m->conv_index[p] = (int)m->convs.size();
  m->convs.push_back(k);
}

// 1x1 conv (kind 1) or 2x2 stride-2 conv (kind 2) routed through the MFMA conv kernel with taps = 1
void add_pw(kp2d_model* m, const std::string& p, int ci, int co, bool bias, int kind) {
  const int k = kind == 2 ? 2 : 1;
  add_spec(m, p + ".weight", {co, ci, k, k});
  if (bias) add_spec(m, p + ".bias", {co});
  ConvPack c;
  c.name = p; c.bn = false; c.bias = bias; c.kind = kind; c.taps = 1;
  c.cin = kind == 2 ? 4 * ci : ci; c.cout = co;
  m->conv_index[p] = (int)m->convs.size();
  m->convs.push_back(c);
}

// SegFormerAttentionModule(c) (modules/segformer.py:209-220); PreNorm registers fn before norm
void add_attention_module(kp2d_model* m, const std::string& p, int c) {
  add_pw(m, p + ".att.fn.to_q", c, c, false, 1);
  add_pw(m, p + ".att.fn.to_kv", c, 2 * c, false, 2);
  add_pw(m, p + ".att.fn.to_out", c, c, false, 1);
  add_spec(m, p + ".att.norm.g", {1, c, 1, 1});
  add_spec(m, p + ".att.norm.b", {1, c, 1, 1});
  const int h = 2 * c;
  add_pw(m, p + ".mff.fn.net.0", c, h, true, 1);
  add_spec(m, p + ".mff.fn.net.1.net.0.weight", {h, 1, 3, 3});
  add_spec(m, p + ".mff.fn.net.1.net.0.bias", {h});
  add_pw(m, p + ".mff.fn.net.1.net.1", h, h, true, 1);
  add_pw(m, p + ".mff.fn.net.3", h, c, true, 1);
  add_spec(m, p + ".mff.norm.g", {1, c, 1, 1});
  add_spec(m, p + ".mff.norm.b", {1, c, 1, 1});
  m->vecs[p + ".att.norm.g"].n = c;
  m->vecs[p + ".att.norm.b"].n = c;
  m->vecs[p + ".mff.norm.g"].n = c;
  m->vecs[p + ".mff.norm.b"].n = c;
  m->vecs[p + ".mff.fn.net.1.net.0.weight"].n = 9 * h;   // repacked [9][h]
  m->vecs[p + ".mff.fn.net.1.net.0.bias"].n = h;
}

int describe(kp2d_model* m) {
  const kp2d_config& g = m->cfg;
  const int c1 = m->c1, c2 = m->c2, c3 = m->c3, c4 = m->c4, c5 = m->c5, d1 = m->d1;
  const bool v3 = g.version == 3;
  // backbone (encoders.py:20-99).  conv1a is packed separately (Cin = 3, or 1 for use_color=False).
  add_spec(m, "backbone.conv1a.conv.weight", {c1, g.in_channels, 3, 3});
  add_spec(m, "backbone.conv1a.bn.weight", {c1});
  add_spec(m, "backbone.conv1a.bn.bias", {c1});
  add_spec(m, "backbone.conv1a.bn.running_mean", {c1});
  add_spec(m, "backbone.conv1a.bn.running_var", {c1});
  add_cbr(m, "backbone.conv1b", c1, c2);
  add_cbr(m, "backbone.conv2a", c2, c2);
  add_cbr(m, "backbone.conv2b", c2, c3);
  add_cbr(m, "backbone.conv3a", c3, c3);
  add_cbr(m, "backbone.conv3b", c3, c4);
  add_cbr(m, "backbone.conv4a", c4, c4);
  add_cbr(m, "backbone.conv4b", c4, c4);
  if (v3) {
    add_cbr(m, "score_loc_head.convDa", c4, c4);
    add_conv(m, "score_loc_head.convDb", c4, 3);
  } else {
    add_cbr(m, "score_head.convDa", c4, c4);
    add_conv(m, "score_head.convDb", c4, 1);
    add_cbr(m, "loc_head.convDa", c4, c4);
    add_conv(m, "loc_head.convDb", c4, 2);
    const bool tc0 = g.upscale_method == KP2D_UP_CONVTRANSPOSE;
    if (tc0) add_tconv(m, "desc_head.upsample", c3 * 4);   // registered first (heads.py:55-56)
    add_cbr(m, "desc_head.convA", c4, c4);
    add_conv(m, "desc_head.convB", c4, c3 * 4, /*shuffle=*/!tc0);
    add_cbr(m, "desc_head.confAa", c3 + c4, c4);
    add_conv(m, "desc_head.confBb", c4, g.nfeatures);
  }
  const int ch = c5, cexp = c4 + c3;
  const int last_in = v3 ? ch / 2 : ch;
  // V3 depth: the last CBR is half a width wider and a third 3x3 conv (featD, no bias) reads the middle slice
  const int trunk_out = (v3 && g.depth) ? ch + ch / 2 : ch;
  if (g.use_attention && (ch > 256 || (ch % 16)))
    return fail(KP2D_ERR_UNSUPPORTED, "attention width %d (built for <= 256, multiple of 16)", ch);
  const bool tc = g.upscale_method == KP2D_UP_CONVTRANSPOSE;
  if (tc && (d1 % 16)) return fail(KP2D_ERR_UNSUPPORTED, "convtranspose upsampling needs channel_dims[5] %% 16 == 0");
  auto seg_like_head = [&](const std::string& P_, int c_out, int width) {
    const std::string L = P_ + ".convs.";
    if (g.use_attention) {
      add_cbr(m, L + "0", c4, ch);
      add_attention_module(m, L + "1", ch);
      add_attention_module(m, L + "2", ch);
      add_cbr(m, L + "3", ch, d1, !tc);
      add_cbr(m, L + "4", ch + d1 / 4, ch);
      add_cbr(m, L + "5", ch, d1, !tc);
      add_cbr(m, L + "6", cexp, width);
      add_conv(m, L + "7", P_ == "seg_head" ? last_in : ch, c_out);
    } else {
      add_cbr(m, L + "0", c4, ch);
      add_cbr(m, L + "1", ch, ch);
      add_cbr(m, L + "2", ch, ch);
      add_cbr(m, L + "3", ch, ch);
      add_cbr(m, L + "4", ch, d1, !tc);
      add_cbr(m, L + "5", ch + d1 / 4, ch);
      add_cbr(m, L + "6", ch, d1, !tc);
      add_cbr(m, L + "7", cexp, width);
      add_conv(m, L + "8", P_ == "seg_head" ? last_in : ch, c_out);
    }
  };
  auto upsamplers = [&](const std::string& P_) {   // registered after convs / featB / featD (segmentation.py:113-118)
    if (tc) { add_tconv(m, P_ + ".upsample", d1); add_tconv(m, P_ + ".upsample2", d1); }
  };
  seg_like_head("seg_head", g.n_classes, trunk_out);
  if (!v3) upsamplers("seg_head");
  if (v3) {
    add_conv(m, "seg_head.featB", ch / 2, g.nfeatures);
    if (g.depth) {   // Conv2d(dim_split, 1, bias=False): segmentation.py:281-284
      add_spec(m, "seg_head.featD.weight", {1, ch / 2, 3, 3});
      ConvPack c;
      c.name = "seg_head.featD"; c.bn = false; c.bias = false; c.cin = ch / 2; c.cout = 1;
      m->conv_index[c.name] = (int)m->convs.size();
      m->convs.push_back(c);
    }
    upsamplers("seg_head");
  } else if (g.depth) {
    seg_like_head("depth_head", 1, ch);   // kp2dtiny.py:402-437: a second full segmentation head with one output
    upsamplers("depth_head");
  }
  add_cbr(m, "vlad_head.convlad1", c4, g.encoder_dim);
  add_cbr(m, "vlad_head.convlad2", g.encoder_dim, g.encoder_dim);
  add_cbr(m, "vlad_head.convlad3", g.encoder_dim, g.encoder_dim);
  {
    // The first CBR of every head reads the same backbone map: one launch computes them all (rows of the parts
    // back to back), the heads then read channel slices of its output.  The attention seg heads keep their own
    // launch (their first CBR feeds a LayerNorm, which wants a dense tensor).
    ConvPack mg;
    mg.name = "heads.first"; mg.bn = true; mg.cin = c4;
    auto part = [&](const std::string& n) { mg.parts.emplace_back(n, m->convs[m->conv_index.at(n)].cout); mg.cout += mg.parts.back().second; };
    if (v3) part("score_loc_head.convDa");
    else { part("score_head.convDa"); part("loc_head.convDa"); part("desc_head.convA"); }
    if (!g.use_attention && m->conv_index.count("seg_head.convs.0")) part("seg_head.convs.0");
    part("vlad_head.convlad1");
    if (mg.parts.size() >= 2) { m->conv_index[mg.name] = (int)m->convs.size(); m->convs.push_back(mg); }
  }
  const bool has_vlad = g.global_descriptor == KP2D_GD_NETVLAD && !g.remove_netvlad;
  if (has_vlad) {
    add_spec(m, "vlad_head.netvlad.centroids", {g.num_clusters, g.encoder_dim});
    add_spec(m, "vlad_head.netvlad.conv.weight", {g.num_clusters, g.encoder_dim, 1, 1});
  } else if (g.global_descriptor == KP2D_GD_GEM) {
    add_spec(m, "vlad_head.netvlad.p", {1});
    m->vecs["vlad_head.netvlad.p"].n = 1;
  } else if (g.global_descriptor == KP2D_GD_CONVAP) {
    add_pw(m, "vlad_head.netvlad.channel_pool", g.encoder_dim, g.encoder_dim, true, 1);
  }

  // blob layout
  size_t off = 0;
  auto take = [&](size_t floats) { size_t o = off; off = align_up(off + floats, ALIGN / 4); return o; };
  m->conv1a_w = take((size_t)9 * g.in_channels * c1);
  m->conv1a_sc = take(c1);
  m->conv1a_sh = take(c1);
  m->conv1a_ws = take(1);
  for (auto& c : m->convs) {
    if (c.cin % 4) return fail(KP2D_ERR_UNSUPPORTED, "%s: input channels %d not a multiple of 4", c.name.c_str(), c.cin);
    if (c.shuffle && (c.cout % 16)) return fail(KP2D_ERR_UNSUPPORTED, "%s: pixel-shuffle conv needs cout %% 16 == 0", c.name.c_str());
    c.kc = (c.cin % 16 == 0) ? 16 : ((c.cin % 8 == 0) ? 8 : 16);
    c.npad = c.cout <= 32 ? 32 : (c.cout + 63) / 64 * 64;
    c.w_off = take(c.w_floats());
    c.sc_off = take(c.npad);
    c.sh_off = take(c.npad);
    c.w16_off = take(c.w16_floats());
    if (c.npad >= 64) c.w16n_off = take(c.w16_floats());
    if (c.npad >= 64 && c.kind == 0 && c.taps == 9) c.w16t_off = take(c.w16_floats());
    c.sc16_off = take(c.npad);
    if (c.head()) c.wd_off = take(c.wd_floats());
  }
  for (auto& kv : m->vecs) kv.second.off = take(kv.second.n);
  if (has_vlad) {
    m->vlad_wa = take((size_t)g.num_clusters * g.encoder_dim);
    m->vlad_cent = take((size_t)g.num_clusters * g.encoder_dim);
  }
  m->blob_floats = off;
  return KP2D_OK;
}

const std::vector<float>* host_get(const kp2d_model* m, const std::string& key) {
  auto it = m->host.find(key);
  return it == m->host.end() ? nullptr : &it->second;
}

// BatchNorm2d eval: y = (x - mean) / sqrt(var + 1e-5) * gamma + beta  ->  y = x * scale + shift
void bn_fold(const kp2d_model* m, const std::string& p, int co, float* scale, float* shift) {
  const auto& g = *host_get(m, p + ".weight");
  const auto& b = *host_get(m, p + ".bias");
  const auto& mu = *host_get(m, p + ".running_mean");
  const auto& var = *host_get(m, p + ".running_var");
  for (int c = 0; c < co; ++c) {
    const float s = g[c] / std::sqrt(var[c] + 1e-5f);
    scale[c] = s;
    shift[c] = b[c] - mu[c] * s;
  }
}

int pack(kp2d_model* m, std::vector<float>& blob) {
  for (const auto& s : m->specs)
    if (!host_get(m, s.key)) return fail(KP2D_ERR_WEIGHT, "missing tensor '%s'", s.key.c_str());
  blob.assign(m->blob_floats, 0.f);
  const int c1 = m->c1;
  {
    const auto& w = *host_get(m, "backbone.conv1a.conv.weight");   // [c1][cin][3][3]
    const int nk = 9 * m->cfg.in_channels;
    for (int co = 0; co < c1; ++co)
      for (int k = 0; k < nk; ++k) blob[m->conv1a_w + (size_t)k * c1 + co] = w[(size_t)co * nk + k];
    bn_fold(m, "backbone.conv1a.bn", c1, &blob[m->conv1a_sc], &blob[m->conv1a_sh]);
    // the fused first layer splits these weights as w 2^e (conv3x3_f16.hip STEM): e as for every other layer
    float wmax = 0.f;
    for (float v : w) wmax = std::max(wmax, std::fabs(v));
    int e16 = 11;
    while (e16 > -96 && wmax * std::ldexp(1.0f, e16) > 32768.0f) --e16;
    blob[m->conv1a_ws] = std::ldexp(1.0f, e16);
  }
  for (const auto& c : m->convs) {
    std::vector<float> wvirt;
    if (c.tconv) {
      // virtual 3x3 weight [4*co + 2a + b][ci][ty][tx] of the transposed convolution (see add_tconv)
      const auto& wt = *host_get(m, c.name + ".transposed_conv.weight");   // [ci][co][ky][kx]
      const int cq4 = c.cout / 4;
      wvirt.assign((size_t)c.cout * c.cin * 9, 0.f);
      auto kmap = [](int par, int t) { return par == 0 ? (t == 1 ? 1 : -1) : (t == 1 ? 2 : (t == 2 ? 0 : -1)); };
      for (int co = 0; co < cq4; ++co)
        for (int a = 0; a < 2; ++a)
          for (int b = 0; b < 2; ++b)
            for (int ci = 0; ci < c.cin; ++ci)
              for (int ty = 0; ty < 3; ++ty)
                for (int tx = 0; tx < 3; ++tx) {
                  const int ky = kmap(a, ty), kx = kmap(b, tx);
                  if (ky < 0 || kx < 0) continue;
                  wvirt[((size_t)(4 * co + 2 * a + b) * c.cin + ci) * 9 + ty * 3 + tx] =
                      wt[(((size_t)ci * cq4 + co) * 3 + ky) * 3 + kx];
                }
    }
    std::vector<float> sc(c.cout), sh(c.cout);
    if (!c.parts.empty()) {   // rows of the parts, back to back
      wvirt.reserve((size_t)c.cout * c.cin * 9);
      int row = 0;
      for (const auto& pt : c.parts) {
        const auto& wp = *host_get(m, pt.first + ".conv.weight");
        wvirt.insert(wvirt.end(), wp.begin(), wp.end());
        bn_fold(m, pt.first + ".bn", pt.second, sc.data() + row, sh.data() + row);
        row += pt.second;
      }
    }
    const auto& w = (c.tconv || !c.parts.empty()) ? wvirt : *host_get(m, c.name + (c.bn ? ".conv.weight" : ".weight"));   // [cout][ci][k][k]
    if (!c.parts.empty()) {
    } else if (c.tconv) {
      std::vector<float> s4(c.cout / 4), h4(c.cout / 4);
      bn_fold(m, c.name + ".bn", c.cout / 4, s4.data(), h4.data());
      for (int i = 0; i < c.cout; ++i) { sc[i] = s4[i / 4]; sh[i] = h4[i / 4]; }
    } else if (c.bn) {
      bn_fold(m, c.name + ".bn", c.cout, sc.data(), sh.data());
    } else {
      const std::vector<float>* b = c.bias ? host_get(m, c.name + ".bias") : nullptr;
      for (int i = 0; i < c.cout; ++i) { sc[i] = 1.f; sh[i] = b ? (*b)[i] : 0.f; }
    }
    const int ng = c.npad <= 32 ? 32 : 64;          // channels per workgroup group
    const int ngroups = c.npad / ng;
    const int nchunk = (c.cin + c.kc - 1) / c.kc;
    const int cq = c.cout / 4;
    for (int q = 0; q < c.npad; ++q) {
      // packed position q -> original output channel (PixelShuffle: out[c,2h+i,2w+j] = in[4c+2i+j,h,w])
      int co = -1;
      if (q < c.cout) co = c.shuffle ? 4 * (q % cq) + (q / cq) : q;
      blob[c.sc_off + q] = co >= 0 ? sc[co] : 0.f;
      blob[c.sh_off + q] = co >= 0 ? sh[co] : 0.f;
      if (co < 0) continue;
      const int grp = q / ng, n = q % ng;
      for (int ci = 0; ci < c.cin; ++ci) {
        const int chk = ci / c.kc, kk = ci % c.kc;
        if (c.kind == 0) {
          for (int tap = 0; tap < 9; ++tap) {
            const size_t dst = c.w_off + ((((size_t)grp * nchunk + chk) * 9 + tap) * ng + n) * c.kc + kk;
            blob[dst] = w[((size_t)co * c.cin + ci) * 9 + tap];
          }
        } else {
          const size_t dst = c.w_off + (((size_t)grp * nchunk + chk) * ng + n) * c.kc + kk;
          if (c.kind == 1) {
            blob[dst] = w[(size_t)co * c.cin + ci];
          } else {
            // GEMM k = dy*2C + dx*C + cc  <-  weight[co][cc][dy][dx]   (C = cin/4)
            const int Cq = c.cin / 4, dy = ci / (2 * Cq), dx = (ci / Cq) & 1, cc = ci % Cq;
            blob[dst] = w[(((size_t)co * Cq + cc) * 2 + dy) * 2 + dx];
          }
        }
      }
    }
    (void)ngroups;
    if (c.head())   // dot-product form of the head layers: [chunk][tap][4][16], zero rows / columns as padding
      for (int co = 0; co < c.cout; ++co)
        for (int ci = 0; ci < c.cin; ++ci)
          for (int tap = 0; tap < 9; ++tap)
            blob[c.wd_off + ((((size_t)(ci / 16) * 9 + tap) * 4 + co) * 16) + ci % 16] = w[((size_t)co * c.cin + ci) * 9 + tap];
    // split-fp16 pack (conv3x3.hip PREC 1): K walked in chunks of 16; each row is 16 hi halves then 16 lo
    // halves of w * 2^e; the epilogue scale carries the 2^-e.  e = 11 keeps the lo half of ordinary weights a normal
    // fp16; a layer with large weights (|w| * 2^11 would pass the fp16 range: |w| >= 16) takes the largest e that keeps
    // |w| * 2^e <= 2^15, so no checkpoint can turn a weight into inf (hi) / -inf (lo) silently.  Powers of two: the
    // products and the fp32 accumulation are the same bits up to the exponent, whatever e is.
    {
      float wmax = 0.f;
      for (float v : w) {
        if (!std::isfinite(v)) return fail(KP2D_ERR_WEIGHT, "%s: non-finite weight value", c.name.c_str());
        wmax = std::max(wmax, std::fabs(v));
      }
      int e16 = 11;
      while (e16 > -96 && wmax * std::ldexp(1.0f, e16) > 32768.0f) --e16;
      const float wscale = std::ldexp(1.0f, e16), wunscale = std::ldexp(1.0f, -e16);
      _Float16* h16 = reinterpret_cast<_Float16*>(&blob[c.w16_off]);
      const int nchunk16 = (c.cin + 15) / 16;
      for (int q = 0; q < c.npad; ++q) {
        int co = -1;
        if (q < c.cout) co = c.shuffle ? 4 * (q % cq) + (q / cq) : q;
        blob[c.sc16_off + q] = co >= 0 ? sc[co] * wunscale : 0.f;
        if (co < 0) continue;
        const int grp = q / ng, n = q % ng;
        for (int ci = 0; ci < c.cin; ++ci) {
          const int chk = ci / 16, kk = ci % 16;
          for (int tap = 0; tap < c.taps; ++tap) {
            float wv;
            if (c.kind == 0) wv = w[((size_t)co * c.cin + ci) * 9 + tap];
            else if (c.kind == 1) wv = w[(size_t)co * c.cin + ci];
            else {
              const int Cq = c.cin / 4, dy = ci / (2 * Cq), dx = (ci / Cq) & 1, cc = ci % Cq;
              wv = w[(((size_t)co * Cq + cc) * 2 + dy) * 2 + dx];
            }
            wv *= wscale;
            const _Float16 hi = (_Float16)wv;
            const _Float16 lo = (_Float16)(wv - (float)hi);
            // 3x3 layers: the nine taps of a chunk sit in slot order {0,1,3,4,2,5,6,7,8} (conv3x3_f16.hip pairs
            // slots (0,1) (2,3) (4,5) (6,7) into one K = 32 MFMA each; slot 8 is the single)
            static const int kSlot[9] = {0, 1, 4, 2, 3, 5, 6, 7, 8};
            const int slot = c.kind == 0 ? kSlot[tap] : tap;
            const size_t row = ((((size_t)grp * nchunk16 + chk) * c.taps + slot) * ng + n) * 32;   // in halves
            h16[row + kk] = hi;
            h16[row + 16 + kk] = lo;
            if (c.npad >= 64) {   // 32-channel groups of the same rows
              _Float16* n16 = reinterpret_cast<_Float16*>(&blob[c.w16n_off]);
              const size_t rown = ((((size_t)(q / 32) * nchunk16 + chk) * c.taps + slot) * 32 + (q % 32)) * 32;
              n16[rown + kk] = hi;
              n16[rown + 16 + kk] = lo;
            }
            if (c.w16t_off) {     // tap (dy, dx) in the slot of tap (dx, dy): what a tile that walks the map transposed multiplies
              _Float16* t16 = reinterpret_cast<_Float16*>(&blob[c.w16t_off]);
              const int slot_t = kSlot[3 * (tap % 3) + tap / 3];
              const size_t rowt = ((((size_t)grp * nchunk16 + chk) * c.taps + slot_t) * ng + n) * 32;
              t16[rowt + kk] = hi;
              t16[rowt + 16 + kk] = lo;
            }
          }
        }
      }
    }
  }
  for (const auto& kv : m->vecs) {
    const auto& src = *host_get(m, kv.first);
    const std::string& key = kv.first;
    if (key.size() > 13 && key.compare(key.size() - 13, 13, ".net.0.weight") == 0 && key.find(".net.1.") != std::string::npos) {
      const int h = kv.second.n / 9;                       // depthwise [h][1][3][3] -> [9][h]
      for (int c = 0; c < h; ++c)
        for (int t = 0; t < 9; ++t) blob[kv.second.off + (size_t)t * h + c] = src[(size_t)c * 9 + t];
    } else {
      std::copy(src.begin(), src.end(), blob.begin() + kv.second.off);
    }
  }
  if (m->cfg.global_descriptor == KP2D_GD_NETVLAD && !m->cfg.remove_netvlad) {
    const auto& wa = *host_get(m, "vlad_head.netvlad.conv.weight");
    const auto& ce = *host_get(m, "vlad_head.netvlad.centroids");
    std::copy(wa.begin(), wa.end(), blob.begin() + m->vlad_wa);
    std::copy(ce.begin(), ce.end(), blob.begin() + m->vlad_cent);
  }
  return KP2D_OK;
}

// ------------------------------------------------------------------------------------------------
// launch plan
// ------------------------------------------------------------------------------------------------
struct Plan {
  kp2d_model* m;
  hipStream_t stream;
  char* ws;
  Arena arena;
  bool dry = false;       // only size the arena
  int B, H, W;
  int b0 = 0;             // first frame of this sub-batch in the caller's batch
  const float* seg_ptr = nullptr;   // this sub-batch's slice of the caller's seg output and of the class-id map
  long long* seg_ids = nullptr;     // (kp2d_set_seg_ids): the layer that writes seg also writes its per-pixel argmax
  int nlanes = 1;         // stream lanes of this forward (conv3x3_wsm.hip sizes its grid by it)
  int rc = KP2D_OK;
  // Independent layers of one level as ONE launch (conv3x3_f16.hip::conv3x3_f16x3_multi_kernel; small grids only): between
  // group_begin() and group_end() the 3x3 split-fp16 launches are collected instead of enqueued.  Their inputs must not be
  // released — and no tap taken — before group_end(): the caller's job (build()).
  const float* stem_x = nullptr;   // the frames, when conv1b's launch computes conv1a itself (build())
  bool grouping = false;
  bool no_levels = false;   // dry runs: size the head-by-head schedule too (plan_bytes_uncached takes the larger)
  std::vector<ConvArgs> pending;
  std::vector<std::string> pending_names;
  void group_begin() {
    if (dry || rc != KP2D_OK || m->profiling || m->tap_dst || !m->multi_launch) return;      // (profiles and taps: one launch per layer)
    grouping = true;
  }
  void group_end() {
    grouping = false;
    if (pending.empty()) return;
    int e = -1000;
    if (pending.size() >= 2 && rc == KP2D_OK) e = launch_conv3x3_f16x3_multi(pending.data(), (int)pending.size(), stream);
    if (e == -1000) {
      for (size_t i = 0; i < pending.size() && rc == KP2D_OK; ++i)
        check(launch_conv3x3(pending[i], 16, stream), pending_names[i].c_str());
    } else {
      check(e, pending_names[0].c_str());
    }
    pending.clear();
    pending_names.clear();
  }

  // kp2d_set_tap: copy activation `a` (this sub-batch's frames) to the caller's planar [B,C,H,W] buffer
  void tap(const std::string& name, const Act& a) {
    if (dry || rc != KP2D_OK || !m->tap_dst || name != m->tap_name) return;
    const size_t per = (size_t)a.C * a.H * a.W;
    if (((size_t)b0 + B) * per > m->tap_cap) { rc = fail(KP2D_ERR_ARG, "tap '%s': buffer holds %zu floats, needs %zu", name.c_str(), m->tap_cap, ((size_t)b0 + B) * per); return; }
    if (a.fmt == 1) check(launch_s16p_to_nchw(ptr(a), m->tap_dst + (size_t)b0 * per, B, a.C, a.H, a.W, a.PS ? a.PS : a.C, a.CO, stream), name.c_str());
    else check(launch_nhwc_to_nchw(ptr(a), m->tap_dst + (size_t)b0 * per, B, a.C, a.H * a.W, a.PS ? a.PS : a.C, a.CO, stream), name.c_str());
  }

  Act alloc(int C, int H_, int W_) {
    Act a;
    a.C = C; a.H = H_; a.W = W_;
    a.bytes = (size_t)B * H_ * W_ * C * sizeof(float);
    a.off = arena.alloc(a.bytes);
    if (a.off == (size_t)-1 && rc == KP2D_OK) rc = fail(KP2D_ERR_WORKSPACE, "workspace exhausted");
    return a;
  }
  void release(const Act& a) { if (a.bytes) arena.release(a.off, a.bytes); }
  static Act view(const Act& parent, int c, int o) {   // channels [o, o + c) of parent; released by releasing the parent
    Act v = parent;
    v.bytes = 0; v.C = c; v.PS = parent.PS ? parent.PS : parent.C; v.CO = parent.CO + o;
    return v;
  }
  float* ptr(const Act& a) const { return reinterpret_cast<float*>(ws + a.off); }

  void prof_begin(const std::string& layer, const char* kernel, double flops, double bytes) {
    if (!m->profiling || dry) return;
    if (m->prof_used == m->prof.size()) {
      ProfRec r;
      (void)hipEventCreate(&r.e0);
      (void)hipEventCreate(&r.e1);
      m->prof.push_back(r);
    }
    ProfRec& r = m->prof[m->prof_used];
    r.layer = layer; r.kernel = kernel; r.flops = flops; r.bytes = bytes;
    (void)hipEventRecord(r.e0, stream);
  }
  void prof_end() {
    if (!m->profiling || dry) return;
    (void)hipEventRecord(m->prof[m->prof_used].e1, stream);
    ++m->prof_used;
  }
  void check(int e, const char* what) {
    if (e != 0 && rc == KP2D_OK) rc = fail(KP2D_ERR_HIP, "%s: launch failed (%d: %s)", what, e,
                                           e > 0 ? hipGetErrorString((hipError_t)e) : "unsupported shape");
  }

  static ConvSrc dense(const float* p, const Act& t, int c, int o) {
    ConvSrc s{};
    const int ps = t.PS ? t.PS : t.C;
    s.p = p; s.c = c; s.o = o + t.CO;
    s.ps = ps; s.rs = (long)t.W * ps; s.bs = (long)t.H * t.W * ps;
    s.fmt = t.fmt;
    return s;
  }
  // arguments of one conv launch; false (rc set) when the plan and the layer disagree
  bool conv_args(const std::string& name, const ConvSrc& s0, const ConvSrc& s1, int act, int store, float* out0, int os0,
                 int oo0, float* out1, int os1, int oo1, int nsplit, int Hc, int Wc, ConvArgs& a) {
    const ConvPack& c = m->convs[m->conv_index.at(name)];
    a = ConvArgs{};
    a.in0 = s0; a.in1 = s1; a.taps = c.taps;
    const bool split = m->precision == KP2D_PREC_F16X3;
    { static const int dbg = getenv("KP2D_DBG") ? atoi(getenv("KP2D_DBG")) : 0; a.dbg = dbg; }
    a.prec = split ? 1 : 0;
    a.ids_out = (store == ST_NCHW && seg_ids && out0 && out0 == seg_ptr && nsplit == c.cout && c.npad == 32) ? seg_ids : nullptr;
    a.wsm_min = m->wsm_min;
    a.wsm_grid = m->wsm_grid;
    a.wsm_tr = m->wsm_tr;
    a.ws_min = m->ws_min;
    a.wsm_lanes = nlanes;
    a.s16_min = m->s16_min;
    // S16P tensors beyond the 32-channel stage are read and written by conv3x3_wsm.hip only: build() fixed the layout
    // after asking conv3x3_wsm_would_run, the launcher then skips its item-count policy
    a.wsm_force = ((s0.fmt == 1 && !(c.cin == 32 && s1.c == 0) && store != ST_NCHW) || store == ST_S16P_SHUFFLE || store == ST_MIX16 ||
                   (store == ST_S16P && c.npad >= 64)) ? 1 : 0;
    if (stem_x && name == "backbone.conv1b") {
      a.stem_x = stem_x; a.stem_w = m->blob + m->conv1a_w; a.stem_scale = m->blob + m->conv1a_sc; a.stem_shift = m->blob + m->conv1a_sh;
      a.stem_wscale = m->blob + m->conv1a_ws; a.stem_act = m->cfg.leaky_relu ? ACT_LEAKY : ACT_RELU;
    }
    a.w = m->blob + (split ? c.w16_off : c.w_off);
    a.w_tr = (split && c.w16t_off) ? m->blob + c.w16t_off : nullptr;
    a.tiles_x = (Wc + 15) / 16; a.tiles_y = (Hc + 15) / 16;
    // Small grids (a frame or two at a time): a 64-channel-group launch would leave most CUs idle and each of its
    // few workgroups is a long serial chain; 32-channel groups double the workgroups and halve their length.
    // (a forced warp-specialised form — kp2d_set_option("wsm_min_items"), the parity tests — keeps its 64-channel groups)
    const bool wsm_forced = m->wsm_min > 0 && (long)((Wc + 31) / 32) * ((Hc + 15) / 16) * B * (c.npad / 64) >= m->wsm_min;
    if (split && c.npad >= 64 && m->small_grid_ng32 && !wsm_forced && s0.fmt == 0 && !a.wsm_force &&      // (S16P tensors: 64-channel groups)
        (long)a.tiles_x * a.tiles_y * B * (c.npad / 64) < 256) {
      a.w = m->blob + c.w16n_off;
      a.ng32 = 1;
    }
    a.scale = m->blob + (split ? c.sc16_off : c.sc_off);
    a.shift = m->blob + c.sh_off;
    a.out0 = out0; a.os0 = os0; a.oo0 = oo0; a.out1 = out1; a.os1 = os1; a.oo1 = oo1;
    a.B = B; a.H = Hc; a.W = Wc; a.cin = c.cin; a.cout = c.cout; a.npad = c.npad;
    a.act = act; a.store = store; a.nsplit = nsplit;
    if (s0.c + s1.c != c.cin) { rc = fail(KP2D_ERR_ARG, "%s: plan feeds %d channels, layer expects %d", name.c_str(), s0.c + s1.c, c.cin); return false; }
    return true;
  }
  // score / loc / depth heads: 1-4 output channels as an HBM-bound dot-product kernel (exact fp32 in both modes)
  bool head_dot(const ConvPack& c, const ConvArgs& a) const {
    return c.head() && a.store == ST_NCHW && a.in1.c == 0 && a.act != ACT_SOFTMAX_C && m->head_dot;
  }
  void head_dot_args(const ConvPack& c, ConvArgs& a) const {
    a.prec = 0;
    a.w = m->blob + c.wd_off;
    a.scale = m->blob + c.sc_off;
  }
  // core launch: sources already described
  void conv_src(const std::string& name, const ConvSrc& s0, const ConvSrc& s1, int act, int store, float* out0, int os0,
                int oo0, float* out1, int os1, int oo1, int nsplit, int Hc, int Wc) {
    if (rc != KP2D_OK || dry) return;
    const ConvPack& c = m->convs[m->conv_index.at(name)];
    ConvArgs a;
    if (!conv_args(name, s0, s1, act, store, out0, os0, oo0, out1, os1, oo1, nsplit, Hc, Wc, a)) return;
    const bool split = m->precision == KP2D_PREC_F16X3;
    const double px = (double)B * Hc * Wc;
    if (head_dot(c, a)) {
      head_dot_args(c, a);
      prof_begin(name, "conv3x3_head", 2.0 * 9 * c.cin * c.cout * px, 4.0 * px * (c.cin + c.cout) + 4.0 * 9 * c.cin * c.cout);
      check(launch_head3x3(a, stream), name.c_str());
      prof_end();
      return;
    }
    if (grouping && split && c.taps == 9 && pending.size() < 4) {
      pending.push_back(a);
      pending_names.push_back(name);
      return;
    }
    const char* fam = split ? (c.taps == 9 ? "conv3x3_f16x3" : "conv1x1_f16x3")
                            : (c.taps == 9 ? (c.kc == 16 ? "conv3x3_f32<16>" : "conv3x3_f32<8>") : "conv1x1_f32");
    if (a.stem_x)      // conv1a computed inside this launch: its products count, its input is the 3-channel frame
      prof_begin(name, fam, 2.0 * 9 * (3.0 * 16 + c.cin * c.cout) * px, 4.0 * px * (3 + c.cout / 4.0) + 4.0 * 9 * c.cin * c.cout);
    else
    prof_begin(name, fam, 2.0 * c.taps * c.cin * c.cout * px, 4.0 * px * (c.cin + c.cout) + 4.0 * c.taps * c.cin * c.cout);
    check(launch_conv3x3(a, split ? 16 : c.kc, stream), name.c_str());
    if (m->profiling && !dry) m->prof[m->prof_used].kernel += conv3x3_last_variant();      // which tile form ran
    prof_end();
  }
  // KP2DTinyV2's score head (-> 1 channel, sigmoid) and location head (-> 2, tanh): planar outputs, one launch for both
  // when both run as dot-product kernels (per-layer profiling keeps them apart)
  void head_pair(const std::string& n0, const Act& in0, int act0, float* out0, const std::string& n1, const Act& in1, int act1,
                 float* out1, int Hc, int Wc) {
    if (rc != KP2D_OK || dry) return;
    const ConvPack& c0 = m->convs[m->conv_index.at(n0)];
    const ConvPack& c1 = m->convs[m->conv_index.at(n1)];
    ConvArgs a0, a1;
    const ConvSrc none0 = dense(ptr(in0), in0, 0, 0), none1 = dense(ptr(in1), in1, 0, 0);
    if (!conv_args(n0, dense(ptr(in0), in0, in0.C, 0), none0, act0, ST_NCHW, out0, 0, 0, nullptr, 0, 0, c0.cout, Hc, Wc, a0)) return;
    if (!conv_args(n1, dense(ptr(in1), in1, in1.C, 0), none1, act1, ST_NCHW, out1, 0, 0, nullptr, 0, 0, c1.cout, Hc, Wc, a1)) return;
    // few frames only: at 64 frames the two launches overlap their tails and the pair is 0.4 % of the step slower
    // (21.16k vs 21.24k frames/s, three alternating runs); at one frame it saves a 4-us launch (0.273 -> 0.265 ms)
    static const bool pair_env = !(getenv("KP2D_HEAD_PAIR") && getenv("KP2D_HEAD_PAIR")[0] == '0');      // (A/B knob)
    const bool pair_on = pair_env && (long)((Wc + 15) / 16) * ((Hc + 3) / 4) * B < 1024;
    if (!pair_on || m->profiling || !head_dot(c0, a0) || !head_dot(c1, a1) || c0.cout != 1 || c1.cout != 2) {
      conv(n0, in0, in0.C, 0, nullptr, act0, ST_NCHW, out0, 0, 0, nullptr, 0, 0, c0.cout, Hc, Wc);
      conv(n1, in1, in1.C, 0, nullptr, act1, ST_NCHW, out1, 0, 0, nullptr, 0, 0, c1.cout, Hc, Wc);
      return;
    }
    head_dot_args(c0, a0);
    head_dot_args(c1, a1);
    check(launch_head3x3_pair(a0, a1, stream), n0.c_str());
  }
  // generic conv over dense NHWC activations: in1 may be null (no concat).  Channel slices via (c0, o0).
  void conv(const std::string& name, const Act& in0, int c0, int o0, const Act* in1, int act, int store,
            float* out0, int os0, int oo0, float* out1, int os1, int oo1, int nsplit, int Hc, int Wc) {
    if (rc != KP2D_OK || dry) return;
    ConvSrc s0 = dense(ptr(in0), in0, c0, o0);
    ConvSrc s1 = in1 ? dense(ptr(*in1), *in1, in1->C, 0) : dense(ptr(in0), in0, 0, 0);
    conv_src(name, s0, s1, act, store, out0, os0, oo0, out1, os1, oo1, nsplit, Hc, Wc);
  }
  // 1x1 conv -> NHWC activation
  Act pw(const std::string& name, const Act& in, int act, int store = ST_NHWC) {
    const ConvPack& c = m->convs[m->conv_index.at(name)];
    Act out{};
    if (store == ST_NHWC_POOL) {
      out = alloc(c.cout, in.H / 2, in.W / 2);
      conv(name, in, in.C, 0, nullptr, act, store, nullptr, 0, 0, dry ? nullptr : ptr(out), c.cout, 0, 0, in.H, in.W);
    } else {
      out = alloc(c.cout, in.H, in.W);
      conv(name, in, in.C, 0, nullptr, act, ST_NHWC, dry ? nullptr : ptr(out), c.cout, 0, nullptr, 0, 0, 0, in.H, in.W);
    }
    return out;
  }
  Act layernorm(const std::string& prefix, const Act& in) {
    Act out = alloc(in.C, in.H, in.W);
    if (rc != KP2D_OK || dry) return out;
    LnArgs a{ptr(in), m->blob + m->vecs.at(prefix + ".g").off, m->blob + m->vecs.at(prefix + ".b").off, ptr(out),
             (long)B * in.H * in.W, in.C};
    const double px = (double)B * in.H * in.W;
    prof_begin(prefix, "channel_layernorm", 8.0 * px * in.C, 8.0 * px * in.C);
    check(launch_channel_layernorm(a, stream), prefix.c_str());
    prof_end();
    return out;
  }
  // SegFormerAttentionModule.forward (modules/segformer.py:217-220); `pool` folds the following MaxPool2d(2,2)
  Act attention_module(const std::string& p, const Act& x, bool pool) {
    const int C = x.C, h = x.H, w = x.W;
    Act ln1 = layernorm(p + ".att.norm", x);
    Act q = pw(p + ".att.fn.to_q", ln1, ACT_NONE);
    Act kv = alloc(2 * C, h / 2, w / 2);
    if (rc == KP2D_OK && !dry) {
      // 2x2 stride-2 conv == 1x1 conv over [row 2Y | row 2Y+1], each row-view a 2C-channel "pixel" (x, x+1)
      ConvSrc s0{};
      s0.p = ptr(ln1); s0.c = 2 * C; s0.o = 0; s0.ps = 2 * C; s0.rs = 2L * w * C; s0.bs = (long)h * w * C;
      ConvSrc s1 = s0;
      s1.p = ptr(ln1) + (long)w * C;
      conv_src(p + ".att.fn.to_kv", s0, s1, ACT_NONE, ST_NHWC, ptr(kv), 2 * C, 0, nullptr, 0, 0, 0, h / 2, w / 2);
    }
    release(ln1);
    Act ao = alloc(C, h, w);
    if (rc == KP2D_OK && !dry) {
      const int heads = 4;
      AttnArgs a{ptr(q), ptr(kv), ptr(ao), B, h * w, (h / 2) * (w / 2), C, heads, 1.0f / std::sqrt((float)(C / heads))};
      a.prec = m->precision == KP2D_PREC_F16X3 ? 1 : 0;
      const double st = (double)B * h * w * (h / 2) * (w / 2);
      prof_begin(p + ".att.fn", (a.prec == 1 && C / heads <= 16) ? "attention_f16x3" : "attention", 4.0 * st * C, 4.0 * B * ((double)2 * h * w * C + (h / 2) * (w / 2) * 2.0 * C));
      check(launch_attention(a, stream), (p + ".att.fn").c_str());
      prof_end();
    }
    release(q);
    release(kv);
    Act t = pw(p + ".att.fn.to_out", ao, ACT_NONE);
    release(ao);
    tap(p + ".att", t);
    Act ln2 = layernorm(p + ".mff.norm", t);
    release(t);
    Act f0 = pw(p + ".mff.fn.net.0", ln2, ACT_NONE);
    release(ln2);
    if (mff_fusable(C) && f0.C == 128) {
      Act f3 = mff_tail(p, f0, C, h, w, pool);
      release(f0);
      tap(p + ".mff", f3);
      return f3;
    }
    Act f1 = alloc(f0.C, h, w);
    if (rc == KP2D_OK && !dry) {
      DwArgs a{ptr(f0), m->blob + m->vecs.at(p + ".mff.fn.net.1.net.0.weight").off,
               m->blob + m->vecs.at(p + ".mff.fn.net.1.net.0.bias").off, ptr(f1), B, h, w, f0.C};
      const double px = (double)B * h * w;
      prof_begin(p + ".mff.fn.net.1.net.0", "dwconv3x3", 18.0 * px * f0.C, 8.0 * px * f0.C);
      check(launch_dwconv3x3(a, stream), (p + ".mff.dw").c_str());
      prof_end();
    }
    release(f0);
    Act f2 = pw(p + ".mff.fn.net.1.net.1", f1, ACT_GELU);
    release(f1);
    Act f3 = pw(p + ".mff.fn.net.3", f2, ACT_NONE, pool ? ST_NHWC_POOL : ST_NHWC);
    release(f2);
    tap(p + ".mff", f3);         // pooled when the module folds the following MaxPool2d
    return f3;
  }
  // the same module with MixFeedForward's tail as ONE launch (mff_tail.hip): f16x3 arithmetic, 64 -> 128 -> 64 widths
  bool mff_fusable(int C) const { return m->mff_fused && m->precision == KP2D_PREC_F16X3 && C == 64; }
  Act mff_tail(const std::string& p, const Act& f0, int C, int h, int w, bool pool) {
    Act f3 = alloc(C, pool ? h / 2 : h, pool ? w / 2 : w);
    if (rc == KP2D_OK && !dry) {
      const ConvPack& c1 = m->convs[m->conv_index.at(p + ".mff.fn.net.1.net.1")];
      const ConvPack& c3 = m->convs[m->conv_index.at(p + ".mff.fn.net.3")];
      MffTailArgs a{};
      a.h = ptr(f0);
      a.wdw = m->blob + m->vecs.at(p + ".mff.fn.net.1.net.0.weight").off;
      a.bdw = m->blob + m->vecs.at(p + ".mff.fn.net.1.net.0.bias").off;
      a.w1 = m->blob + c1.w16_off; a.sc1 = m->blob + c1.sc16_off; a.sh1 = m->blob + c1.sh_off;
      a.w3 = m->blob + c3.w16_off; a.sc3 = m->blob + c3.sc16_off; a.sh3 = m->blob + c3.sh_off;
      a.out = ptr(f3); a.B = B; a.H = h; a.W = w; a.pool = pool ? 1 : 0;
      const double px = (double)B * h * w;
      prof_begin(p + ".mff.fn.net.1-3", "mff_tail", px * (18.0 * 128 + 2.0 * 128 * 128 + 2.0 * 128 * 64), 4.0 * px * (128 + (pool ? 16 : 64)));
      check(launch_mff_tail(a, stream), (p + ".mff tail").c_str());
      prof_end();
    }
    return f3;
  }
  // CBR -> NHWC activation (optionally pooled / pooled+full / pixel-shuffled)
  Act cbr(const std::string& name, const Act& in0, const Act* in1, int store, Act* pooled = nullptr) {
    const ConvPack& c = m->convs[m->conv_index.at(name)];
    const int act = m->cfg.leaky_relu ? ACT_LEAKY : ACT_RELU;
    const int Hc = in0.H, Wc = in0.W;
    Act out{};
    if (store == ST_NHWC) {
      out = alloc(c.cout, Hc, Wc);
      conv(name, in0, in0.C, 0, in1, act, store, dry ? nullptr : ptr(out), c.cout, 0, nullptr, 0, 0, 0, Hc, Wc);
    } else if (store == ST_NHWC_POOL) {
      out = alloc(c.cout, Hc / 2, Wc / 2);
      conv(name, in0, in0.C, 0, in1, act, store, nullptr, 0, 0, dry ? nullptr : ptr(out), c.cout, 0, 0, Hc, Wc);
    } else if (store == ST_NHWC_BOTH) {
      out = alloc(c.cout, Hc, Wc);
      *pooled = alloc(c.cout, Hc / 2, Wc / 2);
      conv(name, in0, in0.C, 0, in1, act, store, dry ? nullptr : ptr(out), c.cout, 0, dry ? nullptr : ptr(*pooled),
           c.cout, 0, 0, Hc, Wc);
    } else if (store == ST_S16P) {
      out = alloc(c.cout, Hc, Wc);
      out.fmt = 1;
      conv(name, in0, in0.C, 0, in1, act, store, dry ? nullptr : ptr(out), c.cout, 0, nullptr, 0, 0, 0, Hc, Wc);
    } else if (store == ST_S16P_POOL) {
      out = alloc(c.cout, Hc / 2, Wc / 2);
      out.fmt = 1;
      conv(name, in0, in0.C, 0, in1, act, store, nullptr, 0, 0, dry ? nullptr : ptr(out), c.cout, 0, 0, Hc, Wc);
    } else if (store == ST_S16P_BOTH) {
      out = alloc(c.cout, Hc, Wc);
      *pooled = alloc(c.cout, Hc / 2, Wc / 2);
      out.fmt = pooled->fmt = 1;
      conv(name, in0, in0.C, 0, in1, act, store, dry ? nullptr : ptr(out), c.cout, 0, dry ? nullptr : ptr(*pooled),
           c.cout, 0, 0, Hc, Wc);
    } else if (store == ST_S16P_SHUFFLE) {
      out = alloc(c.cout / 4, Hc * 2, Wc * 2);
      out.fmt = 1;
      conv(name, in0, in0.C, 0, in1, act, store, dry ? nullptr : ptr(out), c.cout / 4, 0, nullptr, 0, 0, 0, Hc, Wc);
    } else {  // ST_SHUFFLE
      out = alloc(c.cout / 4, Hc * 2, Wc * 2);
      conv(name, in0, in0.C, 0, in1, act, store, dry ? nullptr : ptr(out), c.cout / 4, 0, nullptr, 0, 0, 0, Hc, Wc);
    }
    tap(name, out);              // ST_NHWC_POOL: the pooled tensor (the full-resolution one is never written)
    return out;
  }
};

struct FwdOut {
  const uint8_t* frames = nullptr;   // kp2d_forward_frames: uint8 [B,Hs,Ws,3]; x is null then
  int Hs = 0, Ws = 0;
  const float* x;
  float *score, *shift, *feat, *seg, *vlad, *depth;
};

// KP2DTinyV2.forward (kp2dtiny.py:552-591) / KP2DTinyV3.forward (:906-957) as a launch sequence
void build(Plan& P, const FwdOut& o, uint32_t flags) {
  kp2d_model* m = P.m;
  const kp2d_config& g = m->cfg;
  const bool v3 = g.version == 3;
  const int lk = g.leaky_relu ? ACT_LEAKY : ACT_RELU;
  const int H = P.H, W = P.W, B = P.B;

  // ---- backbone (encoders.py:105-129) ----
  // Big grids: conv1a inside conv1b's launch (conv3x3_f16.hip STEM) — its output, the largest tensor of the forward after
  // `skip`, is never written.  Float frames, RGB, 16 -> 32 first stage, split-fp16 arithmetic, a pooled conv1b on the
  // warp-specialised form; a tap on conv1a keeps the two launches (the fused layer has no output to copy).
  // The first layer in the split-fp16 arithmetic (RGB frames, 16 channels): one set of bits whether it runs fused, as its own
  // launch, or straight from uint8 frames — so a forward's results do not depend on the grid size that picks the form.
  const bool c1a_split = m->stem_fusion != 0 && m->precision == KP2D_PREC_F16X3 && g.in_channels == 3 && m->c1 == 16;
  const bool stem = c1a_split && m->stem_fusion == 1 && !o.frames && m->c2 == 32 && g.downsample >= 2 && conv3x3_ws_would_run(B, H, W, m->ws_min) &&
                    !(m->tap_dst && m->tap_name == "backbone.conv1a");
  Act t1a = P.alloc(m->c1, H, W);      // (allocated either way: the workspace size must not depend on the input kind or on a tap)
  if (stem) P.stem_x = o.x;
  if (!stem && !P.dry && P.rc == KP2D_OK) {
    Conv1aArgs a{};
    a.x = o.x; a.w = m->blob + m->conv1a_w; a.scale = m->blob + m->conv1a_sc; a.shift = m->blob + m->conv1a_sh;
    a.out = P.ptr(t1a); a.B = B; a.H = H; a.W = W; a.cout = m->c1; a.act = lk; a.cin = g.in_channels;
    const double px = (double)B * H * W;
    if (c1a_split) {
      P.prof_begin("backbone.conv1a", o.frames ? "conv1a_mfma_u8" : "conv1a_mfma", 2.0 * 27 * m->c1 * px,
                   (o.frames ? 3.0 * B * o.Hs * o.Ws : 12.0 * px) + 4.0 * px * m->c1);
      P.check(launch_conv1a_mfma(a, m->blob + m->conv1a_ws, o.frames, o.Hs, o.Ws, P.stream), "backbone.conv1a (split-fp16)");
    } else if (o.frames) {
      P.prof_begin("backbone.conv1a", "conv1a_u8", 2.0 * 27 * m->c1 * px, 3.0 * B * o.Hs * o.Ws + 4.0 * px * m->c1);
      P.check(launch_conv1a_u8(a, o.frames, o.Hs, o.Ws, P.stream), "backbone.conv1a (uint8 frames)");
    } else {
      P.prof_begin("backbone.conv1a", "conv1a", 2.0 * 9 * g.in_channels * m->c1 * px, 4.0 * px * (g.in_channels + m->c1));
      P.check(launch_conv1a(a, P.stream), "backbone.conv1a");
    }
    P.prof_end();
  }
  P.tap("backbone.conv1a", t1a);
  // The 32-channel stage conv1b -> conv2a -> conv2b -> conv3a -> conv3b with its four inner tensors kept SPLIT (S16P,
  // kp2d_kernels.h): the consumers copy their operand images HBM -> LDS without a vector instruction (conv3x3_s16.hip; these
  // layers are bound by HBM and by their staging, not by the matrix cores).  Same values bit for bit.  Decided here, for
  // the whole chain, because the layout has exactly one reader and two writers: S configs (16 -> 32 -> 32 -> 32 -> 64, two
  // pools), split-fp16 arithmetic, and a grid big enough for the persistent forms of both ends.
  const bool s16 = m->precision == KP2D_PREC_F16X3 && g.downsample == 2 && m->c1 == 16 && m->c2 == 32 && m->c3 == 32 && m->c4 == 64 &&
                   m->s16_min >= 0 && conv3x3_ws_would_run(B, H, W, m->ws_min) &&
                   conv3x3_s16_would_run(B, H / 2, W / 2, P.nlanes, m->s16_min, m->wsm_grid);
  Act p1 = P.cbr("backbone.conv1b", t1a, nullptr, s16 ? ST_S16P_POOL : (g.downsample >= 2 ? ST_NHWC_POOL : ST_NHWC));
  P.release(t1a);
  Act t2a = P.cbr("backbone.conv2a", p1, nullptr, s16 ? ST_S16P : ST_NHWC);
  P.release(p1);
  Act t2b = P.cbr("backbone.conv2b", t2a, nullptr, s16 ? ST_S16P : (g.downsample >= 3 ? ST_NHWC_POOL : ST_NHWC));
  P.release(t2a);
  Act t3a = P.cbr("backbone.conv3a", t2b, nullptr, s16 ? ST_S16P : ST_NHWC);
  P.release(t2b);
  const bool only_enc = (flags & KP2D_FWD_ONLY_ENCODER) != 0;   // only_encoder(): skip every head but the VPR encoder
  // First CBR of every head in one launch ("heads.first", see describe()); first(name) hands out its channel slices.
  static const bool merge_env = !(getenv("KP2D_MERGE_HEADS") && getenv("KP2D_MERGE_HEADS")[0] == '0');
  // Where a head's own launch would be a small grid (a frame or two per call) the five launches are five serial latencies
  // (0.42 -> 0.37 ms per frame).  On big grids the merged layer is ONE launch of the warp-specialised form with five times
  // the rounds (its start-up and drain paid once: conv family 333 -> 343 TFLOP/s at 64 x 240 x 320) against strided slice
  // reads in the five consumers: +0.1 ... +0.6 % at 64 frames, +1.4 % at 32, +0.9 % at 16, +0.6 % at 480 x 640, +0.9 % N
  // (profiles/r4_ab_merged_heads.txt); V3 (three parts), fp32 arithmetic and 30 x 40 head maps measured -0.2 ... -0.8 %
  // and keep their own launches.  KP2D_MERGE_HEADS: 0 never, 2 always.
  const int Hc = H >> g.downsample, Wc = W >> g.downsample;      // the cell grid (backbone output)
  const bool small_grid = (long)((Hc + 15) / 16) * ((Wc + 15) / 16) * P.B < 256;
  static const bool merge_always = getenv("KP2D_MERGE_HEADS") && getenv("KP2D_MERGE_HEADS")[0] == '2';
  // (round 5: 30 x 40 head maps too once the merged layer — five times the work items of one head's — runs on the
  // warp-specialised form: 64 frames of 120 x 160: five launches of 0.031 ms -> one of 0.102, +0.5 ... +2 % end to end)
  const int merged_groups = m->conv_index.count("heads.first") ? (m->convs[m->conv_index.at("heads.first")].npad / 64) : 0;
  const bool big_wsm = m->precision == KP2D_PREC_F16X3 && !v3 && m->wsm_min >= 0 &&
                       ((long)Hc * Wc >= 60 * 80 ||
                        ((long)Hc * Wc >= 30 * 40 && merged_groups >= 2 &&
                         conv3x3_wsm_would_run(B, Hc, Wc, merged_groups, P.nlanes, m->wsm_min, m->wsm_grid, 2)));
  const bool merged = merge_env && (small_grid || merge_always || big_wsm) && !only_enc && m->conv_index.count("heads.first");
  // Big grids of the plain V2 S configuration: S16P is the layout of EVERY tensor a split-fp16 3x3 layer of the warp-specialised
  // form reads — conv3b's two outputs, conv4a / 4b, the merged first layer's desc / seg / vlad slices, both pixel-shuffled
  // tensors, convs.5, convlad2 — so those layers' staging waves only issue LDS-DMA copies (conv3x3_wsm.hip IN16).  fp32 NHWC
  // stays where another kernel reads: the score / location slices (exact dot products, head3x3.hip), convs.1's pooled output
  // and convs.2 / .3 (30 x 40 maps: general kernels), confAa's and convs.7's outputs (confBb / convs.8, planar outputs),
  // convlad3's (NetVLAD).  Same values bit for bit (a consumer multiplies the halves its own staging would have produced).
  // Decided once, on the form running for the smallest converted layer (conv4a); a tap keeps its layer readable either way.
  static const bool s16all_env = !(getenv("KP2D_S16ALL") && getenv("KP2D_S16ALL")[0] == '0');      // (A/B knob)
  const int Hq = H / 4, Wq = W / 4;
  const bool s16_all = s16 && s16all_env && m->s16_all && !v3 && !only_enc && !g.use_attention && !g.depth &&
                       g.upscale_method != KP2D_UP_CONVTRANSPOSE && m->c5 == 64 && m->d1 == 128 && g.encoder_dim == 64 &&
                       m->wsm_min >= 0 && m->wsm_tr == 0 && merged &&
                       Wq / 2 >= 32 &&      // (convs.4 writes its pixel-shuffled S16P output from a W / 8 map: the form's least width
                       (long)((Wq / 2 + 31) / 32) * ((Hq / 2 + 15) / 16) * B * 2 >= 8 &&      //  and its least grid, eight work items)
                       conv3x3_wsm_would_run(B, Hq, Wq, 1, P.nlanes, m->wsm_min, m->wsm_grid, 1);
  Act xp{};
  Act skip = P.cbr("backbone.conv3b", t3a, nullptr, s16_all ? ST_S16P_BOTH : ST_NHWC_BOTH, &xp);   // downsample >= 1 always
  P.release(t3a);
  Act t4a = P.cbr("backbone.conv4a", xp, nullptr, s16_all ? ST_S16P : ST_NHWC);
  P.release(xp);
  Act xb = P.cbr("backbone.conv4b", t4a, nullptr, s16_all ? ST_S16P : ST_NHWC);
  P.release(t4a);
  const int H2 = skip.H, W2 = skip.W;
  if (xb.H != Hc || xb.W != Wc) { P.rc = fail(KP2D_ERR_ARG, "plan: cell grid %dx%d, expected %dx%d", xb.H, xb.W, Hc, Wc); return; }

  // the two planar outputs behind a 64-channel S16P tensor (conv3x3_s16.hip's planar form: cout <= 32 plain logits)
  static const bool planar_env = !(getenv("KP2D_S16PLANAR") && getenv("KP2D_S16PLANAR")[0] == '0');      // (A/B knob)
  // (not the class logits when the forward also writes the dense class map: the argmax over channels that sit in 32 different
  // lanes — DPP rotations per pixel — made that layer 0.147 -> 0.189 ms; the general kernel finds it in its LDS tile)
  auto s16_planar = [&](int cout, bool with_ids = false) {
    return s16_all && planar_env && !with_ids && cout <= 32 && m->c4 == 64 && m->c5 == 64 && !(W2 & 3);
  };
  Act mx{}, mxs{};
  int mx_split = 1 << 30;      // first channel of the merged layer kept in the S16P tensor mxs (s16_all: behind score | loc)
  if (merged && s16_all) {
    const ConvPack& cf = m->convs[m->conv_index.at("heads.first")];
    mx_split = cf.parts[0].second + cf.parts[1].second;
    mx = P.alloc(mx_split, Hc, Wc);
    mxs = P.alloc(cf.cout - mx_split, Hc, Wc);
    mxs.fmt = 1;
    P.conv("heads.first", xb, xb.C, 0, nullptr, lk, ST_MIX16, P.dry ? nullptr : P.ptr(mx), mx.C, 0, P.dry ? nullptr : P.ptr(mxs), mxs.C, 0,
           mx_split, Hc, Wc);
  } else if (merged) {
    mx = P.cbr("heads.first", xb, nullptr, ST_NHWC);
  }
  auto first = [&](const std::string& name) -> Act {
    if (merged) {
      int o = 0;
      for (const auto& pt : m->convs[m->conv_index.at("heads.first")].parts) {
        if (pt.first == name) {
          Act v = o < mx_split ? Plan::view(mx, pt.second, o) : Plan::view(mxs, pt.second, o - mx_split);
          P.tap(name, v);
          return v;
        }
        o += pt.second;
      }
    }
    return P.cbr(name, xb, nullptr, ST_NHWC);
  };
  // NetVLAD / GeM / ConvAP / encoder map behind vlad_head.convlad3 (vpr.py:78-89, netvlad.py:79-106)
  // (keep: when the tail runs on the side stream its scratch must outlive the plan's next allocations — released by the caller)
  auto vlad_tail = [&](const Act& v3a, std::vector<Act>* keep = nullptr) {
    const int S = Hc * Wc, K = g.num_clusters, C = g.encoder_dim;
    if (only_enc || g.remove_netvlad) {
      // vpr.py:84-87: remove_netvlad (to_export) returns the encoder map itself whatever the pooler;
      // only_encoder=True returns l2(map).  Both leave as the NCHW map.
      if (!P.dry && P.rc == KP2D_OK) {
        if (!g.remove_netvlad) P.check(launch_l2norm_channels(P.ptr(v3a), (long)B * S, C, P.stream), "vlad_head.l2");
        P.check(launch_nhwc_to_nchw(P.ptr(v3a), o.vlad, B, C, S, C, 0, P.stream), "vlad_head (encoder map)");
      }
    } else if (g.global_descriptor == KP2D_GD_GEM) {
      if (!P.dry && P.rc == KP2D_OK) {
        PoolArgs a{P.ptr(v3a), m->blob + m->vecs.at("vlad_head.netvlad.p").off, o.vlad, B, C, Hc, Wc};
        P.prof_begin("vlad_head.netvlad", "gem", 4.0 * B * S * C, 4.0 * B * S * C);
        P.check(launch_gem(a, P.stream), "vlad_head.netvlad (GeM)");
        P.prof_end();
      }
    } else if (g.global_descriptor == KP2D_GD_CONVAP) {
      Act cp = P.pw("vlad_head.netvlad.channel_pool", v3a, ACT_NONE);
      if (!P.dry && P.rc == KP2D_OK) {
        PoolArgs a{P.ptr(cp), nullptr, o.vlad, B, C, Hc, Wc};
        P.prof_begin("vlad_head.netvlad", "convap_pool", 1.0 * B * S * C, 4.0 * B * S * C);
        P.check(launch_convap_pool(a, P.stream), "vlad_head.netvlad (ConvAP)");
        P.prof_end();
      }
      if (keep) keep->push_back(cp);
      else P.release(cp);
    } else {
      const int ns = netvlad_nsplit(S);
      const int tps = netvlad_tiles_per_slab(S, B);
      Act part{};
      part.bytes = (size_t)B * (ns * tps + (tps > 1 ? 1 : 0)) * ((size_t)K * C + K) * sizeof(float);   // tile mode: + the ordered sums
      part.off = P.arena.alloc(part.bytes);
      if (part.off == (size_t)-1 && P.rc == KP2D_OK) P.rc = fail(KP2D_ERR_WORKSPACE, "workspace exhausted");
      if (!P.dry && P.rc == KP2D_OK) {
        VladArgs a{};
        a.x = P.ptr(v3a); a.wa = m->blob + m->vlad_wa; a.cent = m->blob + m->vlad_cent;
        a.part = P.ptr(part); a.out = o.vlad; a.B = B; a.S = S; a.C = C; a.K = K; a.nsplit = ns; a.tps = tps;
        a.prec = m->precision == KP2D_PREC_F16X3 ? 1 : 0;
        P.prof_begin("vlad_head.netvlad", "netvlad", 2.0 * 2 * K * C * (double)B * S, 4.0 * B * ((double)S * C + K * C));
        P.check(launch_netvlad(a, P.stream), "vlad_head.netvlad");
        P.prof_end();
      }
      if (keep) keep->push_back(part);
      else P.arena.release(part.off, part.bytes);
    }
  };
  // Small grids, the plain V2 configuration (PixelShuffle upsampling, no attention, no depth head): the heads level by
  // level instead of head by head.  A frame's forward is a chain of dependent launches of ~8-10 us each whatever they compute;
  // the layers of different heads that wait for the same predecessor go out as ONE launch (Plan::group_begin / group_end),
  // so the heads cost the length of the longest chain (the segmentation head's eight layers), not the sum of all chains:
  // 17 launches -> 12 behind the merged first layer.  Same kernels, same arithmetic, per layer.
  // (a dry run sizes the workspace for whichever schedule keeps more tensors alive — P.no_levels picks; profiles and taps
  // take the layers one launch at a time)
  const bool levels = merged && small_grid && !v3 && !g.use_attention && !g.depth && g.upscale_method != KP2D_UP_CONVTRANSPOSE &&
                      m->precision == KP2D_PREC_F16X3 && m->multi_launch && !P.no_levels && !s16_all &&
                      (P.dry || (!m->profiling && !m->tap_dst));
  if (levels) {
    const std::string L = "seg_head.convs.";
    Act s1 = first("score_head.convDa"), l1 = first("loc_head.convDa"), d1 = first("desc_head.convA");
    Act g0 = first(L + "0"), v1 = first("vlad_head.convlad1");
    P.head_pair("score_head.convDb", s1, ACT_SIGMOID, o.score, "loc_head.convDb", l1, ACT_TANH, o.shift, Hc, Wc);
    const ConvPack& cB = m->convs[m->conv_index.at("desc_head.convB")];
    // level 1
    Act d2 = P.alloc(cB.cout / 4, H2, W2);
    P.group_begin();
    P.conv("desc_head.convB", d1, d1.C, 0, nullptr, ACT_NONE, ST_SHUFFLE, P.dry ? nullptr : P.ptr(d2), d2.C, 0, nullptr, 0, 0, 0, Hc, Wc);
    Act g1 = P.cbr(L + "1", g0, nullptr, ST_NHWC_POOL);
    Act v2 = P.cbr("vlad_head.convlad2", v1, nullptr, ST_NHWC);
    P.group_end();
    // level 2
    P.group_begin();
    Act d3 = P.cbr("desc_head.confAa", d2, &skip, ST_NHWC);
    Act g2 = P.cbr(L + "2", g1, nullptr, ST_NHWC);
    Act v3a = P.cbr("vlad_head.convlad3", v2, nullptr, ST_NHWC);
    P.group_end();
    P.release(d2);
    P.release(g1);
    P.release(v2);
    // The VPR head is done with its convolutions two launches before the descriptor head and six before the segmentation
    // head: its pooling (NetVLAD: three launches, ~25 us of a frame's ~230) goes to a side stream and runs BESIDE the rest
    // (fork / join by events).  Its input and scratch stay allocated until the join (the dry run sizes the workspace the
    // same way).
    // Not under stream capture: replayed as graphs with several frames in flight (pipeline.FrameStream) the extra branch
    // costs the overlap BETWEEN frames — 10.3k -> 4.6k frames/s (profiles/r5_ab_side_stream.txt); a plain forward gains 5 %.
    std::vector<Act> vlad_keep;
    // The stream is created on first use, not with the model: HIP spreads a process's streams over four hardware queues,
    // and a stream that exists — used or not — took one from pipeline.BatchStream's two (64-frame batches, two steps in
    // flight: 25.0k -> 23.6k frames/s with an idle side stream in the process, back at 24.9k with GPU_MAX_HW_QUEUES=8).
    bool side = m->side_overlap && !P.dry && P.rc == KP2D_OK;
    if (side) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(P.stream, &cs) != hipSuccess) { (void)hipGetLastError(); side = false; }
      else if (cs != hipStreamCaptureStatusNone) side = false;
    }
    if (side && !m->side_stream) {
      if (hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&m->side_fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&m->side_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (m->side_stream) (void)hipStreamDestroy(m->side_stream);
        m->side_stream = nullptr;
        m->side_overlap = false;
        side = false;
      }
    }
    hipStream_t main_stream = P.stream;
    if (side) {
      P.check((int)hipEventRecord(m->side_fork, main_stream), "side stream fork");
      P.check((int)hipStreamWaitEvent(m->side_stream, m->side_fork, 0), "side stream fork");
      P.stream = m->side_stream;
    }
    vlad_tail(v3a, &vlad_keep);
    if (side) {
      P.check((int)hipEventRecord(m->side_join, m->side_stream), "side stream join");
      P.stream = main_stream;
    }
    // level 3
    P.group_begin();
    P.conv("desc_head.confBb", d3, d3.C, 0, nullptr, ACT_NONE, ST_NCHW, o.feat, 0, 0, nullptr, 0, 0, g.nfeatures, H2, W2);
    Act g3 = P.cbr(L + "3", g2, nullptr, ST_NHWC);
    P.group_end();
    P.release(d3);
    P.release(g2);
    // the rest of the segmentation head is the critical path: one layer per launch
    Act g4 = P.cbr(L + "4", g3, nullptr, ST_SHUFFLE);
    P.release(g3);
    Act g5 = P.cbr(L + "5", g4, &xb, ST_NHWC);
    P.release(g4);
    Act g6 = P.cbr(L + "6", g5, nullptr, ST_SHUFFLE);
    P.release(g5);
    Act g7 = P.cbr(L + "7", g6, &skip, ST_NHWC);
    P.release(g6);
    P.conv(L + "8", g7, g7.C, 0, nullptr, ACT_NONE, ST_NCHW, o.seg, 0, 0, nullptr, 0, 0, g.n_classes, H2, W2);
    P.release(g7);
    if (side) P.check((int)hipStreamWaitEvent(main_stream, m->side_join, 0), "side stream join");
    for (const Act& k : vlad_keep) P.arena.release(k.off, k.bytes);
    P.release(v3a);
    P.release(mx);
    P.release(xb);
    P.release(skip);
    return;
  }
  // ---- score / location heads (heads.py:28-35; sigmoid/tanh kp2dtiny.py:574-575, :927-935) ----
  if (only_enc) {
  } else if (v3) {
    Act s1 = first("score_loc_head.convDa");
    P.conv("score_loc_head.convDb", s1, s1.C, 0, nullptr, ACT_SIGMOID0_TANH, ST_NCHW, o.score, 0, 0, o.shift, 0, 0, 1, Hc, Wc);
    P.release(s1);
  } else {
    Act s1 = first("score_head.convDa");
    Act l1 = first("loc_head.convDa");
    P.head_pair("score_head.convDb", s1, ACT_SIGMOID, o.score, "loc_head.convDb", l1, ACT_TANH, o.shift, Hc, Wc);
    P.release(s1);
    P.release(l1);
    // ---- descriptor head (heads.py:91-104) ----
    Act d1 = first("desc_head.convA");
    const ConvPack& cB = m->convs[m->conv_index.at("desc_head.convB")];
    Act d2 = P.alloc(cB.cout / 4, H2, W2);
    if (g.upscale_method == KP2D_UP_CONVTRANSPOSE) {
      // convB at the cell grid, then the transposed-conv upsampler as a pixel-shuffled 3x3 conv (heads.py:96-98)
      Act db = P.alloc(cB.cout, Hc, Wc);
      P.conv("desc_head.convB", d1, d1.C, 0, nullptr, ACT_NONE, ST_NHWC, P.dry ? nullptr : P.ptr(db), db.C, 0, nullptr, 0, 0, 0, Hc, Wc);
      P.conv("desc_head.upsample", db, db.C, 0, nullptr, lk, ST_SHUFFLE, P.dry ? nullptr : P.ptr(d2), d2.C, 0, nullptr, 0, 0, 0, Hc, Wc);
      P.release(db);
    } else {
      if (s16_all) d2.fmt = 1;
      P.conv("desc_head.convB", d1, d1.C, 0, nullptr, ACT_NONE, s16_all ? ST_S16P_SHUFFLE : ST_SHUFFLE, P.dry ? nullptr : P.ptr(d2), d2.C, 0, nullptr, 0, 0, 0, Hc, Wc);
    }
    P.release(d1);
    P.tap("desc_head.convB", d2);     // the pixel-shuffled / transposed-conv upsampled tensor (heads.py:96-98)
    Act d3 = P.cbr("desc_head.confAa", d2, &skip, s16_planar(g.nfeatures) ? ST_S16P : ST_NHWC);
    P.release(d2);
    P.conv("desc_head.confBb", d3, d3.C, 0, nullptr, ACT_NONE, ST_NCHW, o.feat, 0, 0, nullptr, 0, 0, g.nfeatures, H2, W2);
    P.release(d3);
  }

  // ---- segmentation head: segmentation.py:126-157 (V2), :321-347 (V3), :442-466 (V2 att), :588-619 (V3 att) ----
  // trunk(prefix) runs everything up to the last CBR(c_exp -> width) and returns it plus the name of the final conv
  // CBR(ch -> d1) + 2x upsampling: PixelShuffle folded into the store, or (to_mcu) the CBR at its own resolution
  // followed by TransposedConvUpsampleModel as a second, pixel-shuffled conv (segmentation.py:139-147)
  auto upconv = [&](const std::string& cname, const std::string& uname, const Act& in) -> Act {
    if (g.upscale_method != KP2D_UP_CONVTRANSPOSE) return P.cbr(cname, in, nullptr, s16_all ? ST_S16P_SHUFFLE : ST_SHUFFLE);
    Act t = P.cbr(cname, in, nullptr, ST_NHWC);
    Act u = P.cbr(uname, t, nullptr, ST_SHUFFLE);
    P.release(t);
    return u;
  };
  auto trunk = [&](const std::string& hp, std::string* last) -> Act {
    const std::string L = hp + ".convs.";
    Act g5{};
    int i;   // index of the second-to-last shuffle CBR
    if (g.use_attention) {
      Act g0 = P.cbr(L + "0", xb, nullptr, ST_NHWC);
      Act a1 = P.attention_module(L + "1", g0, /*pool=*/true);
      P.release(g0);
      Act a2 = P.attention_module(L + "2", a1, false);
      P.release(a1);
      Act g4 = upconv(L + "3", hp + ".upsample", a2);
      P.release(a2);
      g5 = P.cbr(L + "4", g4, &xb, ST_NHWC);
      P.release(g4);
      i = 5;
    } else {
      Act g0 = first(L + "0");
      Act g1 = P.cbr(L + "1", g0, nullptr, ST_NHWC_POOL);
      P.release(g0);
      Act g2 = P.cbr(L + "2", g1, nullptr, ST_NHWC);
      P.release(g1);
      Act g3 = P.cbr(L + "3", g2, nullptr, ST_NHWC);
      P.release(g2);
      Act g4 = upconv(L + "4", hp + ".upsample", g3);
      P.release(g3);
      g5 = P.cbr(L + "5", g4, &xb, s16_all ? ST_S16P : ST_NHWC);
      P.release(g4);
      i = 6;
    }
    Act g6 = upconv(L + std::to_string(i), hp + ".upsample2", g5);
    P.release(g5);
    Act g7 = P.cbr(L + std::to_string(i + 1), g6, &skip, (hp == "seg_head" && s16_planar(g.n_classes, P.seg_ids != nullptr)) ? ST_S16P : ST_NHWC);
    P.release(g6);
    *last = L + std::to_string(i + 2);
    return g7;
  };
  if (!only_enc) {
    std::string last;
    Act g7 = trunk("seg_head", &last);
    if (v3) {
      const int half = m->c5 / 2;   // dim_split = c_hidden // 2 (segmentation.py:190, :339-343)
      P.conv("seg_head.featB", g7, half, 0, nullptr, ACT_NONE, ST_NCHW, o.feat, 0, 0, nullptr, 0, 0, g.nfeatures, H2, W2);
      if (g.depth)   // depth = featD(seg[:, half:2*half]).sigmoid()  (segmentation.py:340-341, kp2dtiny.py:956)
        P.conv("seg_head.featD", g7, half, half, nullptr, ACT_SIGMOID, ST_NCHW, o.depth, 0, 0, nullptr, 0, 0, 1, H2, W2);
      const bool sm = (flags & KP2D_FWD_EVAL) && !g.remove_softmax;
      P.conv(last, g7, half, g7.C - half, nullptr, sm ? ACT_SOFTMAX_C : ACT_NONE, ST_NCHW, o.seg, 0, 0, nullptr, 0, 0,
             g.n_classes, H2, W2);
    } else {
      P.conv(last, g7, g7.C, 0, nullptr, ACT_NONE, ST_NCHW, o.seg, 0, 0, nullptr, 0, 0, g.n_classes, H2, W2);
    }
    P.release(g7);
  }
  if (!only_enc && !v3 && g.depth) {   // depth = depth_head(x, skip).sigmoid()  (kp2dtiny.py:588-590)
    std::string last;
    Act g7 = trunk("depth_head", &last);
    P.conv(last, g7, g7.C, 0, nullptr, ACT_SIGMOID, ST_NCHW, o.depth, 0, 0, nullptr, 0, 0, 1, H2, W2);
    P.release(g7);
  }

  // ---- VPR head (vpr.py:78-89) + NetVLAD (netvlad.py:79-106) ----
  {
    Act v1 = first("vlad_head.convlad1");
    Act v2 = P.cbr("vlad_head.convlad2", v1, nullptr, s16_all ? ST_S16P : ST_NHWC);
    P.release(v1);
    Act v3a = P.cbr("vlad_head.convlad3", v2, nullptr, ST_NHWC);
    P.release(v2);
    vlad_tail(v3a);
    P.release(v3a);
  }
  if (merged) P.release(mx);
  if (merged) P.release(mxs);
  P.release(xb);
  P.release(skip);
}

int validate_shape(const kp2d_model* m, int B, int H, int W) {
  if (B < 1) return fail(KP2D_ERR_ARG, "B must be >= 1");
  // the segmentation head pools the cell grid once more (segmentation.py:134): H, W divisible by 2 * cell
  const int q = 2 << m->cfg.downsample;
  if (H < 16 || W < 16 || (H % q) || (W % q)) return fail(KP2D_ERR_ARG, "H and W must be multiples of %d and >= 16 (got %dx%d)", q, H, W);
  return KP2D_OK;
}

size_t plan_bytes(kp2d_model* m, int Bc, int H, int W);

// Frames per internal sub-batch.  Measured on MI355X (profiles/r1_*): the path is compute-bound, so bigger
// launches win (64 frames at once: 5.4k frames/s vs 3.7k with 10-frame sub-batches that keep intermediates
// inside the Infinity Cache but leave the 30x40 layers with 60 workgroups for 256 CUs).  The automatic
// choice therefore only caps the workspace (4 GiB), it does not chase cache residency.
int auto_chunk(const kp2d_model* m, int B, int H, int W) {
  if (m->chunk_frames > 0) return std::min(B, m->chunk_frames);
  const size_t per_frame = plan_bytes(const_cast<kp2d_model*>(m), 1, H, W);
  if (per_frame == 0) return 1;
  const size_t cap = (size_t)4 << 30;
  return (int)std::max<size_t>(1, std::min<size_t>((size_t)B, cap / per_frame));
}

size_t plan_bytes_uncached(kp2d_model* m, int Bc, int H, int W);

// dry-run planning costs ~0.1 ms of host time; the result only depends on (frames, H, W), so it is memoised
size_t plan_bytes(kp2d_model* m, int Bc, int H, int W) {
  const uint64_t key = ((uint64_t)Bc << 40) ^ ((uint64_t)H << 20) ^ (uint64_t)W;
  auto it = m->plan_cache.find(key);
  if (it != m->plan_cache.end()) return it->second;
  const size_t v = plan_bytes_uncached(m, Bc, H, W);
  m->plan_cache[key] = v;
  return v;
}

size_t plan_bytes_uncached(kp2d_model* m, int Bc, int H, int W) {
  Plan P{};
  P.m = m; P.stream = nullptr; P.ws = nullptr; P.dry = true; P.B = Bc; P.H = H; P.W = W;
  P.arena.reset((size_t)1 << 46);
  FwdOut o{};
  build(P, o, 0);
  if (P.rc != KP2D_OK) return 0;
  // the level-by-level schedule of small grids and the head-by-head one keep different tensors alive: room for either
  Plan Q{};
  Q.m = m; Q.stream = nullptr; Q.ws = nullptr; Q.dry = true; Q.B = Bc; Q.H = H; Q.W = W; Q.no_levels = true;
  Q.arena.reset((size_t)1 << 46);
  build(Q, o, 0);
  return Q.rc == KP2D_OK ? std::max(P.arena.high, Q.arena.high) : 0;
}

}  // namespace

// ================================================================================================
extern "C" {

const char* kp2d_last_error(void) { return g_err.c_str(); }
extern "C++" {
namespace kp2d { void set_last_error(const char* msg) { g_err = msg ? msg : ""; } }   // for the other API files
}
int32_t kp2d_abi_version(void) { return KP2D_ABI_VERSION; }

int kp2d_create(const kp2d_config* cfg, kp2d_model** out) {
  if (!cfg || !out) return fail(KP2D_ERR_ARG, "null argument");
  // ABI evolution: the struct grew by in_channels; the shorter form (everything up to upscale_method) means RGB
  constexpr int32_t kOldSize = (int32_t)offsetof(kp2d_config, in_channels);
  if (cfg->struct_size != (int32_t)sizeof(kp2d_config) && cfg->struct_size != kOldSize)
    return fail(KP2D_ERR_ARG, "kp2d_config.struct_size mismatch");
  const int cin0 = (cfg->struct_size == kOldSize || cfg->in_channels == 0) ? 3 : cfg->in_channels;
  if (cin0 != 3 && cin0 != 1) return fail(KP2D_ERR_UNSUPPORTED, "in_channels=%d (3 and 1 are built)", cin0);
  if (cin0 == 1 && cfg->version != 3) return fail(KP2D_ERR_ARG, "in_channels=1 is KP2DTinyV3(use_color=False); V2 reads RGB");
  if (cfg->global_descriptor < KP2D_GD_NETVLAD || cfg->global_descriptor > KP2D_GD_CONVAP) return fail(KP2D_ERR_ARG, "bad global_descriptor");
  if (cfg->version != 2 && cfg->version != 3) return fail(KP2D_ERR_ARG, "version must be 2 or 3");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(KP2D_ERR_HIP, "no HIP device visible: this library has no CPU path");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(KP2D_ERR_ARG, "device %d out of range (%d visible)", cfg->device, ndev);
  auto* m = new kp2d_model();
  std::memcpy(&m->cfg, cfg, (size_t)cfg->struct_size);
  m->cfg.struct_size = (int32_t)sizeof(kp2d_config);
  m->cfg.in_channels = cin0;
  m->c1 = cfg->channel_dims[0]; m->c2 = cfg->channel_dims[1]; m->c3 = cfg->channel_dims[2];
  m->c4 = cfg->channel_dims[3]; m->c5 = cfg->channel_dims[4]; m->d1 = cfg->channel_dims[5];
  if (m->c1 % 16 || m->c1 > 256) { delete m; return fail(KP2D_ERR_UNSUPPORTED, "channel_dims[0]=%d (conv1a kernels need a multiple of 16, <= 256)", cfg->channel_dims[0]); }
  if (cfg->nfeatures != 32 && cfg->nfeatures != 64 && cfg->nfeatures != 128) { delete m; return fail(KP2D_ERR_UNSUPPORTED, "nfeatures=%d (32, 64 and 128 are built)", cfg->nfeatures); }
  if (cfg->downsample != 2 && cfg->downsample != 3) { delete m; return fail(KP2D_ERR_UNSUPPORTED, "downsample=%d (2 and 3 are built)", cfg->downsample); }
  if (cfg->n_classes < 1 || cfg->n_classes > 32) { delete m; return fail(KP2D_ERR_UNSUPPORTED, "n_classes must be in [1,32]"); }
  int rc = describe(m);
  if (rc != KP2D_OK) { delete m; return rc; }
  const char* nlanes = getenv("KP2D_LANES");
  if (nlanes) m->lanes = std::max(1, std::min(8, atoi(nlanes)));
  m->lanes_default = m->lanes;
  if (getenv("KP2D_SIDE") && getenv("KP2D_SIDE")[0] == '0') m->side_overlap = false;      // (A/B knob)
  *out = m;
  return KP2D_OK;
}

void kp2d_destroy(kp2d_model* m) {
  if (!m) return;
  if (m->blob) (void)hipFree(m->blob);
  for (auto& r : m->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (auto st : m->lane_streams) (void)hipStreamDestroy(st);
  for (auto ev : m->lane_events) (void)hipEventDestroy(ev);
  if (m->fork_event) (void)hipEventDestroy(m->fork_event);
  if (m->side_stream) (void)hipStreamDestroy(m->side_stream);
  if (m->side_fork) (void)hipEventDestroy(m->side_fork);
  if (m->side_join) (void)hipEventDestroy(m->side_join);
  delete m;
}

int kp2d_num_weights(const kp2d_model* m) { return m ? (int)m->specs.size() : 0; }

int kp2d_weight_info(const kp2d_model* m, int index, const char** key, int64_t shape[4], int* ndim) {
  if (!m || index < 0 || index >= (int)m->specs.size()) return fail(KP2D_ERR_ARG, "weight index out of range");
  const WeightSpec& s = m->specs[index];
  if (key) *key = s.key.c_str();
  if (ndim) *ndim = (int)s.shape.size();
  if (shape) for (size_t i = 0; i < s.shape.size() && i < 4; ++i) shape[i] = s.shape[i];
  return KP2D_OK;
}

int kp2d_set_weight(kp2d_model* m, const char* key, const float* host, const int64_t* shape, int ndim) {
  if (!m || !key || !host || (!shape && ndim > 0)) return fail(KP2D_ERR_ARG, "null argument");
  std::string k(key);
  if (k.size() > 20 && k.compare(k.size() - 20, 20, ".num_batches_tracked") == 0) return KP2D_OK;
  auto it = m->spec_index.find(k);
  if (it == m->spec_index.end()) return fail(KP2D_ERR_WEIGHT, "unexpected key '%s'", key);
  const WeightSpec& s = m->specs[it->second];
  bool same = (int)s.shape.size() == ndim;
  for (int i = 0; same && i < ndim; ++i) same = s.shape[i] == shape[i];
  if (!same) return fail(KP2D_ERR_WEIGHT, "shape mismatch for '%s'", key);
  m->host[k].assign(host, host + s.numel());
  m->finalized = false;
  return KP2D_OK;
}

int kp2d_finalize_weights(kp2d_model* m) {
  if (!m) return fail(KP2D_ERR_ARG, "null model");
  std::vector<float> blob;
  int rc = pack(m, blob);
  if (rc != KP2D_OK) return rc;
  DeviceGuard guard(m->cfg.device);
  if (!m->blob) HIP_TRY(hipMalloc((void**)&m->blob, m->blob_floats * sizeof(float)));
  HIP_TRY(hipMemcpy(m->blob, blob.data(), m->blob_floats * sizeof(float), hipMemcpyHostToDevice));
  m->finalized = true;
  return KP2D_OK;
}

size_t kp2d_packed_bytes(const kp2d_model* m) { return m ? m->blob_floats * sizeof(float) : 0; }

int kp2d_export_packed(const kp2d_model* m, void* dev_dst, void* stream) {
  if (!m || !dev_dst) return fail(KP2D_ERR_ARG, "null argument");
  if (!m->finalized) return fail(KP2D_ERR_STATE, "weights not finalised");
  DeviceGuard guard(m->cfg.device);
  HIP_TRY(hipMemcpyAsync(dev_dst, m->blob, m->blob_floats * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return KP2D_OK;
}

int kp2d_import_packed(kp2d_model* m, const void* dev_src, void* stream) {
  if (!m || !dev_src) return fail(KP2D_ERR_ARG, "null argument");
  DeviceGuard guard(m->cfg.device);
  if (!m->blob) HIP_TRY(hipMalloc((void**)&m->blob, m->blob_floats * sizeof(float)));
  HIP_TRY(hipMemcpyAsync(m->blob, dev_src, m->blob_floats * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  m->finalized = true;
  return KP2D_OK;
}

// Sub-batch schedule shared by kp2d_workspace_bytes and kp2d_forward: `lanes` concurrent streams, each working
// through ceil(nchunks / lanes) sub-batches of `chunk` frames in its own slice of the workspace.
static void schedule(const kp2d_model* m, int B, int H, int W, int* lanes, int* chunk) {
  int nl = m->profiling ? 1 : std::max(1, m->lanes);
  nl = std::min(nl, B);
  int c = std::min(auto_chunk(m, B, H, W), (B + nl - 1) / nl);
  *lanes = nl;
  *chunk = std::max(1, c);
}

size_t kp2d_vlad_dim(const kp2d_model* m, int H, int W) {
  if (!m) return 0;
  const kp2d_config& g = m->cfg;
  if (g.remove_netvlad) return (size_t)g.encoder_dim * (H >> g.downsample) * (W >> g.downsample);   // vpr.py:84
  if (g.global_descriptor != KP2D_GD_NETVLAD) return (size_t)g.encoder_dim * 16;
  return (size_t)g.num_clusters * g.encoder_dim;
}

size_t kp2d_workspace_bytes(const kp2d_model* m, int B, int H, int W) {
  if (!m || validate_shape(m, B, H, W) != KP2D_OK) return 0;
  kp2d_model* mm = const_cast<kp2d_model*>(m);
  // size for the largest lane count this handle may use (profiling toggles lanes to 1, which needs less)
  const bool prof = mm->profiling;
  mm->profiling = false;
  int nl, chunk;
  schedule(m, B, H, W, &nl, &chunk);
  mm->profiling = prof;
  const size_t per = align_up(plan_bytes(mm, chunk, H, W));
  int nl1, chunk1;
  mm->profiling = true;
  schedule(m, B, H, W, &nl1, &chunk1);
  mm->profiling = prof;
  return std::max(per * nl, align_up(plan_bytes(mm, chunk1, H, W)));
}

static int forward_impl(kp2d_model* m, const float* x, const uint8_t* frames, int Hs, int Ws, int B, int H, int W,
                        uint32_t flags, float* score, float* shift, float* feat, float* seg, float* vlad, float* depth,
                        void* workspace, size_t workspace_bytes, void* stream) {
  const bool only_enc = (flags & KP2D_FWD_ONLY_ENCODER) != 0;
  if (!m || (!x && !frames) || !vlad || !workspace) return fail(KP2D_ERR_ARG, "null argument");
  if (frames && (Hs < 1 || Ws < 1)) return fail(KP2D_ERR_ARG, "bad source frame size %dx%d", Hs, Ws);
  if (frames && (m->cfg.in_channels != 3 || m->c1 != 16))
    return fail(KP2D_ERR_UNSUPPORTED, "kp2d_forward_frames needs an RGB model with a 16-channel first layer (use kp2d_preprocess + kp2d_forward)");
  if (!only_enc && (!score || !shift || !feat || !seg)) return fail(KP2D_ERR_ARG, "null argument");
  if (!only_enc && m->cfg.depth && !depth) return fail(KP2D_ERR_ARG, "depth=1 model needs the depth output");
  if (!m->finalized) return fail(KP2D_ERR_STATE, "weights not finalised (kp2d_finalize_weights / kp2d_import_packed)");
  int rc = validate_shape(m, B, H, W);
  if (rc != KP2D_OK) return rc;
  if ((uintptr_t)workspace % ALIGN) return fail(KP2D_ERR_WORKSPACE, "workspace must be %zu-byte aligned", ALIGN);
  if (m->seg_ids_dst && !only_enc && m->seg_ids_cap < (size_t)B * (2 * (H >> m->cfg.downsample)) * (2 * (W >> m->cfg.downsample)))
    return fail(KP2D_ERR_ARG, "kp2d_set_seg_ids: buffer of %zu ids is too small for this forward", m->seg_ids_cap);
  int nl, chunk;
  schedule(m, B, H, W, &nl, &chunk);
  const size_t per = align_up(plan_bytes(m, chunk, H, W));
  if (per == 0) return KP2D_ERR_WORKSPACE;
  if (workspace_bytes < per * nl) return fail(KP2D_ERR_WORKSPACE, "workspace %zu B < required %zu B", workspace_bytes, per * nl);
  const kp2d_config& g = m->cfg;
  // cell grid (score / shift) and dense grid (feat / seg / depth: one pixel-shuffle above the cell grid)
  const size_t Hc = H >> g.downsample, Wc = W >> g.downsample, H2 = 2 * Hc, W2 = 2 * Wc;
  hipStream_t caller = (hipStream_t)stream;
  m->prof_used = 0;
  m->prof_stream = caller;
  DeviceGuard guard(g.device);
  // fork: lanes 1.. run on internal streams that start after everything already queued on the caller's stream
  if (nl > 1) {
    while ((int)m->lane_streams.size() < nl - 1) {
      hipStream_t st; hipEvent_t ev;
      // (KP2D_LANE_PRIORITY=-1: lane streams from the high-priority pool of hardware queues — an A/B knob, profiles/r5_hw_queues.txt)
      static const int lane_prio = getenv("KP2D_LANE_PRIORITY") ? atoi(getenv("KP2D_LANE_PRIORITY")) : 0;
      if (lane_prio != 0) HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lane_prio));
      else HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      m->lane_streams.push_back(st); m->lane_events.push_back(ev);
    }
    if (!m->fork_event) HIP_TRY(hipEventCreateWithFlags(&m->fork_event, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(m->fork_event, caller));
    for (int k = 1; k < nl; ++k) HIP_TRY(hipStreamWaitEvent(m->lane_streams[k - 1], m->fork_event, 0));
  }
  int ci = 0;
  rc = KP2D_OK;
  for (int b0 = 0; b0 < B && rc == KP2D_OK; b0 += chunk, ++ci) {
    const int lane = ci % nl;
    Plan P{};
    P.m = m; P.stream = lane == 0 ? caller : m->lane_streams[lane - 1];
    P.ws = (char*)workspace + (size_t)lane * per; P.dry = false;
    P.B = std::min(chunk, B - b0); P.H = H; P.W = W; P.b0 = b0;
    P.nlanes = std::min(nl, (B + chunk - 1) / chunk);
    P.arena.reset(per);
    FwdOut o{};
    o.x = x ? x + (size_t)b0 * g.in_channels * H * W : nullptr;
    o.frames = frames ? frames + (size_t)b0 * Hs * Ws * 3 : nullptr;
    o.Hs = Hs; o.Ws = Ws;
    o.score = score ? score + (size_t)b0 * Hc * Wc : nullptr;
    o.shift = shift ? shift + (size_t)b0 * 2 * Hc * Wc : nullptr;
    o.feat = feat ? feat + (size_t)b0 * g.nfeatures * H2 * W2 : nullptr;
    o.seg = seg ? seg + (size_t)b0 * g.n_classes * H2 * W2 : nullptr;
    P.seg_ptr = o.seg;
    P.seg_ids = (m->seg_ids_dst && o.seg && !only_enc) ? reinterpret_cast<long long*>(m->seg_ids_dst) + (size_t)b0 * H2 * W2 : nullptr;
    o.vlad = vlad + (size_t)b0 * (only_enc ? (size_t)g.encoder_dim * Hc * Wc : kp2d_vlad_dim(m, H, W));
    o.depth = depth ? depth + (size_t)b0 * H2 * W2 : nullptr;
    build(P, o, flags);
    rc = P.rc;
  }
  // join: the caller's stream continues only after every lane has drained.  This also runs when a launch failed
  // part-way: kernels already queued on the internal streams still write the caller's workspace and outputs, so the
  // caller's stream (and whoever frees those buffers in stream order after the error) must be ordered behind them.
  std::string first_err = rc != KP2D_OK ? g_err : std::string();
  for (int k = 1; k < nl; ++k) {
    hipError_t e = hipEventRecord(m->lane_events[k - 1], m->lane_streams[k - 1]);
    if (e == hipSuccess) e = hipStreamWaitEvent(caller, m->lane_events[k - 1], 0);
    if (e != hipSuccess && rc == KP2D_OK) rc = fail(KP2D_ERR_HIP, "lane join: %s", hipGetErrorString(e));
  }
  if (!first_err.empty()) g_err = first_err;
  return rc;
}

int kp2d_forward(kp2d_model* m, const float* x, int B, int H, int W, uint32_t flags, float* score, float* shift,
                 float* feat, float* seg, float* vlad, float* depth, void* workspace, size_t workspace_bytes,
                 void* stream) {
  if (!x) return fail(KP2D_ERR_ARG, "null argument");
  return forward_impl(m, x, nullptr, 0, 0, B, H, W, flags, score, shift, feat, seg, vlad, depth, workspace, workspace_bytes, stream);
}

int kp2d_forward_frames(kp2d_model* m, const uint8_t* frames, int B, int Hs, int Ws, int H, int W, uint32_t flags,
                        float* score, float* shift, float* feat, float* seg, float* vlad, float* depth, void* workspace,
                        size_t workspace_bytes, void* stream) {
  if (!frames) return fail(KP2D_ERR_ARG, "null argument");
  return forward_impl(m, nullptr, frames, Hs, Ws, B, H, W, flags, score, shift, feat, seg, vlad, depth, workspace, workspace_bytes, stream);
}

int kp2d_post(kp2d_model* m, const float* score, const float* shift, const float* feat, const float* seg, int B,
              int H, int W, int Hc, int Wc, int feat_c, int Hf, int Wf, int seg_c, int Hs, int Ws, float* score_out,
              float* coord, float* desc, int64_t* seg_ids, int sample_segmentation, void* stream) {
  if (!m || !score || !shift || !score_out || !coord) return fail(KP2D_ERR_ARG, "null argument");
  if (desc && !feat) return fail(KP2D_ERR_ARG, "desc requested without feat");
  // seg == NULL with seg_ids: the ids are already there (the forward wrote them, kp2d_set_seg_ids) and are left alone
  if (seg_ids && !seg && sample_segmentation) return fail(KP2D_ERR_ARG, "sampled class ids need the seg tensor");
  if (seg_ids && !seg) seg_ids = nullptr;
  DeviceGuard guard(m->cfg.device);
  PostArgs a{};
  a.score_in = score; a.shift = shift; a.feat = feat; a.score_out = score_out; a.coord = coord; a.desc = desc;
  a.B = B; a.C = feat_c; a.Hc = Hc; a.Wc = Wc; a.Hf = Hf; a.Wf = Wf; a.H = H; a.W = W;
  a.cell = 1 << m->cfg.downsample;
  a.cross_ratio = 2.0f;   // kp2dtiny.py:339
  if (!desc) a.C = 32;
  if (seg_ids && !sample_segmentation) {        // the dense argmax does not depend on the decoded coordinates: one launch
    const ArgmaxArgs g{seg, seg_ids, B, seg_c, Hs * Ws};
    const int e = launch_post_seg(a, g, (hipStream_t)stream);
    if (e) return fail(e < 0 ? KP2D_ERR_UNSUPPORTED : KP2D_ERR_HIP, "post / argmax kernel: %d (descriptor channels %d)", e, feat_c);
    return KP2D_OK;
  }
  int e = launch_post(a, (hipStream_t)stream);
  if (e) return fail(e < 0 ? KP2D_ERR_UNSUPPORTED : KP2D_ERR_HIP, "post kernel: %d (descriptor channels %d)", e, feat_c);
  if (seg_ids && sample_segmentation) {
    SegSampleArgs g{seg, coord, seg_ids, B, seg_c, Hs, Ws, Hc, Wc, H, W};
    e = launch_seg_sample_argmax(g, (hipStream_t)stream);
    if (e) return fail(KP2D_ERR_HIP, "sampled argmax kernel: %d", e);
  } else if (seg_ids) {
    ArgmaxArgs g{seg, seg_ids, B, seg_c, Hs * Ws};
    e = launch_seg_argmax(g, (hipStream_t)stream);
    if (e) return fail(KP2D_ERR_HIP, "argmax kernel: %d", e);
  }
  return KP2D_OK;
}

int kp2d_select_topk(const float* score, int B, int n, int k, float thr, int32_t* idx, float* val, int32_t* count,
                     void* stream) {
  if (!score || !idx || !count) return fail(KP2D_ERR_ARG, "null argument");
  if (B < 1 || n < 1) return fail(KP2D_ERR_ARG, "empty score map (B=%d, n=%d)", B, n);
  if (k < 1) return fail(KP2D_ERR_ARG, "k must be >= 1 (pass k = n for \"every cell above the threshold\")");
  DeviceGuard guard(score, (hipStream_t)stream);
  TopkArgs a{score, B, n, k, thr, idx, val, count};
  int e = launch_topk(a, (hipStream_t)stream);
  if (e) return fail(KP2D_ERR_HIP, "topk kernel: %d", e);
  return KP2D_OK;
}

int kp2d_select_keypoints(const float* score, const float* coord, const float* desc, int B, int C, int n, int k, float thr,
                          int32_t* idx, float* val, int32_t* count, float* pts, float* dsel, void* stream) {
  if (!score || !coord || !desc || !idx || !count || !pts || !dsel) return fail(KP2D_ERR_ARG, "null argument");
  if (B < 1 || n < 1 || C < 1) return fail(KP2D_ERR_ARG, "empty score map (B=%d, n=%d, C=%d)", B, n, C);
  if (k < 1) return fail(KP2D_ERR_ARG, "k must be >= 1 (pass k = n for \"every cell above the threshold\")");
  DeviceGuard guard(score, (hipStream_t)stream);
  // Two launches.  (The gather inside the top-k kernel — the sorted keys are still in LDS — was built and measured: one
  // workgroup per frame fetching k * C scattered values is 20 us slower at a single frame than the 250 workgroups of
  // the gather kernel, 0.276 -> 0.295 ms per frame.)
  const TopkArgs a{score, B, n, k, thr, idx, val, count};
  int e = launch_topk(a, (hipStream_t)stream);
  if (e) return fail(KP2D_ERR_HIP, "topk kernel: %d", e);
  const GatherArgs g{coord, desc, idx, pts, dsel, B, C, n, k};
  e = launch_gather(g, (hipStream_t)stream);
  if (e) return fail(KP2D_ERR_HIP, "gather kernel: %d", e);
  return KP2D_OK;
}

int kp2d_gather_keypoints(const float* coord, const float* desc, const int32_t* idx, int B, int C, int n, int k,
                          float* pts, float* dsel, void* stream) {
  if (!coord || !desc || !idx || !pts || !dsel) return fail(KP2D_ERR_ARG, "null argument");
  DeviceGuard guard(coord, (hipStream_t)stream);
  GatherArgs a{coord, desc, idx, pts, dsel, B, C, n, k};
  int e = launch_gather(a, (hipStream_t)stream);
  if (e) return fail(KP2D_ERR_HIP, "gather kernel: %d", e);
  return KP2D_OK;
}

int kp2d_preprocess(const uint8_t* frames, int B, int Hs, int Ws, float* x, int H, int W, void* stream) {
  if (!frames || !x || B < 1 || Hs < 1 || Ws < 1 || H < 1 || W < 1) return fail(KP2D_ERR_ARG, "bad preprocess arguments");
  DeviceGuard guard(x, (hipStream_t)stream);     // the OUTPUT is always device memory (frames may be pinned host memory)
  int e = launch_preprocess(frames, x, B, Hs, Ws, H, W, (hipStream_t)stream);
  if (e) return fail(KP2D_ERR_HIP, "preprocess kernel: %d", e);
  return KP2D_OK;
}

// scratch layout of the matcher: [train_best u64 B*max1][rnn_idx i32 B*max1][rnn_dist f32 B*max1][partials]
static constexpr int kMatchSlices = 16;
static bool match_wants_slices(int B, int max0, int max1) {
  // few pairs: the train range of a query is split over several workgroups (match.hip knn2()); needs partial arrays
  return (long)((std::min(max0, max1) + 63) / 64) * B < 512;
}
size_t kp2d_match_scratch_bytes(int B, int max0, int max1) {
  if (B < 1 || max0 < 1 || max1 < 1) return 0;
  size_t n = (size_t)B * max1 * 16;
  if (match_wants_slices(B, max0, max1)) n += (size_t)kMatchSlices * B * std::max(max0, max1) * 12;
  return (n + 255) & ~(size_t)255;
}

int kp2d_match_descriptors_ex(const float* d0, const int32_t* n0, const float* d1, const int32_t* n1, int B, int max0,
                              int max1, int C, float ratio, const int32_t* cls0, const int32_t* cls1, uint32_t flags,
                              int32_t* nn_idx, float* nn_dist, float* nn_dist2, int32_t* match_q, float* match_d,
                              void* scratch, size_t scratch_bytes, void* stream) {
  if (!d0 || !n0 || !d1 || !n1 || !nn_idx || !nn_dist || !nn_dist2 || !match_q || !match_d || !scratch)
    return fail(KP2D_ERR_ARG, "null argument");
  if (B < 1 || max0 < 1 || max1 < 1) return fail(KP2D_ERR_ARG, "empty match problem");
  if ((cls0 == nullptr) != (cls1 == nullptr)) return fail(KP2D_ERR_ARG, "class ids must be given for both sides or neither");
  if (flags & ~(uint32_t)KP2D_MATCH_MUTUAL) return fail(KP2D_ERR_ARG, "unknown match flags 0x%x", flags);
  if (scratch_bytes < (size_t)B * max1 * 16) return fail(KP2D_ERR_WORKSPACE, "match scratch %zu B < required %zu B (kp2d_match_scratch_bytes)", scratch_bytes, kp2d_match_scratch_bytes(B, max0, max1));
  if ((uintptr_t)scratch % 8) return fail(KP2D_ERR_WORKSPACE, "match scratch must be 8-byte aligned");
  DeviceGuard guard(d0, (hipStream_t)stream);
  MatchArgs a{d0, d1, n0, n1, B, max0, max1, C, ratio, nn_idx, nn_dist, nn_dist2,
              reinterpret_cast<unsigned long long*>(scratch), match_q, match_d};
  a.cls0 = cls0; a.cls1 = cls1;
  a.mutual = (flags & KP2D_MATCH_MUTUAL) ? 1 : 0;
  char* p = reinterpret_cast<char*>(scratch) + (size_t)B * max1 * 8;
  a.rnn_idx = reinterpret_cast<int32_t*>(p); p += (size_t)B * max1 * 4;
  a.rnn_dist = reinterpret_cast<float*>(p); p += (size_t)B * max1 * 4;
  const size_t left = scratch_bytes - (size_t)B * max1 * 16;
  const size_t per_slice = (size_t)B * std::max(max0, max1) * 12;
  if (match_wants_slices(B, max0, max1) && left >= 2 * per_slice) {
    a.part_slices = (int)std::min<size_t>(kMatchSlices, left / per_slice);
    const size_t n = (size_t)a.part_slices * B * std::max(max0, max1);
    a.part_idx = reinterpret_cast<int32_t*>(p);
    a.part_d = reinterpret_cast<float*>(p + n * 4);
    a.part_d2 = reinterpret_cast<float*>(p + n * 8);
  }
  int e = launch_match(a, (hipStream_t)stream);
  if (e) return fail(e < 0 ? KP2D_ERR_UNSUPPORTED : KP2D_ERR_HIP, "match kernels: %d (descriptor width %d)", e, C);
  return KP2D_OK;
}

int kp2d_match_descriptors(const float* d0, const int32_t* n0, const float* d1, const int32_t* n1, int B, int max0,
                           int max1, int C, float ratio, int32_t* nn_idx, float* nn_dist, float* nn_dist2,
                           int32_t* match_q, float* match_d, void* scratch, void* stream) {
  if (!d0 || !n0 || !d1 || !n1 || !nn_idx || !nn_dist || !nn_dist2 || !match_q || !match_d || !scratch)
    return fail(KP2D_ERR_ARG, "null argument");
  if (B < 1 || max0 < 1 || max1 < 1) return fail(KP2D_ERR_ARG, "empty match problem");
  DeviceGuard guard(d0, (hipStream_t)stream);
  MatchArgs a{d0, d1, n0, n1, B, max0, max1, C, ratio, nn_idx, nn_dist, nn_dist2,
              reinterpret_cast<unsigned long long*>(scratch), match_q, match_d};
  int e = launch_match(a, (hipStream_t)stream);
  if (e) return fail(e < 0 ? KP2D_ERR_UNSUPPORTED : KP2D_ERR_HIP, "match kernels: %d (descriptor width %d)", e, C);
  return KP2D_OK;
}

int kp2d_match_pairs(const int32_t* match_q, const float* match_d, const float* pts0, const float* pts1, int B, int max0,
                     int max1, float* pairs, int32_t* idx, float* dist, int32_t* count, void* stream) {
  if (!match_q || !count || (dist && !match_d) || (pairs && (!pts0 || !pts1))) return fail(KP2D_ERR_ARG, "null argument");
  if (B < 1 || max0 < 1 || max1 < 1) return fail(KP2D_ERR_ARG, "empty match problem");
  DeviceGuard guard(match_q, (hipStream_t)stream);
  PairsArgs a{match_q, match_d, pts0, pts1, B, max0, max1, pairs, idx, dist, count};
  int e = launch_match_pairs(a, (hipStream_t)stream);
  if (e) return fail(KP2D_ERR_HIP, "match pairs kernel: %d", e);
  return KP2D_OK;
}

int kp2d_match_topk_pairs(int mode, const int32_t* match_q, const int64_t* matches0, const float* val, const float* pts0,
                          const float* pts1, int B, int max0, int max1, int k, float* pairs, int32_t* idx, float* out_val,
                          int32_t* count, void* scratch, size_t scratch_bytes, void* stream) {
  if (mode != KP2D_TOPK_BF && mode != KP2D_TOPK_LG) return fail(KP2D_ERR_ARG, "mode is KP2D_TOPK_BF or KP2D_TOPK_LG");
  if (!val || !count || !scratch || (mode == KP2D_TOPK_BF ? !match_q : !matches0) || (pairs && (!pts0 || !pts1)))
    return fail(KP2D_ERR_ARG, "null argument");
  if (B < 1 || max0 < 1 || max1 < 1) return fail(KP2D_ERR_ARG, "empty match problem");
  const int n = mode == KP2D_TOPK_BF ? max1 : max0;
  const int kcap = (k <= 0 || k > n) ? n : k;
  if (scratch_bytes < kp2d_match_topk_scratch_bytes(B, max0, max1)) return fail(KP2D_ERR_WORKSPACE, "match top-k scratch too small");
  if (reinterpret_cast<uintptr_t>(scratch) & 3) return fail(KP2D_ERR_ARG, "match top-k scratch must be 4-byte aligned");
  DeviceGuard guard(val, (hipStream_t)stream);
  TopkPairsArgs a{};
  a.mode = mode; a.match_q = match_q; a.matches0 = reinterpret_cast<const long long*>(matches0); a.val = val;
  a.pts0 = pts0; a.pts1 = pts1; a.B = B; a.n = n; a.max0 = max0; a.max1 = max1; a.kcap = kcap;
  a.keys = static_cast<float*>(scratch);
  a.sel = reinterpret_cast<int32_t*>(a.keys + (size_t)B * n);
  a.pairs = pairs; a.idx = idx; a.out_val = out_val; a.count = count;
  int e = launch_match_topk_pairs(a, (hipStream_t)stream);
  if (e) return fail(KP2D_ERR_HIP, "match top-k pairs: %d", e);
  return KP2D_OK;
}

size_t kp2d_match_topk_scratch_bytes(int B, int max0, int max1) {
  const size_t n = (size_t)(max0 > max1 ? max0 : max1);
  return (size_t)(B > 0 ? B : 0) * n * 8;      // keys [B][n] float + selection [B][<= n] int32
}

int kp2d_set_profiling(kp2d_model* m, int on) {
  if (!m) return fail(KP2D_ERR_ARG, "null model");
  m->profiling = on != 0;
  m->prof_used = 0;
  return KP2D_OK;
}

int kp2d_profile_count(kp2d_model* m) {
  if (!m) return fail(KP2D_ERR_ARG, "null model");
  if (!m->profiling) return fail(KP2D_ERR_STATE, "profiling is off");
  if (m->prof_used) {
    hipError_t e = hipEventSynchronize(m->prof[m->prof_used - 1].e1);
    if (e != hipSuccess) return fail(KP2D_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(e));
  }
  return (int)m->prof_used;
}

int kp2d_profile_get(kp2d_model* m, int index, const char** layer, const char** kernel, float* ms, double* flops,
                     double* bytes) {
  if (!m || index < 0 || (size_t)index >= m->prof_used) return fail(KP2D_ERR_ARG, "profile index out of range");
  ProfRec& r = m->prof[index];
  if (layer) *layer = r.layer.c_str();
  if (kernel) *kernel = r.kernel.c_str();
  if (flops) *flops = r.flops;
  if (bytes) *bytes = r.bytes;
  if (ms) HIP_TRY(hipEventElapsedTime(ms, r.e0, r.e1));
  return KP2D_OK;
}

int kp2d_set_precision(kp2d_model* m, int mode) {
  if (!m || (mode != KP2D_PREC_FP32 && mode != KP2D_PREC_F16X3)) return fail(KP2D_ERR_ARG, "precision must be KP2D_PREC_FP32 or KP2D_PREC_F16X3");
  m->precision = mode;
  m->plan_cache.clear();      // (which layers run merged depends on the arithmetic mode: plan sizes are memoised per mode)
  return KP2D_OK;
}

int kp2d_get_precision(const kp2d_model* m) { return m ? m->precision : KP2D_ERR_ARG; }

int kp2d_set_tap(kp2d_model* m, const char* layer, float* dst, size_t capacity_floats) {
  if (!m) return fail(KP2D_ERR_ARG, "null model");
  if (!layer || !dst) { m->tap_name.clear(); m->tap_dst = nullptr; m->tap_cap = 0; return KP2D_OK; }
  m->tap_name = layer; m->tap_dst = dst; m->tap_cap = capacity_floats;
  return KP2D_OK;
}

int kp2d_set_option(kp2d_model* m, const char* key, long value) {
  if (!m || !key) return fail(KP2D_ERR_ARG, "bad argument");
  const std::string k = key;
  m->plan_cache.clear();      // (an option may change the plan: sizes are memoised per setting)
  if (k == "wsm_min_items") {
    if (value > 0x7fffffffL || value < -1) return fail(KP2D_ERR_ARG, "wsm_min_items out of range");
    m->wsm_min = (int)value;
    return KP2D_OK;
  }
  if (k == "s16_min_items") {     // conv3x3_s16.hip (split activations through the backbone's 32-channel stage): 0 automatic, N from N tiles, -1 never
    if (value > 0x7fffffffL || value < -1) return fail(KP2D_ERR_ARG, "s16_min_items out of range");
    m->s16_min = (int)value;
    return KP2D_OK;
  }
  if (k == "mff_fused") {         // 1 (default): depthwise 3x3 -> 1x1 -> GELU -> 1x1 of the attention modules' MixFeedForward as one launch
    if (value < 0 || value > 1) return fail(KP2D_ERR_ARG, "mff_fused is 0 or 1");
    m->mff_fused = value != 0;
    return KP2D_OK;
  }
  if (k == "stem_fusion") {       // 1 (default): conv1a in split-fp16 products, inside conv1b's launch on big grids; 2: never fused; 0: exact-fp32 FMA kernels
    if (value < 0 || value > 2) return fail(KP2D_ERR_ARG, "stem_fusion is 0, 1 or 2");
    m->stem_fusion = (int)value;
    return KP2D_OK;
  }
  if (k == "side_overlap") {      // 1 (default): single frames run NetVLAD on a side stream beside the segmentation head; 0: in line
    if (value < 0 || value > 1) return fail(KP2D_ERR_ARG, "side_overlap is 0 or 1");
    m->side_overlap = value != 0;
    if (!m->side_overlap && m->side_stream) {      // give the stream (and the hardware queue it maps to) back
      DeviceGuard guard(m->cfg.device);
      (void)hipStreamSynchronize(m->side_stream);
      (void)hipStreamDestroy(m->side_stream);
      (void)hipEventDestroy(m->side_fork);
      (void)hipEventDestroy(m->side_join);
      m->side_stream = nullptr; m->side_fork = m->side_join = nullptr;
    }
    return KP2D_OK;
  }
  if (k == "s16_all") {      // 1 (default): S16P tensors between the warp-specialised 3x3 layers of big grids; 0: only inside the 32-channel stage
    if (value < 0 || value > 1) return fail(KP2D_ERR_ARG, "s16_all is 0 or 1");
    m->s16_all = value != 0;
    m->plan_cache.clear();
    return KP2D_OK;
  }
  if (k == "multi_launch") {      // 1 (default): layers of different heads that wait for the same predecessor run as one launch on small grids
    if (value < 0 || value > 1) return fail(KP2D_ERR_ARG, "multi_launch is 0 or 1");
    m->multi_launch = value != 0;
    return KP2D_OK;
  }
  if (k == "ws_min_tiles") {
    if (value < 0 || value > 0x7fffffffL) return fail(KP2D_ERR_ARG, "ws_min_tiles out of range");
    m->ws_min = (int)value;
    return KP2D_OK;
  }
  if (k == "lanes") {      // stream lanes of one forward: 0 = the default (KP2D_LANES, else 2); 1 when the CALLER keeps several batches in flight
    if (value < 0 || value > 8) return fail(KP2D_ERR_ARG, "lanes is 0 (default) .. 8");
    m->lanes = value == 0 ? m->lanes_default : (int)value;
    return KP2D_OK;
  }
  if (k == "wsm_transposed") {      // conv3x3_wsm.hip: tiles walk the map transposed — 0 never (default), 1 always, 2 where cheaper
    if (value < 0 || value > 2) return fail(KP2D_ERR_ARG, "wsm_transposed is 0, 1 or 2");
    m->wsm_tr = (int)value;
    return KP2D_OK;
  }
  if (k == "wsm_grid") {
    if (value < 0 || value > 65536) return fail(KP2D_ERR_ARG, "wsm_grid out of range");
    m->wsm_grid = (int)value;
    return KP2D_OK;
  }
  return fail(KP2D_ERR_ARG, "unknown option '%s'", key);
}

int kp2d_set_seg_ids(kp2d_model* m, int64_t* ids, size_t capacity) {
  if (!m) return fail(KP2D_ERR_ARG, "null model");
  m->seg_ids_dst = ids;
  m->seg_ids_cap = ids ? capacity : 0;
  return KP2D_OK;
}

int kp2d_set_chunk_frames(kp2d_model* m, int frames) {
  if (!m || frames < 0) return fail(KP2D_ERR_ARG, "bad argument");
  m->chunk_frames = frames;
  return KP2D_OK;
}

}  // extern "C"
