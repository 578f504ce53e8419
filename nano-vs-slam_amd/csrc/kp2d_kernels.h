// Internal launch interface between the C-ABI host code (kp2d_api.cpp) and the HIP kernels.
// Not part of the public ABI (that is include/kp2d.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kp2d {

// ---- activation / epilogue selectors -------------------------------------------------------
enum Act : int {
  ACT_NONE = 0,
  ACT_LEAKY = 1,           // LeakyReLU(0.01)       modules/base.py:33
  ACT_RELU = 2,            // ReLU (to_mcu configs) modules/base.py:35
  ACT_SIGMOID = 3,         // score head            models/kp2dtiny.py:574
  ACT_TANH = 4,            // loc head              models/kp2dtiny.py:575
  ACT_SIGMOID0_TANH = 5,   // V3 fused score/loc: ch0 sigmoid, ch1..2 tanh  models/kp2dtiny.py:927-935
  ACT_SOFTMAX_C = 6,       // V3 eval: Softmax2d over classes                models/kp2dtiny.py:942-943
  ACT_GELU = 7,            // exact-erf GELU inside MixFeedForward           modules/segformer.py:185
};

enum Store : int {
  ST_NHWC = 0,             // out0[pixel][os0] (+oo0), full resolution
  ST_NHWC_POOL = 1,        // out1 = MaxPool2d(2,2) of the activation only
  ST_NHWC_BOTH = 2,        // out0 full-res AND out1 pooled (conv3b: skip + x)
  ST_SHUFFLE = 3,          // PixelShuffle(2) folded into the store: out0 is the 2H x 2W NHWC tensor
  ST_NCHW = 4,             // API-facing planar output; channels [0,nsplit) -> out0, [nsplit,cout) -> out1
  ST_S16P = 5,             // out0 is an S16P tensor (below), full resolution
  ST_S16P_POOL = 6,        // out1 = MaxPool2d(2,2) of the activation as an S16P tensor (conv1b)
  ST_S16P_BOTH = 7,        // out0 full-res AND out1 pooled, both S16P tensors (conv3b: skip + x)
  ST_S16P_SHUFFLE = 8,     // PixelShuffle(2) folded into the store, out0 the 2H x 2W S16P tensor (cout / 4 a multiple of 32)
  ST_MIX16 = 9,            // 64-channel groups below channel `nsplit`: fp32 NHWC into out0 (os0 channels); from `nsplit` on: the
                           // S16P tensor out1 (os1 channels, its chunk 0 = channel nsplit) — the heads' merged first layer, whose
                           // score / location slices are read by the fp32 dot-product kernels and the rest by split-fp16 convs
};
// S16P ("split, planar rows"): an activation kept as the fp16 halves the split-fp16 kernels multiply, x = hi + lo with
// hi = fp16(x), lo = fp16(x - hi) — per frame [C / 16 chunks][H][plane: hi | lo][W][16 halves], the same bytes as fp32 NHWC.
// A tile row of one plane is contiguous, so the consumer copies its LDS operand image straight from HBM (conv3x3_s16.hip).
// Only between layers of one forward (workspace tensors); C a multiple of 16.  Readers: conv3x3_s16.hip (32 input channels, the
// layer's weights resident in LDS) and conv3x3_wsm.hip's IN16 form (any whole number of chunks, one or two S16P sources).

// One 3x3 / stride 1 / pad 1 (taps = 9) or 1x1 (taps = 1) convolution over an NHWC activation that may be
// the channel-concat of two tensors (torch.cat([up, skip], 1): heads.py:99, segmentation.py:141,149).
// A source is addressed as ptr + b*bs + y*rs + x*ps + o + c, so strided views work too: the 2x2 stride-2
// to_kv conv (modules/segformer.py:93-95) is a 1x1 conv over two row-views of the full-resolution tensor.
struct ConvSrc { const float* p; int c, o; long bs, rs, ps; int fmt; };   // channels taken, first channel, strides (floats); fmt 1: an S16P tensor (bs only)
struct ConvArgs {
  ConvSrc in0, in1;
  int taps;                           // 9 or 1
  int prec;                           // 0: exact fp32 MFMA, 1: split-fp16 3xMFMA (weights packed as hi|lo halves)
  const float* w;                     // packed [group][cin_pad/KC][taps][ng][KC]
  const float* scale;                 // [npad]  BN: gamma/sqrt(var+eps); bias conv: 1
  const float* shift;                 // [npad]  BN: beta - mean*scale;   bias conv: bias
  float* out0; int os0, oo0;
  float* out1; int os1, oo1;
  int B, H, W;                        // conv resolution
  int cin, cout, npad;
  int act, store, nsplit;
  int tiles_x, tiles_y;
  int ng32;                           // 1: w holds 32-channel groups although npad >= 64 (small grids)
  int wsm_min;                        // least (tile, group) work items for the warp-specialised multi-chunk form; 0: automatic (KP2D_WSM, else one per workgroup); < 0: never
  const float* w_tr;                  // the 64-channel-group pack with the taps transposed (dy <-> dx), nullptr: none (conv3x3_wsm.hip: transposed tiles)
  int wsm_tr;                         // conv3x3_wsm.hip: tiles walk the map transposed (tile rows = map columns).  In: 0 never, 1 always, 2 where cheaper; the launcher hands the kernel its decision (0 / 1)
  int wsm_lanes;                      // stream lanes launching side by side (the form takes CUs / lanes workgroups)
  long long* ids_out;                 // ST_NCHW, one channel group: also write argmax over the stored channels per pixel, [B][H][W] int64 (nullptr: no)
  int ws_min;                         // least tiles for the warp-specialised conv1b form (0: 1024)
  int wsm_grid;                       // most workgroups of that form per launch; 0: KP2D_WSM_GRID or one per CU
  // conv1b's warp-specialised form with conv1a computed by its staging waves (conv3x3_f16.hip STEM): the frames [B,3,H,W] and
  // conv1a's weights [27][16] / folded BatchNorm; in0 is then unused.  nullptr: conv1a is its own launch
  const float* stem_x; const float* stem_w; const float* stem_scale; const float* stem_shift; const float* stem_wscale; int stem_act;   // stem_wscale: device pointer to 2^e
  int s16_min;                        // conv3x3_s16.hip: least work items for the form (0: automatic, three rounds per workgroup)
  int wsm_force;                      // the plan fixed this layer's tensor layouts on conv3x3_wsm.hip running it (S16P in or out): no item-count policy
  int dbg;                            // timing ablations only (KP2D_DBG): 1 skip the epilogue, 2 skip LDS commit, 4 skip global loads, 8 skip MFMA, 64 skip only the epilogue's global stores
};

struct Conv1aArgs {                   // backbone.conv1a: NCHW frame in -> NHWC out, Cin = 3 (RGB) or 1 (use_color=False)
  const float* x;                     // [B,cin,H,W]
  const float* w;                     // [9*cin][cout]  (k = ci*9 + dy*3 + dx)
  int cin;
  const float* scale; const float* shift;
  float* out; int B, H, W, cout, act;
};

int launch_conv3x3(const ConvArgs& a, int kc, hipStream_t s);
// tile form the calling thread's last launch_conv3x3* call chose ("<2,1,16>", "ws", "wsm", ...; "" for the fp32 / 1x1 kernels):
// kp2d_profile_get reports it behind the kernel family, so a test can assert WHICH kernel it covered
const char* conv3x3_last_variant();
void conv3x3_note_variant(const char* v);
int launch_conv1a(const Conv1aArgs& a, hipStream_t s);
int launch_conv1a_u8(const Conv1aArgs& a, const unsigned char* frames, int Hs, int Ws, hipStream_t s);   // frame front-end fused in
// conv3x3_f16.hip: conv1a in the split-fp16 arithmetic of the fused first layer (STEM), bit-identical to it; frames != null: uint8 frames in
int launch_conv1a_mfma(const Conv1aArgs& a, const float* wscale_dev, const unsigned char* frames, int Hs, int Ws, hipStream_t s);
int launch_head3x3_pair(const ConvArgs& a0, const ConvArgs& a1, hipStream_t s);   // a 1-channel and a 2-channel head, one launch
int launch_head3x3(const ConvArgs& a, hipStream_t s);         // head3x3.hip: taps = 9, cout <= 4, planar outputs (exact fp32 dot products)
int launch_conv3x3_f16x3(const ConvArgs& a, hipStream_t s);   // conv3x3_f16.hip: taps = 9, prec = 1 (16x16x32 MFMA)
// 2-4 independent small-grid layers as one launch; -1000: not all of them are layers of the single-frame form
int launch_conv3x3_f16x3_multi(const ConvArgs* list, int n, hipStream_t s);
// conv3x3_wsm.hip: the same layers, 64-channel groups, warp-specialised and persistent; -1000 = not eligible / fewer than min_items work items
int launch_conv3x3_f16x3_wsm(const ConvArgs& a, hipStream_t s, int n_item);   // n_item: 64 (64-channel groups) or 32 (32-channel layers)
// conv3x3_s16.hip: 32-input-channel layers whose input is an S16P tensor (in0.fmt == 1); -1000: in0 is not S16P
int launch_conv3x3_f16x3_s16(const ConvArgs& a, hipStream_t s);
// would that form run for a B x H x W map (the plan decides the activation layout of conv1b .. conv3a's outputs by it)
bool conv3x3_s16_would_run(int B, int H, int W, int lanes, int min_items, int grid_opt);
// would the warp-specialised conv1b form (conv3x3_f16.hip, the only producer of a pooled S16P tensor) run
bool conv3x3_ws_would_run(int B, int H, int W, int ws_min);
// kp2d_set_tap on an S16P tensor: channels [c0, c0 + C) of a Ct-channel tensor -> planar fp32
int launch_s16p_to_nchw(const float* in, float* out, int B, int C, int H, int W, int Ct, int c0, hipStream_t s);
// would conv3x3_wsm.hip's automatic policy take a 64-channel-group layer of `groups` groups on a B x H x W map
bool conv3x3_wsm_would_run(int B, int H, int W, int groups, int lanes, int wsm_min, int grid_opt, int full_rounds);

// ---- NetVLAD (modules/aggregators/netvlad.py:79-106) ---------------------------------------
struct VladArgs {
  const float* x;        // encoder output, NHWC [B][S][C]
  const float* wa;       // soft-assign 1x1 conv weight [K][C]
  const float* cent;     // centroids [K][C]
  float* part;           // workspace [B][nsplit][K*C + K]
  float* out;            // [B][K*C]
  int B, S, C, K, nsplit;
  int tps = 1;           // > 1: one workgroup per 64-pixel tile, tps tiles per slab (netvlad_tiles_per_slab); part holds nsplit*tps rows
  int prec = 0;          // 1: the soft-assignment logits as split-fp16 products (f16x3 mode); the aggregation is exact fp32 in both
};
int launch_netvlad(const VladArgs& a, hipStream_t s);
struct PoolArgs { const float* x; const float* p; float* out; int B, C, Hc, Wc; };   // GeM (p = exponent) / ConvAP pooling
int launch_gem(const PoolArgs& a, hipStream_t s);
int launch_convap_pool(const PoolArgs& a, hipStream_t s);
int netvlad_nsplit(int S);
int netvlad_tiles_per_slab(int S, int B);

// ---- post-processing (models/kp2dtiny.py:593-647 / 959-1015) -------------------------------
struct PostArgs {
  const float* score_in;  // [B,1,Hc,Wc] sigmoid score
  const float* shift;     // [B,2,Hc,Wc] tanh shift
  const float* feat;      // [B,C,Hf,Wf] dense descriptors (NCHW)
  float* score_out;       // [B,1,Hc,Wc]
  float* coord;           // [B,2,Hc,Wc] pixels, ch0 = x
  float* desc;            // [B,C,Hc,Wc] sampled + L2 normalised (nullptr: training mode, no sampling)
  int B, C, Hc, Wc, Hf, Wf, H, W, cell;
  float cross_ratio;
};
int launch_post(const PostArgs& a, hipStream_t s);

struct ArgmaxArgs { const float* seg; int64_t* ids; int B, C, HW; };
int launch_seg_argmax(const ArgmaxArgs& a, hipStream_t s);
int launch_post_seg(const PostArgs& a, const ArgmaxArgs& g, hipStream_t s);      // both in one launch when shapes allow
struct SegSampleArgs { const float* seg; const float* coord; int64_t* ids; int B, C, Hs, Ws, Hc, Wc, H, W; };
int launch_seg_sample_argmax(const SegSampleArgs& a, hipStream_t s);

// ---- keypoint selection (callers K1/K2/K3) ----------------------------------
struct TopkArgs {
  const float* score;   // [B][n]
  int B, n, k;
  float thr;            // keep score > thr; pass -inf for plain top-k
  int32_t* idx;         // [B][k] flat cell indices, score desc / index asc on ties; -1 padded
  float* val;           // [B][k] scores (0 padded), may be nullptr
  int32_t* count;       // [B]
};
int launch_topk(const TopkArgs& a, hipStream_t s);

struct GatherArgs {     // gather coords / descriptors of selected cells into [B][k][2] / [B][k][C]
  const float* coord; const float* desc; const int32_t* idx;
  float* pts; float* dsel; int B, C, n, k;
};
int launch_gather(const GatherArgs& a, hipStream_t s);

// ---- SegFormerAttentionModule pieces (modules/segformer.py:63-220) ---------------------------
struct LnArgs { const float* x; const float* g; const float* b; float* y; long npix; int C; };
int launch_channel_layernorm(const LnArgs& a, hipStream_t s);

struct DwArgs { const float* x; const float* w; const float* bias; float* y; int B, H, W, C; };   // w: [9][C]
int launch_dwconv3x3(const DwArgs& a, hipStream_t s);

// MixFeedForward's tail fused (mff_tail.hip): depthwise 3x3 + bias -> 1x1 (128 -> 128) + bias -> GELU -> 1x1 (128 -> 64) + bias [-> MaxPool2d(2,2)]
struct MffTailArgs {
  const float* h;                                       // net.0's output, NHWC [B][H][W][128]
  const float* wdw; const float* bdw;                   // depthwise weights [9][128], bias [128]
  const float* w1; const float* sc1; const float* sh1;  // net.1.net.1: split-fp16 pack (64-channel groups), scale (2^-e), bias
  const float* w3; const float* sc3; const float* sh3;  // net.3
  float* out;                                           // [B][H][W][64], or pooled [B][H/2][W/2][64]
  int B, H, W, pool;
};
int launch_mff_tail(const MffTailArgs& a, hipStream_t s);

struct AttnArgs {
  const float* q;      // [B][S][C]   (to_q output, NHWC)
  const float* kv;     // [B][T][2C]  (to_kv output: k = channels [0,C), v = [C,2C))
  float* out;          // [B][S][C]
  int B, S, T, C, heads;
  float scale;         // (C/heads)^-0.5
  // row strides / channel offsets in floats; 0 = the dense defaults above (q_stride C, kv_stride 2C, k_off 0,
  // v_off C, out_stride C).  The LightGlue blocks read q, k, v as slices of one [q|k|v] or [qk|v] row.
  int q_stride = 0, kv_stride = 0, k_off = 0, v_off = 0, out_stride = 0;
  int prec = 0;        // 1: split-fp16 operands on v_mfma_f32_32x32x16_f16 (head dim <= 16), 0: exact fp32 MFMA
  const int32_t* tcount = nullptr;   // [B] keys that exist in batch item b's key / value sequence (the rest of its T rows is padding), null: all T
  int kv_bshift = 0;   // keys / values of batch item b come from item (b + kv_bshift) % B (LightGlue cross attention
                       // of both directions in one launch: items [0,B/2) are image 0, [B/2,B) image 1)
};
int launch_attention(const AttnArgs& a, hipStream_t s);

void set_last_error(const char* msg);   // thread-local message behind kp2d_last_error() (kp2d_api.cpp)

// ---- LightGlue matcher (lightglue/lightglue.py; kernels in lightglue.hip) -----------------------
struct LgPosArgs {
  const float* k0; const float* k1;         // keypoints [B][M][2] / [B][N][2], pixels (x, y)
  const float* size0; const float* size1;   // image size [B][2] (w, h) or null: 1 + max - min of the keypoints
  const float* wr;                          // posenc.Wr.weight [hd/2][2]
  float* cs;                                // [B*M + B*N][hd]: cos(hd/2) | sin(hd/2) per token
  int B, M, N, hd;
};
int launch_lg_posenc(const LgPosArgs& a, hipStream_t s);

enum LgEpi { LG_EPI_NONE = 0, LG_EPI_ROTARY = 1, LG_EPI_LNGELU = 2, LG_EPI_RESID = 3 };
struct LgLinArgs {
  const float* x0; const float* x1;         // input = [x0 row (k0 wide) | x1 row (k1 wide)]; x1 unused when k1 == 0
  int k0, k1, xs0, xs1;                     // widths and row strides (floats)
  const float* w;                           // W^T packed [K][nout] (nout padded to a multiple of 32)
  const float* bias;                        // [nout] or null
  float* out; int os, oo;                   // output row stride / column offset
  int rows, nout, nvalid;                   // nvalid <= nout columns are stored
  int epi;
  const float* cs; int hd, rot_cols;        // ROTARY: per-row cos|sin, head dim, leading columns that rotate
  const float* ln_g; const float* ln_b;     // LNGELU: LayerNorm affine over the nout columns
  const float* res; int rs;                 // RESID: out = res + y
};
int launch_lg_linear(const LgLinArgs& a, hipStream_t s);

// fused tail of a Self/CrossBlock for D = 32: x += ffn(cat[x, out_proj(ctx)])  (lightglue.py:260-261 / :322-326)
struct LgTailArgs {
  float* x; const float* ctx;           // [rows][D] in place / attention context (null: projection only)
  // split-fp16 matrix-core images of W (lightglue_api.cpp mfma_image()) and fp32 biases
  const void* io; const float* bo;       // out_proj / to_out  [D -> D]
  const void* i1; const float* b1;       // ffn.0              [2D -> 2D]
  const float* ln_g; const float* ln_b;  // ffn.1
  const void* i2; const float* b2;       // ffn.3              [2D -> D]
  int rows, D;
  // optional: the NEXT token-wise projection of the updated x in the same launch (the cross block's [to_qk | to_v], the
  // next layer's Wqkv with its rotary epilogue, or the final projection): out[row][0..nvalid) = x W^T + b
  const void* in = nullptr; const float* bn = nullptr;    // image [D -> nn] (nn = 64 or 96, padded), bias [nn]
  float* on = nullptr; int nn = 0, nos = 0, nvalid = 0;   // output, its row stride, columns stored
  const float* cs = nullptr; int hd = 0, rot_cols = 0;    // rotary: per-row cos | sin, head dim, leading columns that rotate
};
int launch_lg_tail(const LgTailArgs& a, hipStream_t s);

struct LgAssignArgs {
  const float* fz;                          // [B*M + B*N][fs]: final_proj(x) / D^0.25 in [0,D), matchability logit at D
  int fs, D, B, M, N;
  float* scores;                            // [B][M+1][N+1] log assignment (out)
  // scratch, per 64 x 64 tile of the inner block: [B][ceil(N/64)][M] for rows, [B][ceil(M/64)][N] for columns
  float* rp_m; float* rp_s; float* cp_m; float* cp_s;      // log-sum-exp partials of sim (maximum, sum of exp)
  float* rmax; int* rarg; float* cmax; int* carg;          // max / argmax partials of the final scores
  float th;                                 // filter_threshold
  const int32_t* cnt0 = nullptr; const int32_t* cnt1 = nullptr;   // [B] keypoints that exist in each set (rows past them are padding), null: M / N
  int64_t* matches0; int64_t* matches1; float* mscores0; float* mscores1;
};
int launch_lg_assign(const LgAssignArgs& a, hipStream_t s);

// ---- descriptor matching (src/visual_odometry/feature_matcher.py:89-98, 179-209) ---------------
struct MatchArgs {
  const float* d0; const float* d1;         // [B][max0][C] query / [B][max1][C] train descriptors
  const int32_t* n0; const int32_t* n1;     // [B] valid rows per pair
  int B, max0, max1, C;
  float ratio;
  int32_t* nn_idx; float* nn_dist; float* nn_dist2;   // [B][max0] k=2 neighbours of every query
  unsigned long long* train_best;                      // [B][max1] scratch: (distance bits << 32 | query)
  int32_t* match_q; float* match_d;                    // [B][max1] query matched to each train row (-1: none)
  // optional (kp2d_match_descriptors_ex):
  const int32_t* cls0 = nullptr; const int32_t* cls1 = nullptr;   // [B][max0] / [B][max1] class ids: a query sees its own class only
  int mutual = 0;                                      // 1: mutual nearest neighbours instead of ratio test + one-to-one
  int32_t* rnn_idx = nullptr; float* rnn_dist = nullptr;          // [B][max1] nearest query of every train row (mutual)
  int32_t* part_idx = nullptr; float* part_d = nullptr; float* part_d2 = nullptr;   // [slices][B][max(max0,max1)] partial results
  int part_slices = 0;                                 // train-range slices the partial arrays can hold (few pairs: more workgroups)
};
struct PairsArgs {       // compaction of the matched rows of every pair (train order)
  const int32_t* match_q; const float* match_d;        // [B][max1]
  const float* pts0; const float* pts1;                // [B][max0][2] / [B][max1][2] keypoints (x, y); may be null with pairs
  int B, max0, max1;
  float* pairs;          // [B][max1][4]  x0, y0, x1, y1 of match i (null: skip)
  int32_t* idx;          // [B][max1][2]  (query row, train row) (null: skip)
  float* dist;           // [B][max1]     distance (null: skip)
  int32_t* count;        // [B]
};
int launch_match_pairs(const PairsArgs& a, hipStream_t s);
struct TopkPairsArgs {   // at most kcap matched pairs per frame pair, best first (the VO loop's top_k_matches)
  int mode;              // 0: BF (match_q [B][n] + match distance, smaller is better; n = max1)   1: LightGlue (matches0 [B][n] int64 + matching score; n = max0)
  const int32_t* match_q; const long long* matches0; const float* val;
  const float* pts0; const float* pts1;
  int B, n, max0, max1, kcap;
  float* keys; int32_t* sel;       // scratch [B][n] / [B][kcap]
  float* pairs; int32_t* idx; float* out_val; int32_t* count;
};
int launch_match_topk_pairs(const TopkPairsArgs& a, hipStream_t s);
int launch_match(const MatchArgs& a, hipStream_t s);

// ---- small layout / elementwise kernels -----------------------------------------------------
int launch_preprocess(const unsigned char* src, float* dst, int B, int Hs, int Ws, int H, int W, hipStream_t s);
int launch_l2norm_channels(float* x, long npix, int C, hipStream_t s);
int launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, int istride, int ioff, hipStream_t s);

}  // namespace kp2d
