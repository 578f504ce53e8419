/* kp2d.h — C ABI of the MI355X-native kp2dtiny multi-task inference path.
 *
 * The reference (ETH-PBL/Nano-VS-SLAM) is pure Python: its boundary for this path is the
 * torch.nn.Module surface of KP2DTinyV2 / KP2DTinyV3 (src/kp2dtiny/models/kp2dtiny.py:284-1015).
 * This header is the layer UNDER that surface: the entry points a maintainer binds (ctypes stub in
 * INTEGRATION.md) so that Module.forward / Module.post_processing and the callers' keypoint
 * selectors run as hand-written gfx950 kernels.  Each function names the reference code it replaces.
 *
 * Conventions
 *   - plain C types only; every tensor is a raw pointer + sizes; float32 unless stated
 *   - "dev" pointers are HIP device pointers on the model's device, "host" pointers are CPU memory
 *   - API tensors are NCHW exactly as the reference returns them
 *   - every call returns 0 (KP2D_OK) or a negative kp2d_status; kp2d_last_error() gives the text
 *   - all device work is enqueued on the caller's stream; no call synchronises the device except
 *     kp2d_finalize_weights / kp2d_import_packed (one-time uploads) and kp2d_profile_* readers
 *   - the library never allocates caller-visible memory: outputs and the workspace are caller-owned
 *   - one handle per (device, stream); a handle is not thread-safe
 */
#ifndef KP2D_H_
#define KP2D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KP2D_ABI_VERSION 1

typedef enum kp2d_status {
  KP2D_OK = 0,
  KP2D_ERR_ARG = -1,          /* bad argument (null pointer, bad shape, H/W not divisible by 8 ...)      */
  KP2D_ERR_UNSUPPORTED = -2,  /* configuration outside the built path (see DESIGN.md "out of scope")     */
  KP2D_ERR_STATE = -3,        /* call order: weights not finalised, profiling off, ...                    */
  KP2D_ERR_WEIGHT = -4,       /* unknown key / wrong shape / missing tensor at finalise                    */
  KP2D_ERR_WORKSPACE = -5,    /* workspace too small or misaligned                                         */
  KP2D_ERR_HIP = -6           /* a HIP runtime call or kernel launch failed                                */
} kp2d_status;

typedef struct kp2d_model kp2d_model; /* opaque */

/* Constructor arguments of KP2DTinyV2.__init__ (kp2dtiny.py:301-319) / KP2DTinyV3.__init__ (:680-702)
 * that change the arithmetic.  get_config()/tiny_factory() (:221-281) live in the Python host layer. */
typedef struct kp2d_config {
  int32_t struct_size;      /* sizeof(kp2d_config), for ABI evolution                                   */
  int32_t version;          /* 2 = KP2DTinyV2, 3 = KP2DTinyV3                                            */
  int32_t channel_dims[6];  /* c1,c2,c3,c4,c5,d1                                                         */
  int32_t nfeatures;        /* descriptor channels                                                       */
  int32_t n_classes;        /* nClasses                                                                  */
  int32_t num_clusters;     /* NetVLAD K                                                                 */
  int32_t encoder_dim;      /* NetVLAD C                                                                 */
  int32_t downsample;       /* 2 for every S/N config (cell = 4)                                         */
  int32_t use_attention;    /* SegFormerAttentionModule x2 in the seg head                               */
  int32_t leaky_relu;       /* 1: LeakyReLU(0.01), 0: ReLU                                               */
  int32_t remove_softmax;   /* V3 only (kp2dtiny.py:698,942)                                             */
  int32_t device;           /* HIP device ordinal                                                        */
  int32_t global_descriptor;/* KP2D_GD_NETVLAD / KP2D_GD_GEM / KP2D_GD_CONVAP (vpr.py:53-76)                     */
  int32_t remove_netvlad;   /* to_export configs: "vlad" is the encoder map [B,enc,H/4,W/4] (vpr.py:84)          */
  int32_t depth;            /* depth=True: V2 second seg-like head, V3 third slice + featD (kp2dtiny.py:402-437)  */
  int32_t upscale_method;   /* KP2D_UP_PIXELSHUFFLE / KP2D_UP_CONVTRANSPOSE (to_mcu, kp2dtiny.py:271-273; base.py:80-117)  */
  int32_t in_channels;      /* 3 = RGB frames; 1 = KP2DTinyV3(use_color=False) (kp2dtiny.py:718-721); 0 means 3.  A caller built
                               against the struct without this field (struct_size 84) gets 3.                               */
} kp2d_config;
#define KP2D_UP_PIXELSHUFFLE 0
#define KP2D_UP_CONVTRANSPOSE 1
#define KP2D_GD_NETVLAD 0
#define KP2D_GD_GEM 1
#define KP2D_GD_CONVAP 2

/* kp2d_forward flags */
#define KP2D_FWD_EVAL 1u    /* model.training is False: V3 applies Softmax2d to seg (kp2dtiny.py:942-943) */
#define KP2D_FWD_ONLY_ENCODER 2u /* model.only_encoder(x) (kp2dtiny.py:515-518, vpr.py:78-89): backbone + convlad1-3 only;
                                    vlad = channel-wise L2-normalised encoder map [B,enc,Hc,Wc] (raw map when
                                    remove_netvlad); score/shift/feat/seg/depth are not written and may be NULL */

const char* kp2d_last_error(void);
int32_t kp2d_abi_version(void);

/* ---- lifetime ------------------------------------------------------------------------------- */
/* replaces: KP2DTinyV2(**conf, nClasses=..) / KP2DTinyV3(..) construction (eval_multitask.py:150-159) */
int kp2d_create(const kp2d_config* cfg, kp2d_model** out);
void kp2d_destroy(kp2d_model* m);

/* ---- weights: the state_dict is the wire format ---------------------------- */
/* enumerate the tensors the model expects, in the reference's registration order */
int kp2d_num_weights(const kp2d_model* m);
int kp2d_weight_info(const kp2d_model* m, int index, const char** key, int64_t shape[4], int* ndim);
/* replaces: model.load_state_dict(sd) (eval_multitask.py:161-167, demo.py:12-15); host float32, C-contiguous.
 * BatchNorm num_batches_tracked entries are not part of the arithmetic and are ignored if passed. */
int kp2d_set_weight(kp2d_model* m, const char* key, const float* host, const int64_t* shape, int ndim);
/* fold BatchNorm into per-channel scale/shift, re-lay conv weights for the kernels, upload.  Blocking. */
int kp2d_finalize_weights(kp2d_model* m);
/* the packed device blob, for the one-off RCCL broadcast rank 0 -> all ranks */
size_t kp2d_packed_bytes(const kp2d_model* m);
int kp2d_export_packed(const kp2d_model* m, void* dev_dst, void* stream);
int kp2d_import_packed(kp2d_model* m, const void* dev_src, void* stream);

/* ---- forward --------------------------------------------------------------------------------- */
/* scratch the caller must provide for a (B,H,W) call; 256-byte aligned device memory */
size_t kp2d_workspace_bytes(const kp2d_model* m, int B, int H, int W);
/* replaces: KP2DTinyV2.forward (kp2dtiny.py:552-591) / KP2DTinyV3.forward (:906-957).
 *   x      [B,3,H,W]  RGB in [-1,1] ([B,1,H,W] for in_channels = 1); H, W divisible by 8
 *   score  [B,1,H/4,W/4]  sigmoid, un-bordered      shift [B,2,H/4,W/4]  tanh ("coord" key of forward)
 *   feat   [B,nfeatures,H/2,W/2] dense descriptors   seg   [B,n_classes,H/2,W/2] logits (V3 eval: probabilities)
 *   vlad   [B,kp2d_vlad_dim]: NetVLAD K*C; GeM / ConvAP encoder_dim*16; remove_netvlad: [B,encoder_dim,H/4,W/4] */
size_t kp2d_vlad_dim(const kp2d_model* m, int H, int W);
/*   depth  [B,1,H/2,W/2] sigmoid, only for depth=1 models (NULL otherwise) */
int kp2d_forward(kp2d_model* m, const float* x, int B, int H, int W, uint32_t flags, float* score, float* shift,
                 float* feat, float* seg, float* vlad, float* depth, void* workspace, size_t workspace_bytes,
                 void* stream);

/* kp2d_forward with the frame front-end as the first layer's prologue: frames is uint8 [B,Hs,Ws,3]
 * on the device; /255, the bilinear resize to (H, W) and .sub(0.5).mul(2) (src/evaluation/visual_odometry.py:77-87)
 * happen while conv1a stages its input tile, so the float [B,3,H,W] frame is never written.  Bit-identical to
 * kp2d_preprocess followed by kp2d_forward.  RGB models with a 16-channel first layer (every S / N / F config);
 * others return KP2D_ERR_UNSUPPORTED (use the two calls). */
int kp2d_forward_frames(kp2d_model* m, const uint8_t* frames, int B, int Hs, int Ws, int H, int W, uint32_t flags,
                        float* score, float* shift, float* feat, float* seg, float* vlad, float* depth, void* workspace,
                        size_t workspace_bytes, void* stream);

/* replaces: post_processing (kp2dtiny.py:593-625 / :959-993).  `desc` and `seg_ids` may be NULL when the
 * module is in training mode (the reference skips sampling: kp2dtiny.py:615).
 *   score_out [B,1,Hc,Wc] border-zeroed   coord [B,2,Hc,Wc] pixels (ch0 = x)
 *   desc [B,C,Hc,Wc] bilinearly sampled, unit norm   seg_ids [B,1,Hs,Ws] int64 argmax over seg's channels */
int kp2d_post(kp2d_model* m, const float* score, const float* shift, const float* feat, const float* seg, int B,
              int H, int W, int Hc, int Wc, int feat_c, int Hf, int Wf, int seg_c, int Hs, int Ws, float* score_out,
              float* coord, float* desc, int64_t* seg_ids, int sample_segmentation, void* stream);
/* sample_segmentation != 0 (model.sample_segmentation, kp2dtiny.py:634-639): seg_ids is [B,1,Hc,Wc], the class of the
 * nearest seg pixel at each cell's coordinate, instead of the dense [B,1,Hs,Ws] argmax. */

/* replaces the callers' selectors: threshold + top-k on the cell grid, batched and on device
 * (evaluation/visual_odometry.py:105-117 K1, evaluation/descriptor.py:12-36 K2,
 *  gluefactory/models/extractors/kp2dtiny.py:38-42 K3).  Order: score descending, flat index ascending.
 *   score [B,n]; idx [B,k] (-1 padded); val [B,k] or NULL; count [B]; thr = -INFINITY for plain top-k.
 *   Any k >= 1: the reference's "no cap" (top_k <= 0: every cell above thr, frontend.py:122) is k = n.  k <= 16384
 *   selects and sorts in LDS; larger k sorts in place in the idx row (slower, same result). */
int kp2d_select_topk(const float* score, int B, int n, int k, float thr, int32_t* idx, float* val, int32_t* count,
                     void* stream);
/* gather the selected cells: pts [B,k,2] (x,y), dsel [B,k,C]; rows of padded (-1) entries are zero */
int kp2d_gather_keypoints(const float* coord, const float* desc, const int32_t* idx, int B, int C, int n, int k,
                          float* pts, float* dsel, void* stream);

/* kp2d_select_topk + kp2d_gather_keypoints as one call (still two launches): what every caller of the selectors
 * does next (visual_odometry.py:113-117 indexes coord / feat with the selection; extractors/kp2dtiny.py:41-42 gathers
 * keypoints and descriptors).  Same outputs, bit for bit, as the two calls. */
int kp2d_select_keypoints(const float* score, const float* coord, const float* desc, int B, int C, int n, int k, float thr,
                          int32_t* idx, float* val, int32_t* count, float* pts, float* dsel, void* stream);

/* replaces the per-frame front-end of inference() (src/evaluation/visual_odometry.py:77-87): kornia.image_to_tensor
 * / 255, kornia bilinear resize (align_corners=False), .sub(0.5).mul(2).  frames: uint8 [B,Hs,Ws,3] on the device;
 * x: float32 [B,3,H,W]. */
int kp2d_preprocess(const uint8_t* frames, int B, int Hs, int Ws, float* x, int H, int W, void* stream);

/* replaces: BfFeatureMatcher.match = cv2.BFMatcher(NORM_L2).knnMatch(k=2) + goodMatchesOneToOne
 * (src/visual_odometry/feature_matcher.py:89-98, :179-209), batched over B frame pairs, on device.
 *   d0 [B,max0,C] query descriptors, n0 [B] valid rows; d1 [B,max1,C] train descriptors, n1 [B]; C in {32,64,128}
 *   nn_idx / nn_dist / nn_dist2 [B,max0]  nearest train row, its L2 distance, second-nearest distance
 *     (nn_idx alone = cv2.BFMatcher(NORM_L2, crossCheck=False).match, src/evaluation/descriptor.py:132-134)
 *     For ANY finite input: with >= 256 train rows the search ranks on split-fp16 matrix-core keys and decides on an
 *     exact pass, which needs every row's norm in [0.5, 2^15] (unit-norm descriptors are); a workgroup that meets a
 *     row outside that range scans its rows with the exact arithmetic instead (slower, same answer).  On equal fp32
 *     distances the lower train index wins, except that among THREE train rows within ~1e-6 of each other (not
 *     identical) the matrix-core form may return either of the two nearest as nn_idx; nn_dist / nn_dist2 are exact.
 *   match_q [B,max1]  the query kept for each train row after ratio test + one-to-one filtering (-1: none)
 *   match_d [B,max1]  its distance
 *   scratch: B*max1*8 bytes of device memory */
int kp2d_match_descriptors(const float* d0, const int32_t* n0, const float* d1, const int32_t* n1, int B, int max0,
                           int max1, int C, float ratio, int32_t* nn_idx, float* nn_dist, float* nn_dist2,
                           int32_t* match_q, float* match_d, void* scratch, void* stream);
/* The same with the matcher variants of the reference's callers:
 *   cls0 [B,max0] / cls1 [B,max1] (both or neither): per-row class ids — a query only sees train rows of its own class.
 *     Replaces VisualOdometry.match_semantic (src/visual_odometry/visual_odometry.py:347-380: one BF match per class
 *     id, 28 of them per frame) with ONE launch.  A query whose class has fewer than two train rows gets no match (the
 *     reference's knnMatch(k=2) has no second neighbour there and match_semantic skips the class).  NOTE: as shipped,
 *     the reference's match_semantic unpacks two values from a matcher that returns three, so every class lands in its
 *     bare `except` and it returns no matches at all; this implements what the loop is written to do.
 *   flags & KP2D_MATCH_MUTUAL: match_q[t] = q iff t is q's nearest train row AND q is t's nearest query; no ratio test
 *     (cv2.BFMatcher(NORM_L2, crossCheck=True).match, src/evaluation/descriptor.py:221-222).
 *   scratch: kp2d_match_scratch_bytes(B, max0, max1) bytes, 8-byte aligned (B*max1*16 is the least accepted; the rest
 *     lets a search with few pairs spread one query's train rows over several workgroups). */
#define KP2D_MATCH_MUTUAL 1u
size_t kp2d_match_scratch_bytes(int B, int max0, int max1);
int kp2d_match_descriptors_ex(const float* d0, const int32_t* n0, const float* d1, const int32_t* n1, int B, int max0,
                              int max1, int C, float ratio, const int32_t* cls0, const int32_t* cls1, uint32_t flags,
                              int32_t* nn_idx, float* nn_dist, float* nn_dist2, int32_t* match_q, float* match_d,
                              void* scratch, size_t scratch_bytes, void* stream);
/* The matched rows of every pair as compact lists in train order (what the VO loop takes to the host instead of every
 * keypoint and descriptor: visual_odometry.py:270-284 kps0 = prev_keypoints[idxs0], kps1 = kps_cur[idxs1]):
 *   pairs [B,max1,4] (x0, y0, x1, y1) from pts0 [B,max0,2] / pts1 [B,max1,2]; idx [B,max1,2] (query row, train row);
 *   dist [B,max1]; count [B].  pairs / idx / dist may each be NULL. */
int kp2d_match_pairs(const int32_t* match_q, const float* match_d, const float* pts0, const float* pts1, int B, int max0,
                     int max1, float* pairs, int32_t* idx, float* dist, int32_t* count, void* stream);

/* The VO loop's top_k_matches cap on the device, fused with the compaction above (replaces
 * src/visual_odometry/visual_odometry.py:272-283 — BF branch: np.argpartition(score, k)[:k], the k SMALLEST distances —
 * and :26-32 + :260-266 — LightGlue branch: get_matches_scores(...) then scores.topk(k), the k LARGEST matching scores):
 *   mode KP2D_TOPK_BF: match_q [B,max1] + val = match_d [B,max1] (kp2d_match_descriptors); pair = (pts0[match_q[t]], pts1[t])
 *   mode KP2D_TOPK_LG: matches0 [B,max0] int64 + val = matching_scores0 [B,max0] (kp2d_lg_forward);
 *                      pair = (pts0[q], pts1[matches0[q]])
 *   k <= 0: every match.  At most kcap = min(k, n) pairs per frame pair (n = max1 / max0), BEST FIRST, equal values by lower
 *   source row (the reference's order within its k survivors is unspecified; callers use the set).
 *   pairs [B,kcap,4] (x0, y0, x1, y1), idx [B,kcap,2] (row in set 0, row in set 1; -1 past count), out_val [B,kcap]
 *   (distance / score), count [B]; pairs / idx / out_val may each be NULL.
 *   scratch: kp2d_match_topk_scratch_bytes(B, max0, max1) bytes of device memory. */
#define KP2D_TOPK_BF 0
#define KP2D_TOPK_LG 1
size_t kp2d_match_topk_scratch_bytes(int B, int max0, int max1);
int kp2d_match_topk_pairs(int mode, const int32_t* match_q, const int64_t* matches0, const float* val, const float* pts0,
                          const float* pts1, int B, int max0, int max1, int k, float* pairs, int32_t* idx, float* out_val,
                          int32_t* count, void* scratch, size_t scratch_bytes, void* stream);

/* ---- measurement ------------------------------------------------------------------------------ */
/* when on, every kernel launch of kp2d_forward is bracketed by HIP events on the caller's stream */
int kp2d_set_profiling(kp2d_model* m, int on);
/* number of launches recorded by the last kp2d_forward; blocks until those events have completed */
int kp2d_profile_count(kp2d_model* m);
/* one record: layer name, kernel family, elapsed ms, algorithmic FLOPs and HBM bytes of that launch */
int kp2d_profile_get(kp2d_model* m, int index, const char** layer, const char** kernel, float* ms, double* flops,
                     double* bytes);
/* Arithmetic of the convolution kernels (both accumulate in fp32 and meet the 1e-3 / index-identity bar):
 *   KP2D_PREC_FP32   exact fp32 on v_mfma_f32_32x32x2_f32 (bit-for-bit an fp32 fma chain)
 *   KP2D_PREC_F16X3  split fp16: x*w = xh*wh + xh*wl + xl*wh on v_mfma_f32_16x16x32_f16 (3x3 convolutions; the 1x1
 *                    convolutions and the attention kernel use v_mfma_f32_32x32x16_f16), fp32 accumulate, fp32-grade error
 *                    (default; see DESIGN.md "Numerics").  Both weight packs are resident; switching is free. */
#define KP2D_PREC_FP32 0
#define KP2D_PREC_F16X3 1
int kp2d_set_precision(kp2d_model* m, int mode);
int kp2d_get_precision(const kp2d_model* m);
/* Parity aid: the next kp2d_forward calls also copy ONE intermediate activation, as planar [B,C,H,W] fp32, to dst
 * (device memory, capacity in floats; a forward that needs more fails with KP2D_ERR_ARG).  layer = a CBR's state-dict
 * prefix ("backbone.conv1a", "backbone.conv3b", "vlad_head.convlad3", ...; a layer whose MaxPool2d is folded into its
 * store yields the pooled tensor) or "<attention module>.att" / ".mff" (modules/segformer.py:217-220).  This is how
 * the tests compare the kernels with the reference's recorded intermediates (tests/golden *_taps fixtures) layer by
 * layer.  layer = NULL or dst = NULL switches it off. */
int kp2d_set_tap(kp2d_model* m, const char* layer, float* dst, size_t capacity_floats);
/* The next kp2d_forward / kp2d_forward_frames calls (not ONLY_ENCODER) also write the dense class map — the argmax over
 * the class planes of `seg`, what post_processing computes first (kp2dtiny.py:609 / :975) — to ids [B,1,H2,W2] int64
 * (device memory, capacity in elements), from the epilogue of the layer that writes `seg` while its tile is still in LDS.
 * kp2d_post with seg = NULL and seg_ids = that buffer then leaves the ids as they are instead of reading `seg` again.
 * ids = NULL switches it off.  Only valid while `seg` is unchanged between the two calls (the Python host checks the
 * tensor's identity and version counter). */
int kp2d_set_seg_ids(kp2d_model* m, int64_t* ids, size_t capacity);
/* frames per internal sub-batch (0 = automatic).  Intermediates of one sub-batch stay in the 256 MB Infinity Cache. */
int kp2d_set_chunk_frames(kp2d_model* m, int frames);
/* Tuning knobs of the engine (never needed for correct results; used by the A/B scripts and the parity tests to force a
 * kernel form).  Keys:
 *   "wsm_min_items"  least number of (16 x 32 pixel tile, 64-channel group) work items of a launch for the
 *                    warp-specialised persistent form of the multi-chunk 3x3 layers (conv3x3_wsm.hip);
 *                    0 = automatic (KP2D_WSM if set, else one item per workgroup of the launch), -1 = never.
 *   "ws_min_tiles"   least 16 x 32 pixel tiles of a launch for the warp-specialised form of backbone.conv1b
 *                    (conv3x3_f16x3_ws_kernel); 0 = default (1024).
 *   "wsm_grid"       most workgroups of that form per launch (0 = KP2D_WSM_GRID if set, else CUs / stream lanes).
 *   "lanes"          stream lanes one forward splits its batch over (sub-batches run side by side on internal streams;
 *                    a workspace sized before the change stays valid only for lane counts <= the one it was sized
 *                    for): 0 = default (KP2D_LANES if set, else 2).  A caller that keeps two batches in flight on two
 *                    streams of its own (pipeline.BatchStream: each with its own workspace) sets 1 — the two
 *                    forwards then fill each other's launch tails, which two lanes of ONE forward (same layer at the
 *                    same time) cannot: 22.9k -> 23.4k frames/s at 64 x 240 x 320.
 *   "wsm_transposed" that form's tiles walk the map transposed (tile rows = map columns, the taps of the weight pack
 *                    transposed to match): 0 = never (default), 1 = always, 2 = where the matrix-time model says it is
 *                    cheaper (30 x 40 maps: 3 x 1 tiles instead of 2 x 2).  It sums the nine taps in another order, so
 *                    results differ from every other tile form in the last bits — which is why it is opt-in: with it
 *                    off, outputs are bit-identical whatever the batch size, lane count or tile form
 *                    ("conv3x3_f16x3<wsm>t" in the profile).
 *   "s16_all"        1 (default): big grids keep every tensor the warp-specialised 3x3 layers read as the fp16 halves of the
 *                    split (LDS-DMA staging, conv3x3_wsm.hip; bit-identical); 0: only inside the backbone's 32-channel stage.
 *   "side_overlap"   1 (default): a plain single-frame forward runs NetVLAD on a model-owned side stream beside the
 *                    segmentation head (never under stream capture; the stream is created on first use); 0: in line — and
 *                    the stream is destroyed (a process that keeps several streams busy wants the hardware queue back).
 *   "stem_fusion", "s16_min_items", "multi_launch", "mff_fused": README.md's table of knobs.
 * Unknown keys return KP2D_ERR_ARG.  kp2d_profile_get reports the tile form each conv launch took behind its kernel
 * family ("conv3x3_f16x3<wsm>", "conv3x3_f16x3<2,1,16>", ...). */
int kp2d_set_option(kp2d_model* m, const char* key, long value);

#ifdef __cplusplus
}
#endif
#endif /* KP2D_H_ */
