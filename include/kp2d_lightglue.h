/* C ABI of the MI355X-native LightGlue matcher (BASELINE config 5).
 *
 * Replaces, for inference, the reference's `LightGlue(conf, weights_path)` torch module
 * (lightglue/lightglue.py:418-614; constructed at src/visual_odometry/visual_odometry.py:149-155 and by
 * gluefactory's `matchers.lightglue`, gluefactory/configs/kp2dtiny_S+lightglue_homography.yaml:25-31): eval mode,
 * flash = False, depth_confidence = width_confidence = -1 (the reference configs' values; early stopping and point
 * pruning are not built and the host layer refuses them).
 *
 * Conventions are those of include/kp2d.h: plain pointers and sizes, device pointers for tensors, caller-provided
 * workspace, work enqueued on the caller's stream, int status (0 = OK, kp2d_status codes, kp2d_last_error()).
 */
#ifndef KP2D_LIGHTGLUE_H
#define KP2D_LIGHTGLUE_H

#include "kp2d.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kp2d_lg kp2d_lg; /* opaque */

/* LightGlue.default_conf entries that change the arithmetic (lightglue.py:419-438; lightglue_configs.py:1-22) */
typedef struct kp2d_lg_config {
  int32_t struct_size;     /* sizeof(kp2d_lg_config)                                               */
  int32_t input_dim;       /* descriptor width of the extractor; != descriptor_dim adds input_proj */
  int32_t descriptor_dim;  /* D: 32 (configs S, A) or 64 (F); built for D % 32 == 0, D <= 64       */
  int32_t n_layers;        /* 4 in every reference config                                          */
  int32_t num_heads;       /* 4                                                                    */
  int32_t device;          /* HIP device ordinal                                                   */
} kp2d_lg_config;

/* replaces: LightGlue(conf) construction */
int kp2d_lg_create(const kp2d_lg_config* cfg, kp2d_lg** out);
void kp2d_lg_destroy(kp2d_lg* m);

/* weights: the reference module's state_dict is the wire format (keys in registration order, lightglue.py:444-470) */
int kp2d_lg_num_weights(const kp2d_lg* m);
int kp2d_lg_weight_info(const kp2d_lg* m, int index, const char** key, int64_t shape[4], int* ndim);
/* replaces: self.load_state_dict(torch.load(weights_path)) (lightglue.py:472-473); host float32, C-contiguous */
int kp2d_lg_set_weight(kp2d_lg* m, const char* key, const float* host, const int64_t* shape, int ndim);
int kp2d_lg_finalize_weights(kp2d_lg* m);

/* scratch for a batch of B image pairs with M / N keypoints each; 256-byte aligned device memory */
size_t kp2d_lg_workspace_bytes(const kp2d_lg* m, int B, int M, int N);

/* replaces: LightGlue.forward(data) (lightglue.py:484-614).
 *   kpts0 [B,M,2] / kpts1 [B,N,2]   keypoints in pixels (x, y)                   data["keypoints0/1"]
 *   desc0 [B,M,input_dim] / desc1 [B,N,input_dim]                                 data["descriptors0/1"]
 *   size0 / size1 [B,2] (w, h) or NULL: 1 + max - min of the keypoints            data["view0/1"]["image_size"]
 *   filter_threshold                                                              conf.filter_threshold
 * outputs (any of matches / scores / ref_desc may be NULL):
 *   log_assignment [B,M+1,N+1]   pred["log_assignment"]
 *   matches0 [B,M] / matches1 [B,N] int64 (-1: unmatched), mscores0 [B,M] / mscores1 [B,N]
 *   ref_desc0 [B,M,D] / ref_desc1 [B,N,D]  the last layer's descriptors (pred["ref_descriptors0/1"][:, 0]) */
int kp2d_lg_forward(kp2d_lg* m, const float* kpts0, const float* kpts1, const float* desc0, const float* desc1,
                    const float* size0, const float* size1, int B, int M, int N, float filter_threshold,
                    float* log_assignment, int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1,
                    float* ref_desc0, float* ref_desc1, void* workspace, size_t workspace_bytes, void* stream);

/* The same on PADDED keypoint sets: n0 / n1 [B] int32 device arrays say how many rows of kpts0 / desc0 (kpts1 / desc1) exist,
 * the remaining rows of the M / N are padding.  This is what lets the matcher sit inside a replayed HIP graph behind a
 * threshold + top-k selection whose count changes from frame to frame (the reference hands LightGlue exactly the selected
 * rows, src/visual_odometry/visual_odometry.py:198-258: a different tensor shape every frame).  Padding rows are no keys in
 * any attention, carry no assignment mass, and come out as matches = -1 / scores = 0; the valid rows' results are those of
 * kp2d_lg_forward on the n0 / n1 rows alone (up to fp32 summation order).  log_assignment entries of padding rows / columns
 * are -inf (inner block) or unspecified (border).  size0 / size1 are required (the default would look at every row). */
int kp2d_lg_forward_counts(kp2d_lg* m, const float* kpts0, const float* kpts1, const float* desc0, const float* desc1,
                           const float* size0, const float* size1, const int32_t* n0, const int32_t* n1, int B, int M, int N,
                           float filter_threshold, float* log_assignment, int64_t* matches0, int64_t* matches1,
                           float* mscores0, float* mscores1, float* ref_desc0, float* ref_desc1, void* workspace,
                           size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KP2D_LIGHTGLUE_H */
