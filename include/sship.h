/* sship.h - C ABI of libsuperslam_hip.so: the MI355X (gfx950) deep-feature front-end for SuperSLAM.
 *
 * This header is the drop-in boundary.  Every entry point replaces a piece of the reference's
 * TensorRT/CUDA inference layer (paths relative to /root/reference); the C++ adapter a maintainer adds
 * on the reference side (classes SuperPoint / LightGlue implementing IFeatureExtractor / IFeatureMatcher
 * over these calls) is shown in INTEGRATION.md and shipped as include/superslam_hip/.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch/OpenCV types cross the ABI;
 *   - every function returns an int status (0 = SSHIP_OK); nothing throws or longjmps across the ABI;
 *     sship_last_error() returns a thread-local message for the last non-zero status;
 *   - "_dev" pointers are HIP device pointers, all others host pointers;
 *   - `stream` is a hipStream_t passed as void*; NULL is the legacy default stream itself (= torch's default
 *     stream), so consecutive asynchronous calls made with NULL - extractor then matcher - are ordered with each
 *     other and with the caller's default-stream work.  The synchronous host-image / host-keypoint entry points
 *     use the handle's own (blocking) stream and return after synchronising it.  Multi-threaded callers: the legacy
 *     default stream synchronises with EVERY blocking stream of the process, i.e. a NULL-stream batch call on one
 *     thread serialises with another thread's handles (the loop-closure matcher / EigenPlaces).  The handle streams
 *     do not synchronise with each other, so threads that use the synchronous entry points (the reference's usage)
 *     run concurrently; a thread that drives the asynchronous batch API next to them should pass its own
 *     hipStreamNonBlocking stream (or hipStreamPerThread) instead of NULL;
 *   - external dtypes follow the reference engines (scripts/rebuild_engines.sh:85-92,108-115):
 *     image u8 (normalised to [0,1] on device), scores f32, descriptors f16, kpts f32,
 *     matches0 i32, mscores0 f32.  Internal accumulation is f32.
 *   - handles are single-threaded; sship_lg_weights is immutable and shareable across handles/threads
 *     (the LightGlueEngine / shared_engine() analogue, include/LightGlue.h:28-31,44).
 *
 * Environment
 *   The library reads exactly two environment variables, each once per process:
 *     SUPERSLAM_HIP_DEVICE=<n>   device ordinal sship_init(-1) / the first call of a thread binds (default 0; the multi-process
 *                                scripts set it from LOCAL_RANK);
 *     SSHIP_RCCL_LIBRARY=<path>  the RCCL library sship_comm_* binds at run time instead of the process's own / librccl.so.1.
 *   Nothing else: there is ONE kernel per layer and no run-time kernel selection, so a stray variable cannot move a SuperSLAM
 *   process onto a slower or looser path.  Profiling is an API (sship_set_profiling), not a variable.  The A/B switches of the
 *   development history (SUPERSLAM_HIP_CONV*, _ATTN*, _FFN*, _LG_*, SSHIP_*_TRACE) and the kernels they select exist only in the
 *   developer build superslam_amd/lib/variants/dev.so (`python -m superslam_amd.build --dev`, -DSSHIP_DEV_SWITCHES=1), which the
 *   A/B tests and scripts load explicitly.
 */
#ifndef SSHIP_H_
#define SSHIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSHIP_VERSION 100
#define SSHIP_DESC_DIM 256 /* include/SuperPoint.h:76 descriptor_dim */

typedef enum {
  SSHIP_OK = 0,
  SSHIP_ERR_INVALID = 1,        /* bad argument */
  SSHIP_ERR_HIP = 2,            /* HIP runtime failure */
  SSHIP_ERR_IO = 3,             /* weights file missing / malformed */
  SSHIP_ERR_NOMEM = 4,
  SSHIP_ERR_POOL_EXHAUSTED = 5, /* src/SuperPoint.cc:724-727 */
  SSHIP_ERR_NO_DEVICE = 6       /* no gfx950 device: the library never falls back to the CPU */
} sship_status;

/* ------------------------------------------------------------------------------------------------
 * Runtime
 * ---------------------------------------------------------------------------------------------- */
/* Select the HIP device (replaces the implicit cudaSetDevice(0) of SuperPoint::initialize,
 * src/SuperPoint.cc:38-67).  Fails with SSHIP_ERR_NO_DEVICE when no GPU is visible. */
int sship_init(int device);
int sship_version(void);
const char* sship_last_error(void);
/* level: 0 trace .. 4 error; the adapter forwards to SLOG_* (include/Logging.h:21-26). */
void sship_set_log_callback(void (*cb)(int level, const char* msg));
int sship_device_synchronize(void);

/* ------------------------------------------------------------------------------------------------
 * Descriptor pool - include/DescriptorPool.h:13-91, src/DescriptorPool.cc:10-38
 * N device slots of max_keypoints*dim fp16; LIFO free-list (FreeList, DescriptorPool.h:25-44).
 * ---------------------------------------------------------------------------------------------- */
typedef struct sship_pool sship_pool;
int sship_pool_create(int num_slots, int max_keypoints, int dim, sship_pool** out);
/* Frees the device slots and drops the creator's reference.  The bookkeeping (free-list, mutex) is reference counted
 * like the reference's shared_ptr<FreeList> (DescriptorPool.h:71-75): every DeviceDescriptors handle holds one
 * reference (sship_pool_retain when the handle is made, sship_pool_release + sship_pool_release_ref in its deleter),
 * so a handle may outlive the pool's owner; its data pointer then dangles exactly as in the reference
 * (DescriptorPool.cc:27-32) but releasing it is safe. */
void sship_pool_destroy(sship_pool* pool);
void sship_pool_retain(sship_pool* pool);
void sship_pool_release_ref(sship_pool* pool);
int sship_pool_acquire(sship_pool* pool);            /* FreeList::acquire: slot index or -1 when exhausted */
void sship_pool_release(sship_pool* pool, int slot); /* FreeList::release */
int sship_pool_in_use(const sship_pool* pool);       /* FreeList::in_use */
void* sship_pool_slot_ptr(const sship_pool* pool, int slot); /* DescriptorPool::slot_ptr (NULL if out of range) */

/* ------------------------------------------------------------------------------------------------
 * Descriptor gather - 1:1 with launch_gather_descriptors, include/DescriptorGather.h:12-20,
 * src/DescriptorGather.cu:14-82.  grid_fp16_dev is [channels, grid_h, grid_w] (CHW) fp16; cell_h/cell_w
 * are device int arrays; out is [num_keypoints, channels] fp16, each row L2-normalised
 * (fp32 sum of squares, rsqrt(sum + 1e-12), round-to-nearest fp16).  num_keypoints <= 0 is a no-op.
 * The _hwc variant reads a channels-last grid [grid_h, grid_w, channels] (one contiguous 512-B row per
 * keypoint - the layout the HIP SuperPoint produces internally).
 * ---------------------------------------------------------------------------------------------- */
int sship_gather_normalize(const void* grid_fp16_dev, int channels, int grid_h, int grid_w,
                           const int* cell_h_dev, const int* cell_w_dev, int num_keypoints,
                           void* out_fp16_dev, void* stream);
int sship_gather_normalize_hwc(const void* grid_fp16_dev, int channels, int grid_h, int grid_w,
                               const int* cell_h_dev, const int* cell_w_dev, int num_keypoints,
                               void* out_fp16_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Detector post-processing stages (exposed so each can be parity-tested bit-exactly)
 * ---------------------------------------------------------------------------------------------- */
/* utils/convert_superpoint_to_onnx.py:82-87: pooled = max_pool2d(s, 2r+1, 1, r); s = (s==pooled)?s:0. */
int sship_nms(const float* scores_dev, int batch, int h, int w, int radius, float* out_dev, void* stream);
/* src/SuperPoint.cc:696-719 on one device score map: strict `score > thr` (thr is a double) inside the
 * border, descending (score, h, w) order, first max_kp; kp (x = w*input_w/score_w, y = h*input_h/score_h,
 * score) triples, cells = min(h/8, desc_h-1), min(w/8, desc_w-1).  All outputs are device arrays
 * ([3*max_kp] f32, [max_kp] i32, [max_kp] i32, [1] i32); n_candidates_dev may be NULL. */
int sship_select_topk(const float* scores_dev, int score_h, int score_w, int input_h, int input_w,
                      double thr, int border, int max_kp, int desc_h, int desc_w, float* kp_xys_dev,
                      int* cell_h_dev, int* cell_w_dev, int* n_dev, int* n_candidates_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SuperPoint extractor - include/SuperPoint.h:36-54, src/SuperPoint.cc
 * ---------------------------------------------------------------------------------------------- */
typedef struct sship_sp sship_sp;
typedef struct {
  const char* weights_path;  /* safetensors with the reference's state-dict keys (conv1a.weight ...).  Takes
                                the place of SuperPoint's engine_file (src/SuperSLAM.cc:72-78). */
  int max_keypoints;         /* superpoint.max_keypoints (600 in the KITTI YAML) */
  double keypoint_threshold; /* superpoint.keypoint_threshold (0.005) */
  int remove_borders;        /* superpoint.remove_borders (4) */
  int nms_radius;            /* exporter --nms-radius (4) */
  int pool_slots;            /* include/SuperPoint.h:77-78 descriptor_pool_slots (8); 0 -> 8 */
  int max_batch;             /* images per call the workspaces are sized for (2 = stereo); 0 -> 2 */
} sship_sp_config;

/* One image's extraction result.  kp_xys is caller-owned [3*max_keypoints] (x, y, score per keypoint -
 * cv::KeyPoint(x, y, 1, -1, score), src/SuperPoint.cc:715).  desc_dev points into pool slot `slot`
 * ([n, 256] fp16 row-major, L2-normalised); the caller owns the slot and returns it with
 * sship_pool_release(sship_sp_pool(sp), slot).  n == 0 -> slot = -1, desc_dev = NULL (success). */
typedef struct {
  float* kp_xys;
  int n;
  void* desc_dev;
  int slot;
} sship_features;

int sship_sp_create(const sship_sp_config* cfg, sship_sp** out); /* ctor + initialize() */
void sship_sp_destroy(sship_sp* sp);
sship_pool* sship_sp_pool(sship_sp* sp);
int sship_sp_max_keypoints(const sship_sp* sp);

/* SuperPoint::extract (src/SuperPoint.cc:895-899 -> infer_device :597-676): host u8 image, 1 or 3 (BGR)
 * channels, row stride in bytes.  Synchronous. */
int sship_sp_extract(sship_sp* sp, const uint8_t* img, int h, int w, int stride, int channels,
                     sship_features* out);
/* SuperPoint::extract_stereo (src/SuperPoint.cc:902-908 -> infer_device_stereo :754-892): one batch-2
 * pass; the pair must share resolution (:762-765). */
int sship_sp_extract_stereo(sship_sp* sp, const uint8_t* left, const uint8_t* right, int h, int w,
                            int stride, int channels, sship_features* out_left, sship_features* out_right);
/* Decode-ahead upload ring for dataset runners.  No reference counterpart: the reference stages every frame
 * in-line (clone + convertTo + memcpy into one pinned buffer + H2D, src/SuperPoint.cc:768-795); here `depth` stereo frames live in
 * pinned host memory, so a decoder thread writes pixels straight into a slot (sship_sp_ring_host), starts its H2D on the ring's
 * own copy stream (sship_sp_ring_upload - the one call that may run on another thread than the handle's owner), and the
 * tracking thread extracts from the uploaded slot (sship_sp_extract_stereo_ring = sship_sp_extract_stereo without the host
 * copy and with the upload already overlapped with the previous frame's compute).  Images are [h][w*channels] u8, 1 or 3 (BGR)
 * channels; the caller keeps slot reuse behind the extract call that consumes it. */
int sship_sp_ring_create(sship_sp* sp, int depth, int h, int w, int channels);
uint8_t* sship_sp_ring_host(sship_sp* sp, int slot, int image /* 0 left, 1 right */);
int sship_sp_ring_upload(sship_sp* sp, int slot);
int sship_sp_extract_stereo_ring(sship_sp* sp, int slot, sship_features* out_left, sship_features* out_right);
/* Cross-frame pipelining for the per-frame path (no reference counterpart: the reference runs extract -> match -> estimator
 * strictly in sequence, src/StereoFrontEnd.cc:10-48).  sship_sp_ring_submit ENQUEUES the whole extraction of an uploaded slot
 * (network, selection, descriptor head into two freshly acquired pool slots, D2H of keypoints / counts into the slot's own
 * pinned buffers) on the extractor's stream and returns at once; the later sship_sp_extract_stereo_ring(slot) only waits for
 * that work's completion event and hands the results out.  Called right after frame t's extraction has returned - before frame
 * t's LightGlue match - it lets frame t+1's SuperPoint kernels share the GPU with frame t's matcher (the matcher's launches
 * cover a fraction of the CUs at one pair).  Same thread as every other call on this handle; at most one submission per slot.
 * Constraints while a submission is pending (submitted, not yet collected):
 *   - sship_sp_ring_upload(slot) on that slot returns SSHIP_ERR_INVALID: the queued network still reads the slot's device
 *     frame (collect first, then refill);
 *   - the synchronous extractor calls (sship_sp_extract*, sship_sp_extract_stereo_ring of another slot) run on the handle's
 *     own stream and are ordered behind the submission;
 *   - sship_sp_extract_batch_device / sship_frontend_batch_device on a caller-supplied NON-BLOCKING stream share the handle's
 *     activations with the submission and are NOT ordered with it: do not overlap them with a pending submission;
 *   - sship_sp_destroy with a submission pending waits for it and returns its pool slots.
 * Stage timings: sship_sp_ring_submit restarts the calling thread's stage marks, so under pipelining
 * sship_get_stage_timings mixes frame t+1's extraction stages with frame t's match stages - profile un-pipelined. */
int sship_sp_ring_submit(sship_sp* sp, int slot);
/* SuperPoint::infer host path (src/SuperPoint.cc:322-348,427-528): keypoints + CV_32F [n,256] descriptors
 * on the host.  kp_xys [3*max_kp], desc_f32 [max_kp*256]. */
int sship_sp_infer_host(sship_sp* sp, const uint8_t* img, int h, int w, int stride, int channels,
                        float* kp_xys, float* desc_f32, int* n);

/* Throughput path: `batch` grayscale u8 images already resident in HBM ([batch, h, w] contiguous); results
 * stay on the device: desc [batch, max_kp, 256] f16, kp [batch, max_kp, 3] f32, n [batch] i32.
 * Asynchronous on `stream`; no host synchronisation anywhere inside. */
int sship_sp_extract_batch_device(sship_sp* sp, const uint8_t* imgs_dev, int batch, int h, int w,
                                  void* desc_out_dev, float* kp_out_dev, int* n_out_dev, void* stream);

/* Dense outputs of the network, in the reference engine's layouts (scripts/rebuild_engines.sh:88-97):
 * scores f32 [batch, 8*(h/8), 8*(w/8)] after NMS, descriptors f16 [batch, 256, h/8, w/8] (CHW,
 * L2-normalised).  logits_dev (optional) receives the raw detector logits f32 [batch, 65, h/8, w/8].
 * Any output pointer may be NULL. */
int sship_sp_dense(sship_sp* sp, const uint8_t* imgs_dev, int batch, int h, int w, float* scores_dev,
                   void* desc_grid_dev, float* logits_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LightGlue matcher - include/LightGlue.h:28-63, src/LightGlue.cc
 * ---------------------------------------------------------------------------------------------- */
typedef struct sship_lg_weights sship_lg_weights; /* LightGlueEngine analogue (refcounted, immutable) */
typedef struct sship_lg sship_lg;

int sship_lg_weights_load(const char* safetensors_path, sship_lg_weights** out);
void sship_lg_weights_retain(sship_lg_weights* w);
void sship_lg_weights_release(sship_lg_weights* w);

/* LightGlue(engine, image_width, image_height) + initialize().  max_keypoints bounds n0/n1 (the reference
 * engine's profile max is 1024, scripts/rebuild_engines.sh:118); max_pairs sizes the batched workspaces. */
int sship_lg_create(sship_lg_weights* w, int image_width, int image_height, int max_keypoints,
                    int max_pairs, sship_lg** out);
void sship_lg_destroy(sship_lg* lg);

/* Keypoint normalisation, src/LightGlue.cc:241-251: (pt - (W/2,H/2)) / (max(W,H)/2).  kp_xy has `stride`
 * floats per keypoint (2 or 3); out is [n,2]. */
int sship_lg_normalize_keypoints(const sship_lg* lg, const float* kp_xy, int stride, int n, float* out);

/* Device match, src/LightGlue.cc:377-457: host keypoints (pixel coordinates, `kp_stride` floats apart),
 * descriptors resident in pool slots ([n,256] fp16).  Outputs on the host: matches0 [n0] (index into set 1
 * or -1), mscores0 [n0].  n0 == 0 or n1 == 0 -> SSHIP_ERR_INVALID (the reference returns an empty result). */
int sship_lg_match_device(sship_lg* lg, const float* kp0, int kp_stride0, int n0, const void* desc0_dev,
                          const float* kp1, int kp_stride1, int n1, const void* desc1_dev,
                          int32_t* matches0, float* mscores0);
/* Host-descriptor match, src/LightGlue.cc:285-324: CV_32F [n,256] descriptors, converted to fp16 and
 * uploaded internally (the loop-closure overload). */
int sship_lg_match_host(sship_lg* lg, const float* kp0, int kp_stride0, int n0, const float* desc0_f32,
                        const float* kp1, int kp_stride1, int n1, const float* desc1_f32,
                        int32_t* matches0, float* mscores0);
/* Throughput path: `pairs` independent problems, everything device-resident and asynchronous.
 * kp_dev [2*pairs, max_kp, 3] (pixel x, y, score), n_dev [2*pairs], desc_dev [2*pairs, max_kp, 256] f16,
 * image 2p is set 0 and image 2p+1 is set 1 of pair p.  Outputs matches0_dev / mscores0_dev
 * [pairs, max_kp]; rows >= n are -1 / 0. */
int sship_lg_match_batch_device(sship_lg* lg, const float* kp_dev, const int* n_dev, const void* desc_dev,
                                int pairs, int32_t* matches0_dev, float* mscores0_dev, void* stream);
/* Test-only introspection of the matcher (no reference counterpart; used by the parity suite to compare the internals
 * with the oracle layer by layer - the product never calls these).
 * sship_lg_debug_set_layers: the NEXT match call on this handle (one-shot) runs only the first n_layers (1..9) transformer
 *   layers and skips the assignment (matches0 = -1, mscores0 = 0); the call after it is a full match again.
 * sship_lg_debug_read: after a match call, copy state of this handle to the host as f32 (device-synchronising):
 *   SSHIP_LG_DEBUG_X    residual stream of sequence `index` (2p = set 0, 2p+1 = set 1 of pair p): out[rows][256]
 *   SSHIP_LG_DEBUG_SIM  assignment similarity md0 md1^T of pair `index`: out[rows][cols]
 *   SSHIP_LG_DEBUG_KPTS normalised keypoints of sequence `index` (src/LightGlue.cc:241-251 on the device): out[rows][2]
 *   SSHIP_LG_DEBUG_ROPE rotary table of sequence `index`: out[rows][64] = 32 (cos, sin) pairs */
enum { SSHIP_LG_DEBUG_X = 0, SSHIP_LG_DEBUG_SIM = 1, SSHIP_LG_DEBUG_KPTS = 2, SSHIP_LG_DEBUG_ROPE = 3 };
int sship_lg_debug_set_layers(sship_lg* lg, int n_layers);
int sship_lg_debug_read(sship_lg* lg, int what, int index, int rows, int cols, float* out);
/* Test-only: copy one encoder activation of the extractor's LAST call to the host as raw fp16 (channels-last [batch][h_l][w_l][c_l]),
 * device-synchronising.  layer: 1 conv1b (+pool), 2 conv2a, 3 conv2b (+pool), 4 conv3a, 5 conv3b (+pool), 6 conv4a, 7 conv4b (the ids of
 * sship_sp_bench_layer).  Used by the parity suite to compare single layers (e.g. the Winograd variant of conv2a / conv2b) with a CPU
 * convolution of the previous layer's activation; the product never calls it.  `bytes` must not exceed the activation's size.
 * Layer 2 (conv2a) is NOT materialised by the shipped library: conv2a -> conv2b -> pool run as one kernel whose intermediate map never leaves the CU
 * (csrc/conv_fuse2.hip); its buffer holds the last two-launch run's map (developer build, SUPERSLAM_HIP_CONV2=split / SUPERSLAM_HIP_CONV64=wino) or nothing. */
int sship_sp_debug_activation(sship_sp* sp, int layer, void* out_host, unsigned long long bytes);
/* Match post-processing, src/LightGlue.cc:326-363: ascending i, skip -1, distance = 1 - score.
 * Returns the number of matches (>= 0). */
int sship_filter_matches(const int32_t* matches0, const float* mscores0, int n0, int* query_idx,
                         int* train_idx, float* distance);
/* LightGlue::descriptors_to_host, src/LightGlue.cc:460-475: fp16 [count, dim] device -> f32 host. */
int sship_desc_to_host(const void* desc_dev, int count, int dim, float* out_f32);

/* ------------------------------------------------------------------------------------------------
 * EigenPlaces place recogniser - include/EigenPlaces.h:19-40, src/EigenPlaces.cc
 * ResNet-18 trunk + L2Norm / GeM / Linear(512, 512) / L2Norm (utils/convert_eigenplaces_to_onnx.py:54-60), used once per
 * keyframe by the loop-closure thread.  weights_path: safetensors of the hub model's state_dict (keys backbone.*,
 * aggregation.*, what utils/convert_eigenplaces_to_onnx.py:99 saves) - it takes the place of the .engine file.
 * sship_ep_infer is the device half of EigenPlaces::compute_global_descriptor (src/EigenPlaces.cc:147-174): the caller hands
 * the HOST-preprocessed fp32 [3, input_h, input_w] tensor (src/EigenPlaces.cc:123-145 runs on the host in the reference too;
 * include/superslam_hip/place_recognizer.hpp restates it) and receives the L2-normalised 512-d descriptor.  Synchronous.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sship_ep sship_ep;
int sship_ep_create(const char* weights_path, int input_w, int input_h, sship_ep** out);
void sship_ep_destroy(sship_ep* ep);
int sship_ep_descriptor_dim(const sship_ep* ep);
int sship_ep_infer(sship_ep* ep, const float* chw_host, float* desc_out);
/* EigenPlaces::compute_global_descriptor for the image itself (src/EigenPlaces.cc:147-174 including :123-145): u8 image, 1 channel or
 * 3 = BGR, row stride in bytes, any size.  The image is uploaded as u8 and preprocessed ON THE DEVICE (fixed-point 8-bit bilinear resize to
 * input_w x input_h, x 1/255, ImageNet mean / std: bit-identical to sship_ep_preprocess), then the network runs as in sship_ep_infer -
 * the descriptor equals sship_ep_infer(sship_ep_preprocess(img)) bit for bit.
 *   sship_ep_infer_u8         host image, host descriptor; synchronous (returns after the handle's stream has drained);
 *   sship_ep_infer_u8_device  device image, device descriptor [512] f32, asynchronous on `stream` (NULL = legacy default stream).  One
 *                             call in flight per handle (the activations live in the handle: calls of one handle must be ordered, on one
 *                             stream or by events).  The FIRST call with a new source size (h, w) allocates and uploads that size's resize
 *                             tables (hipMalloc + a blocking copy; not legal under stream capture - warm each size up first); the tables
 *                             are immutable afterwards, so later calls are purely asynchronous.  The descriptor's bits do not depend on
 *                             the device's CU count or partition mode (the split-K factors are functions of the layer shapes only). */
int sship_ep_infer_u8(sship_ep* ep, const uint8_t* img, int h, int w, int stride, int channels, float* desc_out);
int sship_ep_infer_u8_device(sship_ep* ep, const uint8_t* img_dev, int h, int w, int stride, int channels, float* desc_out_dev, void* stream);
/* Measurement hook: `iters` back-to-back sship_ep_infer_u8_device calls on the handle's stream over a resident image; average ms. */
int sship_ep_bench(sship_ep* ep, const uint8_t* img_dev, int h, int w, int stride, int channels, int iters, float* avg_ms);
/* EigenPlaces::preprocess (src/EigenPlaces.cc:123-145) on the host, no GPU: u8 image (1 channel or 3 = BGR, row stride in bytes) ->
 * fp32 [3, input_h, input_w]: GRAY2RGB / BGR2RGB, cv::resize INTER_LINEAR (OpenCV's 8-bit fixed-point path), x 1/255,
 * ImageNet mean / std.  Exported so that every binding shares one implementation. */
int sship_ep_preprocess(const uint8_t* img, int h, int w, int stride, int channels, int input_w, int input_h, float* chw_out);

/* ------------------------------------------------------------------------------------------------
 * Fused front-end step: what StereoFrontEnd::process asks of the two interfaces per frame
 * (src/StereoFrontEnd.cc:14,33): SuperPoint on L and R (one batch) + gather x2 + one LightGlue match,
 * for `pairs` stereo pairs at once, device-resident, no host synchronisation.  imgs_dev is
 * [2*pairs, h, w] u8 ordered L0, R0, L1, R1, ...  Output shapes as in the two batch calls above.
 * ---------------------------------------------------------------------------------------------- */
int sship_frontend_batch_device(sship_sp* sp, sship_lg* lg, const uint8_t* imgs_dev, int pairs, int h, int w,
                                void* desc_out_dev, float* kp_out_dev, int* n_out_dev,
                                int32_t* matches0_dev, float* mscores0_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU exchange (BASELINE configs 3 and 5).  No reference counterpart: the reference is single-GPU.
 * Frames / pairs / cameras are sharded over one process per GPU with replicated weights and NO data-path collective; the
 * one exchange step is a fixed-stride all-gather of every rank's padded results into the image of the shared
 * DescriptorPool (include/DescriptorPool.h:13-91: [count, 256] fp16 rows per frame) on every rank.  RCCL (xGMI) is bound
 * at run time; without it these calls return SSHIP_ERR_NO_DEVICE and the rest of the library is unaffected.
 *   rank 0:      sship_comm_unique_id(id)  ->  hand the 128 bytes to the other ranks out of band (file, MPI, torch store, ...)
 *   every rank:  sship_init(device); sship_comm_create(id, rank, world, &comm)       (collective: all ranks must call)
 *   per batch:   sship_gather_features_rccl(comm, desc, kp, n, units, max_kp, desc_all, kp_all, n_all, stream)
 * Buffers are device memory: local desc [units, max_kp, 256] f16, kp [units, max_kp, 3] f32, n [units] i32; the *_all buffers
 * hold world x units units in rank order (rank r's block at r * units); ranks with fewer real units pad with n = 0.
 * The three all-gathers go out as ONE grouped RCCL step on `stream` (asynchronous; NULL = the default stream).
 * ---------------------------------------------------------------------------------------------- */
typedef struct sship_comm sship_comm;
int sship_comm_unique_id(void* id_out_128);
int sship_comm_create(const void* id_128, int rank, int world, sship_comm** out);
void sship_comm_destroy(sship_comm* comm);
int sship_comm_rank(const sship_comm* comm);
int sship_comm_world(const sship_comm* comm);
int sship_gather_features_rccl(sship_comm* comm, const void* desc_local_dev, const float* kp_local_dev, const int* n_local_dev,
                               int units_per_rank, int max_keypoints, void* desc_all_dev, float* kp_all_dev, int* n_all_dev,
                               void* stream);

/* Per-stage device timings (ms) of the calling thread's last call sequence made with profiling enabled
 * (sship_set_profiling(1) inserts hipEvents; off by default; level 2 adds one event per SuperPoint layer launch - labels
 * "<scope>:<stage>/<layer>", e.g. "sp_gpu_infer:encoder/conv1a+conv1b+pool" - the IN-SITU launch durations bench.py's roofline
 * line reports; a profiled batch call also keeps a device copy of its input so that sship_sp_bench_layer re-launches on real pixels).  Labels are "<scope>:<stage>" where <scope> is the
 * reference's own SUPERSLAM_PROFILE label the stage belongs to - sp_gpu_infer (src/SuperPoint.cc:639),
 * sp_extract_stereo (:904), fe_lg_stereo_match (src/StereoFrontEnd.cc:32) - and <stage> this library's finer split
 * (encoder, heads, select, gather, posenc_qkv0, layers_x9, assign_filter); summing a scope's stages gives the
 * reference's figure.  At level 2 the SuperPoint stages are reported ONLY as their per-launch entries ("sp_gpu_infer:encoder/conv2a",
 * ...): no entry carries the bare "<scope>:<stage>" label, a stage's time is the sum over its "<scope>:<stage>/..." entries.  Timers are per thread.  Returns the number of stages. */
void sship_set_profiling(int level);
int sship_get_stage_timings(const char** labels, float* ms, int max_stages);

/* Measurement hook for bench.py's roofline line: re-launch ONE layer of the network `iters` times on the
 * handle's stream, bracketed by hipEvents on that same stream, over the activations left by the previous
 * sship_sp_* call of shape (batch, h, w); *avg_ms = mean launch duration.  layer: 0 conv1a, 1 conv1b (+pool),
 * 2 conv2a, 3 conv2b (+pool), 4 conv3a, 5 conv3b (+pool), 6 conv4a, 7 conv4b, 8 convPa, 9 convPb, 10 convDa,
 * 11 convDb.  *macs receives the layer's multiply-accumulate count for that shape. */
int sship_sp_bench_layer(sship_sp* sp, int layer, int batch, int h, int w, int iters, float* avg_ms, double* macs);
/* (layer ids 12-14 are the memory-bound stages of the same handle: 12 = softmax + depth-to-space + NMS + threshold +
 * candidate compaction, 13 = top-k, 14 = descriptor head at the selected keypoints; *macs = 0 for them.
 * 15 = conv2a + conv2b + pool as the ONE launch throughput batches run instead of layers 2 and 3 (csrc/conv_fuse2.hip; *macs = both layers').)
 *
 * Same for one stage of the matcher, over the state the last match call left on this handle, timed with hipEvents on the
 * handle's stream: 0 first Wqkv projection, 1 self attention, 2 cross attention (both directions), 3 SelfBlock FFN + the
 * fused [to_qk|to_v] projection, 4 CrossBlock FFN + the fused next Wqkv, 5 last CrossBlock FFN + final_proj + matchability,
 * 6 assignment pass 1 (similarity tiles + row / column log-sum-exp), 7 assignment pass 2 (similarity tiles + row / column
 * arg-max of the double log-softmax scores). */
int sship_lg_bench_stage(sship_lg* lg, int stage, int iters, float* avg_ms);

/* Measurement aid for the roofline line: the v_mfma_f32_32x32x16_f16 rate (TFLOP/s) the device sustains from registers
 * for ~5 ms on every CU, with zero (random_operands = 0) or random fp16 operands.  The chip clocks to its power budget,
 * so the random-operand figure (about 1.6 PFLOP/s on MI355X) - not the 2.5 PFLOP/s datasheet peak - is what a real
 * convolution can approach.  No reference counterpart (the reference has no measurement API). */
int sship_mfma_probe(int random_operands, float* tflops);

#ifdef __cplusplus
}
#endif
#endif /* SSHIP_H_ */
