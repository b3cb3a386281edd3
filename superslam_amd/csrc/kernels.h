// Internal kernel launch interface of libsuperslam_hip (not part of the C ABI).
#pragma once
#include "common.h"

namespace sship {

constexpr int kMaxKp = 4096;   // upper bound on max_keypoints (top-k sorts in LDS)
constexpr int kLogitStride = 68;   // detector logits: 65 channels in a 68-wide fp32 row (272 B, 16-B aligned; was 80: 15 % fewer bytes through convPb -> k_nms_tile)

// ---- sp_kernels.hip ----
void launch_conv1a(const uint8_t* img, const float* w, const float* bias, _Float16* out, int B, int H, int W,
                   hipStream_t s);
struct NmsArgs {
  const float* logits;   // [B, Hc, Wc, ls] (loader 0)
  int ls;
  const float* scores_in;  // [B, H, W] (loader 1)
  int B, H, W;             // score-map shape
  int radius;
  float thr_f;             // smallest float f with (double)f > thr  ->  keep iff s >= thr_f
  int border;
  unsigned long long* cand;  // [B, cap] or null
  int* cand_count;           // [B]
  int cap;
  float* scores_out;         // [B, H, W] post-NMS or null
  float* scores_raw_out;     // [B, H, W] pre-NMS (softmax + depth-to-space) or null
};
float threshold_as_float(double thr);
void launch_nms_tile(int loader, const NmsArgs& a, hipStream_t s);

struct TopkArgs {
  const unsigned long long* cand;  // [B, cap]
  const int* cand_count;           // [B]
  int* reset_count;                // [B] or null: the kernel zeroes the image's candidate counter once it has read it (the next call's NMS starts from 0 without a memset launch)
  int cap, max_kp;
  int score_w;                     // W of the score map (key idx = h*W + w)
  float scale_x, scale_y;          // input_w / score_w, input_h / score_h (float division on the host)
  int desc_h, desc_w;
  float* kp_xys;                   // [B, max_kp, 3]
  int* cell_h;                     // [B, max_kp]
  int* cell_w;                     // [B, max_kp]
  int* n_out;                      // [B]
  int* n_cand_out;                 // [B] or null
};
void launch_topk(const TopkArgs& a, int B, hipStream_t s);
void launch_threshold_scan(const float* scores, int H, int W, float thr_f, int border, unsigned long long* cand,
                           int* cand_count, int cap, hipStream_t s);
void launch_gather_hwc(bool raw, const _Float16* grid, int C, int gh, int gw, size_t img_stride, const int* cell_h,
                       const int* cell_w, const int* n_dev, int n_host, int max_kp, int B, _Float16* out,
                       hipStream_t s);
void launch_gather_chw(const _Float16* grid, int C, int gh, int gw, const int* cell_h, const int* cell_w, int n,
                       _Float16* out, hipStream_t s);
void launch_desc_dense_chw(const _Float16* raw, int cells_per_img, int B, _Float16* out, hipStream_t s);
void launch_logits_chw(const float* in, int ls, int cells_per_img, int B, float* out, hipStream_t s);
void launch_bgr2gray(const uint8_t* in, int n, uint8_t* out, hipStream_t s);

// ---- sp_convs.hip : MFMA implicit-GEMM layers of SuperPoint ----
struct ConvW {          // one packed conv / linear layer on the device
  _Float16* w = nullptr;  // packed A-fragment order (igemm.h)
  _Float16* w_wino = nullptr;  // optional: Winograd F(2x2,3x3)-transformed weights of a 64 -> 64 3x3 layer (conv_wino.hip)
  _Float16* w_q = nullptr;  // optional second form: 64-row cout tiles over 32-channel chunks (conv_pp128.hip); convPb: the plain [80][256] matrix
  float* bias = nullptr;  // [cout_pad]
  int cin = 0, cout = 0, cout_pad = 0, ks = 1, ct = 64;
};
// in: channels-last fp16 [B,H,W,cin]; out: [B,Ho,Wo,cout] fp16 (pool: floor(/2)).
hipError_t sp_conv3x3_strip(const ConvW& w, const _Float16* in, _Float16* out, int B, int H, int W, bool pool,
                            hipStream_t s);
hipError_t sp_conv1ab_fused(const ConvW& w1b, const _Float16* w1a_frag, const float* b1a, const uint8_t* img,
                            _Float16* out, int B, int H, int W, hipStream_t s);
hipError_t sp_conv3x3_pp(const ConvW& w, const _Float16* in, _Float16* out, int B, int H, int W, bool pool, hipStream_t s);
bool sp_conv3x3_pp128_fits(int B, int H, int W, int cin);
hipError_t sp_conv3x3_pp128(const ConvW& w, const _Float16* in, _Float16* out, int B, int H, int W, bool pool, hipStream_t s);
// conv_fuse2.hip: conv2a -> conv2b -> max-pool in one launch (rolling window over 30-column strips, weights in registers); bit-identical to the
// two launches of sp_conv3x3_pp.  fits(): the shape is supported and (unless `any_batch`) the batch fills the chip with strip segments
bool sp_conv2ab_fused_fits(int B, int H, int W, bool any_batch);
hipError_t sp_conv2ab_fused(const ConvW& wa, const ConvW& wb, const _Float16* in, _Float16* out, int B, int H, int W, hipStream_t s);
// the same rolling-window kernel as ONE 64 -> 128 layer (conv3a): all 128 output channels in one launch, weights in registers; bit-identical to sp_conv3x3_pp
bool sp_conv3a_roll_fits(int B, int H, int W);
hipError_t sp_conv3a_roll(const ConvW& w, const _Float16* in, _Float16* out, int B, int H, int W, hipStream_t s);
hipError_t sp_conv1ab_pp(const ConvW& w1b, const _Float16* w1a_frag, const float* b1a, const uint8_t* img, _Float16* out,
                         int B, int H, int W, hipStream_t s);
// conv_wino.hip: Winograd F(2x2, 3x3) for 64 -> 64 channels (SUPERSLAM_HIP_CONV64=wino)
bool sp_conv3x3_wino_fits(int H, int W, int cin, int cout);
hipError_t sp_conv3x3_wino(const _Float16* upack, const float* bias, const _Float16* in, _Float16* out, int B, int H, int W, bool pool, hipStream_t s);
hipError_t sp_conv1x1_f16(const ConvW& w, const _Float16* in, _Float16* out, int B, int H, int W, hipStream_t s);
// compute units of the current device (cached; persistent kernels launch one workgroup per CU). probe.hip
int cu_count();

// probe.hip
hipError_t mfma_probe(bool random_operands, float* tflops);

hipError_t launch_desc_head_sparse(const ConvW& da32, const ConvW& db32, const _Float16* a4b, int Hc, int Wc, const int* cell_h,
                                   const int* cell_w, const int* n_dev, int max_kp, int B, _Float16* out, size_t out_img_stride,
                                   hipStream_t s);
void launch_desc_head_gather(const ConvW& db32, const _Float16* da, int Hc, int Wc, const int* cell_h, const int* cell_w,
                             const int* n_dev, int max_kp, int B, _Float16* out, size_t out_img_stride, hipStream_t s);
hipError_t sp_conv1x1_f32(const ConvW& w, const _Float16* in, float* out, int ostride, int B, int H, int W,
                          hipStream_t s);

// ---- ep_kernels.hip : EigenPlaces (ResNet-18 + GeM) ----
// ws: split-K workspace of ws_bytes (ep_splitk_workspace_bytes) or null = never split; the split is clamped to what fits
hipError_t ep_conv(const ConvW& w, const _Float16* in, _Float16* out, const _Float16* res, int H, int W, bool relu, bool decim,
                   hipStream_t s, float* ws = nullptr, size_t ws_bytes = 0);
size_t ep_splitk_workspace_bytes(int in_h, int in_w);
void launch_ep_resize_norm(const uint8_t* src, int stride, int ch, const int* tab, int out_w, int out_h, float* out, hipStream_t s);
void launch_ep_im2col(const float* x, int H, int W, int Ho, int Wo, _Float16* out, hipStream_t s);
void launch_ep_maxpool(const _Float16* in, int H, int W, int Ho, int Wo, _Float16* out, hipStream_t s);
// fused stem + ReLU + max-pool (round 6): wfrag = [2][11][64][8] fp16 A fragments (k = (c, ky, kx padded to 8)), bias fp32 [64]
void launch_ep_stem_pool(const float* x, int H, int W, int Ho, int Wo, int Hp, int Wp, const _Float16* wfrag, const float* bias, _Float16* out,
                         hipStream_t s);
size_t ep_tail_ws_floats();   // the tail's workspace: partial GeM sums of up to 32 workgroups + the [512] pre-normalisation outputs
void launch_ep_tail(const _Float16* feat, int npix, float p, const float* wt, const float* bias, float* ws, int* counters, float* out,
                    hipStream_t s);

// ---- lg_kernels.hip ----
struct LgDims {
  int S;    // sequences (2 * pairs)
  int NP;   // padded tokens per sequence (multiple of 32)
};
void launch_lg_prep(const float* kp, int kp_stride, int kp_seq_stride, const int* lens, int max_kp, int* lens_clamped,
                    const _Float16* desc, size_t desc_seq_stride, const float* wr, float img_w, float img_h, LgDims d,
                    _Float16* x, float* rope, float* kpn, hipStream_t s);
hipError_t lg_linear_heads(const ConvW& w, const _Float16* x, LgDims d, int rope_segs, int t_seg,
                           const float* rope, _Float16* q, _Float16* k, _Float16* vt, hipStream_t s);
void launch_lg_attention(const _Float16* q, const _Float16* k, const _Float16* vt, const int* lens, LgDims d,
                         bool cross, _Float16* ctx, hipStream_t s, bool shared_gpu = false);
// lg_attn_res.hip: throughput batches, the keys of a (sequence, head) resident in LDS
bool lg_attention_res_fits(LgDims d);
void launch_lg_attention_res(const _Float16* q, const _Float16* k, const _Float16* vt, const int* lens, LgDims d, bool cross,
                             _Float16* ctx, hipStream_t s);
// prefetch (both launches below): up to three packed layers the NEXT FFN launch streams; latency mode pulls them into L2 with surplus workgroups
hipError_t launch_lg_proj_heads(const ConvW& next, _Float16* x, LgDims d, int rope_segs, int t_seg, const float* rope, _Float16* q,
                                _Float16* k, _Float16* vt, hipStream_t s, const ConvW* const* prefetch = nullptr);
void launch_lg_ffn(const ConvW& w0, const ConvW& w3, const float* gamma, const float* beta, const _Float16* ctx,
                   _Float16* x, LgDims d, const ConvW* next, bool heads, int rope_segs, int t_seg, const float* rope,
                   _Float16* q, _Float16* k, _Float16* vt, _Float16* out, const float* match_w, float match_b,
                   float* logsig, hipStream_t s, const ConvW* const* prefetch = nullptr);
void launch_lg_sim(const _Float16* md, const int* lens, LgDims d, float* sim, hipStream_t s);
void launch_lg_assign(const _Float16* md, const float* logsig, const int* lens, LgDims d, float* ws, float* pcol, int max_kp,
                      int32_t* matches0, float* mscores0, float thr, int stage, hipStream_t s);

}  // namespace sship
