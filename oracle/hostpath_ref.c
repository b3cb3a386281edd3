/* CPU restatement of the reference's host-side hot-path arithmetic.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): built into oracle/_build/liboracle_c.so and
 * loaded only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Each function cites the reference lines it restates (paths relative to /root/reference):
 *   orc_nms_maxpool      utils/convert_superpoint_to_onnx.py:82-87   (9x9 max-pool NMS)
 *   orc_select_topk      src/SuperPoint.cc:696-719                   (threshold scan, sort, top-k, cells)
 *   orc_gather_normalize src/DescriptorGather.cu:14-56               (nearest-cell gather + renorm)
 *   orc_normalize_kpts   src/LightGlue.cc:241-251                    (LightGlue keypoint normalisation)
 *   orc_filter_matches   src/LightGlue.cc:326-363                    (-1 filter, distance = 1 - score)
 *   orc_half_to_float    src/LightGlue.cc:460-475                    (descriptors_to_host widening)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- IEEE binary16 <-> binary32, round-to-nearest-even (what __float2half / __half2float do) ---- */
static float h2f(uint16_t h) {
  uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu, u;
  if (e == 0) {
    if (m == 0) u = s;
    else { /* subnormal */
      int sh = 0;
      while (!(m & 0x400u)) { m <<= 1; ++sh; }
      m &= 0x3ffu;
      u = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
    }
  } else if (e == 31) u = s | 0x7f800000u | (m << 13);
  else u = s | ((e + 112) << 23) | (m << 13);
  float f; memcpy(&f, &u, 4); return f;
}
static uint16_t f2h(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t s = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(s | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
  if (x >= 0x477ff000u) return (uint16_t)(s | 0x7c00u);           /* rounds to inf */
  if (x < 0x33000001u) return (uint16_t)s;                         /* underflow to 0 */
  int e = (int)(x >> 23) - 127;
  uint32_t m = (x & 0x7fffffu) | 0x800000u;
  int shift = (e < -14) ? (13 + (-14 - e)) : 13;
  uint32_t half_m = m >> shift, rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (half_m & 1))) ++half_m;
  uint32_t he = (e < -14) ? 0 : (uint32_t)(e + 15);
  uint32_t out = (e < -14) ? half_m : ((he << 10) + (half_m - 0x400u));
  return (uint16_t)(s | out);
}
void orc_half_to_float(const uint16_t* in, float* out, long n) { for (long i = 0; i < n; ++i) out[i] = h2f(in[i]); }
void orc_float_to_half(const float* in, uint16_t* out, long n) { for (long i = 0; i < n; ++i) out[i] = f2h(in[i]); }

/* convert_superpoint_to_onnx.py:82-87 : pooled = max_pool2d(s, 2r+1, stride 1, pad r) (-inf pad);
 * s = (s == pooled) ? s : 0.  Direct window max, no separable shortcut, so it is an independent check. */
void orc_nms_maxpool(const float* s, int H, int W, int r, float* out) {
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float m = -INFINITY;
      int y0 = y - r < 0 ? 0 : y - r, y1 = y + r >= H ? H - 1 : y + r;
      int x0 = x - r < 0 ? 0 : x - r, x1 = x + r >= W ? W - 1 : x + r;
      for (int yy = y0; yy <= y1; ++yy)
        for (int xx = x0; xx <= x1; ++xx) { float v = s[(long)yy * W + xx]; if (v > m) m = v; }
      float v = s[(long)y * W + x];
      out[(long)y * W + x] = (v == m) ? v : 0.0f;
    }
}

typedef struct { float score; int h, w; } cand_t;
/* std::greater<std::pair<float, std::pair<int,int>>> : descending score, ties by larger h, then larger w
 * (SuperPoint.cc:703). */
static int cand_cmp(const void* a, const void* b) {
  const cand_t *x = (const cand_t*)a, *y = (const cand_t*)b;
  if (x->score != y->score) return x->score > y->score ? -1 : 1;
  if (x->h != y->h) return x->h > y->h ? -1 : 1;
  if (x->w != y->w) return x->w > y->w ? -1 : 1;
  return 0;
}
/* SuperPoint.cc:696-719.  `thr` is the double keypoint_threshold_; `score > thr` promotes the float
 * score to double (strict >).  Returns N; writes kp (x, y, score) triples, (h, w) score coords and cells. */
int orc_select_topk(const float* scores, int score_h, int score_w, int input_h, int input_w,
                    double thr, int border, int max_kp, int desc_h, int desc_w,
                    float* kp_xys, int* hw, int* cell_h, int* cell_w, int* n_candidates) {
  long cap = (long)score_h * score_w, m = 0;
  cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (size_t)(cap > 0 ? cap : 1));
  for (int h = border; h < score_h - border; ++h)
    for (int w = border; w < score_w - border; ++w) {
      float sc = scores[(long)h * score_w + w];
      if ((double)sc > thr) { c[m].score = sc; c[m].h = h; c[m].w = w; ++m; }
    }
  qsort(c, (size_t)m, sizeof(cand_t), cand_cmp);
  int n = (int)(m < max_kp ? m : max_kp);
  const float scale_x = (float)input_w / score_w;   /* static_cast<float>(input_width_) / score_width */
  const float scale_y = (float)input_h / score_h;
  for (int i = 0; i < n; ++i) {
    kp_xys[3 * i + 0] = c[i].w * scale_x;           /* cv::KeyPoint(w*scale_x, h*scale_y, 1, -1, score) */
    kp_xys[3 * i + 1] = c[i].h * scale_y;
    kp_xys[3 * i + 2] = c[i].score;
    if (hw) { hw[2 * i] = c[i].h; hw[2 * i + 1] = c[i].w; }
    int ch = c[i].h / 8, cw = c[i].w / 8;           /* :717-718 nearest cell, clamped */
    cell_h[i] = ch < desc_h - 1 ? ch : desc_h - 1;
    cell_w[i] = cw < desc_w - 1 ? cw : desc_w - 1;
  }
  if (n_candidates) *n_candidates = (int)m;
  free(c);
  return n;
}

/* DescriptorGather.cu:14-56.  grid fp16 [C, gh, gw]; fp32 sum of squares; inv = rsqrtf(sum + 1e-12f);
 * out[n*C + c] = half(v * inv).  The CUDA kernel reduces per-thread partials through a shared-memory
 * tree (256 threads, one channel each at C = 256); `tree` != 0 reproduces that association order,
 * `tree` == 0 sums sequentially.  rsqrtf is evaluated as 1/sqrtf in fp32 (CUDA's rsqrtf is <= 2 ulp). */
void orc_gather_normalize(const uint16_t* grid, int C, int gh, int gw, const int* cell_h, const int* cell_w,
                          int n, uint16_t* out, int tree) {
  const long plane = (long)gh * gw;
  float part[256];
  for (int i = 0; i < n; ++i) {
    const long base = (long)cell_h[i] * gw + cell_w[i];
    float sum;
    if (tree) {
      for (int t = 0; t < 256; ++t) {
        float p = 0.0f;
        for (int c = t; c < C; c += 256) { float v = h2f(grid[c * plane + base]); p += v * v; }
        part[t] = p;
      }
      for (int s = 128; s > 0; s >>= 1) for (int t = 0; t < s; ++t) part[t] += part[t + s];
      sum = part[0];
    } else {
      sum = 0.0f;
      for (int c = 0; c < C; ++c) { float v = h2f(grid[c * plane + base]); sum += v * v; }
    }
    const float inv = 1.0f / sqrtf(sum + 1e-12f);
    for (int c = 0; c < C; ++c) out[(long)i * C + c] = f2h(h2f(grid[c * plane + base]) * inv);
  }
}

/* LightGlue.cc:241-251: (pt - (W/2, H/2)) / (max(W,H)/2), all in float with /2.0f. */
void orc_normalize_kpts(const float* kp_xy, int stride, int n, int image_w, int image_h, float* out) {
  const float scale = (float)(image_w > image_h ? image_w : image_h) / 2.0f;
  const float cx = (float)image_w / 2.0f, cy = (float)image_h / 2.0f;
  for (int i = 0; i < n; ++i) {
    out[2 * i + 0] = (kp_xy[(long)stride * i + 0] - cx) / scale;
    out[2 * i + 1] = (kp_xy[(long)stride * i + 1] - cy) / scale;
  }
}

/* LightGlue.cc:326-363: ascending i, skip matches0[i] < 0, DMatch{i, j, 1 - mscores0[i]}. */
int orc_filter_matches(const int32_t* matches0, const float* mscores0, int n0,
                       int* query, int* train, float* distance) {
  int k = 0;
  for (int i = 0; i < n0; ++i) {
    int j = matches0[i];
    if (j < 0) continue;
    query[k] = i; train[k] = j; distance[k] = 1.0f - mscores0[i]; ++k;
  }
  return k;
}
