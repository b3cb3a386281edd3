// C ABI of libsuperslam_hip.so (include/sship.h): weights, workspaces, streams and the launch sequences of
// the SuperPoint extractor, the LightGlue matcher and the fused front-end step.  gfx950 only; there is no CPU
// fallback anywhere in this library - without a GPU every entry point fails with SSHIP_ERR_NO_DEVICE.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sship.h"
#include "../../include/superslam_hip/place_recognizer.hpp"
#include "kernels.h"

namespace sship {

static thread_local std::string g_err;
static void (*g_log_cb)(int, const char*) = nullptr;
static int g_profiling = 0;  // 0 off, 1 stage marks, 2 stage marks + one mark per SuperPoint layer launch (sship_set_profiling)

void set_error(const std::string& msg) {
  g_err = msg;
  if (g_log_cb) g_log_cb(4, msg.c_str());
}
void log_msg(int level, const char* fmt, ...) {
  if (!g_log_cb) return;
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_log_cb(level, buf);
}
static int fail(int code, const std::string& msg) {
  set_error(msg);
  return code;
}

// ------------------------------------------------------------------------------------------------
// device memory helper
// ------------------------------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t ensure(size_t n) {
    if (n <= bytes) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; bytes = 0;
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  template <class T> T* as() const { return static_cast<T*>(p); }
};
struct PinBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~PinBuf() { if (p) (void)hipHostFree(p); }
  hipError_t ensure(size_t n) {
    if (n <= bytes) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr; bytes = 0;
    hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// ------------------------------------------------------------------------------------------------
// safetensors reader (8-byte LE header length, JSON header, raw little-endian tensor data)
// ------------------------------------------------------------------------------------------------
struct Tensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  size_t numel() const { size_t n = 1; for (auto d : shape) n *= (size_t)d; return n; }
};
typedef std::map<std::string, Tensor> StateDict;

static float half_bits_to_float(uint16_t h) {
  const uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f;
  uint32_t m = h & 0x3ffu, u;
  if (e == 0) {
    if (m == 0) u = s;
    else {
      int sh = 0;
      while (!(m & 0x400u)) { m <<= 1; ++sh; }
      u = s | ((uint32_t)(113 - sh) << 23) | ((m & 0x3ffu) << 13);
    }
  } else if (e == 31) u = s | 0x7f800000u | (m << 13);
  else u = s | ((e + 112) << 23) | (m << 13);
  float f; memcpy(&f, &u, 4); return f;
}

static bool load_safetensors(const std::string& path, StateDict& sd, std::string& err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { err = "cannot open weights file '" + path + "'"; return false; }
  uint64_t hlen = 0;
  f.read(reinterpret_cast<char*>(&hlen), 8);
  if (!f || hlen == 0 || hlen > (1ull << 28)) { err = "bad safetensors header in '" + path + "'"; return false; }
  std::string js(hlen, '\0');
  f.read(&js[0], (std::streamsize)hlen);
  std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  // Minimal parser for {"name": {"dtype": "F32", "shape": [..], "data_offsets": [a, b]}, ...}
  size_t i = 0;
  auto skip_ws = [&]() { while (i < js.size() && (js[i] == ' ' || js[i] == '\n' || js[i] == '\t' || js[i] == '\r')) ++i; };
  auto parse_str = [&](std::string& out) -> bool {
    skip_ws();
    if (i >= js.size() || js[i] != '"') return false;
    ++i; out.clear();
    while (i < js.size() && js[i] != '"') { if (js[i] == '\\' && i + 1 < js.size()) ++i; out.push_back(js[i++]); }
    ++i; return true;
  };
  auto skip_value = [&]() {  // skip a JSON value (used for __metadata__)
    skip_ws();
    int depth = 0; bool in_str = false;
    for (; i < js.size(); ++i) {
      const char c = js[i];
      if (in_str) { if (c == '\\') ++i; else if (c == '"') in_str = false; continue; }
      if (c == '"') in_str = true;
      else if (c == '{' || c == '[') ++depth;
      else if (c == '}' || c == ']') { if (--depth == 0) { ++i; return; } }
      else if ((c == ',') && depth == 0) return;
    }
  };
  skip_ws();
  if (i >= js.size() || js[i] != '{') { err = "safetensors header is not an object"; return false; }
  ++i;
  while (true) {
    skip_ws();
    if (i < js.size() && js[i] == '}') break;
    std::string name;
    if (!parse_str(name)) { err = "safetensors header parse error"; return false; }
    skip_ws();
    if (js[i] != ':') { err = "safetensors header parse error (:)"; return false; }
    ++i;
    if (name == "__metadata__") { skip_value(); }
    else {
      skip_ws();
      if (js[i] != '{') { err = "safetensors entry parse error"; return false; }
      ++i;
      std::string dtype; std::vector<int64_t> shape; uint64_t off0 = 0, off1 = 0;
      while (true) {
        skip_ws();
        if (js[i] == '}') { ++i; break; }
        std::string key;
        if (!parse_str(key)) { err = "safetensors entry key parse error"; return false; }
        skip_ws(); ++i;  // ':'
        skip_ws();
        if (key == "dtype") { parse_str(dtype); }
        else if (key == "shape" || key == "data_offsets") {
          ++i;  // '['
          std::vector<int64_t> vals;
          while (true) {
            skip_ws();
            if (js[i] == ']') { ++i; break; }
            if (js[i] == ',') { ++i; continue; }
            vals.push_back(strtoll(js.c_str() + i, nullptr, 10));
            while (i < js.size() && (isdigit((unsigned char)js[i]) || js[i] == '-')) ++i;
          }
          if (key == "shape") shape = vals;
          else if (vals.size() == 2) { off0 = (uint64_t)vals[0]; off1 = (uint64_t)vals[1]; }
        } else skip_value();
        skip_ws();
        if (js[i] == ',') ++i;
      }
      Tensor t; t.shape = shape;
      const size_t n = t.numel();
      if (off1 > blob.size() || off1 < off0) { err = "tensor '" + name + "' exceeds file"; return false; }
      t.data.resize(n);
      if (dtype == "F32" && off1 - off0 == n * 4) memcpy(t.data.data(), blob.data() + off0, n * 4);
      else if (dtype == "F16" && off1 - off0 == n * 2) {
        const uint16_t* h = reinterpret_cast<const uint16_t*>(blob.data() + off0);
        for (size_t k = 0; k < n; ++k) t.data[k] = half_bits_to_float(h[k]);
      } else if (dtype == "I64" || dtype == "I32" || dtype == "U8" || dtype == "BOOL" || dtype == "I16" || dtype == "I8") {
        skip_ws();
        if (i < js.size() && js[i] == ',') ++i;
        continue;  // bookkeeping tensors (BatchNorm.num_batches_tracked ...): not weights
      } else { err = "tensor '" + name + "': unsupported dtype " + dtype; return false; }
      sd[name] = std::move(t);
    }
    skip_ws();
    if (i < js.size() && js[i] == ',') ++i;
  }
  return true;
}

static const Tensor* find_tensor(const StateDict& sd, const std::string& name, std::initializer_list<int64_t> shape,
                                 std::string& err) {
  auto it = sd.find(name);
  if (it == sd.end()) { err = "missing tensor '" + name + "'"; return nullptr; }
  if (it->second.shape != std::vector<int64_t>(shape)) { err = "tensor '" + name + "' has the wrong shape"; return nullptr; }
  return &it->second;
}

// ------------------------------------------------------------------------------------------------
// weight packing into MFMA A-fragment order (igemm.h):
//   [cout_blk][cin_chunk][ky][kx][kstep][mtile][lane][8]  with
//   cout = cb*CT + mt*32 + (lane & 31),  cin = chunk*64 + ks*16 + (lane >> 5)*8 + e
// `w` is [cout][cin][ks][ks] row-major (PyTorch Conv2d / Linear); row_map/scale allow host-side row
// permutation and folding of constant factors.
// ------------------------------------------------------------------------------------------------
// tile_interleave: M-tile mt of row block cb holds logical rows (mt * nblocks + cb) * 32 .. + 31 instead of
// cb * ct + mt * 32 .. (the fused projection of the LightGlue FFN kernel: every wave then owns one 32-row tile of
// EACH of the q / k / v segments, so all waves have the same epilogue work).  The bias stays in logical order.
static std::vector<_Float16> pack_conv(const float* w, int cout, int cin, int ks, int ct, int chunk, const std::vector<int>* row_map,
                                       const std::vector<float>* row_scale, bool tile_interleave);
static int upload_conv(const float* w, const float* bias, int cout, int cin, int ks, int ct, ConvW& out,
                       const std::vector<int>* row_map = nullptr, const std::vector<float>* row_scale = nullptr,
                       bool tile_interleave = false) {
  const int cout_pad = (cout + ct - 1) / ct * ct, mt_n = ct / 32, nchunk = cin / 64;
  const int nblocks = cout_pad / ct;
  std::vector<_Float16> pk((size_t)cout_pad * cin * ks * ks);
  size_t o = 0;
  for (int cb = 0; cb < cout_pad / ct; ++cb)
    for (int ch = 0; ch < nchunk; ++ch)
      for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx)
          for (int kstep = 0; kstep < 4; ++kstep)
            for (int mt = 0; mt < mt_n; ++mt)
              for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                  const int co = tile_interleave ? (mt * nblocks + cb) * 32 + (lane & 31) : cb * ct + mt * 32 + (lane & 31);
                  const int ci = ch * 64 + kstep * 16 + (lane >> 5) * 8 + e;
                  float v = 0.f;
                  if (co < cout) {
                    const int src = row_map ? (*row_map)[co] : co;
                    v = w[(((size_t)src * cin + ci) * ks + ky) * ks + kx];
                    if (row_scale) v *= (*row_scale)[co];
                  }
                  pk[o++] = (_Float16)v;
                }
  std::vector<float> bp(cout_pad, 0.f);
  for (int co = 0; co < cout; ++co) {
    const int src = row_map ? (*row_map)[co] : co;
    bp[co] = bias ? bias[src] * (row_scale ? (*row_scale)[co] : 1.f) : 0.f;
  }
  // a failure part-way leaves nothing behind (the caller's error path only frees what `out` still points to)
  auto fail_free = [&](hipError_t e, const char* what) {
    if (out.w) (void)hipFree(out.w);
    if (out.bias) (void)hipFree(out.bias);
    out.w = nullptr; out.bias = nullptr;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return (int)SSHIP_ERR_HIP;
  };
  out.w = nullptr; out.bias = nullptr;
  hipError_t e;
  if ((e = hipMalloc(reinterpret_cast<void**>(&out.w), pk.size() * sizeof(_Float16))) != hipSuccess) return fail_free(e, "upload_conv: hipMalloc(weights)");
  if ((e = hipMalloc(reinterpret_cast<void**>(&out.bias), bp.size() * sizeof(float))) != hipSuccess) return fail_free(e, "upload_conv: hipMalloc(bias)");
  if ((e = hipMemcpy(out.w, pk.data(), pk.size() * sizeof(_Float16), hipMemcpyHostToDevice)) != hipSuccess) return fail_free(e, "upload_conv: hipMemcpy(weights)");
  if ((e = hipMemcpy(out.bias, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return fail_free(e, "upload_conv: hipMemcpy(bias)");
  out.cin = cin; out.cout = cout; out.cout_pad = cout_pad; out.ks = ks; out.ct = ct;
  return SSHIP_OK;
}
static void free_conv(ConvW& c) {
  if (c.w) (void)hipFree(c.w);
  if (c.w_q) (void)hipFree(c.w_q);
  if (c.w_wino) (void)hipFree(c.w_wino);
  if (c.bias) (void)hipFree(c.bias);
  c.w = nullptr; c.w_q = nullptr; c.w_wino = nullptr; c.bias = nullptr;
}
// the same fragment order with a K chunk of `chunk` channels: [cout_blk][cin / chunk][ky][kx][k-step chunk / 16][m-tile][lane][8]
static std::vector<_Float16> pack_conv(const float* w, int cout, int cin, int ks, int ct, int chunk, const std::vector<int>* row_map,
                                       const std::vector<float>* row_scale, bool /*tile_interleave*/) {
  const int cout_pad = (cout + ct - 1) / ct * ct, mt_n = ct / 32;
  std::vector<_Float16> pk((size_t)cout_pad * cin * ks * ks);
  size_t o = 0;
  for (int cb = 0; cb < cout_pad / ct; ++cb)
    for (int ch = 0; ch < cin / chunk; ++ch)
      for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx)
          for (int kstep = 0; kstep < chunk / 16; ++kstep)
            for (int mt = 0; mt < mt_n; ++mt)
              for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                  const int co = cb * ct + mt * 32 + (lane & 31);
                  const int ci = ch * chunk + kstep * 16 + (lane >> 5) * 8 + e;
                  float v = 0.f;
                  if (co < cout) {
                    const int src = row_map ? (*row_map)[co] : co;
                    v = w[(((size_t)src * cin + ci) * ks + ky) * ks + kx];
                    if (row_scale) v *= (*row_scale)[co];
                  }
                  pk[o++] = (_Float16)v;
                }
  return pk;
}
// Winograd F(2x2, 3x3) weights of a 64 -> 64 layer for conv_wino.hip: U_p = (G g G^T)[xi][nu], p = 4 xi + nu, computed in fp32 and
// rounded to fp16, packed as MFMA A fragments [p 16][chunk 4][m 2][lane 64][8]: cout = 32 m + (lane & 31), cin = 16 chunk + 8 (lane >> 5) + e
[[maybe_unused]] static int upload_wino(const float* w, ConvW& out) {
  static const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
  std::vector<_Float16> pk((size_t)16 * 4 * 2 * 512);
  for (int co = 0; co < 64; ++co)
    for (int ci = 0; ci < 64; ++ci) {
      const float* g = w + ((size_t)co * 64 + ci) * 9;
      float Gg[4][3], U[4][4];
      for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 3; ++k) Gg[i][k] = G[i][0] * g[0 * 3 + k] + G[i][1] * g[1 * 3 + k] + G[i][2] * g[2 * 3 + k];
      for (int i = 0; i < 4; ++i)
        for (int jn = 0; jn < 4; ++jn) U[i][jn] = Gg[i][0] * G[jn][0] + Gg[i][1] * G[jn][1] + Gg[i][2] * G[jn][2];
      const int m = co >> 5, chunk = ci >> 4, lane = (co & 31) + 32 * ((ci & 15) >> 3), e = ci & 7;
      for (int p = 0; p < 16; ++p) pk[(((size_t)p * 4 + chunk) * 2 + m) * 512 + lane * 8 + e] = (_Float16)U[p >> 2][p & 3];
    }
  SSHIP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&out.w_wino), pk.size() * sizeof(_Float16)));
  SSHIP_HIP_CHECK(hipMemcpy(out.w_wino, pk.data(), pk.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  return SSHIP_OK;
}
// second packing of a 128-input-channel 3x3 layer for conv_pp128.hip (64-row cout tiles, 32-channel chunks)
static int upload_conv_q(const float* w, int cout, int cin, ConvW& out) {
  const std::vector<_Float16> pk = pack_conv(w, cout, cin, 3, 64, 32, nullptr, nullptr, false);
  SSHIP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&out.w_q), pk.size() * sizeof(_Float16)));
  SSHIP_HIP_CHECK(hipMemcpy(out.w_q, pk.data(), pk.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  return SSHIP_OK;
}
static int upload_floats(const float* src, size_t n, float** dst) {
  *dst = nullptr;
  SSHIP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(dst), n * sizeof(float)));
  if (hipError_t e = hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice)) {
    (void)hipFree(*dst); *dst = nullptr;
    set_error(std::string("upload_floats: ") + hipGetErrorString(e));
    return SSHIP_ERR_HIP;
  }
  return SSHIP_OK;
}

static std::atomic<bool> g_inited{false};
static std::atomic<int> g_device{-1};  // the device sship_init selected; every API entry binds the calling thread to it
static std::mutex g_init_mu;           // sship_init may race between the tracking and the loop-closure thread
// HIP's current device is per thread and defaults to 0: a second caller thread (the reference's loop-closure worker,
// SuperSLAM.cc:129-133) on a rank that owns device != 0 would otherwise launch on the wrong GPU.
static inline void bind_thread() {
  static thread_local bool bound = false;
  const int dev = g_device.load(std::memory_order_acquire);
  if (!bound && dev >= 0) { (void)hipSetDevice(dev); bound = true; }
}
static int require_device() {
  if (g_inited.load(std::memory_order_acquire)) return SSHIP_OK;
  return sship_init(-1);
}

// ------------------------------------------------------------------------------------------------
// stage timers (off by default)
// ------------------------------------------------------------------------------------------------
// One timer per calling THREAD (thread_local): the tracking thread and the loop-closure thread each own a handle
//  and may both have profiling on; sship_get_stage_timings reports the calling thread's last sequence.
// Labels are "<reference profile scope>:<stage>" - the scopes are the reference's own SUPERSLAM_PROFILE labels
// (sp_gpu_infer src/SuperPoint.cc:639, sp_extract_stereo :904, fe_lg_stereo_match src/StereoFrontEnd.cc:32), the part
// after the colon is this library's finer split of that scope.
struct StageTimer {
  std::vector<std::pair<const char*, hipEvent_t>> marks;
  std::vector<std::pair<std::string, float>> last;
  // No destructor work: a thread_local's destructor can run after the HIP runtime has been torn down (process exit,
  // or a thread that outlives hipDeviceReset) and hipEventDestroy there is undefined; at most 64 events leak per thread.
  void begin(hipStream_t s) { if (!g_profiling) return; clear(); mark("start", s); }
  // a matcher entered on its own (no extractor call in front of it on this thread) opens its own sequence
  void begin_if_idle(hipStream_t s) { if (g_profiling && marks.empty()) mark("start", s); }
  void mark(const char* label, hipStream_t s) {
    if (!g_profiling) return;
    if (marks.size() >= 64) {  // nobody collected: do not accumulate events without bound; re-arm so the next interval
      clear();                 // is measured from here, not from an arbitrary earlier stage
      hipEvent_t e0;
      if (hipEventCreate(&e0) == hipSuccess) { (void)hipEventRecord(e0, s); marks.push_back({"start", e0}); }
    }
    hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s); marks.push_back({label, e});
  }
  // per-layer marks (level 2 only)
  void mark_fine(const char* label, hipStream_t s) { if (g_profiling >= 2) mark(label, s); }
  void clear() { for (auto& m : marks) (void)hipEventDestroy(m.second); marks.clear(); }
  void collect() {
    if (marks.size() < 2) return;
    (void)hipEventSynchronize(marks.back().second);
    last.clear();
    for (size_t i = 1; i < marks.size(); ++i) {
      float ms = 0.f; (void)hipEventElapsedTime(&ms, marks[i - 1].second, marks[i].second);
      last.push_back({marks[i].first, ms});
    }
    clear();
  }
};
static thread_local StageTimer g_timer;

}  // namespace sship

using namespace sship;

// ====================================================================================================
// runtime
// ====================================================================================================
extern "C" int sship_version(void) { return SSHIP_VERSION; }
extern "C" const char* sship_last_error(void) { return g_err.c_str(); }
extern "C" void sship_set_log_callback(void (*cb)(int, const char*)) { g_log_cb = cb; }
extern "C" void sship_set_profiling(int level) { g_profiling = level < 0 ? 0 : level > 2 ? 2 : level; }
extern "C" int sship_get_stage_timings(const char** labels, float* ms, int max_stages) {
  bind_thread();
  g_timer.collect();
  int n = 0;
  for (auto& kv : g_timer.last) {
    if (n >= max_stages) break;
    labels[n] = kv.first.c_str(); ms[n] = kv.second; ++n;
  }
  return n;
}
extern "C" int sship_init(int device) {
  std::lock_guard<std::mutex> init_guard(g_init_mu);
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(SSHIP_ERR_NO_DEVICE, "no HIP device visible: libsuperslam_hip has no CPU path");
  if (device < 0) {
    const char* env = getenv("SUPERSLAM_HIP_DEVICE");
    device = env ? atoi(env) : -1;
  }
  if (device >= 0) {
    if (device >= count) return fail(SSHIP_ERR_INVALID, "device index out of range");
    SSHIP_HIP_CHECK(hipSetDevice(device));
  }
  int cur = 0;
  SSHIP_HIP_CHECK(hipGetDevice(&cur));
  hipDeviceProp_t prop;
  SSHIP_HIP_CHECK(hipGetDeviceProperties(&prop, cur));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(SSHIP_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  g_device.store(cur, std::memory_order_release);
  g_inited.store(true, std::memory_order_release);
  log_msg(2, "sship: device %d %s, %d CUs", cur, prop.gcnArchName, prop.multiProcessorCount);
  return SSHIP_OK;
}
extern "C" int sship_device_synchronize(void) {
  bind_thread();
  SSHIP_HIP_CHECK(hipDeviceSynchronize());
  return SSHIP_OK;
}

// ====================================================================================================
// descriptor pool  (include/DescriptorPool.h:13-91, src/DescriptorPool.cc:10-38)
// ====================================================================================================
// The bookkeeping (free-list + mutex) is reference counted separately from the device slots, like the reference's
// shared_ptr<FreeList> (DescriptorPool.h:71-75): a DeviceDescriptors handle may outlive the extractor that owns the pool.
// sship_pool_destroy frees the DEVICE memory (DescriptorPool.cc:27-32 - a surviving handle's data pointer dangles exactly
// as in the reference) and drops the owner's reference; the struct itself dies with the last handle reference, so a late
// sship_pool_release never touches freed host memory.
struct sship_pool {
  int max_keypoints = 0, dim = 0;
  size_t slot_bytes = 0;
  std::vector<void*> slots;     // nullptr once destroyed
  std::vector<int> free_slots;  // LIFO (FreeList)
  mutable std::mutex mu;        // handles may be released from another thread (async keyframe copies)
  std::atomic<int> refs{1};     // owner + one per live handle (sship_pool_retain)
};
extern "C" int sship_pool_create(int num_slots, int max_keypoints, int dim, sship_pool** out) {
  bind_thread();
  if (!out || num_slots <= 0 || max_keypoints <= 0 || dim <= 0) return fail(SSHIP_ERR_INVALID, "pool_create: bad arguments");
  if (int rc = require_device()) return rc;
  auto* p = new sship_pool();
  p->max_keypoints = max_keypoints; p->dim = dim;
  p->slot_bytes = (size_t)max_keypoints * dim * sizeof(_Float16);
  p->slots.assign(num_slots, nullptr);
  for (int i = num_slots - 1; i >= 0; --i) p->free_slots.push_back(i);  // DescriptorPool.h:27-29
  for (int i = 0; i < num_slots; ++i)
    if (hipMalloc(&p->slots[i], p->slot_bytes) != hipSuccess) {
      sship_pool_destroy(p);
      return fail(SSHIP_ERR_NOMEM, "pool_create: hipMalloc failed");
    }
  *out = p;
  return SSHIP_OK;
}
extern "C" void sship_pool_retain(sship_pool* pool) {
  if (pool) pool->refs.fetch_add(1, std::memory_order_relaxed);
}
extern "C" void sship_pool_release_ref(sship_pool* pool) {
  if (pool && pool->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete pool;
}
extern "C" void sship_pool_destroy(sship_pool* pool) {
  bind_thread();
  if (!pool) return;
  {
    std::lock_guard<std::mutex> g(pool->mu);
    for (void*& s : pool->slots) { if (s) (void)hipFree(s); s = nullptr; }
  }
  sship_pool_release_ref(pool);
}
extern "C" int sship_pool_acquire(sship_pool* pool) {
  bind_thread();
  if (!pool) return -1;
  std::lock_guard<std::mutex> g(pool->mu);
  if (pool->free_slots.empty()) return -1;
  const int s = pool->free_slots.back();
  pool->free_slots.pop_back();
  return s;
}
extern "C" void sship_pool_release(sship_pool* pool, int slot) {
  bind_thread();
  if (!pool || slot < 0 || slot >= (int)pool->slots.size()) return;
  std::lock_guard<std::mutex> g(pool->mu);
  pool->free_slots.push_back(slot);
}
extern "C" int sship_pool_in_use(const sship_pool* pool) {
  bind_thread();
  if (!pool) return 0;
  std::lock_guard<std::mutex> g(pool->mu);
  return (int)pool->slots.size() - (int)pool->free_slots.size();
}
extern "C" void* sship_pool_slot_ptr(const sship_pool* pool, int slot) {
  bind_thread();
  if (!pool || slot < 0 || slot >= (int)pool->slots.size()) return nullptr;
  std::lock_guard<std::mutex> g(pool->mu);
  return pool->slots[slot];
}

// ====================================================================================================
// gather / nms / select stage entry points
// ====================================================================================================
extern "C" int sship_gather_normalize(const void* grid, int channels, int gh, int gw, const int* cell_h,
                                      const int* cell_w, int n, void* out, void* stream) {
  bind_thread();
  if (n <= 0) return SSHIP_OK;  // DescriptorGather.cu:69
  if (!grid || !cell_h || !cell_w || !out || channels <= 0) return fail(SSHIP_ERR_INVALID, "gather_normalize: null argument");
  if (int rc = require_device()) return rc;
  launch_gather_chw(static_cast<const _Float16*>(grid), channels, gh, gw, cell_h, cell_w, n,
                    static_cast<_Float16*>(out), static_cast<hipStream_t>(stream));
  SSHIP_HIP_CHECK(hipGetLastError());
  return SSHIP_OK;
}
extern "C" int sship_gather_normalize_hwc(const void* grid, int channels, int gh, int gw, const int* cell_h,
                                          const int* cell_w, int n, void* out, void* stream) {
  bind_thread();
  if (n <= 0) return SSHIP_OK;
  if (!grid || !cell_h || !cell_w || !out) return fail(SSHIP_ERR_INVALID, "gather_normalize_hwc: null argument");
  if (channels <= 0 || channels > 256 || channels % 4) return fail(SSHIP_ERR_INVALID, "gather_normalize_hwc: channels must be <= 256 and a multiple of 4");
  if (int rc = require_device()) return rc;
  launch_gather_hwc(false, static_cast<const _Float16*>(grid), channels, gh, gw, 0, cell_h, cell_w, nullptr, n, n, 1,
                    static_cast<_Float16*>(out), static_cast<hipStream_t>(stream));
  SSHIP_HIP_CHECK(hipGetLastError());
  return SSHIP_OK;
}
extern "C" int sship_nms(const float* scores, int batch, int h, int w, int radius, float* out, void* stream) {
  bind_thread();
  if (!scores || !out || batch <= 0 || h <= 0 || w <= 0) return fail(SSHIP_ERR_INVALID, "nms: bad arguments");
  if (radius < 0 || radius > 8) return fail(SSHIP_ERR_INVALID, "nms: radius must be in [0, 8]");
  if (int rc = require_device()) return rc;
  NmsArgs a{};
  a.scores_in = scores; a.B = batch; a.H = h; a.W = w; a.radius = radius; a.scores_out = out;
  launch_nms_tile(1, a, static_cast<hipStream_t>(stream));
  SSHIP_HIP_CHECK(hipGetLastError());
  return SSHIP_OK;
}
extern "C" int sship_select_topk(const float* scores, int score_h, int score_w, int input_h, int input_w, double thr,
                                 int border, int max_kp, int desc_h, int desc_w, float* kp_xys, int* cell_h,
                                 int* cell_w, int* n_dev, int* n_cand_dev, void* stream) {
  bind_thread();
  if (!scores || !kp_xys || !cell_h || !cell_w || !n_dev || score_h <= 0 || score_w <= 0)
    return fail(SSHIP_ERR_INVALID, "select_topk: bad arguments");
  if (max_kp <= 0 || max_kp > kMaxKp) return fail(SSHIP_ERR_INVALID, "select_topk: max_kp must be in [1, 4096]");
  if (int rc = require_device()) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int cap = score_h * score_w;
  DevBuf cand, cnt;
  SSHIP_HIP_CHECK(cand.ensure((size_t)cap * 8));
  SSHIP_HIP_CHECK(cnt.ensure(4));
  SSHIP_HIP_CHECK(hipMemsetAsync(cnt.p, 0, 4, s));
  launch_threshold_scan(scores, score_h, score_w, threshold_as_float(thr), border, cand.as<unsigned long long>(),
                        cnt.as<int>(), cap, s);
  TopkArgs t{};
  t.cand = cand.as<unsigned long long>(); t.cand_count = cnt.as<int>(); t.cap = cap; t.max_kp = max_kp;
  t.score_w = score_w;
  t.scale_x = static_cast<float>(input_w) / score_w;  // SuperPoint.cc:707-708
  t.scale_y = static_cast<float>(input_h) / score_h;
  t.desc_h = desc_h; t.desc_w = desc_w; t.kp_xys = kp_xys; t.cell_h = cell_h; t.cell_w = cell_w; t.n_out = n_dev;
  t.n_cand_out = n_cand_dev;
  launch_topk(t, 1, s);
  SSHIP_HIP_CHECK(hipGetLastError());
  SSHIP_HIP_CHECK(hipStreamSynchronize(s));  // scratch buffers die with this scope
  return SSHIP_OK;
}

// ====================================================================================================
// SuperPoint
// ====================================================================================================
struct sship_sp {
  sship_sp_config cfg{};
  hipStream_t stream = nullptr;
  float* w1a = nullptr;  // [9][64] tap-major fp32 (fp16-rounded values)  (stand-alone conv1a kernel)
  float* b1a = nullptr;
  _Float16* w1a_frag = nullptr;   // conv1a as MFMA A fragments [2][64][8] (K = 9 taps zero-padded to 16)
  _Float16* w1a_fragb = nullptr;  // ping-pong kernel: taps 0..4 | taps 5..8 + the bias split into fp16 hi/lo parts (see sship_sp_create)
  ConvW c1b, c2a, c2b, c3a, c3b, c4a, c4b, cPa, cPb, cDa, cDb;
  ConvW cDb32;  // convDb packed in 32-row blocks (one per wave of k_desc_head_gather)
  sship_pool* pool = nullptr;
  // activations (channels-last fp16), sized for (B, H, W)
  int wsB = 0, wsH = 0, wsW = 0;
  DevBuf img, a1a, a1b, a2a, a2b, a3a, a3b, a4a, a4b, aPa, aDa, draw, logits, cand, cand_count;
  const void* cand_zero_ptr = nullptr; int cand_zero_n = 0;  // cand_count[0 .. cand_zero_n) of this allocation are known to be zero (sp_select) ...
  hipStream_t cand_zero_stream = nullptr;                      // ... for work ordered after the last k_topk on THIS stream
  DevBuf kp, cell_h, cell_w, n_dev, desc_stage, gray_in;
  PinBuf h_kp, h_n, h_img;
  int cap = 0;
  float thr_f = 0.f;
  bool img_valid = false;  // sp->img holds the pixels of the last call (host-image paths always; batch path under profiling)
  // decode-ahead upload ring (sship_sp_ring_*): `depth` stereo frames in pinned host memory + their device copies
  struct Ring {
    int depth = 0, h = 0, w = 0, ch = 0;
    size_t img_bytes = 0;
    std::vector<void*> host, dev;     // [depth]: left image then right image, contiguous
    std::vector<hipEvent_t> uploaded;  // [depth]
    hipStream_t copy_stream = nullptr;
    // sship_sp_ring_submit: an extraction enqueued ahead of the call that collects it
    struct Pending { bool active = false; int slots[2] = {-1, -1}; int rc_pool = 0; hipEvent_t done = nullptr; void* h_kp = nullptr; void* h_n = nullptr; };
    std::vector<Pending> pending;      // [depth]
  } ring;
};

static void ring_free(sship_sp* sp);
static void sp_shapes(int H, int W, int& H2, int& W2, int& H4, int& W4, int& Hc, int& Wc) {
  H2 = H / 2; W2 = W / 2; H4 = H2 / 2; W4 = W2 / 2; Hc = H4 / 2; Wc = W4 / 2;  // MaxPool2d(2,2) floors
}

static int sp_ensure(sship_sp* sp, int B, int H, int W) {
  if (B <= sp->wsB && H == sp->wsH && W == sp->wsW) return SSHIP_OK;
  int H2, W2, H4, W4, Hc, Wc;
  sp_shapes(H, W, H2, W2, H4, W4, Hc, Wc);
  if (Hc < 1 || Wc < 1) return fail(SSHIP_ERR_INVALID, "image too small for SuperPoint (needs >= 8x8)");
  const size_t px1 = (size_t)B * H * W, px2 = (size_t)B * H2 * W2, px4 = (size_t)B * H4 * W4, pxc = (size_t)B * Hc * Wc;
  const int mk = sp->cfg.max_keypoints;
  SSHIP_HIP_CHECK(sp->img.ensure(px1));
  SSHIP_HIP_CHECK(sp->gray_in.ensure(px1 * 3));
  SSHIP_HIP_CHECK(sp->a1a.ensure(px1 * 64 * 2));
  SSHIP_HIP_CHECK(sp->a1b.ensure(px2 * 64 * 2));
  SSHIP_HIP_CHECK(sp->a2a.ensure(px2 * 64 * 2));
  SSHIP_HIP_CHECK(sp->a2b.ensure(px4 * 64 * 2));
  SSHIP_HIP_CHECK(sp->a3a.ensure(px4 * 128 * 2));
  SSHIP_HIP_CHECK(sp->a3b.ensure(pxc * 128 * 2));
  SSHIP_HIP_CHECK(sp->a4a.ensure(pxc * 128 * 2));
  SSHIP_HIP_CHECK(sp->a4b.ensure(pxc * 128 * 2));
  SSHIP_HIP_CHECK(sp->aPa.ensure(pxc * 256 * 2));
  SSHIP_HIP_CHECK(sp->aDa.ensure(pxc * 256 * 2));
  SSHIP_HIP_CHECK(sp->draw.ensure(pxc * 256 * 2));
  SSHIP_HIP_CHECK(sp->logits.ensure(pxc * kLogitStride * 4));
  sp->cap = Hc * 8 * Wc * 8;
  SSHIP_HIP_CHECK(sp->cand.ensure((size_t)B * sp->cap * 8));
  SSHIP_HIP_CHECK(sp->cand_count.ensure((size_t)B * 4));
  SSHIP_HIP_CHECK(sp->kp.ensure((size_t)B * mk * 3 * 4));
  SSHIP_HIP_CHECK(sp->cell_h.ensure((size_t)B * mk * 4));
  SSHIP_HIP_CHECK(sp->cell_w.ensure((size_t)B * mk * 4));
  SSHIP_HIP_CHECK(sp->n_dev.ensure((size_t)B * 4));
  SSHIP_HIP_CHECK(sp->h_kp.ensure((size_t)B * mk * 3 * 4));
  SSHIP_HIP_CHECK(sp->h_n.ensure((size_t)B * 4));
  SSHIP_HIP_CHECK(sp->h_img.ensure(px1 * 3));
  sp->wsB = B; sp->wsH = H; sp->wsW = W;
  return SSHIP_OK;
}

// 3x3 conv kernel selection.  Default: the ping-pong kernel (conv_pp.hip) for every layer; SUPERSLAM_HIP_CONV=strip
// runs the lock-step strip kernel instead (conv_strip.hip, kept for A/B runs: profiles/r01_pp_vs_strip.txt).
static int conv_mode() {  // 1 ping-pong (default), 2 strip (developer build only)
  static const int v = [] {
    const char* e = dev_env("SUPERSLAM_HIP_CONV");
    return (SSHIP_DEV_SWITCHES && e && std::string(e) == "strip") ? 2 : 1;
  }();
  return v;
}
static hipError_t conv3(const ConvW& w, const _Float16* in, _Float16* out, int B, int H, int W, bool pool, hipStream_t s) {
#if SSHIP_DEV_SWITCHES
  // SUPERSLAM_HIP_CONV64=wino: conv2a / conv2b as Winograd F(2x2, 3x3) (conv_wino.hip; A/B)
  static const bool wino = [] { const char* e = dev_env("SUPERSLAM_HIP_CONV64"); return e && std::string(e) == "wino"; }();
  if (wino && w.w_wino && sp_conv3x3_wino_fits(H, W, w.cin, w.cout)) return sp_conv3x3_wino(w.w_wino, w.bias, in, out, B, H, W, pool, s);
  return conv_mode() == 1 ? sp_conv3x3_pp(w, in, out, B, H, W, pool, s) : sp_conv3x3_strip(w, in, out, B, H, W, pool, s);
#else
  return sp_conv3x3_pp(w, in, out, B, H, W, pool, s);
#endif
}
// conv2a + conv2b + pool run as ONE launch (conv_fuse2.hip) at every batch size - the strips are cut into as many row segments as it takes to give
// every CU work, so a one-pair call gains as well (0.736 -> 0.710 ms) - unless the shape does not fit the kernel (maps under 8 pixels, images beyond
// 32-bit offsets): then, and in the developer build under SUPERSLAM_HIP_CONV2=split (A/B; tests/test_gpu_alt_paths.py), the two conv3x3_pp launches run.
// Both paths give the same bits.  (sship_sp_debug_activation layer 2 = conv2a's map exists only on the two-launch path.)
static bool conv2_fused(int B, int H2, int W2) {
#if SSHIP_DEV_SWITCHES
  static const std::string mode = [] { const char* e = dev_env("SUPERSLAM_HIP_CONV2"); return std::string(e ? e : ""); }();
  static const bool other = [] { const char* e = dev_env("SUPERSLAM_HIP_CONV64"); return e != nullptr; }();
  if (mode == "split" || other || conv_mode() != 1) return false;
#endif
  return sp_conv2ab_fused_fits(B, H2, W2, true);
}
// conv3a (64 -> 128) runs on conv3x3_pp<64, 64> (two cout tiles).  The rolling-window kernel's one-layer form (conv_fuse2.hip: conv_roll<true>, all 128
// output channels in one launch, weights in registers) is bit-identical and draws the SAME joules (0.7276 vs 0.7274 J per 128-image launch,
// profiles/r06_l_*): it exists in the developer build only (SUPERSLAM_HIP_CONV3A=roll; A/B in tests/test_gpu_alt_paths.py).
static hipError_t conv3a_layer(const ConvW& w, const _Float16* in, _Float16* out, int B, int H, int W, hipStream_t s) {
#if SSHIP_DEV_SWITCHES
  static const bool roll = [] { const char* e = dev_env("SUPERSLAM_HIP_CONV3A"); return e && std::string(e) == "roll"; }();
  if (roll && conv_mode() == 1 && w.cin == 64 && w.cout == 128 && w.ct == 64 && sp_conv3a_roll_fits(B, H, W)) return sp_conv3a_roll(w, in, out, B, H, W, s);
#endif
  return conv3(w, in, out, B, H, W, false, s);
}
static hipError_t conv1ab(sship_sp* sp, const uint8_t* img, _Float16* out, int B, int H, int W, hipStream_t s);
static bool desc_dense_mode() {
  static const bool v = [] { const char* e = dev_env("SUPERSLAM_HIP_DESC"); return e && std::string(e) == "dense"; }();
  return v;
}
// descriptor rows of the selected keypoints of `B` images (cells / counts at the given pointers)
static hipError_t desc_head(sship_sp* sp, int img0, int Hc, int Wc, const int* cell_h, const int* cell_w, const int* n_dev, int B,
                            _Float16* out, size_t out_img_stride, hipStream_t s) {
  const int mk = sp->cfg.max_keypoints;
  if (desc_dense_mode()) {
    launch_desc_head_gather(sp->cDb32, sp->aDa.as<_Float16>() + (size_t)img0 * Hc * Wc * 256, Hc, Wc, cell_h, cell_w, n_dev, mk, B, out,
                            out_img_stride, s);
    return hipGetLastError();
  }
  return launch_desc_head_sparse(sp->cDa, sp->cDb32, sp->a4b.as<_Float16>() + (size_t)img0 * Hc * Wc * 128, Hc, Wc, cell_h, cell_w, n_dev,
                                 mk, B, out, out_img_stride, s);
}

// encoder + both heads up to (logits, raw descriptor grid).  utils/convert_superpoint_to_onnx.py:51-64,77,88.
static int sp_network(sship_sp* sp, const uint8_t* imgs, int B, int H, int W, hipStream_t s, bool dense_desc) {
  int H2, W2, H4, W4, Hc, Wc;
  sp_shapes(H, W, H2, W2, H4, W4, Hc, Wc);
  // conv1a is evaluated inside conv1b's tile staging (conv_pp.hip): the 64-channel full-resolution activation
  // never exists in HBM.
  // level-2 profiling: one mark per launch, labelled "<level-1 label>/<layer>" ("sp_gpu_infer:encoder/conv2a", ...,
  // "sp_gpu_infer:encoder/conv4b", "sp_gpu_infer:heads/convPb", "sp_extract_stereo:select/topk").  At level 2 NO entry
  // carries the bare level-1 label: a stage's level-1 time is the SUM over its "<label>/..." entries (what bench.py does)
  SSHIP_HIP_CHECK(conv1ab(sp, imgs, sp->a1b.as<_Float16>(), B, H, W, s));
  g_timer.mark_fine("sp_gpu_infer:encoder/conv1a+conv1b+pool", s);
  if (conv2_fused(B, H2, W2)) {
    // conv2a -> conv2b -> pool in one launch, the map between them never leaves the CU (conv_fuse2.hip; bit-identical to the two launches below)
    SSHIP_HIP_CHECK(sp_conv2ab_fused(sp->c2a, sp->c2b, sp->a1b.as<_Float16>(), sp->a2b.as<_Float16>(), B, H2, W2, s));
    g_timer.mark_fine("sp_gpu_infer:encoder/conv2a+conv2b+pool", s);
  } else {
    SSHIP_HIP_CHECK(conv3(sp->c2a, sp->a1b.as<_Float16>(), sp->a2a.as<_Float16>(), B, H2, W2, false, s));
    g_timer.mark_fine("sp_gpu_infer:encoder/conv2a", s);
    SSHIP_HIP_CHECK(conv3(sp->c2b, sp->a2a.as<_Float16>(), sp->a2b.as<_Float16>(), B, H2, W2, true, s));
    g_timer.mark_fine("sp_gpu_infer:encoder/conv2b+pool", s);
  }
  SSHIP_HIP_CHECK(conv3a_layer(sp->c3a, sp->a2b.as<_Float16>(), sp->a3a.as<_Float16>(), B, H4, W4, s));
  g_timer.mark_fine("sp_gpu_infer:encoder/conv3a", s);
  SSHIP_HIP_CHECK(conv3(sp->c3b, sp->a3a.as<_Float16>(), sp->a3b.as<_Float16>(), B, H4, W4, true, s));
  g_timer.mark_fine("sp_gpu_infer:encoder/conv3b+pool", s);
  SSHIP_HIP_CHECK(conv3(sp->c4a, sp->a3b.as<_Float16>(), sp->a4a.as<_Float16>(), B, Hc, Wc, false, s));
  g_timer.mark_fine("sp_gpu_infer:encoder/conv4a", s);
  SSHIP_HIP_CHECK(conv3(sp->c4b, sp->a4a.as<_Float16>(), sp->a4b.as<_Float16>(), B, Hc, Wc, false, s));
  g_timer.mark(g_profiling >= 2 ? "sp_gpu_infer:encoder/conv4b" : "sp_gpu_infer:encoder", s);
  SSHIP_HIP_CHECK(conv3(sp->cPa, sp->a4b.as<_Float16>(), sp->aPa.as<_Float16>(), B, Hc, Wc, false, s));
  g_timer.mark_fine("sp_gpu_infer:heads/convPa", s);
  SSHIP_HIP_CHECK(sp_conv1x1_f32(sp->cPb, sp->aPa.as<_Float16>(), sp->logits.as<float>(), kLogitStride, B, Hc, Wc, s));
  // the dense descriptor branch (convDa, convDb) is only materialised for the dense API; extraction evaluates both
  // layers at the selected keypoints (k_desc_head_sparse).  SUPERSLAM_HIP_DESC=dense: dense convDa + gather (A/B runs).
  if (dense_desc || desc_dense_mode()) SSHIP_HIP_CHECK(conv3(sp->cDa, sp->a4b.as<_Float16>(), sp->aDa.as<_Float16>(), B, Hc, Wc, false, s));
  if (dense_desc) SSHIP_HIP_CHECK(sp_conv1x1_f16(sp->cDb, sp->aDa.as<_Float16>(), sp->draw.as<_Float16>(), B, Hc, Wc, s));
  g_timer.mark(g_profiling >= 2 ? "sp_gpu_infer:heads/convPb" : "sp_gpu_infer:heads", s);
  return SSHIP_OK;
}

// heatmap softmax + NMS + threshold -> candidates -> top-k keypoints/cells (all on device).
static int sp_select(sship_sp* sp, int B, int H, int W, float* scores_out, float* kp_out, int* n_out, hipStream_t s) {
  int H2, W2, H4, W4, Hc, Wc;
  sp_shapes(H, W, H2, W2, H4, W4, Hc, Wc);
  // The candidate counters are zeroed by k_topk once it has read them (a 5-us memset launch per call in latency mode otherwise);
  // a memset is needed only for a fresh / regrown buffer or after a call that did not get as far as its k_topk launch.
  // the counters are zero for THIS call only if the k_topk that zeroed them ran on the same stream (a call on another stream is not
  // ordered behind it: it clears its own, as every call did before the reset moved into k_topk)
  if (sp->cand_zero_ptr != sp->cand_count.p || sp->cand_zero_n < B || sp->cand_zero_stream != s)
    SSHIP_HIP_CHECK(hipMemsetAsync(sp->cand_count.p, 0, (size_t)B * 4, s));
  sp->cand_zero_ptr = sp->cand_count.p; sp->cand_zero_n = 0; sp->cand_zero_stream = s;
  NmsArgs a{};
  a.logits = sp->logits.as<float>(); a.ls = kLogitStride; a.B = B; a.H = Hc * 8; a.W = Wc * 8;
  a.radius = sp->cfg.nms_radius; a.thr_f = sp->thr_f; a.border = sp->cfg.remove_borders;
  a.cand = sp->cand.as<unsigned long long>(); a.cand_count = sp->cand_count.as<int>(); a.cap = sp->cap;
  a.scores_out = scores_out;
  launch_nms_tile(0, a, s);
  g_timer.mark_fine("sp_extract_stereo:select/nms_tile", s);
  TopkArgs t{};
  t.cand = a.cand; t.cand_count = a.cand_count; t.cap = sp->cap; t.max_kp = sp->cfg.max_keypoints;
  t.score_w = Wc * 8;
  t.scale_x = static_cast<float>(W) / (Wc * 8);
  t.scale_y = static_cast<float>(H) / (Hc * 8);
  t.desc_h = Hc; t.desc_w = Wc; t.kp_xys = kp_out; t.cell_h = sp->cell_h.as<int>(); t.cell_w = sp->cell_w.as<int>();
  t.n_out = n_out; t.n_cand_out = nullptr; t.reset_count = a.cand_count;
  launch_topk(t, B, s);
  SSHIP_HIP_CHECK(hipGetLastError());
  sp->cand_zero_n = B;
  g_timer.mark(g_profiling >= 2 ? "sp_extract_stereo:select/topk" : "sp_extract_stereo:select", s);
  return SSHIP_OK;
}

static hipError_t conv1ab(sship_sp* sp, const uint8_t* img, _Float16* out, int B, int H, int W, hipStream_t s) {
#if SSHIP_DEV_SWITCHES
  if (conv_mode() == 2) return sp_conv1ab_fused(sp->c1b, sp->w1a_frag, sp->b1a, img, out, B, H, W, s);
#endif
  return sp_conv1ab_pp(sp->c1b, sp->w1a_fragb, sp->b1a, img, out, B, H, W, s);
}

extern "C" int sship_sp_create(const sship_sp_config* cfg, sship_sp** out) {
  bind_thread();
  if (!cfg || !out || !cfg->weights_path) return fail(SSHIP_ERR_INVALID, "sp_create: null argument");
  if (cfg->max_keypoints <= 0 || cfg->max_keypoints > kMaxKp) return fail(SSHIP_ERR_INVALID, "sp_create: max_keypoints must be in [1, 4096]");
  if (cfg->nms_radius < 0 || cfg->nms_radius > 8) return fail(SSHIP_ERR_INVALID, "sp_create: nms_radius must be in [0, 8]");
  if (int rc = require_device()) return rc;
  StateDict sd; std::string err;
  if (!load_safetensors(cfg->weights_path, sd, err)) return fail(SSHIP_ERR_IO, err);
  // every early return below releases what has been uploaded so far (weights, fragments, stream, pool)
  std::unique_ptr<sship_sp, void (*)(sship_sp*)> sp(new sship_sp(), sship_sp_destroy);
  sp->cfg = *cfg;
  if (sp->cfg.pool_slots <= 0) sp->cfg.pool_slots = 8;
  if (sp->cfg.max_batch <= 0) sp->cfg.max_batch = 2;
  sp->cfg.weights_path = nullptr;
  sp->thr_f = threshold_as_float(cfg->keypoint_threshold);
  struct L { const char* name; int cout, cin, ks, ct; ConvW* dst; };
  const L layers[] = {{"conv1b", 64, 64, 3, 64, &sp->c1b}, {"conv2a", 64, 64, 3, 64, &sp->c2a},
                      {"conv2b", 64, 64, 3, 64, &sp->c2b}, {"conv3a", 128, 64, 3, 64, &sp->c3a},
                      {"conv3b", 128, 128, 3, 32, &sp->c3b}, {"conv4a", 128, 128, 3, 32, &sp->c4a},
                      {"conv4b", 128, 128, 3, 32, &sp->c4b}, {"convPa", 256, 128, 3, 32, &sp->cPa},
                      {"convPb", 65, 256, 1, 96, &sp->cPb}, {"convDa", 256, 128, 3, 32, &sp->cDa},
                      {"convDb", 256, 256, 1, 128, &sp->cDb}};
  for (const L& l : layers) {
    const Tensor* w = find_tensor(sd, std::string(l.name) + ".weight", {l.cout, l.cin, l.ks, l.ks}, err);
    const Tensor* b = w ? find_tensor(sd, std::string(l.name) + ".bias", {l.cout}, err) : nullptr;
    if (!w || !b) return fail(SSHIP_ERR_IO, err);
    if (int rc = upload_conv(w->data.data(), b->data.data(), l.cout, l.cin, l.ks, l.ct, *l.dst)) return rc;
    // second packing (64-row cout tiles over 32-channel chunks) for the kernels of conv_pp128.hip; conv1b runs fused with conv1a
    // (conv_pp.hip), convDa keeps the k order the sparse descriptor head shares
    if (l.ks == 3 && std::string(l.name) != "convDa" && std::string(l.name) != "conv1b")
      if (int rc = upload_conv_q(w->data.data(), l.cout, l.cin, *l.dst)) return rc;
#if SSHIP_DEV_SWITCHES
    if (std::string(l.name) == "conv2a" || std::string(l.name) == "conv2b")
      if (int rc = upload_wino(w->data.data(), *l.dst)) return rc;
#endif
    if (std::string(l.name) == "convDb")
      if (int rc = upload_conv(w->data.data(), b->data.data(), l.cout, l.cin, 1, 32, sp->cDb32)) return rc;
    if (std::string(l.name) == "convPb") {  // the streaming kernel reads the plain matrix, [80][256] fp16 (sp_convs.hip: k_convpb_stream)
      std::vector<_Float16> plain((size_t)80 * 256, (_Float16)0.f);
      for (int co = 0; co < 65; ++co)
        for (int ci = 0; ci < 256; ++ci) plain[(size_t)co * 256 + ci] = (_Float16)w->data[(size_t)co * 256 + ci];
      SSHIP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&l.dst->w_q), plain.size() * sizeof(_Float16)));
      SSHIP_HIP_CHECK(hipMemcpy(l.dst->w_q, plain.data(), plain.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    }
  }
  {
    const Tensor* w = find_tensor(sd, "conv1a.weight", {64, 1, 3, 3}, err);
    const Tensor* b = w ? find_tensor(sd, "conv1a.bias", {64}, err) : nullptr;
    if (!w || !b) return fail(SSHIP_ERR_IO, err);
    std::vector<float> wt(576);
    for (int co = 0; co < 64; ++co)
      for (int t = 0; t < 9; ++t) wt[t * 64 + co] = (float)(_Float16)w->data[co * 9 + t];  // fp16 engine weights
    if (int rc = upload_floats(wt.data(), 576, &sp->w1a)) return rc;
    if (int rc = upload_floats(b->data.data(), 64, &sp->b1a)) return rc;
    std::vector<_Float16> fr(2 * 64 * 8);
    for (int mt = 0; mt < 2; ++mt)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int co = mt * 32 + (lane & 31), k = (lane >> 5) * 8 + e;
          fr[(mt * 64 + lane) * 8 + e] = (_Float16)(k < 9 ? w->data[co * 9 + k] : 0.f);
        }
    SSHIP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&sp->w1a_frag), fr.size() * sizeof(_Float16)));
    SSHIP_HIP_CHECK(hipMemcpy(sp->w1a_frag, fr.data(), fr.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    // ping-pong kernel (conv_pp.hip, stage_conv1a): balanced K layout - lanes hh = 0 hold taps 0..4 in slots 0..4, lanes hh = 1
    // taps 5..8 in slots 0..3 and the bias, split into fp16 hi / lo parts, in slots 4 and 5 (their B elements are 1.0)
    std::fill(fr.begin(), fr.end(), (_Float16)0.f);
    for (int mt = 0; mt < 2; ++mt)
      for (int lane = 0; lane < 64; ++lane) {
        const int co = mt * 32 + (lane & 31), hh = lane >> 5;
        _Float16* f = &fr[(mt * 64 + lane) * 8];
        if (hh == 0) {
          for (int e = 0; e < 5; ++e) f[e] = (_Float16)w->data[co * 9 + e];
        } else {
          for (int e = 0; e < 4; ++e) f[e] = (_Float16)w->data[co * 9 + 5 + e];
          const _Float16 hi = (_Float16)b->data[co];
          f[4] = hi;
          f[5] = (_Float16)(b->data[co] - (float)hi);
        }
      }
    SSHIP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&sp->w1a_fragb), fr.size() * sizeof(_Float16)));
    SSHIP_HIP_CHECK(hipMemcpy(sp->w1a_fragb, fr.data(), fr.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  }
  // blocking stream: ordered with the legacy default stream, so a caller that passes stream = NULL (e.g. torch's default
  // stream, whose handle IS 0) still gets its work ordered against ours
  SSHIP_HIP_CHECK(hipStreamCreateWithFlags(&sp->stream, hipStreamDefault));
  if (int rc = sship_pool_create(sp->cfg.pool_slots, sp->cfg.max_keypoints, SSHIP_DESC_DIM, &sp->pool)) return rc;
  *out = sp.release();
  return SSHIP_OK;
}
extern "C" void sship_sp_destroy(sship_sp* sp) {
  bind_thread();
  if (!sp) return;
  (void)hipDeviceSynchronize();
  for (ConvW* c : {&sp->c1b, &sp->c2a, &sp->c2b, &sp->c3a, &sp->c3b, &sp->c4a, &sp->c4b, &sp->cPa, &sp->cPb, &sp->cDa, &sp->cDb, &sp->cDb32})
    free_conv(*c);
  if (sp->w1a) (void)hipFree(sp->w1a);
  if (sp->b1a) (void)hipFree(sp->b1a);
  if (sp->w1a_frag) (void)hipFree(sp->w1a_frag);
  if (sp->w1a_fragb) (void)hipFree(sp->w1a_fragb);
  ring_free(sp);  // BEFORE the pool: an uncollected sship_sp_ring_submit hands its slots back to sp->pool
  if (sp->pool) sship_pool_destroy(sp->pool);
  sp->pool = nullptr;
  if (sp->stream) (void)hipStreamDestroy(sp->stream);
  delete sp;
}
extern "C" int sship_sp_debug_activation(sship_sp* sp, int layer, void* out_host, unsigned long long bytes) {
  bind_thread();
  if (!sp || !out_host) return fail(SSHIP_ERR_INVALID, "sp_debug_activation: null argument");
  DevBuf* bufs[8] = {nullptr, &sp->a1b, &sp->a2a, &sp->a2b, &sp->a3a, &sp->a3b, &sp->a4a, &sp->a4b};
  if (layer < 1 || layer > 7 || !bufs[layer]->p || bytes > bufs[layer]->bytes) return fail(SSHIP_ERR_INVALID, "sp_debug_activation: bad layer / size");
  SSHIP_HIP_CHECK(hipDeviceSynchronize());
  SSHIP_HIP_CHECK(hipMemcpy(out_host, bufs[layer]->p, bytes, hipMemcpyDeviceToHost));
  return SSHIP_OK;
}
extern "C" sship_pool* sship_sp_pool(sship_sp* sp) {
  bind_thread(); return sp ? sp->pool : nullptr; }
extern "C" int sship_sp_max_keypoints(const sship_sp* sp) { return sp ? sp->cfg.max_keypoints : 0; }

extern "C" int sship_sp_extract_batch_device(sship_sp* sp, const uint8_t* imgs, int batch, int h, int w, void* desc_out,
                                             float* kp_out, int* n_out, void* stream) {
  bind_thread();
  if (!sp || !imgs || !desc_out || !kp_out || !n_out || batch <= 0) return fail(SSHIP_ERR_INVALID, "sp_extract_batch_device: bad arguments");
  // stream == NULL is the legacy default stream itself (torch's default stream): back-to-back extractor / matcher calls
  // made with NULL are then ordered with each other and with the caller's own NULL-stream work.  (They used to fall back
  // to each handle's private stream, which are ordered with the NULL stream but not with one another.)
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (int rc = sp_ensure(sp, batch, h, w)) return rc;
  // a profiled call keeps a copy of its input in the handle: sship_sp_bench_layer re-launches layers on the buffers this
  // call leaves behind, and a conv1a+conv1b launch on an all-zero image clocks ~7 % higher than on real pixels (the chip
  // runs at its power limit; low-toggle operands draw less) - the figure would flatter the kernel
  if (g_profiling && imgs != sp->img.as<uint8_t>())
    SSHIP_HIP_CHECK(hipMemcpyAsync(sp->img.p, imgs, (size_t)batch * h * w, hipMemcpyDeviceToDevice, s));
  sp->img_valid = g_profiling != 0 || imgs == sp->img.as<uint8_t>();
  g_timer.begin(s);
  if (int rc = sp_network(sp, imgs, batch, h, w, s, false)) return rc;
  if (int rc = sp_select(sp, batch, h, w, nullptr, kp_out, n_out, s)) return rc;
  int H2, W2, H4, W4, Hc, Wc;
  sp_shapes(h, w, H2, W2, H4, W4, Hc, Wc);
  SSHIP_HIP_CHECK(desc_head(sp, 0, Hc, Wc, sp->cell_h.as<int>(), sp->cell_w.as<int>(), n_out, batch, static_cast<_Float16*>(desc_out),
                            (size_t)sp->cfg.max_keypoints * 256, s));
  g_timer.mark("sp_extract_stereo:gather", s);
  return SSHIP_OK;
}

extern "C" int sship_sp_dense(sship_sp* sp, const uint8_t* imgs, int batch, int h, int w, float* scores, void* desc_grid,
                              float* logits, void* stream) {
  bind_thread();
  if (!sp || !imgs || batch <= 0) return fail(SSHIP_ERR_INVALID, "sp_dense: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);  // NULL = legacy default stream (see sship_sp_extract_batch_device)
  if (int rc = sp_ensure(sp, batch, h, w)) return rc;
  if (int rc = sp_network(sp, imgs, batch, h, w, s, desc_grid != nullptr)) return rc;
  int H2, W2, H4, W4, Hc, Wc;
  sp_shapes(h, w, H2, W2, H4, W4, Hc, Wc);
  if (scores) {
    NmsArgs a{};
    a.logits = sp->logits.as<float>(); a.ls = kLogitStride; a.B = batch; a.H = Hc * 8; a.W = Wc * 8;
    a.radius = sp->cfg.nms_radius; a.scores_out = scores;
    launch_nms_tile(0, a, s);
  }
  if (desc_grid) launch_desc_dense_chw(sp->draw.as<_Float16>(), Hc * Wc, batch, static_cast<_Float16*>(desc_grid), s);
  if (logits) launch_logits_chw(sp->logits.as<float>(), kLogitStride, Hc * Wc, batch, logits, s);
  SSHIP_HIP_CHECK(hipGetLastError());
  return SSHIP_OK;
}

extern "C" int sship_mfma_probe(int random_operands, float* tflops) {
  bind_thread();
  if (!tflops) return fail(SSHIP_ERR_INVALID, "mfma_probe: null argument");
  if (int rc = require_device()) return rc;
  SSHIP_HIP_CHECK(mfma_probe(random_operands != 0, tflops));
  return SSHIP_OK;
}

extern "C" int sship_sp_bench_layer(sship_sp* sp, int layer, int batch, int h, int w, int iters, float* avg_ms,
                                    double* macs) {
  bind_thread();
  if (!sp || !avg_ms || iters <= 0 || layer < 0 || layer > 15) return fail(SSHIP_ERR_INVALID, "sp_bench_layer: bad arguments");
  if (batch > sp->wsB || h != sp->wsH || w != sp->wsW) return fail(SSHIP_ERR_INVALID, "sp_bench_layer: run the network at this shape first");
  if (layer <= 1 && !sp->img_valid)
    return fail(SSHIP_ERR_INVALID, "sp_bench_layer: the handle holds no copy of the last input - make one call with sship_set_profiling(1) first "
                                   "(a launch on stale / zero pixels runs at a higher clock than the real one)");
  int H2, W2, H4, W4, Hc, Wc;
  sp_shapes(h, w, H2, W2, H4, W4, Hc, Wc);
  hipStream_t s = sp->stream;
  _Float16 *a1a = sp->a1a.as<_Float16>(), *a1b = sp->a1b.as<_Float16>(), *a2a = sp->a2a.as<_Float16>(),
           *a2b = sp->a2b.as<_Float16>(), *a3a = sp->a3a.as<_Float16>(), *a3b = sp->a3b.as<_Float16>(),
           *a4a = sp->a4a.as<_Float16>(), *a4b = sp->a4b.as<_Float16>(), *aPa = sp->aPa.as<_Float16>(),
           *aDa = sp->aDa.as<_Float16>();
  auto run = [&]() -> hipError_t {
    switch (layer) {
      case 0: launch_conv1a(sp->img.as<uint8_t>(), sp->w1a, sp->b1a, a1a, batch, h, w, s); return hipGetLastError();  // stand-alone (not on the path)
      case 1: return conv1ab(sp, sp->img.as<uint8_t>(), a1b, batch, h, w, s);
      case 15:  // conv2a -> conv2b -> pool in one launch (conv_fuse2.hip): what a throughput batch runs instead of layers 2 and 3
        if (!sp_conv2ab_fused_fits(batch, H2, W2, true)) return hipErrorInvalidValue;
        return sp_conv2ab_fused(sp->c2a, sp->c2b, a1b, a2b, batch, H2, W2, s);
      case 2: return conv3(sp->c2a, a1b, a2a, batch, H2, W2, false, s);
      case 3: return conv3(sp->c2b, a2a, a2b, batch, H2, W2, true, s);
      case 4: return conv3a_layer(sp->c3a, a2b, a3a, batch, H4, W4, s);
      case 5: return conv3(sp->c3b, a3a, a3b, batch, H4, W4, true, s);
      case 6: return conv3(sp->c4a, a3b, a4a, batch, Hc, Wc, false, s);
      case 7: return conv3(sp->c4b, a4a, a4b, batch, Hc, Wc, false, s);
      case 8: return conv3(sp->cPa, a4b, aPa, batch, Hc, Wc, false, s);
      case 9: return sp_conv1x1_f32(sp->cPb, aPa, sp->logits.as<float>(), kLogitStride, batch, Hc, Wc, s);
      case 10: return conv3(sp->cDa, a4b, aDa, batch, Hc, Wc, false, s);
      case 11: return sp_conv1x1_f16(sp->cDb, aDa, sp->draw.as<_Float16>(), batch, Hc, Wc, s);
      case 12: {  // softmax + depth-to-space + NMS + threshold + candidate compaction (k_nms_tile) on the last logits
        if (hipError_t e = hipMemsetAsync(sp->cand_count.p, 0, (size_t)batch * 4, s)) return e;
        sp->cand_zero_n = 0;  // this stage leaves its counts behind (stage 13 reads them): the next extraction clears them itself
        NmsArgs a{};
        a.logits = sp->logits.as<float>(); a.ls = kLogitStride; a.B = batch; a.H = Hc * 8; a.W = Wc * 8;
        a.radius = sp->cfg.nms_radius; a.thr_f = sp->thr_f; a.border = sp->cfg.remove_borders;
        a.cand = sp->cand.as<unsigned long long>(); a.cand_count = sp->cand_count.as<int>(); a.cap = sp->cap;
        launch_nms_tile(0, a, s);
        return hipGetLastError();
      }
      case 13: {  // top-k over the candidates left by the last selection (k_topk)
        TopkArgs t{};
        t.cand = sp->cand.as<unsigned long long>(); t.cand_count = sp->cand_count.as<int>(); t.cap = sp->cap;
        t.max_kp = sp->cfg.max_keypoints; t.score_w = Wc * 8;
        t.scale_x = static_cast<float>(w) / (Wc * 8); t.scale_y = static_cast<float>(h) / (Hc * 8);
        t.desc_h = Hc; t.desc_w = Wc; t.kp_xys = sp->kp.as<float>(); t.cell_h = sp->cell_h.as<int>(); t.cell_w = sp->cell_w.as<int>();
        t.n_out = sp->n_dev.as<int>(); t.n_cand_out = nullptr;
        launch_topk(t, batch, s);
        return hipGetLastError();
      }
      default:  // 14: descriptor head at the selected keypoints (k_desc_head_sparse) into the staging rows
        if (hipError_t e = sp->desc_stage.ensure((size_t)batch * sp->cfg.max_keypoints * 512)) return e;
        return desc_head(sp, 0, Hc, Wc, sp->cell_h.as<int>(), sp->cell_w.as<int>(), sp->n_dev.as<int>(), batch, sp->desc_stage.as<_Float16>(),
                         (size_t)sp->cfg.max_keypoints * 256, s);
    }
  };
  if (layer == 13 || layer == 14) {  // these read the selection's outputs: produce them once on this handle's own buffers
    const int keep = layer;
    layer = 12; SSHIP_HIP_CHECK(run());
    layer = 13; SSHIP_HIP_CHECK(run());
    layer = keep;
  }
  const double px[12] = {(double)h * w, (double)h * w, (double)H2 * W2, (double)H2 * W2, (double)H4 * W4, (double)H4 * W4,
                         (double)Hc * Wc, (double)Hc * Wc, (double)Hc * Wc, (double)Hc * Wc, (double)Hc * Wc, (double)Hc * Wc};
  const double mpp[12] = {9.0 * 64, 576.0 * 64 + 9.0 * 64 /* conv1a fused */, 576.0 * 64, 576.0 * 64, 576.0 * 128, 1152.0 * 128, 1152.0 * 128,
                          1152.0 * 128, 1152.0 * 256, 256.0 * 65, 1152.0 * 256, 256.0 * 256};
  if (macs) *macs = layer < 12 ? px[layer] * mpp[layer] * batch : layer == 15 ? (px[2] * mpp[2] + px[3] * mpp[3]) * batch : 0.0;
  SSHIP_HIP_CHECK(run());  // warm
  hipEvent_t e0, e1;
  SSHIP_HIP_CHECK(hipEventCreate(&e0));
  SSHIP_HIP_CHECK(hipEventCreate(&e1));
  SSHIP_HIP_CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) SSHIP_HIP_CHECK(run());
  SSHIP_HIP_CHECK(hipEventRecord(e1, s));
  SSHIP_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  SSHIP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *avg_ms = ms / iters;
  return SSHIP_OK;
}

// host-image front: upload (pinned) -> gray -> batch path into pool slots -> D2H keypoints.
static int sp_extract_device(sship_sp* sp, const uint8_t* gray, int B, int h, int w, sship_features* const* outs);
static int sp_extract_host(sship_sp* sp, const uint8_t* const* imgs, int B, int h, int w, int stride, int channels,
                           sship_features* const* outs) {
  if (channels != 1 && channels != 3) return fail(SSHIP_ERR_INVALID, "image must have 1 or 3 channels");
  if (h <= 0 || w <= 0 || stride < w * channels) return fail(SSHIP_ERR_INVALID, "bad image geometry");
  if (int rc = sp_ensure(sp, B, h, w)) return rc;
  hipStream_t s = sp->stream;
  const size_t row = (size_t)w * channels, img_bytes = row * h;
  uint8_t* hp = sp->h_img.as<uint8_t>();
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < h; ++y) memcpy(hp + b * img_bytes + y * row, imgs[b] + (size_t)y * stride, row);
  if (channels == 1) {
    SSHIP_HIP_CHECK(hipMemcpyAsync(sp->img.p, hp, img_bytes * B, hipMemcpyHostToDevice, s));
  } else {
    SSHIP_HIP_CHECK(hipMemcpyAsync(sp->gray_in.p, hp, img_bytes * B, hipMemcpyHostToDevice, s));
    launch_bgr2gray(sp->gray_in.as<uint8_t>(), B * h * w, sp->img.as<uint8_t>(), s);
  }
  sp->img_valid = true;
  return sp_extract_device(sp, sp->img.as<uint8_t>(), B, h, w, outs);
}
// `gray`: B u8 images [h][w] resident on the device, ordered with sp->stream.
// Enqueue: network + selection + descriptor head into freshly acquired pool slots + D2H of keypoints / counts into the given
// pinned buffers - everything asynchronous on sp->stream.  slots[b] = -1 where the pool was exhausted (*rc_pool set).
static int sp_extract_enqueue(sship_sp* sp, const uint8_t* gray, int B, int h, int w, int* slots, int* rc_pool, float* h_kp, int* h_n) {
  hipStream_t s = sp->stream;
  g_timer.begin(s);
  *rc_pool = SSHIP_OK;
  for (int b = 0; b < B; ++b) slots[b] = -1;
  if (int rc = sp_network(sp, gray, B, h, w, s, false)) return rc;
  if (int rc = sp_select(sp, B, h, w, nullptr, sp->kp.as<float>(), sp->n_dev.as<int>(), s)) return rc;
  int H2, W2, H4, W4, Hc, Wc;
  sp_shapes(h, w, H2, W2, H4, W4, Hc, Wc);
  const int mk = sp->cfg.max_keypoints;
  // a HIP failure after slots were acquired hands them back: the 8-slot pool must not shrink with every error
  auto give_back = [&](hipError_t e, const char* what) {
    (void)hipStreamSynchronize(s);
    for (int b = 0; b < B; ++b)
      if (slots[b] >= 0) { sship_pool_release(sp->pool, slots[b]); slots[b] = -1; }
    return fail(SSHIP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  };
  for (int b = 0; b < B; ++b) {
    slots[b] = sship_pool_acquire(sp->pool);  // pool_->make(n), SuperPoint.cc:721
    if (slots[b] < 0) { *rc_pool = SSHIP_ERR_POOL_EXHAUSTED; continue; }
    if (hipError_t e = desc_head(sp, b, Hc, Wc, sp->cell_h.as<int>() + (size_t)b * mk, sp->cell_w.as<int>() + (size_t)b * mk,
                                 sp->n_dev.as<int>() + b, 1, static_cast<_Float16*>(sship_pool_slot_ptr(sp->pool, slots[b])), 0, s))
      return give_back(e, "sp_extract: descriptor head");
  }
  g_timer.mark("sp_extract_stereo:gather", s);
  if (hipError_t e = hipMemcpyAsync(h_kp, sp->kp.p, (size_t)B * mk * 12, hipMemcpyDeviceToHost, s)) return give_back(e, "sp_extract: D2H keypoints");
  if (hipError_t e = hipMemcpyAsync(h_n, sp->n_dev.p, (size_t)B * 4, hipMemcpyDeviceToHost, s)) return give_back(e, "sp_extract: D2H counts");
  return SSHIP_OK;
}
// Finish (after the stream / the completion event has been waited for): host copies into the caller's features.
static int sp_extract_finish(sship_sp* sp, int B, const int* slots, int rc_pool, const float* h_kp, const int* h_n, sship_features* const* outs) {
  const int mk = sp->cfg.max_keypoints;
  for (int b = 0; b < B; ++b) {
    const int n = h_n[b];
    outs[b]->n = n; outs[b]->slot = slots[b]; outs[b]->desc_dev = nullptr;
    if (outs[b]->kp_xys) memcpy(outs[b]->kp_xys, h_kp + (size_t)b * mk * 3, (size_t)n * 12);
    if (outs[b]->slot >= 0) {
      if (n == 0) { sship_pool_release(sp->pool, outs[b]->slot); outs[b]->slot = -1; }  // empty handle, SuperPoint.cc:722-723
      else outs[b]->desc_dev = sship_pool_slot_ptr(sp->pool, outs[b]->slot);
    }
  }
  if (rc_pool) return fail(rc_pool, "SuperPoint: descriptor pool exhausted (no free slot)");
  return SSHIP_OK;
}
static int sp_extract_device(sship_sp* sp, const uint8_t* gray, int B, int h, int w, sship_features* const* outs) {
  for (int b = 0; b < B; ++b) { outs[b]->n = 0; outs[b]->desc_dev = nullptr; outs[b]->slot = -1; }
  int slots[2] = {-1, -1}, rc_pool = 0;
  if (B > 2) return fail(SSHIP_ERR_INVALID, "sp_extract: at most two images per host call");
  if (int rc = sp_extract_enqueue(sp, gray, B, h, w, slots, &rc_pool, sp->h_kp.as<float>(), sp->h_n.as<int>())) return rc;
  if (hipError_t e = hipStreamSynchronize(sp->stream)) {
    for (int b = 0; b < B; ++b) if (slots[b] >= 0) sship_pool_release(sp->pool, slots[b]);
    return fail(SSHIP_ERR_HIP, std::string("sp_extract: stream synchronize: ") + hipGetErrorString(e));
  }
  return sp_extract_finish(sp, B, slots, rc_pool, sp->h_kp.as<float>(), sp->h_n.as<int>(), outs);
}

extern "C" int sship_sp_extract(sship_sp* sp, const uint8_t* img, int h, int w, int stride, int channels,
                                sship_features* out) {
  bind_thread();
  if (!sp || !img || !out) return fail(SSHIP_ERR_INVALID, "sp_extract: null argument");
  const uint8_t* imgs[1] = {img};
  sship_features* outs[1] = {out};
  return sp_extract_host(sp, imgs, 1, h, w, stride, channels, outs);
}
extern "C" int sship_sp_extract_stereo(sship_sp* sp, const uint8_t* left, const uint8_t* right, int h, int w, int stride,
                                       int channels, sship_features* out_left, sship_features* out_right) {
  bind_thread();
  if (!sp || !left || !right || !out_left || !out_right) return fail(SSHIP_ERR_INVALID, "sp_extract_stereo: null argument");
  const uint8_t* imgs[2] = {left, right};
  sship_features* outs[2] = {out_left, out_right};
  return sp_extract_host(sp, imgs, 2, h, w, stride, channels, outs);
}
// ---- decode-ahead upload ring (include/sship.h) ----
static void ring_free(sship_sp* sp) {
  auto& r = sp->ring;
  for (void* p : r.host) if (p) (void)hipHostFree(p);
  for (void* p : r.dev) if (p) (void)hipFree(p);
  for (hipEvent_t e : r.uploaded) if (e) (void)hipEventDestroy(e);
  for (auto& pd : r.pending) {
    if (pd.active) {  // submitted, never collected: hand the pool slots back
      if (pd.done) (void)hipEventSynchronize(pd.done);
      for (int b = 0; b < 2; ++b) if (pd.slots[b] >= 0 && sp->pool) sship_pool_release(sp->pool, pd.slots[b]);
    }
    if (pd.done) (void)hipEventDestroy(pd.done);
    if (pd.h_kp) (void)hipHostFree(pd.h_kp);
    if (pd.h_n) (void)hipHostFree(pd.h_n);
  }
  if (r.copy_stream) (void)hipStreamDestroy(r.copy_stream);
  r = sship_sp::Ring();
}
extern "C" int sship_sp_ring_create(sship_sp* sp, int depth, int h, int w, int channels) {
  bind_thread();
  if (!sp || depth < 1 || depth > 16 || h <= 0 || w <= 0 || (channels != 1 && channels != 3)) return fail(SSHIP_ERR_INVALID, "sp_ring_create: bad arguments");
  if (int rc = sp_ensure(sp, 2, h, w)) return rc;
  ring_free(sp);
  auto& r = sp->ring;
  r.depth = depth; r.h = h; r.w = w; r.ch = channels; r.img_bytes = (size_t)h * w * channels;
  r.host.assign(depth, nullptr); r.dev.assign(depth, nullptr); r.uploaded.assign(depth, nullptr);
  r.pending.assign(depth, sship_sp::Ring::Pending());
  auto bail = [&](const char* what, hipError_t e) { ring_free(sp); return fail(SSHIP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); };
  if (hipError_t e = hipStreamCreateWithFlags(&r.copy_stream, hipStreamNonBlocking)) return bail("sp_ring_create: stream", e);
  for (int i = 0; i < depth; ++i) {
    if (hipError_t e = hipHostMalloc(&r.host[i], 2 * r.img_bytes, hipHostMallocDefault)) return bail("sp_ring_create: pinned host frame", e);
    if (hipError_t e = hipMalloc(&r.dev[i], 2 * r.img_bytes)) return bail("sp_ring_create: device frame", e);
    if (hipError_t e = hipEventCreateWithFlags(&r.uploaded[i], hipEventDisableTiming)) return bail("sp_ring_create: event", e);
    if (hipError_t e = hipEventCreateWithFlags(&r.pending[i].done, hipEventDisableTiming)) return bail("sp_ring_create: event", e);
    if (hipError_t e = hipHostMalloc(&r.pending[i].h_kp, (size_t)2 * sp->cfg.max_keypoints * 12, hipHostMallocDefault)) return bail("sp_ring_create: pinned keypoints", e);
    if (hipError_t e = hipHostMalloc(&r.pending[i].h_n, 8, hipHostMallocDefault)) return bail("sp_ring_create: pinned counts", e);
  }
  return SSHIP_OK;
}
extern "C" uint8_t* sship_sp_ring_host(sship_sp* sp, int slot, int image) {
  if (!sp || slot < 0 || slot >= sp->ring.depth || image < 0 || image > 1) return nullptr;
  return static_cast<uint8_t*>(sp->ring.host[slot]) + (size_t)image * sp->ring.img_bytes;
}
extern "C" int sship_sp_ring_upload(sship_sp* sp, int slot) {
  bind_thread();
  if (!sp || slot < 0 || slot >= sp->ring.depth) return fail(SSHIP_ERR_INVALID, "sp_ring_upload: bad slot");
  auto& r = sp->ring;
  // a submitted, uncollected extraction may still be reading r.dev[slot] on sp->stream (directly for 1 channel, through
  // bgr2gray for 3): re-uploading under it would race silently.  Collect first.
  if (r.pending[slot].active)
    return fail(SSHIP_ERR_INVALID, "sp_ring_upload: this slot has a submitted extraction that was not collected (sship_sp_extract_stereo_ring) yet");
  SSHIP_HIP_CHECK(hipMemcpyAsync(r.dev[slot], r.host[slot], 2 * r.img_bytes, hipMemcpyHostToDevice, r.copy_stream));
  SSHIP_HIP_CHECK(hipEventRecord(r.uploaded[slot], r.copy_stream));
  return SSHIP_OK;
}
// the slot's pair as grayscale on the device, ordered with sp->stream; a failed wait / launch is an error, not a partly uploaded frame
static int ring_gray(sship_sp* sp, int slot, const uint8_t** gray) {
  auto& r = sp->ring;
  SSHIP_HIP_CHECK(hipStreamWaitEvent(sp->stream, r.uploaded[slot], 0));
  if (r.ch == 3) {
    launch_bgr2gray(static_cast<const uint8_t*>(r.dev[slot]), 2 * r.h * r.w, sp->img.as<uint8_t>(), sp->stream);
    SSHIP_HIP_CHECK(hipGetLastError());
    *gray = sp->img.as<uint8_t>();
    return SSHIP_OK;
  }
  *gray = static_cast<const uint8_t*>(r.dev[slot]);
  return SSHIP_OK;
}
extern "C" int sship_sp_ring_submit(sship_sp* sp, int slot) {
  bind_thread();
  if (!sp || slot < 0 || slot >= sp->ring.depth) return fail(SSHIP_ERR_INVALID, "sp_ring_submit: bad slot");
  auto& r = sp->ring;
  auto& pd = r.pending[slot];
  if (pd.active) return fail(SSHIP_ERR_INVALID, "sp_ring_submit: this slot already has a submitted extraction (collect it with sship_sp_extract_stereo_ring)");
  if (int rc = sp_ensure(sp, 2, r.h, r.w)) return rc;
  const uint8_t* gray = nullptr;
  if (int rc = ring_gray(sp, slot, &gray)) return rc;
  if (int rc = sp_extract_enqueue(sp, gray, 2, r.h, r.w, pd.slots, &pd.rc_pool, static_cast<float*>(pd.h_kp), static_cast<int*>(pd.h_n))) return rc;
  SSHIP_HIP_CHECK(hipEventRecord(pd.done, sp->stream));
  pd.active = true;
  return SSHIP_OK;
}
extern "C" int sship_sp_extract_stereo_ring(sship_sp* sp, int slot, sship_features* out_left, sship_features* out_right) {
  bind_thread();
  if (!sp || !out_left || !out_right || slot < 0 || slot >= sp->ring.depth) return fail(SSHIP_ERR_INVALID, "sp_extract_stereo_ring: bad arguments");
  auto& r = sp->ring;
  sship_features* outs[2] = {out_left, out_right};
  auto& pd = r.pending[slot];
  if (pd.active) {  // submitted ahead (sship_sp_ring_submit): wait for its completion event only
    pd.active = false;
    for (int b = 0; b < 2; ++b) { outs[b]->n = 0; outs[b]->desc_dev = nullptr; outs[b]->slot = -1; }
    if (hipError_t e = hipEventSynchronize(pd.done)) {
      for (int b = 0; b < 2; ++b) if (pd.slots[b] >= 0) sship_pool_release(sp->pool, pd.slots[b]);
      return fail(SSHIP_ERR_HIP, std::string("sp_extract_stereo_ring: ") + hipGetErrorString(e));
    }
    return sp_extract_finish(sp, 2, pd.slots, pd.rc_pool, static_cast<const float*>(pd.h_kp), static_cast<const int*>(pd.h_n), outs);
  }
  if (int rc = sp_ensure(sp, 2, r.h, r.w)) return rc;
  const uint8_t* gray = nullptr;
  if (int rc = ring_gray(sp, slot, &gray)) return rc;
  return sp_extract_device(sp, gray, 2, r.h, r.w, outs);
}
extern "C" int sship_desc_to_host(const void* desc_dev, int count, int dim, float* out) {
  bind_thread();
  if (count <= 0 || !desc_dev) return SSHIP_OK;  // empty handle -> empty Mat (LightGlue.cc:461-462)
  if (!out || dim <= 0) return fail(SSHIP_ERR_INVALID, "desc_to_host: bad arguments");
  const size_t n = (size_t)count * dim;
  std::vector<uint16_t> tmp(n);
  SSHIP_HIP_CHECK(hipMemcpy(tmp.data(), desc_dev, n * 2, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) out[i] = half_bits_to_float(tmp[i]);
  return SSHIP_OK;
}
extern "C" int sship_sp_infer_host(sship_sp* sp, const uint8_t* img, int h, int w, int stride, int channels, float* kp_xys,
                                   float* desc_f32, int* n) {
  bind_thread();
  if (!sp || !img || !kp_xys || !desc_f32 || !n) return fail(SSHIP_ERR_INVALID, "sp_infer_host: null argument");
  sship_features f{};
  f.kp_xys = kp_xys;
  if (int rc = sship_sp_extract(sp, img, h, w, stride, channels, &f)) return rc;
  *n = f.n;
  int rc = SSHIP_OK;
  if (f.n > 0) rc = sship_desc_to_host(f.desc_dev, f.n, SSHIP_DESC_DIM, desc_f32);
  if (f.slot >= 0) sship_pool_release(sp->pool, f.slot);
  return rc;
}

// ====================================================================================================
// LightGlue
// ====================================================================================================
constexpr int kLgLayers = 9;
struct sship_lg_weights {
  std::mutex mu;
  int refs = 1;
  ConvW qkv[kLgLayers], ffn0_s[kLgLayers], ffn3_s[kLgLayers];  // ffn0_*: out_proj / to_out folded in
  ConvW qkv_t[kLgLayers], cqkv_t[kLgLayers], final_t;            // same projections packed per wave for the FFN tail
  ConvW cqkv[kLgLayers], ffn0_c[kLgLayers], ffn3_c[kLgLayers];
  float *ln_g_s[kLgLayers] = {}, *ln_b_s[kLgLayers] = {}, *ln_g_c[kLgLayers] = {}, *ln_b_c[kLgLayers] = {};
  float* match_w = nullptr;
  float match_b = 0.f;
  float* wr = nullptr;
};
static void lg_weights_free(sship_lg_weights* w) {
  for (int i = 0; i < kLgLayers; ++i) {
    for (ConvW* c : {&w->qkv[i], &w->ffn0_s[i], &w->ffn3_s[i], &w->cqkv[i], &w->ffn0_c[i], &w->ffn3_c[i], &w->qkv_t[i], &w->cqkv_t[i]})
      free_conv(*c);
    for (float* p : {w->ln_g_s[i], w->ln_b_s[i], w->ln_g_c[i], w->ln_b_c[i]}) if (p) (void)hipFree(p);
  }
  free_conv(w->final_t);
  if (w->match_w) (void)hipFree(w->match_w);
  if (w->wr) (void)hipFree(w->wr);
  delete w;
}

extern "C" int sship_lg_weights_load(const char* path, sship_lg_weights** out) {
  bind_thread();
  if (!path || !out) return fail(SSHIP_ERR_INVALID, "lg_weights_load: null argument");
  if (int rc = require_device()) return rc;
  StateDict sd; std::string err;
  if (!load_safetensors(path, sd, err)) return fail(SSHIP_ERR_IO, err);
  // The published checkpoint (superpoint_lightglue.pth) names the blocks self_attn.{i}.* / cross_attn.{i}.*; upstream's
  // LightGlue.__init__ renames them to transformers.{i}.self_attn.* / .cross_attn.* at load time, which is also what
  // matcher.state_dict() (the module the reference exports, utils/convert_lightglue_to_onnx.py:64-74) holds.  Accept both,
  // and an optional "matcher." prefix (a state dict saved from a wrapper module).
  {
    StateDict renamed;
    for (auto& kv : sd) {
      std::string k = kv.first;
      if (k.compare(0, 8, "matcher.") == 0) k = k.substr(8);
      for (const char* blk : {"self_attn.", "cross_attn."}) {
        const size_t bl = strlen(blk);
        if (k.compare(0, bl, blk) == 0) {
          const size_t dot = k.find('.', bl);
          if (dot != std::string::npos && dot > bl && k.find_first_not_of("0123456789", bl) == dot)
            k = "transformers." + k.substr(bl, dot - bl) + "." + std::string(blk, bl - 1) + k.substr(dot);
        }
      }
      renamed[k] = std::move(kv.second);
    }
    sd.swap(renamed);
  }
  sship_lg_weights* w = new sship_lg_weights();
  auto bail = [&](int rc, const std::string& m) { lg_weights_free(w); return fail(rc, m); };
  auto lin = [&](const std::string& name, int cout, int cin, ConvW& dst, const std::vector<int>* map = nullptr,
                 const std::vector<float>* scale = nullptr) -> int {
    const Tensor* wt = find_tensor(sd, name + ".weight", {cout, cin}, err);
    const Tensor* bs = wt ? find_tensor(sd, name + ".bias", {cout}, err) : nullptr;
    if (!wt || !bs) return SSHIP_ERR_IO;
    // 256-row layers use 64-row workgroup tiles (more workgroups for the small token GEMMs), wider ones 128.
    return upload_conv(wt->data.data(), bs->data.data(), cout, cin, 1, cout <= 256 ? 64 : 128, dst, map, scale);
  };
  // FFN of a block with the attention output projection folded into ffn.0:
  //   ffn.0(cat[x, Wo c + bo]) = cat[x, c] [W0a | W0b Wo]^T + (b0 + W0b bo)      (fp64 accumulate on the host)
  // ffn.0' is packed in 64-row blocks and ffn.3 in 32-row blocks: one block per wave of k_lg_ffn.
  auto ffn = [&](const std::string& p, const char* proj, ConvW& d0, ConvW& d3) -> int {
    const Tensor* wo = find_tensor(sd, p + proj + ".weight", {256, 256}, err);
    const Tensor* bo = wo ? find_tensor(sd, p + proj + ".bias", {256}, err) : nullptr;
    const Tensor* w0 = bo ? find_tensor(sd, p + "ffn.0.weight", {512, 512}, err) : nullptr;
    const Tensor* b0 = w0 ? find_tensor(sd, p + "ffn.0.bias", {512}, err) : nullptr;
    const Tensor* w3 = b0 ? find_tensor(sd, p + "ffn.3.weight", {256, 512}, err) : nullptr;
    const Tensor* b3 = w3 ? find_tensor(sd, p + "ffn.3.bias", {256}, err) : nullptr;
    if (!b3) return SSHIP_ERR_IO;
    std::vector<float> wf(512 * 512), bf(512);
    for (int o = 0; o < 512; ++o) {
      const float* row = w0->data.data() + (size_t)o * 512;
      for (int i = 0; i < 256; ++i) wf[(size_t)o * 512 + i] = row[i];
      double bacc = b0->data[o];
      for (int k = 0; k < 256; ++k) bacc += (double)row[256 + k] * bo->data[k];
      bf[o] = (float)bacc;
      for (int i = 0; i < 256; ++i) {
        double a = 0.0;
        for (int k = 0; k < 256; ++k) a += (double)row[256 + k] * wo->data[(size_t)k * 256 + i];
        wf[(size_t)o * 512 + 256 + i] = (float)a;
      }
    }
    if (int rc = upload_conv(wf.data(), bf.data(), 512, 512, 1, 64, d0)) return rc;
    return upload_conv(w3->data.data(), b3->data.data(), 256, 512, 1, 32, d3);
  };
  auto vec = [&](const std::string& name, int n, float** dst) -> int {
    const Tensor* t = find_tensor(sd, name, {n}, err);
    if (!t) return SSHIP_ERR_IO;
    return upload_floats(t->data.data(), n, dst);
  };
  const float kLog2e = 1.4426950408889634f;
  // SelfBlock Wqkv: unflatten(-1, (4, 64, 3)) -> original row f = h*192 + d*3 + c ; new row R = c*256 + h*64 + d.
  std::vector<int> qkv_map(768);
  std::vector<float> qkv_scale(768, 1.f);
  for (int c = 0; c < 3; ++c)
    for (int h = 0; h < 4; ++h)
      for (int d = 0; d < 64; ++d) {
        const int R = c * 256 + h * 64 + d;
        qkv_map[R] = h * 192 + d * 3 + c;
        if (c == 0) qkv_scale[R] = 0.125f * kLog2e;  // softmax(q k^T / sqrt(64)) evaluated with exp2
      }
  // CrossBlock: both qk sides are scaled by 64^-0.25; sqrt(log2 e) on each side turns exp into exp2.
  std::vector<float> cq_scale(256, powf(64.f, -0.25f) * sqrtf(kLog2e));
  std::vector<float> fp_scale(256, powf(256.f, -0.25f));  // mdesc = final_proj(x) / d^0.25
  for (int i = 0; i < kLgLayers; ++i) {
    const std::string ps = "transformers." + std::to_string(i) + ".self_attn.";
    const std::string pc = "transformers." + std::to_string(i) + ".cross_attn.";
    int rc = 0;
    if ((rc = lin(ps + "Wqkv", 768, 256, w->qkv[i], &qkv_map, &qkv_scale))) return bail(rc, err);
    {
      const Tensor* wt = find_tensor(sd, ps + "Wqkv.weight", {768, 256}, err);
      const Tensor* bs = find_tensor(sd, ps + "Wqkv.bias", {768}, err);
      if ((rc = upload_conv(wt->data.data(), bs->data.data(), 768, 256, 1, 96, w->qkv_t[i], &qkv_map, &qkv_scale, true))) return bail(rc, g_err);
    }
    if ((rc = ffn(ps, "out_proj", w->ffn0_s[i], w->ffn3_s[i]))) return bail(rc, err.empty() ? g_err : err);
    if ((rc = vec(ps + "ffn.1.weight", 512, &w->ln_g_s[i]))) return bail(rc, err);
    if ((rc = vec(ps + "ffn.1.bias", 512, &w->ln_b_s[i]))) return bail(rc, err);
    // fused [to_qk ; to_v] -> one 512-row GEMM
    const Tensor* wqk = find_tensor(sd, pc + "to_qk.weight", {256, 256}, err);
    const Tensor* bqk = wqk ? find_tensor(sd, pc + "to_qk.bias", {256}, err) : nullptr;
    const Tensor* wv = bqk ? find_tensor(sd, pc + "to_v.weight", {256, 256}, err) : nullptr;
    const Tensor* bv = wv ? find_tensor(sd, pc + "to_v.bias", {256}, err) : nullptr;
    if (!bv) return bail(SSHIP_ERR_IO, err);
    std::vector<float> wcat(512 * 256), bcat(512), scat(512, 1.f);
    memcpy(wcat.data(), wqk->data.data(), 256 * 256 * 4);
    memcpy(wcat.data() + 256 * 256, wv->data.data(), 256 * 256 * 4);
    memcpy(bcat.data(), bqk->data.data(), 256 * 4);
    memcpy(bcat.data() + 256, bv->data.data(), 256 * 4);
    for (int r = 0; r < 256; ++r) scat[r] = cq_scale[r];
    if ((rc = upload_conv(wcat.data(), bcat.data(), 512, 256, 1, 128, w->cqkv[i], nullptr, &scat))) return bail(rc, g_err);
    if ((rc = upload_conv(wcat.data(), bcat.data(), 512, 256, 1, 64, w->cqkv_t[i], nullptr, &scat, true))) return bail(rc, g_err);
    if ((rc = ffn(pc, "to_out", w->ffn0_c[i], w->ffn3_c[i]))) return bail(rc, err.empty() ? g_err : err);
    if ((rc = vec(pc + "ffn.1.weight", 512, &w->ln_g_c[i]))) return bail(rc, err);
    if ((rc = vec(pc + "ffn.1.bias", 512, &w->ln_b_c[i]))) return bail(rc, err);
  }
  {
    // depth_confidence = -1 -> only log_assignment[n_layers - 1] is evaluated (convert_lightglue_to_onnx.py:71-74)
    const std::string pa = "log_assignment." + std::to_string(kLgLayers - 1) + ".";
    int rc = 0;
    {
      const Tensor* wt = find_tensor(sd, pa + "final_proj.weight", {256, 256}, err);
      const Tensor* bs = find_tensor(sd, pa + "final_proj.bias", {256}, err);
      if ((rc = upload_conv(wt->data.data(), bs->data.data(), 256, 256, 1, 32, w->final_t, nullptr, &fp_scale))) return bail(rc, g_err);
    }
    const Tensor* mw = find_tensor(sd, pa + "matchability.weight", {1, 256}, err);
    const Tensor* mb = mw ? find_tensor(sd, pa + "matchability.bias", {1}, err) : nullptr;
    if (!mb) return bail(SSHIP_ERR_IO, err);
    if ((rc = upload_floats(mw->data.data(), 256, &w->match_w))) return bail(rc, g_err);
    w->match_b = mb->data[0];
    const Tensor* wr = find_tensor(sd, "posenc.Wr.weight", {32, 2}, err);
    if (!wr) return bail(SSHIP_ERR_IO, err);
    if ((rc = upload_floats(wr->data.data(), 64, &w->wr))) return bail(rc, g_err);
  }
  *out = w;
  return SSHIP_OK;
}
extern "C" void sship_lg_weights_retain(sship_lg_weights* w) {
  bind_thread();
  if (!w) return;
  std::lock_guard<std::mutex> g(w->mu);
  ++w->refs;
}
extern "C" void sship_lg_weights_release(sship_lg_weights* w) {
  bind_thread();
  if (!w) return;
  bool last;
  { std::lock_guard<std::mutex> g(w->mu); last = (--w->refs == 0); }
  if (last) lg_weights_free(w);
}

struct sship_lg {
  sship_lg_weights* w = nullptr;
  int image_w = 0, image_h = 0, max_kp = 0, max_pairs = 0, NP = 0;
  hipStream_t stream = nullptr;
  DevBuf x, rope, q, k, vt, ctx, md, logsig, sim, ws;
  DevBuf kp_stage, desc_stage, lens, m0, ms0;
  DevBuf lens_c;  // per-sequence counts clamped to [0, max_kp] by k_lg_prep: what every later kernel of the call reads
  DevBuf kpn;     // normalised keypoints [S*NP][2] f32 (src/LightGlue.cc:241-251 on the device; read back by sship_lg_debug_read)
  int debug_layers = kLgLayers;  // sship_lg_debug_set_layers
  int last_pairs = 0;
  PinBuf h_kp, h_lens, h_m0, h_ms0, h_desc;
  // throughput batches run their transformer layers as two half-batches on two streams (lg_forward): the second stream and the
  // fork / join events
  static constexpr int kAux = 3;
  hipStream_t aux[kAux] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[kAux] = {nullptr, nullptr, nullptr};
};

extern "C" int sship_lg_create(sship_lg_weights* w, int image_w, int image_h, int max_kp, int max_pairs, sship_lg** out) {
  bind_thread();
  if (!w || !out || image_w <= 0 || image_h <= 0) return fail(SSHIP_ERR_INVALID, "lg_create: bad arguments");
  if (max_kp <= 0 || max_kp > kMaxKp) return fail(SSHIP_ERR_INVALID, "lg_create: max_keypoints must be in [1, 4096]");
  if (max_pairs <= 0) max_pairs = 1;
  if (int rc = require_device()) return rc;
  std::unique_ptr<sship_lg> lg(new sship_lg());
  lg->image_w = image_w; lg->image_h = image_h; lg->max_kp = max_kp; lg->max_pairs = max_pairs;
  lg->NP = (max_kp + 31) / 32 * 32;  // 32-token granularity: 600 keypoints -> 608 tokens (128-granular padding was 6.7 % dead work)

  const size_t S = 2 * (size_t)max_pairs, T = S * lg->NP, NP = lg->NP;
  SSHIP_HIP_CHECK(lg->x.ensure(T * 256 * 2));
  SSHIP_HIP_CHECK(lg->rope.ensure(T * 64 * 4));
  SSHIP_HIP_CHECK(lg->q.ensure(T * 256 * 2));
  SSHIP_HIP_CHECK(lg->k.ensure(T * 256 * 2));
  SSHIP_HIP_CHECK(lg->vt.ensure(T * 256 * 2));
  SSHIP_HIP_CHECK(lg->ctx.ensure(T * 256 * 2));
  SSHIP_HIP_CHECK(lg->md.ensure(T * 256 * 2));
  SSHIP_HIP_CHECK(lg->logsig.ensure(T * 4));
  SSHIP_HIP_CHECK(lg->sim.ensure((size_t)max_pairs * NP * NP * 4));
  SSHIP_HIP_CHECK(lg->ws.ensure((size_t)max_pairs * 5 * NP * 4));
  SSHIP_HIP_CHECK(lg->kp_stage.ensure(S * max_kp * 3 * 4));
  SSHIP_HIP_CHECK(lg->desc_stage.ensure(S * max_kp * 256 * 2));
  SSHIP_HIP_CHECK(lg->lens.ensure(S * 4));
  SSHIP_HIP_CHECK(lg->lens_c.ensure(S * 4));
  SSHIP_HIP_CHECK(lg->kpn.ensure(T * 2 * 4));
  SSHIP_HIP_CHECK(lg->m0.ensure((size_t)max_pairs * max_kp * 4));
  SSHIP_HIP_CHECK(lg->ms0.ensure((size_t)max_pairs * max_kp * 4));
  SSHIP_HIP_CHECK(lg->h_kp.ensure(2 * (size_t)max_kp * 3 * 4));
  SSHIP_HIP_CHECK(lg->h_lens.ensure(2 * 4));
  SSHIP_HIP_CHECK(lg->h_m0.ensure((size_t)max_kp * 4));
  SSHIP_HIP_CHECK(lg->h_ms0.ensure((size_t)max_kp * 4));
  SSHIP_HIP_CHECK(lg->h_desc.ensure(2 * (size_t)max_kp * 256 * 2));
  // q/k/vt/ctx of padded tokens must stay finite: start from zeros (prep rewrites x every call).
  SSHIP_HIP_CHECK(hipMemset(lg->q.p, 0, lg->q.bytes));
  SSHIP_HIP_CHECK(hipMemset(lg->k.p, 0, lg->k.bytes));
  SSHIP_HIP_CHECK(hipMemset(lg->vt.p, 0, lg->vt.bytes));
  SSHIP_HIP_CHECK(hipMemset(lg->ctx.p, 0, lg->ctx.bytes));
  // from here on the handle owns streams / events: hand it to sship_lg_destroy on any failure (w is still null: nothing to release)
  std::unique_ptr<sship_lg, void (*)(sship_lg*)> owned(lg.release(), sship_lg_destroy);
  SSHIP_HIP_CHECK(hipStreamCreateWithFlags(&owned->stream, hipStreamDefault));
  for (int i = 0; i < sship_lg::kAux; ++i) {
    SSHIP_HIP_CHECK(hipStreamCreateWithFlags(&owned->aux[i], hipStreamNonBlocking));
    SSHIP_HIP_CHECK(hipEventCreateWithFlags(&owned->ev_join[i], hipEventDisableTiming));
  }
  SSHIP_HIP_CHECK(hipEventCreateWithFlags(&owned->ev_fork, hipEventDisableTiming));
  sship_lg_weights_retain(w);
  owned->w = w;
  *out = owned.release();
  return SSHIP_OK;
}
extern "C" void sship_lg_destroy(sship_lg* lg) {
  bind_thread();
  if (!lg) return;
  (void)hipDeviceSynchronize();
  if (lg->stream) (void)hipStreamDestroy(lg->stream);
  for (int i = 0; i < sship_lg::kAux; ++i) {
    if (lg->aux[i]) (void)hipStreamDestroy(lg->aux[i]);
    if (lg->ev_join[i]) (void)hipEventDestroy(lg->ev_join[i]);
  }
  if (lg->ev_fork) (void)hipEventDestroy(lg->ev_fork);
  sship_lg_weights_release(lg->w);
  delete lg;
}
extern "C" int sship_lg_normalize_keypoints(const sship_lg* lg, const float* kp, int stride, int n, float* out) {
  if (!lg || !kp || !out || stride < 2) return fail(SSHIP_ERR_INVALID, "lg_normalize_keypoints: bad arguments");
  const float scale = std::max(lg->image_w, lg->image_h) / 2.0f;  // LightGlue.cc:242-244
  const float cx = lg->image_w / 2.0f, cy = lg->image_h / 2.0f;
  for (int i = 0; i < n; ++i) {
    out[2 * i + 0] = (kp[(size_t)i * stride + 0] - cx) / scale;
    out[2 * i + 1] = (kp[(size_t)i * stride + 1] - cy) / scale;
  }
  return SSHIP_OK;
}

// Test-only introspection (include/sship.h): truncate the matcher after n layers / read its internal state back.
extern "C" int sship_lg_debug_set_layers(sship_lg* lg, int n_layers) {
  if (!lg || n_layers < 1 || n_layers > kLgLayers) return fail(SSHIP_ERR_INVALID, "lg_debug_set_layers: n_layers must be in [1, 9]");
  lg->debug_layers = n_layers;
  return SSHIP_OK;
}
extern "C" int sship_lg_debug_read(sship_lg* lg, int what, int index, int rows, int cols, float* out) {
  bind_thread();
  if (!lg || !out || rows <= 0 || index < 0) return fail(SSHIP_ERR_INVALID, "lg_debug_read: bad arguments");
  const int NP = lg->NP, S = 2 * lg->last_pairs;
  if (rows > NP) return fail(SSHIP_ERR_INVALID, "lg_debug_read: rows exceeds the padded sequence length");
  SSHIP_HIP_CHECK(hipDeviceSynchronize());
  switch (what) {
    case SSHIP_LG_DEBUG_X: {  // residual stream of sequence `index`: fp16 [NP][256]
      if (index >= S || cols != 256) return fail(SSHIP_ERR_INVALID, "lg_debug_read(X): index/cols");
      std::vector<uint16_t> tmp((size_t)rows * 256);
      SSHIP_HIP_CHECK(hipMemcpy(tmp.data(), lg->x.as<_Float16>() + (size_t)index * NP * 256, tmp.size() * 2, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < tmp.size(); ++i) out[i] = half_bits_to_float(tmp[i]);
      return SSHIP_OK;
    }
    case SSHIP_LG_DEBUG_SIM: {  // assignment similarity of pair `index`: f32 [NP][NP]
      if (index >= S / 2 || cols <= 0 || cols > NP) return fail(SSHIP_ERR_INVALID, "lg_debug_read(SIM): index/cols");
      // the match itself never materialises the matrix: computed here, on demand, from the final projections of the last call
      launch_lg_sim(lg->md.as<_Float16>(), lg->lens_c.as<int>(), LgDims{S, NP}, lg->sim.as<float>(), lg->stream);
      SSHIP_HIP_CHECK(hipStreamSynchronize(lg->stream));
      SSHIP_HIP_CHECK(hipMemcpy2D(out, (size_t)cols * 4, lg->sim.as<float>() + (size_t)index * NP * NP, (size_t)NP * 4, (size_t)cols * 4, rows,
                                  hipMemcpyDeviceToHost));
      return SSHIP_OK;
    }
    case SSHIP_LG_DEBUG_KPTS: {  // normalised keypoints of sequence `index`: f32 [NP][2]
      if (index >= S || cols != 2) return fail(SSHIP_ERR_INVALID, "lg_debug_read(KPTS): index/cols");
      SSHIP_HIP_CHECK(hipMemcpy(out, lg->kpn.as<float>() + (size_t)index * NP * 2, (size_t)rows * 8, hipMemcpyDeviceToHost));
      return SSHIP_OK;
    }
    case SSHIP_LG_DEBUG_ROPE: {  // rotary table of sequence `index`: f32 [NP][32] (cos, sin) pairs
      if (index >= S || cols != 64) return fail(SSHIP_ERR_INVALID, "lg_debug_read(ROPE): index/cols");
      SSHIP_HIP_CHECK(hipMemcpy(out, lg->rope.as<float>() + (size_t)index * NP * 64, (size_t)rows * 256, hipMemcpyDeviceToHost));
      return SSHIP_OK;
    }
    default: return fail(SSHIP_ERR_INVALID, "lg_debug_read: unknown selector");
  }
}

// The matcher proper: `pairs` problems, everything on the device.  9 x (SelfBlock x2 images, CrossBlock), then
// log_assignment[8] + filter_matches.
static int lg_forward(sship_lg* lg, const float* kp, int kp_stride, int kp_seq_stride, const int* lens,
                      const _Float16* desc, size_t desc_seq_stride, int pairs, int32_t* m0, float* ms0, hipStream_t s) {
  const sship_lg_weights* w = lg->w;
  LgDims d{2 * pairs, lg->NP};
  const int T = d.S * d.NP;
  _Float16 *x = lg->x.as<_Float16>(), *q = lg->q.as<_Float16>(), *k = lg->k.as<_Float16>(), *vt = lg->vt.as<_Float16>();
  _Float16* ctx = lg->ctx.as<_Float16>();
  float* rope = lg->rope.as<float>();
  // prep also publishes the counts clamped to [0, max_kp] (a caller-supplied count above max_kp would otherwise read past
  // the descriptor / keypoint stride of its sequence): every kernel below reads the clamped copy.
  launch_lg_prep(kp, kp_stride, kp_seq_stride, lens, lg->max_kp, lg->lens_c.as<int>(), desc, desc_seq_stride, w->wr,
                 (float)lg->image_w, (float)lg->image_h, d, x, rope, lg->kpn.as<float>(), s);
  lens = lg->lens_c.as<int>();
  lg->last_pairs = pairs;
  // 3 launches per block: [projection fused into the previous FFN's tail] -> attention -> FFN(+ next projection).
  static const bool igemm_qkv0 = dev_env("SUPERSLAM_HIP_LG_QKV0") && std::string(dev_env("SUPERSLAM_HIP_LG_QKV0")) == "igemm";  // A/B
  const int n_layers = lg->debug_layers;  // kLgLayers except under sship_lg_debug_set_layers (test-only)
  // the layer stack of pairs [p0, p0 + np) on stream st: every buffer is sequence-major, pairs are independent
  auto layers = [&](int p0, int np, hipStream_t st, bool shared_gpu) -> int {
    const LgDims ds{2 * np, lg->NP};
    const size_t tok = (size_t)2 * p0 * lg->NP;
    _Float16 *xs = x + tok * 256, *qs = q + tok * 256, *ks = k + tok * 256, *vs = vt + tok * 256, *cs = ctx + tok * 256;
    const float* rs = rope + tok * 64;
    const int* ls = lens + 2 * p0;
    // what each FFN launch streams (w0, w3, the fused projection): handed to the launch BEFORE it as a prefetch hint (latency mode)
    auto self_w = [&](int i, const ConvW** o) { o[0] = &w->ffn0_s[i]; o[1] = &w->ffn3_s[i]; o[2] = &w->cqkv_t[i]; };
    auto cross_w = [&](int i, const ConvW** o) { o[0] = &w->ffn0_c[i]; o[1] = &w->ffn3_c[i]; o[2] = i + 1 < kLgLayers ? &w->qkv_t[i + 1] : &w->final_t; };
    const ConvW* pf[3];
    self_w(0, pf);
#if SSHIP_DEV_SWITCHES
    if (igemm_qkv0) SSHIP_HIP_CHECK(lg_linear_heads(w->qkv[0], xs, ds, /*rope_segs=*/2, /*t_seg=*/2, rs, qs, ks, vs, st));
    else
#endif
    SSHIP_HIP_CHECK(launch_lg_proj_heads(w->qkv_t[0], xs, ds, /*rope_segs=*/2, /*t_seg=*/2, rs, qs, ks, vs, st, pf));
    for (int i = 0; i < n_layers; ++i) {
      // SelfBlock (both images of every pair in one launch); its FFN also emits CrossBlock's [to_qk | to_v]
      launch_lg_attention(qs, ks, vs, ls, ds, false, cs, st, shared_gpu);
      cross_w(i, pf);
      launch_lg_ffn(w->ffn0_s[i], w->ffn3_s[i], w->ln_g_s[i], w->ln_b_s[i], cs, xs, ds, &w->cqkv_t[i], true, /*rope_segs=*/0,
                    /*t_seg=*/1, rs, qs, ks, vs, nullptr, nullptr, 0.f, nullptr, st, pf);
      // CrossBlock (qk shared by both directions; sequence s attends to s^1); its FFN emits the next layer's Wqkv,
      // or final_proj + matchability after the last layer
      launch_lg_attention(qs, qs, vs, ls, ds, true, cs, st, shared_gpu);
      if (i + 1 < kLgLayers) {
        self_w(i + 1, pf);
        launch_lg_ffn(w->ffn0_c[i], w->ffn3_c[i], w->ln_g_c[i], w->ln_b_c[i], cs, xs, ds, &w->qkv_t[i + 1], true, 2, 2, rs, qs, ks,
                      vs, nullptr, nullptr, 0.f, nullptr, st, pf);
      } else
        launch_lg_ffn(w->ffn0_c[i], w->ffn3_c[i], w->ln_g_c[i], w->ln_b_c[i], cs, xs, ds, &w->final_t, false, 0, 0, rs, qs, ks, vs,
                      lg->md.as<_Float16>() + tok * 256, w->match_w, w->match_b, lg->logsig.as<float>() + tok, st);
    }
    SSHIP_HIP_CHECK(hipGetLastError());
    return SSHIP_OK;
  };
  // Throughput batches: two half-batches on two streams.  Every LightGlue launch is a whole number of workgroup "rounds" over
  // the 512 workgroup slots plus a partly filled last one (FFN: 1216 tiles = 2.4 rounds at 64 pairs - a third of the launch
  // runs at 37 % occupancy); a second, independent stream of the same kernels fills those tails, and an attention workgroup
  // (70 KB LDS, transcendental-bound) can share a CU with an FFN workgroup (79 KB, matrix-bound) of the other half.
  // Only when each half still qualifies for the throughput kernels; SUPERSLAM_HIP_LG_SPLIT=1 turns it off (A/B runs).
  static const int split_env = dev_env("SUPERSLAM_HIP_LG_SPLIT") ? atoi(dev_env("SUPERSLAM_HIP_LG_SPLIT")) : 0;
  int parts = 1;
  if (split_env >= 1 && split_env <= 1 + sship_lg::kAux) parts = std::min(split_env, pairs);  // forced (1 = off)
  else if ((size_t)2 * (pairs / 2) * lg->NP / 64 >= (size_t)2 * cu_count()) parts = 2;
  if (parts > 1) {
    SSHIP_HIP_CHECK(hipEventRecord(lg->ev_fork, s));
    // Once an auxiliary stream has been forked, every exit - including an error part-way - joins it back into `s` first:
    // work already queued on the (non-blocking) auxiliary streams would otherwise be unordered with the caller's stream and
    // the next call's k_lg_prep could overwrite x / rope / lens_c while an auxiliary stream still reads them.
    int forked = 0, rc = SSHIP_OK;
    hipError_t he = hipSuccess;
    int p0 = pairs / parts;  // part 0 (on s) is launched last: the auxiliary streams are already busy by then
    for (int i = 1; i < parts && rc == SSHIP_OK && he == hipSuccess; ++i) {
      const int np = i + 1 < parts ? pairs / parts : pairs - p0;
      if ((he = hipStreamWaitEvent(lg->aux[i - 1], lg->ev_fork, 0)) != hipSuccess) break;
      forked = i;
      rc = layers(p0, np, lg->aux[i - 1], true);
      p0 += np;
    }
    if (rc == SSHIP_OK && he == hipSuccess) rc = layers(0, pairs / parts, s, true);
    for (int i = 1; i <= forked; ++i) {  // join (also on the error path; best effort there)
      hipError_t e1 = hipEventRecord(lg->ev_join[i - 1], lg->aux[i - 1]);
      if (e1 == hipSuccess) e1 = hipStreamWaitEvent(s, lg->ev_join[i - 1], 0);
      if (e1 != hipSuccess) { (void)hipStreamSynchronize(lg->aux[i - 1]); if (he == hipSuccess) he = e1; }
    }
    if (rc != SSHIP_OK) return rc;
    SSHIP_HIP_CHECK(he);
  } else {
    if (int rc = layers(0, pairs, s, false)) return rc;
  }
  SSHIP_HIP_CHECK(hipGetLastError());
  g_timer.mark("fe_lg_stereo_match:layers_x9", s);
  if (n_layers < kLgLayers) {  // truncated debug run: x after layer n_layers is the product; no assignment
    lg->debug_layers = kLgLayers;  // one-shot: a caller that forgets to reset it must not keep a matcher that matches nothing
    log_msg(3, "sship: LightGlue ran truncated to %d layer(s) (sship_lg_debug_set_layers, test-only); matches0 = -1", n_layers);
    SSHIP_HIP_CHECK(hipMemsetAsync(m0, 0xff, (size_t)pairs * lg->max_kp * 4, s));
    SSHIP_HIP_CHECK(hipMemsetAsync(ms0, 0, (size_t)pairs * lg->max_kp * 4, s));
    return SSHIP_OK;
  }
  // log-assignment + mutual filter straight from md (lg_kernels.hip: k_assign_stream); lg->sim is the partials' scratch
  launch_lg_assign(lg->md.as<_Float16>(), lg->logsig.as<float>(), lens, d, lg->ws.as<float>(), lg->sim.as<float>(), lg->max_kp, m0, ms0,
                   0.1f /* filter_threshold */, 0, s);
  SSHIP_HIP_CHECK(hipGetLastError());
  g_timer.mark("fe_lg_stereo_match:assign_filter", s);
  return SSHIP_OK;
}

extern "C" int sship_lg_match_batch_device(sship_lg* lg, const float* kp, const int* n, const void* desc, int pairs,
                                           int32_t* m0, float* ms0, void* stream) {
  bind_thread();
  if (!lg || !kp || !n || !desc || !m0 || !ms0) return fail(SSHIP_ERR_INVALID, "lg_match_batch_device: null argument");
  if (pairs <= 0 || pairs > lg->max_pairs) return fail(SSHIP_ERR_INVALID, "lg_match_batch_device: pairs exceeds max_pairs");
  hipStream_t s = static_cast<hipStream_t>(stream);  // NULL = legacy default stream: ordered after an extractor call made with NULL
  g_timer.begin_if_idle(s);
  return lg_forward(lg, kp, 3, lg->max_kp * 3, n, static_cast<const _Float16*>(desc), (size_t)lg->max_kp * 256, pairs,
                    m0, ms0, s);
}

// Measurement hook (include/sship.h): one stage of the matcher re-launched `iters` times over the state of the last call.
extern "C" int sship_lg_bench_stage(sship_lg* lg, int stage, int iters, float* avg_ms) {
  bind_thread();
  if (!lg || !avg_ms || iters <= 0 || stage < 0 || stage > 7) return fail(SSHIP_ERR_INVALID, "lg_bench_stage: bad arguments");
  if (lg->last_pairs <= 0) return fail(SSHIP_ERR_INVALID, "lg_bench_stage: run a match on this handle first");
  const sship_lg_weights* w = lg->w;
  const int pairs = lg->last_pairs;
  LgDims d{2 * pairs, lg->NP};
  hipStream_t s = lg->stream;
  _Float16 *x = lg->x.as<_Float16>(), *q = lg->q.as<_Float16>(), *k = lg->k.as<_Float16>(), *vt = lg->vt.as<_Float16>();
  _Float16* ctx = lg->ctx.as<_Float16>();
  float* rope = lg->rope.as<float>();
  const int* lens = lg->lens_c.as<int>();
  const bool call_splits = (size_t)2 * (pairs / 2) * lg->NP / 64 >= (size_t)2 * cu_count();  // lg_forward's two-stream condition
  auto run = [&]() -> hipError_t {
    switch (stage) {
      case 0: return launch_lg_proj_heads(w->qkv_t[0], x, d, 2, 2, rope, q, k, vt, s);
      // the attention kernel the CALL launches: lg_forward runs a throughput batch as two half-batches on two streams and tells the launcher
      // so (shared_gpu: no key split, wave-granular query units).  Rounds 2-5 timed the key-split variant here - a kernel the call never runs.
      case 1: launch_lg_attention(q, k, vt, lens, d, false, ctx, s, call_splits); return hipGetLastError();
      case 2: launch_lg_attention(q, q, vt, lens, d, true, ctx, s, call_splits); return hipGetLastError();
      case 3: launch_lg_ffn(w->ffn0_s[0], w->ffn3_s[0], w->ln_g_s[0], w->ln_b_s[0], ctx, x, d, &w->cqkv_t[0], true, 0, 1, rope, q, k, vt,
                            nullptr, nullptr, 0.f, nullptr, s); return hipGetLastError();
      case 4: launch_lg_ffn(w->ffn0_c[0], w->ffn3_c[0], w->ln_g_c[0], w->ln_b_c[0], ctx, x, d, &w->qkv_t[1], true, 2, 2, rope, q, k, vt,
                            nullptr, nullptr, 0.f, nullptr, s); return hipGetLastError();
      case 5: launch_lg_ffn(w->ffn0_c[8], w->ffn3_c[8], w->ln_g_c[8], w->ln_b_c[8], ctx, x, d, &w->final_t, false, 0, 0, rope, q, k, vt,
                            lg->md.as<_Float16>(), w->match_w, w->match_b, lg->logsig.as<float>(), s); return hipGetLastError();
      case 6: launch_lg_assign(lg->md.as<_Float16>(), lg->logsig.as<float>(), lens, d, lg->ws.as<float>(), lg->sim.as<float>(), lg->max_kp,
                               lg->m0.as<int32_t>(), lg->ms0.as<float>(), 0.1f, 1, s); return hipGetLastError();  // log-sum-exp pass
      default: launch_lg_assign(lg->md.as<_Float16>(), lg->logsig.as<float>(), lens, d, lg->ws.as<float>(), lg->sim.as<float>(), lg->max_kp,
                                lg->m0.as<int32_t>(), lg->ms0.as<float>(), 0.1f, 2, s); return hipGetLastError();  // arg-max pass
    }
  };
  // stages 3 / 4 update the residual stream in place: keep a copy and put it back (the values do not affect the timing, but
  // the handle's state should be what the last match left)
  DevBuf keep;
  const size_t xbytes = (size_t)d.S * d.NP * 512;
  SSHIP_HIP_CHECK(keep.ensure(xbytes));
  SSHIP_HIP_CHECK(hipMemcpyAsync(keep.p, x, xbytes, hipMemcpyDeviceToDevice, s));
  SSHIP_HIP_CHECK(run());  // warm
  hipEvent_t e0, e1;
  SSHIP_HIP_CHECK(hipEventCreate(&e0));
  SSHIP_HIP_CHECK(hipEventCreate(&e1));
  SSHIP_HIP_CHECK(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) SSHIP_HIP_CHECK(run());
  SSHIP_HIP_CHECK(hipEventRecord(e1, s));
  SSHIP_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  SSHIP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  SSHIP_HIP_CHECK(hipMemcpyAsync(x, keep.p, xbytes, hipMemcpyDeviceToDevice, s));
  SSHIP_HIP_CHECK(hipStreamSynchronize(s));
  *avg_ms = ms / iters;
  return SSHIP_OK;
}

static int lg_match_common(sship_lg* lg, const float* kp0, int st0, int n0, const float* kp1, int st1, int n1,
                           int32_t* matches0, float* mscores0) {
  // kpts -> pinned [2, max_kp, 3] (x, y, 0) -> device; descriptors already staged in desc_stage.
  hipStream_t s = lg->stream;
  g_timer.begin_if_idle(s);
  float* hk = lg->h_kp.as<float>();
  const int mk = lg->max_kp;
  for (int i = 0; i < n0; ++i) { hk[3 * i] = kp0[(size_t)i * st0]; hk[3 * i + 1] = kp0[(size_t)i * st0 + 1]; hk[3 * i + 2] = 0.f; }
  for (int i = 0; i < n1; ++i) {
    hk[(size_t)mk * 3 + 3 * i] = kp1[(size_t)i * st1]; hk[(size_t)mk * 3 + 3 * i + 1] = kp1[(size_t)i * st1 + 1];
    hk[(size_t)mk * 3 + 3 * i + 2] = 0.f;
  }
  lg->h_lens.as<int>()[0] = n0; lg->h_lens.as<int>()[1] = n1;
  SSHIP_HIP_CHECK(hipMemcpyAsync(lg->kp_stage.p, hk, 2 * (size_t)mk * 12, hipMemcpyHostToDevice, s));
  SSHIP_HIP_CHECK(hipMemcpyAsync(lg->lens.p, lg->h_lens.p, 8, hipMemcpyHostToDevice, s));
  if (int rc = lg_forward(lg, lg->kp_stage.as<float>(), 3, mk * 3, lg->lens.as<int>(), lg->desc_stage.as<_Float16>(),
                          (size_t)mk * 256, 1, lg->m0.as<int32_t>(), lg->ms0.as<float>(), s))
    return rc;
  SSHIP_HIP_CHECK(hipMemcpyAsync(lg->h_m0.p, lg->m0.p, (size_t)n0 * 4, hipMemcpyDeviceToHost, s));
  SSHIP_HIP_CHECK(hipMemcpyAsync(lg->h_ms0.p, lg->ms0.p, (size_t)n0 * 4, hipMemcpyDeviceToHost, s));
  SSHIP_HIP_CHECK(hipStreamSynchronize(s));
  memcpy(matches0, lg->h_m0.p, (size_t)n0 * 4);
  memcpy(mscores0, lg->h_ms0.p, (size_t)n0 * 4);
  return SSHIP_OK;
}

extern "C" int sship_lg_match_device(sship_lg* lg, const float* kp0, int st0, int n0, const void* desc0, const float* kp1,
                                     int st1, int n1, const void* desc1, int32_t* matches0, float* mscores0) {
  bind_thread();
  if (!lg || !kp0 || !kp1 || !desc0 || !desc1 || !matches0 || !mscores0) return fail(SSHIP_ERR_INVALID, "lg_match_device: null argument");
  if (n0 <= 0 || n1 <= 0) return fail(SSHIP_ERR_INVALID, "lg_match_device: empty keypoint set");  // LightGlue.cc:381-385
  if (n0 > lg->max_kp || n1 > lg->max_kp || st0 < 2 || st1 < 2) return fail(SSHIP_ERR_INVALID, "lg_match_device: n exceeds max_keypoints");
  hipStream_t s = lg->stream;
  // D2D of exactly the slot bytes into the matcher's inputs (LightGlue.cc:425-441)
  SSHIP_HIP_CHECK(hipMemcpyAsync(lg->desc_stage.p, desc0, (size_t)n0 * 512, hipMemcpyDeviceToDevice, s));
  SSHIP_HIP_CHECK(hipMemcpyAsync(lg->desc_stage.as<_Float16>() + (size_t)lg->max_kp * 256, desc1, (size_t)n1 * 512,
                                 hipMemcpyDeviceToDevice, s));
  return lg_match_common(lg, kp0, st0, n0, kp1, st1, n1, matches0, mscores0);
}
extern "C" int sship_lg_match_host(sship_lg* lg, const float* kp0, int st0, int n0, const float* desc0, const float* kp1,
                                   int st1, int n1, const float* desc1, int32_t* matches0, float* mscores0) {
  bind_thread();
  if (!lg || !kp0 || !kp1 || !desc0 || !desc1 || !matches0 || !mscores0) return fail(SSHIP_ERR_INVALID, "lg_match_host: null argument");
  if (n0 <= 0 || n1 <= 0) return fail(SSHIP_ERR_INVALID, "lg_match_host: empty keypoint set");  // LightGlue.cc:294-295
  if (n0 > lg->max_kp || n1 > lg->max_kp || st0 < 2 || st1 < 2) return fail(SSHIP_ERR_INVALID, "lg_match_host: n exceeds max_keypoints");
  hipStream_t s = lg->stream;
  _Float16* hd = lg->h_desc.as<_Float16>();  // store_floats: CV_32F -> engine dtype on the host (LightGlue.cc:227-238)
  for (size_t i = 0; i < (size_t)n0 * 256; ++i) hd[i] = (_Float16)desc0[i];
  for (size_t i = 0; i < (size_t)n1 * 256; ++i) hd[(size_t)lg->max_kp * 256 + i] = (_Float16)desc1[i];
  SSHIP_HIP_CHECK(hipMemcpyAsync(lg->desc_stage.p, hd, (size_t)n0 * 512, hipMemcpyHostToDevice, s));
  SSHIP_HIP_CHECK(hipMemcpyAsync(lg->desc_stage.as<_Float16>() + (size_t)lg->max_kp * 256, hd + (size_t)lg->max_kp * 256,
                                 (size_t)n1 * 512, hipMemcpyHostToDevice, s));
  return lg_match_common(lg, kp0, st0, n0, kp1, st1, n1, matches0, mscores0);
}
extern "C" int sship_filter_matches(const int32_t* matches0, const float* mscores0, int n0, int* q, int* t, float* dist) {
  if (n0 <= 0) return 0;
  if (!matches0 || !q || !t || !dist) return -1;
  int k = 0;
  for (int i = 0; i < n0; ++i) {  // LightGlue.cc:351-361
    const int j = matches0[i];
    if (j < 0) continue;
    q[k] = i; t[k] = j; dist[k] = 1.0f - (mscores0 ? mscores0[i] : 1.0f);
    ++k;
  }
  return k;
}

// ====================================================================================================
// EigenPlaces place recogniser: include/EigenPlaces.h:19-40, src/EigenPlaces.cc:47-174
// ====================================================================================================
struct sship_ep {
  int in_w = 0, in_h = 0;
  hipStream_t stream = nullptr;
  ConvW stem;
  struct Block { ConvW c1, c2, ds; bool has_ds = false; int stride = 1; } blocks[8];
  float* fc_wt = nullptr;   // [in 512][out 512] fp32 (transposed)
  float* fc_b = nullptr;
  float gem_p = 3.f;
  DevBuf d_in, patches, act[4], d_out;   // patches: the im2col matrix of the GEMM stem (developer build's A/B path only)
  DevBuf stem_frag, stem_bias;           // fused stem (round 6): [2][11][64][8] fp16 A fragments, fp32 bias [64]
  PinBuf h_out;
  DevBuf d_ws, d_tail;      // split-K partial sums; the tail's partial GeM sums, [512] pre-normalisation outputs + its arrival counter
  DevBuf d_img;             // u8 entry points: the uploaded image
  PinBuf h_img;
  // resize tables of (src_h, src_w) -> (in_h, in_w), one IMMUTABLE device buffer per source size seen (a dataset has one; a rig a few).
  // A set is written once, before its first use, and never touched again: no call on any stream can observe a table changing under it, so
  // the asynchronous entry point needs no device-wide synchronisation (round 5 re-used one buffer behind a hipDeviceSynchronize).
  struct Tables { int h = 0, w = 0; DevBuf d; };
  std::vector<std::unique_ptr<Tables>> tables;
};
extern "C" void sship_ep_destroy(sship_ep* ep) {
  bind_thread();
  if (!ep) return;
  (void)hipDeviceSynchronize();
  free_conv(ep->stem);
  for (auto& b : ep->blocks) { free_conv(b.c1); free_conv(b.c2); free_conv(b.ds); }
  if (ep->fc_wt) (void)hipFree(ep->fc_wt);
  if (ep->fc_b) (void)hipFree(ep->fc_b);
  if (ep->stream) (void)hipStreamDestroy(ep->stream);
  delete ep;
}
extern "C" int sship_ep_create(const char* weights_path, int input_w, int input_h, sship_ep** out) {
  bind_thread();
  if (!weights_path || !out || input_w < 32 || input_h < 32) return fail(SSHIP_ERR_INVALID, "ep_create: bad arguments");
  if (int rc = require_device()) return rc;
  StateDict sd; std::string err;
  if (!load_safetensors(weights_path, sd, err)) return fail(SSHIP_ERR_IO, err);
  std::unique_ptr<sship_ep, void (*)(sship_ep*)> ep(new sship_ep(), sship_ep_destroy);
  ep->in_w = input_w; ep->in_h = input_h;
  // conv + BatchNorm (eval) folded: w' = w * g / sqrt(var + 1e-5), b' = beta - mean * g / sqrt(var + 1e-5)
  auto folded = [&](const std::string& conv, const std::string& bn, int cout, int cin, int ks, int cin_pad, ConvW& dst) -> int {
    const Tensor* w = find_tensor(sd, conv + ".weight", {cout, cin, ks, ks}, err);
    const Tensor* g = w ? find_tensor(sd, bn + ".weight", {cout}, err) : nullptr;
    const Tensor* be = g ? find_tensor(sd, bn + ".bias", {cout}, err) : nullptr;
    const Tensor* mu = be ? find_tensor(sd, bn + ".running_mean", {cout}, err) : nullptr;
    const Tensor* var = mu ? find_tensor(sd, bn + ".running_var", {cout}, err) : nullptr;
    if (!var) return fail(SSHIP_ERR_IO, err);
    const int kk = ks * ks;
    const bool flat = cin_pad != cin;  // the stem: [cout][cin*ks*ks] rows zero-padded to cin_pad, run as a 1x1 GEMM
    std::vector<float> wf((size_t)cout * (flat ? cin_pad : cin * kk), 0.f), bf(cout);
    for (int co = 0; co < cout; ++co) {
      const float sc = g->data[co] / sqrtf(var->data[co] + 1e-5f);
      bf[co] = be->data[co] - mu->data[co] * sc;
      for (int i = 0; i < cin * kk; ++i) wf[(size_t)co * (flat ? cin_pad : cin * kk) + i] = w->data[(size_t)co * cin * kk + i] * sc;
    }
    if (flat) {
      // the fused stem kernel's A fragments (ep_kernels.hip: k_ep_stem_pool): fragment (m, s), lane (row, kg) = W'[32 m + row][idx = 2 s + kg][kx 0..7],
      // idx = c * 7 + ky; idx = 21 and kx = 7 are zero padding of K
      std::vector<_Float16> fr((size_t)2 * 11 * 64 * 8);
      for (int m = 0; m < 2; ++m)
        for (int st = 0; st < 11; ++st)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
              const int row = lane & 31, idx = 2 * st + (lane >> 5);
              float v = 0.f;
              if (idx < cin * ks && e < ks) v = wf[(size_t)(32 * m + row) * cin_pad + (idx / ks) * kk + (idx % ks) * ks + e];
              fr[((size_t)(m * 11 + st) * 64 + lane) * 8 + e] = (_Float16)v;
            }
      SSHIP_HIP_CHECK(ep->stem_frag.ensure(fr.size() * sizeof(_Float16)));
      SSHIP_HIP_CHECK(hipMemcpy(ep->stem_frag.p, fr.data(), fr.size() * sizeof(_Float16), hipMemcpyHostToDevice));
      SSHIP_HIP_CHECK(ep->stem_bias.ensure(bf.size() * sizeof(float)));
      SSHIP_HIP_CHECK(hipMemcpy(ep->stem_bias.p, bf.data(), bf.size() * sizeof(float), hipMemcpyHostToDevice));
      if (!SSHIP_DEV_SWITCHES) return SSHIP_OK;   // the GEMM form of the stem exists in the developer build only (A/B: SUPERSLAM_HIP_EP_STEM=gemm)
    }
    return upload_conv(wf.data(), bf.data(), cout, flat ? cin_pad : cin, flat ? 1 : ks, 64, dst);
  };
  if (int rc = folded("backbone.0", "backbone.1", 64, 3, 7, 192, ep->stem)) return rc;
  const int planes[4] = {64, 128, 256, 512};
  int cin = 64;
  for (int L = 0; L < 4; ++L)
    for (int b = 0; b < 2; ++b) {
      sship_ep::Block& blk = ep->blocks[L * 2 + b];
      const std::string p = "backbone." + std::to_string(4 + L) + "." + std::to_string(b);
      const int c_in = b == 0 ? cin : planes[L];
      blk.stride = (b == 0 && L > 0) ? 2 : 1;
      if (int rc = folded(p + ".conv1", p + ".bn1", planes[L], c_in, 3, c_in, blk.c1)) return rc;
      if (int rc = folded(p + ".conv2", p + ".bn2", planes[L], planes[L], 3, planes[L], blk.c2)) return rc;
      if (b == 0 && (blk.stride != 1 || c_in != planes[L])) {
        blk.has_ds = true;
        if (int rc = folded(p + ".downsample.0", p + ".downsample.1", planes[L], c_in, 1, c_in, blk.ds)) return rc;
      }
      if (b == 1) cin = planes[L];
    }
  {
    const Tensor* gp = find_tensor(sd, "aggregation.1.p", {1}, err);
    const Tensor* w = gp ? find_tensor(sd, "aggregation.3.weight", {512, 512}, err) : nullptr;
    const Tensor* bb = w ? find_tensor(sd, "aggregation.3.bias", {512}, err) : nullptr;
    if (!bb) return fail(SSHIP_ERR_IO, err);
    ep->gem_p = gp->data[0];
    std::vector<float> wt((size_t)512 * 512);
    for (int j = 0; j < 512; ++j)
      for (int c = 0; c < 512; ++c) wt[(size_t)c * 512 + j] = w->data[(size_t)j * 512 + c];
    if (int rc = upload_floats(wt.data(), wt.size(), &ep->fc_wt)) return rc;
    if (int rc = upload_floats(bb->data.data(), 512, &ep->fc_b)) return rc;
  }
  const int Ho = (input_h - 1) / 2 + 1, Wo = (input_w - 1) / 2 + 1;
  SSHIP_HIP_CHECK(ep->d_in.ensure((size_t)3 * input_h * input_w * 4));
  if (SSHIP_DEV_SWITCHES) SSHIP_HIP_CHECK(ep->patches.ensure((size_t)Ho * Wo * 192 * 2));
  for (auto& a : ep->act) SSHIP_HIP_CHECK(a.ensure((size_t)Ho * Wo * 64 * 2));
  SSHIP_HIP_CHECK(ep->d_out.ensure(512 * 4));
  SSHIP_HIP_CHECK(ep->h_out.ensure(512 * 4));
  SSHIP_HIP_CHECK(ep->d_ws.ensure(ep_splitk_workspace_bytes(input_h, input_w)));
  SSHIP_HIP_CHECK(ep->d_tail.ensure((ep_tail_ws_floats() + 2) * 4));
  SSHIP_HIP_CHECK(hipMemset(ep->d_tail.p, 0, (ep_tail_ws_floats() + 2) * 4));
  SSHIP_HIP_CHECK(hipStreamCreateWithFlags(&ep->stream, hipStreamDefault));
  *out = ep.release();
  return SSHIP_OK;
}
extern "C" int sship_ep_descriptor_dim(const sship_ep* ep) { return ep ? 512 : 0; }
// host half of compute_global_descriptor (no GPU involved): include/superslam_hip/place_recognizer.hpp
extern "C" int sship_ep_preprocess(const uint8_t* img, int h, int w, int stride, int channels, int input_w, int input_h, float* chw_out) {
  if (!img || !chw_out || h <= 0 || w <= 0 || input_w <= 0 || input_h <= 0 || (channels != 1 && channels != 3) || stride < w * channels)
    return fail(SSHIP_ERR_INVALID, "ep_preprocess: bad arguments");
  superslam_hip::eigenplaces_preprocess(superslam_hip::Image{img, h, w, channels, stride}, input_w, input_h, chw_out);
  return SSHIP_OK;
}
// the network from the preprocessed fp32 [3, H, W] tensor in d_in to the descriptor in `desc_dev`, on stream s
static int ep_network(sship_ep* ep, float* desc_dev, hipStream_t s) {
  const int H = ep->in_h, W = ep->in_w;
  int h = (H - 1) / 2 + 1, w = (W - 1) / 2 + 1;
  _Float16* a[4] = {ep->act[0].as<_Float16>(), ep->act[1].as<_Float16>(), ep->act[2].as<_Float16>(), ep->act[3].as<_Float16>()};
  const int hp = (h - 1) / 2 + 1, wp = (w - 1) / 2 + 1;  // MaxPool2d(3, 2, 1)
#if SSHIP_DEV_SWITCHES  // A/B: the stem of rounds 3-5 (im2col -> 1x1 GEMM -> max-pool, three launches)
  static const bool stem_gemm = [] { const char* e = dev_env("SUPERSLAM_HIP_EP_STEM"); return e && std::string(e) == "gemm"; }();
  if (stem_gemm) {
    launch_ep_im2col(ep->d_in.as<float>(), H, W, h, w, ep->patches.as<_Float16>(), s);
    SSHIP_HIP_CHECK(ep_conv(ep->stem, ep->patches.as<_Float16>(), a[0], nullptr, h, w, true, false, s));
    launch_ep_maxpool(a[0], h, w, hp, wp, a[1], s);
  } else
#endif
  launch_ep_stem_pool(ep->d_in.as<float>(), H, W, h, w, hp, wp, ep->stem_frag.as<_Float16>(), ep->stem_bias.as<float>(), a[1], s);
  h = hp; w = wp;
  int cur = 1;  // index of the buffer holding the block input
  for (const auto& blk : ep->blocks) {
    int f[3], k = 0;
    for (int i = 0; i < 4; ++i) if (i != cur) f[k++] = i;  // three free buffers: t, ds, out
    const _Float16* res = a[cur];
    int ho = h, wo = w;
    if (blk.stride == 2) { ho = (h + 1) / 2; wo = (w + 1) / 2; }
    float* ws = ep->d_ws.as<float>();
    const size_t wsb = ep->d_ws.bytes;
    SSHIP_HIP_CHECK(ep_conv(blk.c1, a[cur], a[f[0]], nullptr, h, w, true, blk.stride == 2, s, ws, wsb));
    if (blk.has_ds) {
      SSHIP_HIP_CHECK(ep_conv(blk.ds, a[cur], a[f[1]], nullptr, h, w, false, blk.stride == 2, s));
      res = a[f[1]];
    }
    SSHIP_HIP_CHECK(ep_conv(blk.c2, a[f[0]], a[f[2]], res, ho, wo, true, false, s, ws, wsb));
    cur = f[2]; h = ho; w = wo;
  }
  launch_ep_tail(a[cur], h * w, ep->gem_p, ep->fc_wt, ep->fc_b, ep->d_tail.as<float>(), ep->d_tail.as<int>() + ep_tail_ws_floats(), desc_dev, s);
  SSHIP_HIP_CHECK(hipGetLastError());
  return SSHIP_OK;
}
extern "C" int sship_ep_infer(sship_ep* ep, const float* chw_host, float* desc_out) {
  bind_thread();
  if (!ep || !chw_host || !desc_out) return fail(SSHIP_ERR_INVALID, "ep_infer: null argument");
  hipStream_t s = ep->stream;
  SSHIP_HIP_CHECK(hipMemcpyAsync(ep->d_in.p, chw_host, (size_t)3 * ep->in_h * ep->in_w * 4, hipMemcpyHostToDevice, s));
  if (int rc = ep_network(ep, ep->d_out.as<float>(), s)) return rc;
  SSHIP_HIP_CHECK(hipMemcpyAsync(ep->h_out.p, ep->d_out.p, 512 * 4, hipMemcpyDeviceToHost, s));
  SSHIP_HIP_CHECK(hipStreamSynchronize(s));
  memcpy(desc_out, ep->h_out.p, 512 * 4);
  return SSHIP_OK;
}
// resize tables of (h, w) -> (in_h, in_w).  The first call with a new source size allocates and uploads its set (hipMalloc + a blocking copy of
// ~ 8 (in_w + in_h) ints: not legal under stream capture - run one call per source size before capturing); every later call only looks it up.
constexpr size_t kEpMaxTableSets = 16;
static int ep_tables(sship_ep* ep, int h, int w, const int** tab_out) {
  for (const auto& t : ep->tables)
    if (t->h == h && t->w == w) { *tab_out = t->d.as<int>(); return SSHIP_OK; }
  std::vector<int> t[8];
  superslam_hip::resize_bilinear_coeffs(ep->in_w, w, t[0], t[1], t[2], t[3]);
  superslam_hip::resize_bilinear_coeffs(ep->in_h, h, t[4], t[5], t[6], t[7]);
  std::vector<int> flat;
  for (auto& v : t) flat.insert(flat.end(), v.begin(), v.end());
  if (ep->tables.size() >= kEpMaxTableSets) {  // a stream of ever-changing sizes: drop the oldest set once nothing can still be reading it
    SSHIP_HIP_CHECK(hipDeviceSynchronize());
    ep->tables.erase(ep->tables.begin());
  }
  auto set = std::make_unique<sship_ep::Tables>();
  SSHIP_HIP_CHECK(set->d.ensure(flat.size() * 4));
  SSHIP_HIP_CHECK(hipMemcpy(set->d.p, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));  // a fresh buffer nobody reads yet
  set->h = h; set->w = w;
  *tab_out = set->d.as<int>();
  ep->tables.push_back(std::move(set));
  return SSHIP_OK;
}
static int ep_check_image(const void* img, int h, int w, int stride, int channels, const char* who) {
  if (!img || h <= 0 || w <= 0 || (channels != 1 && channels != 3) || stride < w * channels || h > 16384 || w > 16384)
    return fail(SSHIP_ERR_INVALID, std::string(who) + ": bad image arguments");
  return SSHIP_OK;
}
extern "C" int sship_ep_infer_u8_device(sship_ep* ep, const uint8_t* img_dev, int h, int w, int stride, int channels, float* desc_out_dev,
                                        void* stream) {
  bind_thread();
  if (!ep || !desc_out_dev) return fail(SSHIP_ERR_INVALID, "ep_infer_u8_device: null argument");
  if (int rc = ep_check_image(img_dev, h, w, stride, channels, "ep_infer_u8_device")) return rc;
  const int* tab = nullptr;
  if (int rc = ep_tables(ep, h, w, &tab)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  launch_ep_resize_norm(img_dev, stride, channels, tab, ep->in_w, ep->in_h, ep->d_in.as<float>(), s);
  return ep_network(ep, desc_out_dev, s);
}
extern "C" int sship_ep_infer_u8(sship_ep* ep, const uint8_t* img, int h, int w, int stride, int channels, float* desc_out) {
  bind_thread();
  if (!ep || !desc_out) return fail(SSHIP_ERR_INVALID, "ep_infer_u8: null argument");
  if (int rc = ep_check_image(img, h, w, stride, channels, "ep_infer_u8")) return rc;
  hipStream_t s = ep->stream;
  const size_t row = (size_t)w * channels, bytes = row * h;
  SSHIP_HIP_CHECK(ep->h_img.ensure(bytes));
  SSHIP_HIP_CHECK(ep->d_img.ensure(bytes));
  for (int y = 0; y < h; ++y) memcpy(ep->h_img.as<uint8_t>() + (size_t)y * row, img + (size_t)y * stride, row);  // pinned, dense rows
  SSHIP_HIP_CHECK(hipMemcpyAsync(ep->d_img.p, ep->h_img.p, bytes, hipMemcpyHostToDevice, s));
  if (int rc = sship_ep_infer_u8_device(ep, ep->d_img.as<uint8_t>(), h, w, (int)row, channels, ep->d_out.as<float>(), s)) return rc;
  SSHIP_HIP_CHECK(hipMemcpyAsync(ep->h_out.p, ep->d_out.p, 512 * 4, hipMemcpyDeviceToHost, s));
  SSHIP_HIP_CHECK(hipStreamSynchronize(s));
  memcpy(desc_out, ep->h_out.p, 512 * 4);
  return SSHIP_OK;
}
extern "C" int sship_ep_bench(sship_ep* ep, const uint8_t* img_dev, int h, int w, int stride, int channels, int iters, float* avg_ms) {
  bind_thread();
  if (!ep || !avg_ms || iters <= 0) return fail(SSHIP_ERR_INVALID, "ep_bench: bad arguments");
  if (int rc = ep_check_image(img_dev, h, w, stride, channels, "ep_bench")) return rc;
  hipStream_t s = ep->stream;
  if (int rc = sship_ep_infer_u8_device(ep, img_dev, h, w, stride, channels, ep->d_out.as<float>(), s)) return rc;  // warm (also uploads the tables)
  hipEvent_t e0 = nullptr, e1 = nullptr;
  auto drop = [&](int rc) { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); return rc; };
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return drop(fail(SSHIP_ERR_HIP, "ep_bench: hipEventCreate failed"));
  if (hipEventRecord(e0, s) != hipSuccess) return drop(fail(SSHIP_ERR_HIP, "ep_bench: hipEventRecord failed"));
  for (int i = 0; i < iters; ++i)
    if (int rc = sship_ep_infer_u8_device(ep, img_dev, h, w, stride, channels, ep->d_out.as<float>(), s)) return drop(rc);
  float ms = 0.f;
  if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
    return drop(fail(SSHIP_ERR_HIP, "ep_bench: timing failed"));
  *avg_ms = ms / iters;
  return drop(SSHIP_OK);
}

// ====================================================================================================
// fused front-end step: SuperPoint(batch 2P) + select + gather + LightGlue(P)
// ====================================================================================================
extern "C" int sship_frontend_batch_device(sship_sp* sp, sship_lg* lg, const uint8_t* imgs, int pairs, int h, int w,
                                           void* desc_out, float* kp_out, int* n_out, int32_t* m0, float* ms0,
                                           void* stream) {
  bind_thread();
  if (!sp || !lg || !imgs || !desc_out || !kp_out || !n_out || !m0 || !ms0) return fail(SSHIP_ERR_INVALID, "frontend_batch_device: null argument");
  if (pairs <= 0 || pairs > lg->max_pairs) return fail(SSHIP_ERR_INVALID, "frontend_batch_device: pairs exceeds the matcher's max_pairs");
  if (sp->cfg.max_keypoints != lg->max_kp) return fail(SSHIP_ERR_INVALID, "frontend_batch_device: extractor and matcher disagree on max_keypoints");
  hipStream_t s = static_cast<hipStream_t>(stream);  // one stream for the extractor and the matcher; NULL = legacy default stream
  if (int rc = sship_sp_extract_batch_device(sp, imgs, 2 * pairs, h, w, desc_out, kp_out, n_out, s)) return rc;
  return lg_forward(lg, kp_out, 3, lg->max_kp * 3, n_out, static_cast<const _Float16*>(desc_out), (size_t)lg->max_kp * 256,
                    pairs, m0, ms0, s);
}
