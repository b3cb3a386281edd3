// shard_rccl.hip - the ONE exchange step of the sharded front-end behind the C ABI (BASELINE configs 3 and 5):
// a fixed-stride all-gather of (descriptors, keypoints, counts) over RCCL (xGMI between the GPUs of one node).
//
// The reference is single-GPU and has no counterpart; what the gathered tensors feed is its DescriptorPool
// (include/DescriptorPool.h:13-91: [count, 256] fp16 rows per frame) and the keypoint vectors of include/InferenceInterfaces.h:13-24.
// Everything else on the path is embarrassingly parallel - this file is the only place a collective exists.
//
// RCCL is bound at RUN TIME (dlsym on the process first, then dlopen("librccl.so.1")): a process that already carries an RCCL -
// a Python host under torch.distributed, whose torch/lib/librccl.so is loaded - must not get a second copy of the library next
// to it, and single-GPU users of libsuperslam_hip.so need no RCCL at all.  SSHIP_RCCL_LIBRARY=<path> (include/sship.h, "Environment")
// names the one library to bind instead.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>

#include <mutex>
#include <string>

#include "../../include/sship.h"
#include "common.h"

namespace {

// the slice of rccl.h this file uses (rccl.h:40-43,187,220,260,339,678,923; ABI-stable since NCCL 2.x)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { kNcclSuccess = 0, kNcclInt8 = 0, kNcclInt32 = 2, kNcclHalf = 6, kNcclFloat = 7 };

struct Rccl {
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = RTLD_DEFAULT;
    const char* forced = getenv("SSHIP_RCCL_LIBRARY");
    if (forced && *forced) {
      h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
      if (!h) { const char* e = dlerror(); r.why = std::string("RCCL not found (SSHIP_RCCL_LIBRARY=") + forced + "): " + (e ? e : ""); return; }
    } else if (!dlsym(RTLD_DEFAULT, "ncclAllGather")) {
      h = nullptr;
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
      if (!h) { const char* e = dlerror(); r.why = std::string("RCCL not found (dlopen librccl.so.1): ") + (e ? e : ""); return; }  // dlerror() clears itself: ONE call
    }
    auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) r.why = std::string("RCCL symbol missing: ") + n; return p; };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.ok = r.why.empty();
  });
  return r;
}

int fail(int code, const std::string& msg) {
  sship::set_error(msg);
  return code;
}
int nccl_fail(const char* what, int rc) {
  Rccl& r = rccl();
  return fail(SSHIP_ERR_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}

}  // namespace

struct sship_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

extern "C" int sship_comm_unique_id(void* id_out_128) {
  if (!id_out_128) return fail(SSHIP_ERR_INVALID, "comm_unique_id: null argument");
  Rccl& r = rccl();
  if (!r.ok) return fail(SSHIP_ERR_NO_DEVICE, r.why);
  ncclUniqueId id;
  if (int rc = r.GetUniqueId(&id)) return nccl_fail("ncclGetUniqueId", rc);
  memcpy(id_out_128, &id, sizeof id);
  return SSHIP_OK;
}

extern "C" int sship_comm_create(const void* id_128, int rank, int world, sship_comm** out) {
  if (!id_128 || !out || world < 1 || rank < 0 || rank >= world) return fail(SSHIP_ERR_INVALID, "comm_create: bad arguments");
  Rccl& r = rccl();
  if (!r.ok) return fail(SSHIP_ERR_NO_DEVICE, r.why);
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return fail(SSHIP_ERR_NO_DEVICE, "comm_create: no HIP device bound to this thread (call sship_init first)");
  ncclUniqueId id;
  memcpy(&id, id_128, sizeof id);
  auto* c = new sship_comm();
  c->rank = rank; c->world = world; c->device = dev;
  if (int rc = r.CommInitRank(&c->comm, world, id, rank)) { delete c; return nccl_fail("ncclCommInitRank", rc); }
  *out = c;
  return SSHIP_OK;
}

extern "C" void sship_comm_destroy(sship_comm* c) {
  if (!c) return;
  if (c->comm && rccl().ok) {
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(c->device);  // the calling thread may be bound elsewhere (handles are used from more than one thread)
    (void)rccl().CommDestroy(c->comm);
    if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);  // ... and stays bound where it was
  }
  delete c;
}
extern "C" int sship_comm_rank(const sship_comm* c) { return c ? c->rank : -1; }
extern "C" int sship_comm_world(const sship_comm* c) { return c ? c->world : 0; }

extern "C" int sship_gather_features_rccl(sship_comm* c, const void* desc_local_dev, const float* kp_local_dev, const int* n_local_dev,
                                          int units_per_rank, int max_keypoints, void* desc_all_dev, float* kp_all_dev, int* n_all_dev,
                                          void* stream) {
  if (!c || !c->comm) return fail(SSHIP_ERR_INVALID, "gather_features_rccl: no communicator");
  if (units_per_rank < 0 || max_keypoints <= 0) return fail(SSHIP_ERR_INVALID, "gather_features_rccl: bad sizes");
  if (units_per_rank == 0) return SSHIP_OK;
  if (!desc_local_dev || !kp_local_dev || !n_local_dev || !desc_all_dev || !kp_all_dev || !n_all_dev)
    return fail(SSHIP_ERR_INVALID, "gather_features_rccl: null buffer");
  Rccl& r = rccl();
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(c->device) != hipSuccess) return fail(SSHIP_ERR_NO_DEVICE, "gather_features_rccl: cannot bind the communicator's device");
  struct Rebind { int prev, dev; ~Rebind() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); } } rebind{prev, c->device};  // the caller's device binding survives the call
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t u = (size_t)units_per_rank, k = (size_t)max_keypoints;
  // one grouped step: RCCL fuses the three all-gathers into a single launch / a single pass over the xGMI links
  if (int rc = r.GroupStart()) return nccl_fail("ncclGroupStart", rc);
  int rc1 = r.AllGather(desc_local_dev, desc_all_dev, u * k * SSHIP_DESC_DIM, kNcclHalf, c->comm, s);
  int rc2 = r.AllGather(kp_local_dev, kp_all_dev, u * k * 3, kNcclFloat, c->comm, s);
  int rc3 = r.AllGather(n_local_dev, n_all_dev, u, kNcclInt32, c->comm, s);
  int rce = r.GroupEnd();
  if (rc1) return nccl_fail("ncclAllGather(desc)", rc1);
  if (rc2) return nccl_fail("ncclAllGather(kp)", rc2);
  if (rc3) return nccl_fail("ncclAllGather(n)", rc3);
  if (rce) return nccl_fail("ncclGroupEnd", rce);
  return SSHIP_OK;
}
