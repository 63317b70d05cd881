// Section 7 of the C ABI: the gather of detections over RCCL. Host code only: two ncclAllGather calls on
// KB-sized device buffers. librccl is resolved at run time (dlopen), preferring a copy the process has already loaded,
// so that the library neither links RCCL nor brings a second copy into a PyTorch process.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "cc_internal.h"

using namespace ccamd;

#define CC_HIP(expr)                                                                                         \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess) return set_error(CC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                           __FILE__, __LINE__);                                              \
  } while (0)

namespace {

struct UniqueId {  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value to ncclCommInitRank
  char internal[CC_COMM_ID_BYTES];
};
constexpr int kNcclInt32 = 2;  // ncclDataType_t::ncclInt32

struct RcclApi {
  void* lib = nullptr;
  int (*get_unique_id)(UniqueId*) = nullptr;
  int (*comm_init_rank)(void**, int, UniqueId, int) = nullptr;
  int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*comm_destroy)(void*) = nullptr;
  const char* (*error_string)(int) = nullptr;
  bool ok() const { return get_unique_id && comm_init_rank && all_gather && comm_destroy && error_string; }
};

const RcclApi& rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {  // an RCCL that is already in the process (PyTorch's) wins
      api.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (api.lib) break;
    }
    if (!api.lib)
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
      }
    if (!api.lib) return;
    auto sym = [&](const char* n) { return dlsym(api.lib, n); };
    api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(sym("ncclGetUniqueId"));
    api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(sym("ncclCommInitRank"));
    api.all_gather = reinterpret_cast<decltype(api.all_gather)>(sym("ncclAllGather"));
    api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(sym("ncclCommDestroy"));
    api.error_string = reinterpret_cast<decltype(api.error_string)>(sym("ncclGetErrorString"));
  });
  return api;
}

cc_status need_rccl(const char* who) {
  const RcclApi& r = rccl_api();
  if (!r.ok()) return set_error(CC_ERR_UNSUPPORTED, "%s: librccl is not available (%s)", who, r.lib ? "missing symbols" : "dlopen failed");
  return CC_OK;
}

}  // namespace

struct cc_comm {
  int device = 0, rank = 0, world = 1;
  void* nccl = nullptr;
  hipStream_t stream = nullptr;
  int32_t* d_send = nullptr;
  int32_t* d_recv = nullptr;
  size_t send_cap = 0, recv_cap = 0;  // int32 entries
  std::vector<int32_t> last_offsets;  // result of the last gather (cc_gather_fetch)
  std::vector<cc_rect> last_rects;

  cc_status ensure(size_t send, size_t recv) {
    if (send > send_cap) {
      if (d_send) (void)hipFree(d_send);
      d_send = nullptr;
      send_cap = 0;
      CC_HIP(hipMalloc(reinterpret_cast<void**>(&d_send), send * sizeof(int32_t)));
      send_cap = send;
    }
    if (recv > recv_cap) {
      if (d_recv) (void)hipFree(d_recv);
      d_recv = nullptr;
      recv_cap = 0;
      CC_HIP(hipMalloc(reinterpret_cast<void**>(&d_recv), recv * sizeof(int32_t)));
      recv_cap = recv;
    }
    return CC_OK;
  }
  // all ranks contribute `count` int32 from `h_send`; h_recv receives world * count
  cc_status all_gather_i32(const int32_t* h_send, size_t count, int32_t* h_recv) {
    cc_status st = ensure(count, count * (size_t)world);
    if (st != CC_OK) return st;
    CC_HIP(hipMemcpyAsync(d_send, h_send, count * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    const int rc = rccl_api().all_gather(d_send, d_recv, count, kNcclInt32, nccl, stream);
    if (rc != 0) return set_error(CC_ERR_HIP, "ncclAllGather failed: %s", rccl_api().error_string(rc));
    CC_HIP(hipMemcpyAsync(h_recv, d_recv, count * (size_t)world * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    CC_HIP(hipStreamSynchronize(stream));
    return CC_OK;
  }
};

extern "C" {

void cc_shard_range(int n_items, int rank, int world, int* lo, int* hi) {
  // contiguous blocks in rank order; the first n_items % world ranks hold one item more
  if (world < 1) world = 1;
  const int base = n_items / world, rem = n_items % world;
  const int a = rank * base + std::min(rank, rem);
  if (lo) *lo = a;
  if (hi) *hi = a + base + (rank < rem ? 1 : 0);
}

cc_status cc_comm_unique_id(void* id) {
  if (!id) return set_error(CC_ERR_INVALID_ARG, "cc_comm_unique_id: null buffer");
  cc_status st = need_rccl("cc_comm_unique_id");
  if (st != CC_OK) return st;
  UniqueId u;
  const int rc = rccl_api().get_unique_id(&u);
  if (rc != 0) return set_error(CC_ERR_HIP, "ncclGetUniqueId failed: %s", rccl_api().error_string(rc));
  std::memcpy(id, u.internal, CC_COMM_ID_BYTES);
  return CC_OK;
}

cc_status cc_comm_create(int device, int rank, int world, const void* id, cc_comm** out) {
  if (!out) return set_error(CC_ERR_INVALID_ARG, "cc_comm_create: null output");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return set_error(CC_ERR_INVALID_ARG, "cc_comm_create: rank %d of %d", rank, world);
  if (world > 1 && !id) return set_error(CC_ERR_INVALID_ARG, "cc_comm_create: a unique id is required for more than one rank");
  cc_comm* c = new cc_comm;
  c->device = device;
  c->rank = rank;
  c->world = world;
  if (world > 1) {
    cc_status st = need_rccl("cc_comm_create");
    if (st != CC_OK) {
      delete c;
      return st;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
      delete c;
      return set_error(CC_ERR_NO_DEVICE, "cc_comm_create: no usable HIP device");
    }
    if (device < 0 || device >= n) {
      delete c;
      return set_error(CC_ERR_INVALID_ARG, "cc_comm_create: device %d out of range (devices: %d)", device, n);
    }
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      return set_error(CC_ERR_HIP, "cc_comm_create: %s", hipGetErrorString(e));
    }
    UniqueId u;
    std::memcpy(u.internal, id, CC_COMM_ID_BYTES);
    const int rc = rccl_api().comm_init_rank(&c->nccl, world, u, rank);
    if (rc != 0) {
      (void)hipStreamDestroy(c->stream);
      delete c;
      return set_error(CC_ERR_HIP, "ncclCommInitRank failed: %s", rccl_api().error_string(rc));
    }
  }
  *out = c;
  return CC_OK;
}

void cc_comm_destroy(cc_comm* c) {
  if (!c) return;
  if (c->world > 1) (void)hipSetDevice(c->device);
  if (c->nccl) (void)rccl_api().comm_destroy(c->nccl);
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int cc_comm_rank(const cc_comm* c) { return c ? c->rank : (int)set_error(CC_ERR_INVALID_ARG, "cc_comm_rank: null communicator"); }
int cc_comm_world(const cc_comm* c) { return c ? c->world : (int)set_error(CC_ERR_INVALID_ARG, "cc_comm_world: null communicator"); }

cc_status cc_gather_detections(cc_comm* c, const cc_rect* rects, const int32_t* offsets, int n_frames, cc_rect* out, int cap_rects,
                               int32_t* offsets_out, int cap_frames, int* n_frames_all, int* n_rects_all) {
  if (!c || n_frames < 0 || (n_frames > 0 && !offsets) || !n_frames_all || !n_rects_all)
    return set_error(CC_ERR_INVALID_ARG, "cc_gather_detections: null argument");
  const int n_rects = n_frames > 0 ? offsets[n_frames] - offsets[0] : 0;
  if (n_rects < 0 || (n_rects > 0 && !rects)) return set_error(CC_ERR_INVALID_ARG, "cc_gather_detections: bad offsets");
  for (int f = 0; f < n_frames; f++)
    if (offsets[f + 1] < offsets[f]) return set_error(CC_ERR_INVALID_ARG, "cc_gather_detections: offsets must not decrease");
  const int world = c->world;
  // every rank's {frames, rectangles}
  std::vector<int32_t> headers((size_t)world * 2);
  const int32_t mine[2] = {n_frames, n_rects};
  if (world == 1) {
    headers[0] = mine[0];
    headers[1] = mine[1];
  } else {
    CC_HIP(hipSetDevice(c->device));
    cc_status st = c->all_gather_i32(mine, 2, headers.data());
    if (st != CC_OK) return st;
  }
  long long tot_f = 0, tot_r = 0;
  size_t max_len = 1;
  for (int r = 0; r < world; r++) {
    if (headers[(size_t)r * 2] < 0 || headers[(size_t)r * 2 + 1] < 0) return set_error(CC_ERR_HIP, "cc_gather_detections: corrupt header from rank %d", r);
    tot_f += headers[(size_t)r * 2];
    tot_r += headers[(size_t)r * 2 + 1];
    max_len = std::max(max_len, (size_t)headers[(size_t)r * 2] + 4 * (size_t)headers[(size_t)r * 2 + 1]);
  }
  // payload: per-frame counts, then the rectangles; padded to the longest rank's length
  std::vector<int32_t> payload(max_len, 0), all;
  for (int f = 0; f < n_frames; f++) payload[(size_t)f] = offsets[f + 1] - offsets[f];
  if (n_rects > 0) std::memcpy(payload.data() + n_frames, rects + offsets[0], (size_t)n_rects * sizeof(cc_rect));
  const int32_t* gathered = payload.data();
  if (world > 1) {
    all.resize(max_len * (size_t)world);
    cc_status st = c->all_gather_i32(payload.data(), max_len, all.data());
    if (st != CC_OK) return st;
    gathered = all.data();
  }
  // unpack into the communicator (kept for cc_gather_fetch), then hand out what fits
  c->last_offsets.assign((size_t)tot_f + 1, 0);
  c->last_rects.resize((size_t)tot_r);
  int32_t fo = 0, ro = 0;
  for (int r = 0; r < world; r++) {
    const int32_t* p = gathered + (size_t)r * max_len;
    const int nf = headers[(size_t)r * 2], nr = headers[(size_t)r * 2 + 1];
    long long sum = 0;
    for (int f = 0; f < nf; f++) {
      if (p[f] < 0) return set_error(CC_ERR_HIP, "cc_gather_detections: corrupt payload from rank %d", r);
      sum += p[f];
      c->last_offsets[(size_t)(fo + f + 1)] = ro + (int32_t)sum;
    }
    if (sum != nr) return set_error(CC_ERR_HIP, "cc_gather_detections: payload of rank %d does not match its header", r);
    if (nr > 0) std::memcpy(c->last_rects.data() + ro, p + nf, (size_t)nr * sizeof(cc_rect));
    fo += nf;
    ro += nr;
  }
  *n_frames_all = (int)tot_f;
  *n_rects_all = (int)tot_r;
  return cc_gather_fetch(c, out, cap_rects, offsets_out, cap_frames);
}

cc_status cc_gather_fetch(const cc_comm* c, cc_rect* out, int cap_rects, int32_t* offsets_out, int cap_frames) {
  if (!c) return set_error(CC_ERR_INVALID_ARG, "cc_gather_fetch: null communicator");
  if (c->last_offsets.empty()) return set_error(CC_ERR_INVALID_ARG, "cc_gather_fetch: no gather has completed on this communicator");
  const size_t tot_f = c->last_offsets.size() - 1, tot_r = c->last_rects.size();
  if (tot_f > (size_t)std::max(cap_frames, 0) || tot_r > (size_t)std::max(cap_rects, 0) || !offsets_out || (tot_r > 0 && !out))
    return set_error(CC_ERR_BUFFER_TOO_SMALL, "cc_gather_detections: %zu frames / %zu rectangles in total, room for %d / %d", tot_f, tot_r,
                     cap_frames, cap_rects);
  std::memcpy(offsets_out, c->last_offsets.data(), (tot_f + 1) * sizeof(int32_t));
  if (tot_r > 0) std::memcpy(out, c->last_rects.data(), tot_r * sizeof(cc_rect));
  return CC_OK;
}

}  // extern "C"
