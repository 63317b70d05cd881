// Section 7 of the C ABI: the gather of detections over RCCL. Host code only: two all-gathers of
// KB-sized int32 buffers. librccl is resolved at run time (dlopen), preferring a copy the process has already loaded,
// so that the library neither links RCCL nor brings a second copy into a PyTorch process.
//
// The exchange runs on a Transport: RcclTransport (ncclAllGather on device staging buffers; the product path) or
// SocketTransport (CCAMD_COMM_TRANSPORT=tcp: host buffers over loopback TCP, star through rank 0). The second exists so
// that the world > 1 protocol of cc_gather_detections -- packing, padding, unpacking, the error marker, the
// BUFFER_TOO_SMALL rule -- can be executed by spawned CPU processes in the test suite and during bring-up on a box with
// fewer GPUs than ranks; it needs no device and is never selected unless the environment asks for it.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <random>
#include <thread>
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

bool tcp_transport_requested() {
  const char* e = std::getenv("CCAMD_COMM_TRANSPORT");
  return e && std::strcmp(e, "tcp") == 0;
}

// ---- transports --------------------------------------------------------------------------------------------------
struct Transport {
  virtual ~Transport() {}
  // every rank contributes `count` int32 from `send` (host memory); `recv` receives world * count, in rank order
  virtual cc_status all_gather_i32(const int32_t* send, size_t count, int32_t* recv) = 0;
};

struct RcclTransport : Transport {
  int device = 0, world = 1;
  void* nccl = nullptr;
  hipStream_t stream = nullptr;
  int32_t* d_send = nullptr;
  int32_t* d_recv = nullptr;
  size_t send_cap = 0, recv_cap = 0;  // int32 entries
  ~RcclTransport() override {
    (void)hipSetDevice(device);
    if (nccl) (void)rccl_api().comm_destroy(nccl);
    if (d_send) (void)hipFree(d_send);
    if (d_recv) (void)hipFree(d_recv);
    if (stream) (void)hipStreamDestroy(stream);
  }
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
  cc_status all_gather_i32(const int32_t* h_send, size_t count, int32_t* h_recv) override {
    CC_HIP(hipSetDevice(device));
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

// Loopback TCP, star through rank 0. Blocking calls with a receive timeout, so that a missing peer is an error, not a hang.
struct TcpId {  // what the CC_COMM_ID_BYTES bytes of a tcp id hold
  char magic[8];  // "CCTCP1\0\0"
  uint32_t port;
  unsigned char token[16];
};
constexpr char kTcpMagic[8] = {'C', 'C', 'T', 'C', 'P', '1', 0, 0};

bool send_all(int fd, const void* p, size_t n) {
  const char* c = static_cast<const char*>(p);
  while (n > 0) {
    const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
    if (k <= 0) {
      if (k < 0 && errno == EINTR) continue;
      return false;
    }
    c += k;
    n -= (size_t)k;
  }
  return true;
}
bool recv_all(int fd, void* p, size_t n) {
  char* c = static_cast<char*>(p);
  while (n > 0) {
    const ssize_t k = ::recv(fd, c, n, 0);
    if (k <= 0) {
      if (k < 0 && errno == EINTR) continue;
      return false;
    }
    c += k;
    n -= (size_t)k;
  }
  return true;
}
void set_socket_options(int fd) {
  int one = 1;
  (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
  int secs = 60;
  if (const char* e = std::getenv("CCAMD_COMM_TIMEOUT_S")) secs = std::max(1, std::atoi(e));
  timeval tv{secs, 0};
  (void)setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
  (void)setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
}

struct SocketTransport : Transport {
  int rank = 0, world = 1;
  std::vector<int> peers;  // rank 0: fd of every other rank (index = rank); others: peers[0] = fd to rank 0
  ~SocketTransport() override {
    for (int fd : peers)
      if (fd >= 0) ::close(fd);
  }
  cc_status all_gather_i32(const int32_t* send, size_t count, int32_t* recv) override {
    const uint64_t n = count;
    if (rank != 0) {
      const int fd = peers[0];
      if (!send_all(fd, &n, sizeof(n)) || !send_all(fd, send, count * sizeof(int32_t)))
        return set_error(CC_ERR_IO, "tcp transport: rank %d cannot send to rank 0 (%s)", rank, std::strerror(errno));
      uint64_t ok = 0;
      if (!recv_all(fd, &ok, sizeof(ok))) return set_error(CC_ERR_IO, "tcp transport: rank %d lost rank 0 (%s)", rank, std::strerror(errno));
      if (ok != n) return set_error(CC_ERR_INVALID_ARG, "tcp transport: the ranks disagree on the element count of an all-gather (%llu here)", (unsigned long long)n);
      if (!recv_all(fd, recv, count * (size_t)world * sizeof(int32_t)))
        return set_error(CC_ERR_IO, "tcp transport: rank %d lost rank 0 (%s)", rank, std::strerror(errno));
      return CC_OK;
    }
    bool same = true, io_ok = true;
    if (count) std::memcpy(recv, send, count * sizeof(int32_t));
    std::vector<char> sink;
    for (int r = 1; r < world && io_ok; r++) {
      uint64_t m = 0;
      io_ok = recv_all(peers[(size_t)r], &m, sizeof(m));
      if (!io_ok) break;
      if (m == n) {
        io_ok = recv_all(peers[(size_t)r], recv + (size_t)r * count, count * sizeof(int32_t));
      } else {  // drain what the peer sends, then tell everybody
        same = false;
        // in fixed pieces: the count comes from the peer, and a broken or hostile one must not size an allocation here
        constexpr uint64_t kMaxDrainElems = 1ull << 28;  // 1 GiB of int32: far beyond any gather of detections
        if (m > kMaxDrainElems) {
          errno = EPROTO;
          io_ok = false;
          break;
        }
        sink.resize(1 << 16);
        for (uint64_t left = m * sizeof(int32_t); left > 0 && io_ok;) {
          const size_t piece = (size_t)std::min<uint64_t>(left, sink.size());
          io_ok = recv_all(peers[(size_t)r], sink.data(), piece);
          left -= piece;
        }
      }
    }
    if (!io_ok) return set_error(CC_ERR_IO, "tcp transport: rank 0 lost a peer (%s)", std::strerror(errno));
    const uint64_t verdict = same ? n : ~0ull;
    for (int r = 1; r < world; r++) {
      if (!send_all(peers[(size_t)r], &verdict, sizeof(verdict))) io_ok = false;
      if (same && io_ok && !send_all(peers[(size_t)r], recv, count * (size_t)world * sizeof(int32_t))) io_ok = false;
    }
    if (!io_ok) return set_error(CC_ERR_IO, "tcp transport: rank 0 cannot reach a peer (%s)", std::strerror(errno));
    if (!same) return set_error(CC_ERR_INVALID_ARG, "tcp transport: the ranks disagree on the element count of an all-gather (%llu on rank 0)", (unsigned long long)n);
    return CC_OK;
  }
};

// An id that cc_comm_unique_id has handed out and no communicator has consumed yet. RCCL starts a bootstrap listener
// for every id it creates and nothing retires one that is never used (round 2: a process that asked for two ids and built
// one communicator aborted at exit inside librccl). So at most ONE id is outstanding per process: asking again before
// rank 0 has called cc_comm_create with it returns the SAME bytes, and cc_comm_create on rank 0 consumes it.
struct Outstanding {
  bool valid = false;
  bool tcp = false;
  char id[CC_COMM_ID_BYTES] = {};
  int listen_fd = -1;
};
std::mutex g_id_mu;
Outstanding g_outstanding;

}  // namespace

struct cc_comm {
  int device = 0, rank = 0, world = 1;
  std::unique_ptr<Transport> tr;      // null when world == 1
  std::vector<int32_t> last_offsets;  // result of the last gather (cc_gather_fetch)
  std::vector<cc_rect> last_rects;
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
  const bool tcp = tcp_transport_requested();
  std::lock_guard<std::mutex> lk(g_id_mu);
  if (g_outstanding.valid && g_outstanding.tcp == tcp) {  // not consumed yet: the same id again, no second listener
    std::memcpy(id, g_outstanding.id, CC_COMM_ID_BYTES);
    return CC_OK;
  }
  if (g_outstanding.valid && g_outstanding.listen_fd >= 0) {  // a stale id of the other transport
    ::close(g_outstanding.listen_fd);
    g_outstanding = Outstanding();
  }
  if (tcp) {
    const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return set_error(CC_ERR_IO, "cc_comm_unique_id: socket: %s", std::strerror(errno));
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    a.sin_port = 0;
    socklen_t len = sizeof(a);
    if (::bind(fd, reinterpret_cast<sockaddr*>(&a), sizeof(a)) != 0 || ::listen(fd, 64) != 0 ||
        ::getsockname(fd, reinterpret_cast<sockaddr*>(&a), &len) != 0) {
      const int e = errno;
      ::close(fd);
      return set_error(CC_ERR_IO, "cc_comm_unique_id: cannot listen on 127.0.0.1: %s", std::strerror(e));
    }
    TcpId t{};
    std::memcpy(t.magic, kTcpMagic, 8);
    t.port = ntohs(a.sin_port);
    std::random_device rd;
    for (unsigned char& b : t.token) b = (unsigned char)rd();
    g_outstanding = Outstanding();
    std::memcpy(g_outstanding.id, &t, sizeof(t));
    g_outstanding.listen_fd = fd;
  } else {
    cc_status st = need_rccl("cc_comm_unique_id");
    if (st != CC_OK) return st;
    UniqueId u;
    const int rc = rccl_api().get_unique_id(&u);
    if (rc != 0) return set_error(CC_ERR_HIP, "ncclGetUniqueId failed: %s", rccl_api().error_string(rc));
    g_outstanding = Outstanding();
    std::memcpy(g_outstanding.id, u.internal, CC_COMM_ID_BYTES);
  }
  g_outstanding.valid = true;
  g_outstanding.tcp = tcp;
  std::memcpy(id, g_outstanding.id, CC_COMM_ID_BYTES);
  return CC_OK;
}

static cc_status create_tcp(cc_comm* c, const void* id) {
  TcpId t;
  std::memcpy(&t, id, sizeof(t));
  std::unique_ptr<SocketTransport> tr(new SocketTransport);
  tr->rank = c->rank;
  tr->world = c->world;
  struct Hello {
    unsigned char token[16];
    int32_t rank;
  };
  if (c->rank == 0) {
    int lfd = -1;
    {
      std::lock_guard<std::mutex> lk(g_id_mu);
      if (g_outstanding.valid && g_outstanding.tcp && std::memcmp(g_outstanding.id, id, CC_COMM_ID_BYTES) == 0) {
        lfd = g_outstanding.listen_fd;
        g_outstanding = Outstanding();  // consumed
      }
    }
    if (lfd < 0) return set_error(CC_ERR_INVALID_ARG, "cc_comm_create: rank 0 must use the id its own process got from cc_comm_unique_id (tcp transport)");
    set_socket_options(lfd);  // accept() honours the receive timeout
    tr->peers.assign((size_t)c->world, -1);
    int have = 0;
    while (have < c->world - 1) {
      const int fd = ::accept(lfd, nullptr, nullptr);
      if (fd < 0) {
        if (errno == EINTR) continue;
        const int e = errno;
        ::close(lfd);
        return set_error(CC_ERR_IO, "cc_comm_create: rank 0 waited for %d more rank(s): %s", c->world - 1 - have, std::strerror(e));
      }
      set_socket_options(fd);
      Hello h{};
      if (!recv_all(fd, &h, sizeof(h)) || std::memcmp(h.token, t.token, 16) != 0 || h.rank < 1 || h.rank >= c->world ||
          tr->peers[(size_t)h.rank] >= 0) {
        ::close(fd);  // not one of ours (or a duplicate rank): ignore
        continue;
      }
      tr->peers[(size_t)h.rank] = fd;
      have++;
    }
    ::close(lfd);
  } else {
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    a.sin_port = htons((uint16_t)t.port);
    int fd = -1;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(60);
    for (;;) {
      fd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (fd < 0) return set_error(CC_ERR_IO, "cc_comm_create: socket: %s", std::strerror(errno));
      if (::connect(fd, reinterpret_cast<sockaddr*>(&a), sizeof(a)) == 0) break;
      const int e = errno;
      ::close(fd);
      fd = -1;
      if (std::chrono::steady_clock::now() > deadline)
        return set_error(CC_ERR_IO, "cc_comm_create: rank %d cannot reach rank 0 on 127.0.0.1:%u: %s", c->rank, t.port, std::strerror(e));
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
    set_socket_options(fd);
    Hello h{};
    std::memcpy(h.token, t.token, 16);
    h.rank = c->rank;
    if (!send_all(fd, &h, sizeof(h))) {
      ::close(fd);
      return set_error(CC_ERR_IO, "cc_comm_create: rank %d cannot greet rank 0: %s", c->rank, std::strerror(errno));
    }
    tr->peers.assign(1, fd);
  }
  c->tr = std::move(tr);
  return CC_OK;
}

static cc_status create_rccl(cc_comm* c, const void* id) {
  cc_status st = need_rccl("cc_comm_create");
  if (st != CC_OK) return st;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_error(CC_ERR_NO_DEVICE, "cc_comm_create: no usable HIP device");
  if (c->device < 0 || c->device >= n) return set_error(CC_ERR_INVALID_ARG, "cc_comm_create: device %d out of range (devices: %d)", c->device, n);
  if (c->rank == 0) {  // the id is used now: the next cc_comm_unique_id makes a new one
    std::lock_guard<std::mutex> lk(g_id_mu);
    if (g_outstanding.valid && !g_outstanding.tcp && std::memcmp(g_outstanding.id, id, CC_COMM_ID_BYTES) == 0) g_outstanding = Outstanding();
  }
  std::unique_ptr<RcclTransport> tr(new RcclTransport);
  tr->device = c->device;
  tr->world = c->world;
  hipError_t e = hipSetDevice(c->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&tr->stream, hipStreamNonBlocking);
  if (e != hipSuccess) return set_error(CC_ERR_HIP, "cc_comm_create: %s", hipGetErrorString(e));
  UniqueId u;
  std::memcpy(u.internal, id, CC_COMM_ID_BYTES);
  const int rc = rccl_api().comm_init_rank(&tr->nccl, c->world, u, c->rank);
  if (rc != 0) {
    tr->nccl = nullptr;
    return set_error(CC_ERR_HIP, "ncclCommInitRank failed: %s", rccl_api().error_string(rc));
  }
  c->tr = std::move(tr);
  return CC_OK;
}

cc_status cc_comm_create(int device, int rank, int world, const void* id, cc_comm** out) {
  if (!out) return set_error(CC_ERR_INVALID_ARG, "cc_comm_create: null output");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return set_error(CC_ERR_INVALID_ARG, "cc_comm_create: rank %d of %d", rank, world);
  if (world > 1 && !id) return set_error(CC_ERR_INVALID_ARG, "cc_comm_create: a unique id is required for more than one rank");
  std::unique_ptr<cc_comm> c(new cc_comm);
  c->device = device;
  c->rank = rank;
  c->world = world;
  if (world > 1) {
    // the id says which transport made it; an id of the tcp transport is only honoured when the environment asks for it
    const bool tcp_id = std::memcmp(id, kTcpMagic, 8) == 0;
    if (tcp_id != tcp_transport_requested())
      return set_error(CC_ERR_INVALID_ARG, "cc_comm_create: the id was made by the %s transport but CCAMD_COMM_TRANSPORT selects the other one",
                       tcp_id ? "tcp" : "RCCL");
    const cc_status st = tcp_id ? create_tcp(c.get(), id) : create_rccl(c.get(), id);
    if (st != CC_OK) return st;
  }
  *out = c.release();
  return CC_OK;
}

void cc_comm_destroy(cc_comm* c) { delete c; }

int cc_comm_rank(const cc_comm* c) {
  if (c) return c->rank;
  (void)set_error(CC_ERR_INVALID_ARG, "cc_comm_rank: null communicator");
  return -1;
}
int cc_comm_world(const cc_comm* c) {
  if (c) return c->world;
  (void)set_error(CC_ERR_INVALID_ARG, "cc_comm_world: null communicator");
  return -1;
}

cc_status cc_gather_detections(cc_comm* c, const cc_rect* rects, const int32_t* offsets, int n_frames, cc_rect* out, int cap_rects,
                               int32_t* offsets_out, int cap_frames, int* n_frames_all, int* n_rects_all) {
  if (!c) return set_error(CC_ERR_INVALID_ARG, "cc_gather_detections: null communicator");
  // Validate this rank's arguments, but do NOT return before the first collective: a rank that left early would leave
  // its peers blocked in the all-gather. A rank with bad arguments contributes the header {-1, -1}; every rank sees
  // the marker and all of them return an error together.
  const char* bad = nullptr;
  if (n_frames < 0 || (n_frames > 0 && !offsets) || !n_frames_all || !n_rects_all) bad = "null argument";
  int n_rects = 0;
  if (!bad && n_frames > 0) {
    for (int f = 0; f < n_frames && !bad; f++)
      if (offsets[f + 1] < offsets[f]) bad = "offsets must not decrease";
    if (!bad) {
      n_rects = offsets[n_frames] - offsets[0];
      if (n_rects < 0 || (n_rects > 0 && !rects)) bad = "bad offsets";
    }
  }
  const int world = c->world;
  // every rank's {frames, rectangles}
  std::vector<int32_t> headers((size_t)world * 2);
  const int32_t mine[2] = {bad ? -1 : n_frames, bad ? -1 : n_rects};
  if (world == 1) {
    headers[0] = mine[0];
    headers[1] = mine[1];
  } else {
    cc_status st = c->tr->all_gather_i32(mine, 2, headers.data());
    if (st != CC_OK) return st;
  }
  if (bad) return set_error(CC_ERR_INVALID_ARG, "cc_gather_detections: %s", bad);
  long long tot_f = 0, tot_r = 0;
  size_t max_len = 1;
  for (int r = 0; r < world; r++) {
    if (headers[(size_t)r * 2] < 0 || headers[(size_t)r * 2 + 1] < 0)
      return set_error(CC_ERR_INVALID_ARG, "cc_gather_detections: rank %d reported invalid arguments; no rank gathered", r);
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
    cc_status st = c->tr->all_gather_i32(payload.data(), max_len, all.data());
    if (st != CC_OK) return st;
    gathered = all.data();
  }
  // offsets are int32 (the C ABI's frame offsets): refuse totals they cannot hold -- after the collectives, on every rank alike
  if (tot_f > INT_MAX - 1 || tot_r > INT_MAX)
    return set_error(CC_ERR_OUT_OF_RANGE, "cc_gather_detections: %lld frames / %lld rectangles in total do not fit 32-bit offsets", tot_f, tot_r);
  // unpack into the communicator (kept for cc_gather_fetch), then hand out what fits
  c->last_offsets.assign((size_t)tot_f + 1, 0);
  c->last_rects.resize((size_t)tot_r);
  long long fo = 0, ro = 0;
  for (int r = 0; r < world; r++) {
    const int32_t* p = gathered + (size_t)r * max_len;
    const int nf = headers[(size_t)r * 2], nr = headers[(size_t)r * 2 + 1];
    long long sum = 0;
    for (int f = 0; f < nf; f++) {
      if (p[f] < 0) return set_error(CC_ERR_HIP, "cc_gather_detections: corrupt payload from rank %d", r);
      sum += p[f];
      c->last_offsets[(size_t)(fo + f + 1)] = (int32_t)(ro + sum);
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
