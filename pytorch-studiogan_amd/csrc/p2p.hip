// p2p.hip -- sync-BN's cross-rank reduction FUSED INTO the batch-norm statistics kernel: the ranks exchange their per-channel partial sums by
// writing them straight into each other's HBM (xGMI peer stores into IPC-mapped mailboxes) from inside the kernel that then finalises
// mean / invstd / running statistics -- no collective library call, no extra launch, no host round trip between "statistics" and "normalise".
// Replaces torch.nn.SyncBatchNorm's all_gather of (mean, invstd, count) between its two kernels (reference src/models/model.py:161-165,
// torch/nn/modules/_functions.py SyncBatchNorm.forward) for the 2 x C doubles per layer that a data-parallel BigGAN exchanges ~300 times per step:
// at that size a ring all-reduce is pure latency (launch + proxy + N-1 hops), a one-shot all-to-all of 8-byte stores is one xGMI round trip.
//
// Protocol (cdna_hip_programming.md, publish/consume recipe R2 "the data is the flag"): a double travels as two 8-byte granules
// {tag = epoch : 32 | payload : 32}, each written by ONE system-scope atomic store, so a granule is either absent (old tag) or complete; the
// receiver re-reads its own mailbox until every granule of every sender carries this call's epoch. No fences, no separate flag word.
// Mailbox of rank r: [2 slots][world senders][2 * max_doubles granules]; call number e uses slot e & 1. A rank cannot finish call e before every
// peer has published e, and publishes e + 1 only after finishing e -- so when a fast rank overwrites slot (e + 1) & 1 the slow rank has long
// finished reading call e - 1 from it: two slots suffice. The sum runs over senders in rank order on every rank: replicas stay bit-identical.
// Spins are bounded (a peer that never arrives sets a timeout word the host can read; the call then returns garbage instead of hanging the GPU).
#include "common.h"
#include "../../include/sgamd.h"
#include <hip/hip_runtime.h>
#include <string.h>
#include <vector>

typedef unsigned long long p2p_u64;

struct SgP2P {
  int world, rank;
  long long max_doubles;
  unsigned epoch;                 // last epoch used (host side; every rank issues the same calls in the same order)
  p2p_u64* local;                 // this rank's mailbox (device memory, fine-grained)
  std::vector<p2p_u64*> peers;    // peers[r]: rank r's mailbox as mapped into this process (peers[rank] == local)
  p2p_u64** peers_dev;            // the same table on the device
  unsigned* timeouts_dev;         // [1] number of granule waits that ran out
  bool connected;
};

static inline size_t p2p_mailbox_bytes(int world, long long max_doubles) { return (size_t)2 * world * (size_t)(2 * max_doubles) * sizeof(p2p_u64); }

extern "C" int sg_p2p_create(int world, int rank, long long max_doubles, sg_p2p_t* out, void* handle_out64) {
  SG_CHECK(out && handle_out64 && world > 0 && rank >= 0 && rank < world && max_doubles > 0 && max_doubles <= (1 << 20), "sg_p2p_create: bad arguments");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  SgP2P* p = new SgP2P();
  p->world = world; p->rank = rank; p->max_doubles = max_doubles; p->epoch = 0; p->connected = false; p->peers_dev = nullptr; p->timeouts_dev = nullptr;
  const size_t bytes = p2p_mailbox_bytes(world, max_doubles);
  // fine-grained device memory: peer stores become visible to this device's loads without a cache flush on either side
  if (hipExtMallocWithFlags((void**)&p->local, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    if (hipMalloc((void**)&p->local, bytes) != hipSuccess) { delete p; sg_set_error("sg_p2p_create: mailbox allocation failed"); return -2; }
  }
  if (hipMemset(p->local, 0, bytes) != hipSuccess) { sg_set_error("sg_p2p_create: memset failed"); return -2; }      // tag 0 = "nothing published" (epochs start at 1)
  if (hipMalloc((void**)&p->timeouts_dev, sizeof(unsigned)) != hipSuccess || hipMemset(p->timeouts_dev, 0, sizeof(unsigned)) != hipSuccess) {
    sg_set_error("sg_p2p_create: allocation failed"); return -2;
  }
  hipIpcMemHandle_t h;
  memset(&h, 0, sizeof(h));
  if (world > 1 && hipIpcGetMemHandle(&h, p->local) != hipSuccess) { sg_set_error("sg_p2p_create: hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)"); return -2; }
  memcpy(handle_out64, &h, 64);
  if (hipDeviceSynchronize() != hipSuccess) { sg_set_error("sg_p2p_create: sync failed"); return -2; }
  *out = p;
  return 0;
}

// handles: world x 64 bytes, entry r = what rank r's sg_p2p_create returned (handed around by the host over any channel)
extern "C" int sg_p2p_connect(sg_p2p_t pp, const void* handles) {
  SgP2P* p = (SgP2P*)pp;
  SG_CHECK(p && handles && !p->connected, "sg_p2p_connect: bad arguments");
  p->peers.assign(p->world, nullptr);
  for (int r = 0; r < p->world; r++) {
    if (r == p->rank) { p->peers[r] = p->local; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + 64 * (size_t)r, 64);
    void* ptr = nullptr;
    if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { sg_set_error("sg_p2p_connect: hipIpcOpenMemHandle failed"); return -2; }
    p->peers[r] = (p2p_u64*)ptr;
  }
  if (hipMalloc((void**)&p->peers_dev, sizeof(p2p_u64*) * p->world) != hipSuccess ||
      hipMemcpy(p->peers_dev, p->peers.data(), sizeof(p2p_u64*) * p->world, hipMemcpyHostToDevice) != hipSuccess) {
    sg_set_error("sg_p2p_connect: table upload failed"); return -2;
  }
  p->connected = true;
  return 0;
}

extern "C" int sg_p2p_timeouts(sg_p2p_t pp, int* count) {
  SgP2P* p = (SgP2P*)pp;
  SG_CHECK(p && count, "sg_p2p_timeouts: bad arguments");
  unsigned v = 0;
  if (hipMemcpy(&v, p->timeouts_dev, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) { sg_set_error("sg_p2p_timeouts: copy failed"); return -2; }
  *count = (int)v;
  return 0;
}

extern "C" int sg_p2p_destroy(sg_p2p_t pp) {
  SgP2P* p = (SgP2P*)pp;
  if (!p) return 0;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < (int)p->peers.size(); r++)
    if (r != p->rank && p->peers[r]) (void)hipIpcCloseMemHandle(p->peers[r]);
  if (p->peers_dev) (void)hipFree(p->peers_dev);
  if (p->timeouts_dev) (void)hipFree(p->timeouts_dev);
  if (p->local) (void)hipFree(p->local);
  delete p;
  return 0;
}

// ---- device side ----------------------------------------------------------------------------------------------------------------------------------
#define P2P_SPIN_LIMIT (1u << 22)      // x ~1 us of s_sleep: a few seconds, then give up (the peer died or never issued the call)

__device__ __forceinline__ void p2p_publish(p2p_u64* const* peers, int world, int rank, long long slot_stride, long long sender_stride, int slot, long long g, unsigned epoch,
                                            unsigned payload) {
  const p2p_u64 v = ((p2p_u64)epoch << 32) | payload;
  for (int r = 0; r < world; r++)
    __hip_atomic_store(peers[r] + slot * slot_stride + rank * sender_stride + g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned p2p_consume(const p2p_u64* mine, unsigned epoch, unsigned* timeouts) {
  for (unsigned spins = 0;; spins++) {
    const p2p_u64 x = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned)(x >> 32) == epoch) return (unsigned)x;
    if (spins >= P2P_SPIN_LIMIT) { atomicAdd(timeouts, 1u); return 0x7fc00000u; }      // (a NaN pattern in either half poisons the result visibly)
    __builtin_amdgcn_s_sleep(8);
  }
}
// sum over the ranks of vals[i], i < n, in rank order; one thread per double. Returns the sum for this thread's i.
__device__ __forceinline__ double p2p_allreduce_one(double mine, long long i, p2p_u64* const* peers, int world, int rank, long long max_doubles, unsigned epoch, unsigned* timeouts) {
  const long long sender_stride = 2 * max_doubles, slot_stride = sender_stride * world;
  const int slot = epoch & 1;
  const p2p_u64 bits = (p2p_u64)__double_as_longlong(mine);
  p2p_publish(peers, world, rank, slot_stride, sender_stride, slot, 2 * i, epoch, (unsigned)bits);
  p2p_publish(peers, world, rank, slot_stride, sender_stride, slot, 2 * i + 1, epoch, (unsigned)(bits >> 32));
  const p2p_u64* box = peers[rank] + slot * slot_stride;
  double sum = 0.0;
  for (int s = 0; s < world; s++) {
    const unsigned lo = p2p_consume(box + s * sender_stride + 2 * i, epoch, timeouts);
    const unsigned hi = p2p_consume(box + s * sender_stride + 2 * i + 1, epoch, timeouts);
    sum += __longlong_as_double((long long)(((p2p_u64)hi << 32) | lo));
  }
  return sum;
}

__global__ __launch_bounds__(256) void k_p2p_allreduce_f64(double* buf, int n, p2p_u64* const* peers, int world, int rank, long long max_doubles, unsigned epoch, unsigned* timeouts) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  buf[i] = p2p_allreduce_one(buf[i], i, peers, world, rank, max_doubles, epoch, timeouts);
}
// the fused kernel: this rank's partial sums -> every peer's HBM -> (wait) -> mean / invstd / running statistics of the GLOBAL batch. One thread per channel.
__global__ __launch_bounds__(256) void k_bn_finalize_p2p(const double* partial, double count, int C, float eps, float momentum, float* mean, float* invstd, float* rm, float* rv,
                                                         p2p_u64* const* peers, int world, int rank, long long max_doubles, unsigned epoch, unsigned* timeouts) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double s1 = p2p_allreduce_one(partial[2 * c], 2 * c, peers, world, rank, max_doubles, epoch, timeouts);
  const double s2 = p2p_allreduce_one(partial[2 * c + 1], 2 * c + 1, peers, world, rank, max_doubles, epoch, timeouts);
  // (csrc/norm.hip k_bn_finalize on the summed partials)
  const double m = s1 / count;
  double var = s2 / count - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rm) {
    const double unb = (count > 1.0) ? var * count / (count - 1.0) : var;
    rm[c] = (1.f - momentum) * rm[c] + momentum * (float)m;
    rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unb;
  }
}

static int p2p_ready(SgP2P* p, long long n, const char* what) {
  if (!p || (!p->connected && p->world > 1)) { sg_set_error(what); return -1; }
  if (n > p->max_doubles) { sg_set_error("sg_p2p: more values than the mailbox was created for"); return -1; }
  if (p->world == 1 && !p->connected) {          // a single rank needs no peers: its own mailbox is the whole table
    p->peers.assign(1, p->local);
    if (hipMalloc((void**)&p->peers_dev, sizeof(p2p_u64*)) != hipSuccess || hipMemcpy(p->peers_dev, p->peers.data(), sizeof(p2p_u64*), hipMemcpyHostToDevice) != hipSuccess) {
      sg_set_error("sg_p2p: table upload failed"); return -2;
    }
    p->connected = true;
  }
  p->epoch++;
  if (p->epoch == 0) p->epoch = 2;               // (tag 0 means "empty"; keep the slot parity sequence: 0xffffffff -> 2 would break it only after 4e9 calls -- skip to an even epoch)
  return 0;
}

// in-place sum of n <= max_doubles doubles over the ranks (the backward pass's channel terms), one launch on s
extern "C" int sg_p2p_allreduce_f64(sg_p2p_t pp, double* buf, int n, sg_stream_t s) {
  SgP2P* p = (SgP2P*)pp;
  SG_CHECK(buf && n > 0, "sg_p2p_allreduce_f64: bad arguments");
  if (int rc = p2p_ready(p, n, "sg_p2p_allreduce_f64: not connected")) return rc;
  hipLaunchKernelGGL(k_p2p_allreduce_f64, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, buf, n, (p2p_u64* const*)p->peers_dev, p->world, p->rank, p->max_doubles, p->epoch,
                     p->timeouts_dev);
  SG_LAUNCH_CHECK();
  return 0;
}

// partial [2 C] fp64: this rank's sums (already complete: sg_bn_partial_stats / sg_bn_stats_from_tiles); count = LOCAL rows x world
extern "C" int sg_bn_finalize_p2p(sg_p2p_t pp, const double* partial, double count, int C, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                                  float* running_var, sg_stream_t s) {
  SgP2P* p = (SgP2P*)pp;
  SG_CHECK(partial && mean && invstd && count > 0 && C > 0, "sg_bn_finalize_p2p: bad arguments");
  SG_CHECK((running_mean == nullptr) == (running_var == nullptr), "sg_bn_finalize_p2p: running stats must come as a pair");
  if (int rc = p2p_ready(p, 2ll * C, "sg_bn_finalize_p2p: not connected")) return rc;
  hipLaunchKernelGGL(k_bn_finalize_p2p, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)s, partial, count, C, eps, momentum, mean, invstd, running_mean, running_var,
                     (p2p_u64* const*)p->peers_dev, p->world, p->rank, p->max_doubles, p->epoch, p->timeouts_dev);
  SG_LAUNCH_CHECK();
  return 0;
}

// sync-BN statistics in one call on one stream: local partial sums -> fused exchange + finalize (the peer-store counterpart of sg_bn_stats_sync)
extern "C" int sg_bn_stats_sync_p2p(int dtype, const void* x, int ldx, long long rows, int C, double* partial, sg_p2p_t pp, float eps, float momentum, float* mean, float* invstd,
                                    float* running_mean, float* running_var, sg_stream_t s) {
  SgP2P* p = (SgP2P*)pp;
  SG_CHECK(x && partial && p && rows > 0 && C > 0, "sg_bn_stats_sync_p2p: bad arguments");
  if (hipMemsetAsync(partial, 0, sizeof(double) * 2 * C, (hipStream_t)s) != hipSuccess) { sg_set_error("sg_bn_stats_sync_p2p: memset failed"); return -2; }
  if (int rc = sg_bn_partial_stats(dtype, x, ldx, rows, C, partial, s)) return rc;
  return sg_bn_finalize_p2p(pp, partial, (double)rows * p->world, C, eps, momentum, mean, invstd, running_mean, running_var, s);
}
