// comm.hip -- RCCL entry points of the C ABI: the data-parallel exchanges of the training step without torch.distributed in the
// data path. Replaces DistributedDataParallel's bucketed all-reduce and SyncBatchNorm's statistics exchange
// (reference src/models/model.py:157-180) for a host that drives libsgamd.so directly:
//   sg_comm_unique_id / sg_comm_init_rank / sg_comm_destroy : one communicator per process (one process per GPU; xGMI underneath)
//   sg_allreduce_flat      : in-place sum of a flat fp32 / fp64 device buffer on the caller's stream (the gradient arena, BN terms)
//   sg_bn_stats_sync       : batch-norm statistics of a data-parallel batch in ONE call: per-rank partial sums -> all-reduce ->
//                            mean / invstd / running statistics, all enqueued on the caller's stream (no host sync in between)
// librccl is bound at run time (dlopen of its soname): a process that already carries RCCL through PyTorch-ROCm shares that copy,
// and single-GPU use needs no RCCL at all.
#include "common.h"
#include "../../include/sgamd.h"
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

typedef struct { char internal[128]; } sg_nccl_uid;
typedef void* sg_nccl_comm;
typedef int (*fn_get_uid)(sg_nccl_uid*);
typedef int (*fn_init_rank)(sg_nccl_comm*, int, sg_nccl_uid, int);
typedef int (*fn_destroy)(sg_nccl_comm);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, sg_nccl_comm, hipStream_t);
typedef int (*fn_reduce_scatter)(const void*, void*, size_t, int, int, sg_nccl_comm, hipStream_t);
typedef int (*fn_allgather)(const void*, void*, size_t, int, sg_nccl_comm, hipStream_t);
typedef int (*fn_rank)(const sg_nccl_comm, int*);
typedef int (*fn_count)(const sg_nccl_comm, int*);
typedef const char* (*fn_errstr)(int);

static void* g_rccl = nullptr;
static fn_get_uid p_get_uid; static fn_init_rank p_init_rank; static fn_destroy p_destroy; static fn_allreduce p_allreduce;
static fn_count p_count; static fn_errstr p_errstr; static fn_reduce_scatter p_reduce_scatter; static fn_allgather p_allgather; static fn_rank p_rank;

static int rccl_load() {
  if (g_rccl) return 0;
  // 1. the copy this process already carries (PyTorch-ROCm maps its own torch/lib/librccl.so): RTLD_NOLOAD finds it by soname
  //    without mapping a second, possibly different, RCCL; 2. only then a fresh load.
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (int i = 0; i < 2 && !g_rccl; i++) g_rccl = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  for (int i = 0; i < 3 && !g_rccl; i++) g_rccl = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!g_rccl) { sg_set_error("sg_comm: librccl.so not found"); return -3; }
  p_get_uid = (fn_get_uid)dlsym(g_rccl, "ncclGetUniqueId");
  p_init_rank = (fn_init_rank)dlsym(g_rccl, "ncclCommInitRank");
  p_destroy = (fn_destroy)dlsym(g_rccl, "ncclCommDestroy");
  p_allreduce = (fn_allreduce)dlsym(g_rccl, "ncclAllReduce");
  p_count = (fn_count)dlsym(g_rccl, "ncclCommCount");
  p_errstr = (fn_errstr)dlsym(g_rccl, "ncclGetErrorString");
  p_reduce_scatter = (fn_reduce_scatter)dlsym(g_rccl, "ncclReduceScatter");
  p_allgather = (fn_allgather)dlsym(g_rccl, "ncclAllGather");
  p_rank = (fn_rank)dlsym(g_rccl, "ncclCommUserRank");
  if (!p_get_uid || !p_init_rank || !p_destroy || !p_allreduce || !p_count) {
    sg_set_error("sg_comm: RCCL symbols missing");
    dlclose(g_rccl);
    g_rccl = nullptr;
    return -3;
  }
  return 0;
}
static int rccl_fail(const char* what, int rc) {
  char buf[256];
  snprintf(buf, sizeof(buf), "%s: RCCL error %d (%s)", what, rc, p_errstr ? p_errstr(rc) : "?");
  sg_set_error(buf);
  return -4;
}

extern "C" int sg_comm_unique_id(void* out128) {
  SG_CHECK(out128, "sg_comm_unique_id: null");
  if (int rc = rccl_load()) return rc;
  sg_nccl_uid id;
  if (int rc = p_get_uid(&id)) return rccl_fail("sg_comm_unique_id", rc);
  memcpy(out128, &id, 128);
  return 0;
}
extern "C" int sg_comm_init_rank(const void* id128, int nranks, int rank, sg_comm_t* comm) {
  SG_CHECK(id128 && comm && nranks > 0 && rank >= 0 && rank < nranks, "sg_comm_init_rank: bad arguments");
  if (int rc = rccl_load()) return rc;
  sg_nccl_uid id;
  memcpy(&id, id128, 128);
  sg_nccl_comm c = nullptr;
  if (int rc = p_init_rank(&c, nranks, id, rank)) return rccl_fail("sg_comm_init_rank", rc);
  *comm = c;
  return 0;
}
extern "C" int sg_comm_size(sg_comm_t comm, int* nranks) {
  SG_CHECK(comm && nranks && g_rccl, "sg_comm_size: no communicator");
  if (int rc = p_count(comm, nranks)) return rccl_fail("sg_comm_size", rc);
  return 0;
}
extern "C" int sg_comm_destroy(sg_comm_t comm) {
  if (!comm || !g_rccl) return 0;
  if (int rc = p_destroy(comm)) return rccl_fail("sg_comm_destroy", rc);
  return 0;
}
// in-place sum over the ranks of `comm`; dtype: SG_DTYPE_F32 or SG_DTYPE_F64
extern "C" int sg_allreduce_flat(sg_comm_t comm, void* buf, long long count, int dtype, sg_stream_t s) {
  SG_CHECK(comm && buf && count > 0 && g_rccl, "sg_allreduce_flat: bad arguments / no communicator");
  SG_CHECK(dtype == SG_DTYPE_F32 || dtype == SG_DTYPE_F64, "sg_allreduce_flat: fp32 or fp64 buffers only");
  const int nccl_dt = dtype == SG_DTYPE_F32 ? 7 : 8;      // ncclFloat32 / ncclFloat64 (rccl.h)
  if (int rc = p_allreduce(buf, buf, (size_t)count, nccl_dt, 0 /* ncclSum */, comm, (hipStream_t)s)) return rccl_fail("sg_allreduce_flat", rc);
  return 0;
}
// The two halves of the sharded gradient exchange, in place on the flat arenas: rank r's shard is buf[r * per_rank, (r + 1) * per_rank).
// RCCL's in-place conventions: reduce-scatter with recvbuff == sendbuff + rank * recvcount, all-gather with sendbuff == recvbuff + rank * sendcount.
extern "C" int sg_reduce_scatter_flat(sg_comm_t comm, float* buf, long long per_rank, sg_stream_t s) {
  SG_CHECK(comm && buf && per_rank > 0 && g_rccl && p_reduce_scatter && p_rank, "sg_reduce_scatter_flat: bad arguments / no communicator");
  int rank = 0;
  if (int rc = p_rank(comm, &rank)) return rccl_fail("sg_reduce_scatter_flat", rc);
  if (int rc = p_reduce_scatter(buf, buf + (size_t)rank * per_rank, (size_t)per_rank, 7 /* ncclFloat32 */, 0 /* ncclSum */, comm, (hipStream_t)s))
    return rccl_fail("sg_reduce_scatter_flat", rc);
  return 0;
}
extern "C" int sg_allgather_flat(sg_comm_t comm, float* buf, long long per_rank, sg_stream_t s) {
  SG_CHECK(comm && buf && per_rank > 0 && g_rccl && p_allgather && p_rank, "sg_allgather_flat: bad arguments / no communicator");
  int rank = 0;
  if (int rc = p_rank(comm, &rank)) return rccl_fail("sg_allgather_flat", rc);
  if (int rc = p_allgather(buf + (size_t)rank * per_rank, buf, (size_t)per_rank, 7 /* ncclFloat32 */, comm, (hipStream_t)s)) return rccl_fail("sg_allgather_flat", rc);
  return 0;
}
// Batch statistics of a (data-parallel) batch: partial[2C] fp64 scratch (overwritten). comm == NULL: single rank. `rows` is the
// LOCAL pixel count; the count used for mean / variance and the unbiased running variance is rows * nranks (equal per-rank batches,
// what the reference's SyncBatchNorm sees with a DistributedSampler that drops the ragged tail).
extern "C" int sg_bn_stats_sync(int dtype, const void* x, int ldx, long long rows, int C, double* partial, sg_comm_t comm, float eps, float momentum,
                                float* mean, float* invstd, float* running_mean, float* running_var, sg_stream_t s) {
  SG_CHECK(x && partial && mean && invstd && rows > 0 && C > 0, "sg_bn_stats_sync: bad arguments");
  if (hipMemsetAsync(partial, 0, sizeof(double) * 2 * C, (hipStream_t)s) != hipSuccess) { sg_set_error("sg_bn_stats_sync: memset failed"); return -2; }
  if (int rc = sg_bn_partial_stats(dtype, x, ldx, rows, C, partial, s)) return rc;
  int world = 1;
  if (comm) {
    if (int rc = sg_comm_size(comm, &world)) return rc;
    if (world > 1) { if (int rc = sg_allreduce_flat(comm, partial, 2ll * C, SG_DTYPE_F64, s)) return rc; }
  }
  return sg_bn_finalize(partial, (double)rows * world, C, eps, momentum, mean, invstd, running_mean, running_var, s);
}
