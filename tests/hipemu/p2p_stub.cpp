// The interpreter has one address space and no peers: the peer-store mailboxes of csrc/p2p.hip (IPC-mapped device memory, system-scope atomics) are not
// emulated. The whole-library build links these stand-ins in place of that unit so that the C ABI stays complete; every call fails loudly.
#include "../../include/sgamd.h"
extern "C" void sg_set_error(const char* msg);
static int no() { sg_set_error("sg_p2p_*: not available on the CPU interpreter"); return -1; }
extern "C" int sg_p2p_create(int, int, long long, sg_p2p_t*, void*) { return no(); }
extern "C" int sg_p2p_connect(sg_p2p_t, const void*) { return no(); }
extern "C" int sg_p2p_destroy(sg_p2p_t) { return 0; }
extern "C" int sg_p2p_timeouts(sg_p2p_t, int*) { return no(); }
extern "C" int sg_p2p_allreduce_f64(sg_p2p_t, double*, int, sg_stream_t) { return no(); }
extern "C" int sg_bn_finalize_p2p(sg_p2p_t, const double*, double, int, float, float, float*, float*, float*, float*, sg_stream_t) { return no(); }
extern "C" int sg_bn_stats_sync_p2p(int, const void*, int, long long, int, double*, sg_p2p_t, float, float, float*, float*, float*, float*, sg_stream_t) { return no(); }
