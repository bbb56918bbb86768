// hipemu harness: the few C-ABI symbols the kernel translation units expect from capi.hip (error string, launch profiler), and the
// emulator's switches as plain C functions for ctypes.  TEST INFRASTRUCTURE.
#include <hip/hip_runtime.h>
#include <string>
static std::string g_err;
extern "C" void sg_set_error(const char* msg) { g_err = msg ? msg : ""; }
extern "C" const char* sg_last_error() { return g_err.c_str(); }
extern "C" int sg_prof_begin(hipStream_t, double, int) { return -1; }
extern "C" void sg_prof_end(hipStream_t, int) {}
extern "C" void sg_prof_set_executed(int, double) {}
extern "C" void sg_prof_tag(int, int, double) {}
extern "C" void hipemu_config(int dma_late, int greedy, unsigned seed) {
  hipemu::configure_from_env();
  hipemu::g_cfg.dma_late = dma_late; hipemu::g_cfg.greedy = greedy; hipemu::g_cfg.seed = seed;
}
extern "C" void hipemu_counters(long* out) {
  out[0] = hipemu::g_cfg.launches; out[1] = hipemu::g_cfg.blocks; out[2] = hipemu::g_cfg.mfma; out[3] = hipemu::g_cfg.dma_ops; out[4] = hipemu::g_cfg.tr_reads;
}
