// hipemu runtime translation unit (TEST INFRASTRUCTURE): the interpreter's out-of-line loops (MFMA arithmetic; built -O3) and its switches as plain
// C functions for ctypes. Linked into every emulated library.
#define HIPEMU_IMPL
#include <hip/hip_runtime.h>
extern "C" void hipemu_config(int dma_late, int greedy, unsigned seed) {
  hipemu::configure_from_env();
  hipemu::g_cfg.dma_late = dma_late; hipemu::g_cfg.greedy = greedy; hipemu::g_cfg.seed = seed;
}
extern "C" void hipemu_threads(int n) { hipemu::g_cfg.threads = n; }
extern "C" void hipemu_counters(long* out) {
  out[0] = hipemu::g_cfg.launches; out[1] = hipemu::g_cfg.blocks; out[2] = hipemu::g_cfg.mfma; out[3] = hipemu::g_cfg.dma_ops; out[4] = hipemu::g_cfg.tr_reads;
}
