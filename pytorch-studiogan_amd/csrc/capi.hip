// capi.hip -- error plumbing shared by all C-ABI entry points of libsgamd.so
#include "common.h"
#include "../../include/sgamd.h"
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void sg_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "unknown", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* sg_last_error(void) { return g_err; }
extern "C" int sg_version(void) { return 1; }

// ---- in-library launch profiler (bench.py's roofline leg) --------------------------------------------------------
// When enabled, every launch of the MFMA contraction engine (conv fwd/dgrad/wgrad, gemm) is bracketed by a pair of
// hipEvents recorded on the SAME stream the kernel is launched on; sg_prof_collect() sums the elapsed times and the
// algorithmic FLOPs after the caller has synchronised. Off by default: zero overhead on the normal path.
#define SG_PROF_MAX 16384
static int g_prof_on = 0;
static int g_prof_n = 0;
static hipEvent_t g_ev0[SG_PROF_MAX], g_ev1[SG_PROF_MAX];
static double g_flops[SG_PROF_MAX];
static double g_exec[SG_PROF_MAX];      // FLOPs the launch really executes (quad convolutions: 16 / 36 of the algorithmic count)
static double g_bytes[SG_PROF_MAX];     // algorithmic HBM bytes of the launch (every operand read once, the result written once)
static int g_tag[SG_PROF_MAX];          // which kernel family took the problem (SG_ENG_*)
static int g_kind[SG_PROF_MAX];
static int g_ev_created = 0;

extern "C" int sg_prof_enable(int on) {
  if (on && g_ev_created < SG_PROF_MAX) {
    for (int i = g_ev_created; i < SG_PROF_MAX; i++) {
      if (hipEventCreate(&g_ev0[i]) != hipSuccess || hipEventCreate(&g_ev1[i]) != hipSuccess) { sg_set_error("sg_prof_enable: hipEventCreate failed"); return -2; }
    }
    g_ev_created = SG_PROF_MAX;
  }
  g_prof_on = on;          // bit 0: contraction engine (kinds 0-2), bit 1: HBM-bound families (kinds 3-6), bit 2: the quad convolutions alone (sg_prof_begin_q)
  g_prof_n = 0;
  return 0;
}
// returns the slot index or -1
extern "C" int sg_prof_begin(hipStream_t st, double flops, int kind) {
  if (!(g_prof_on & (kind < 3 ? 1 : 2)) || g_prof_n >= SG_PROF_MAX) return -1;
  const int i = g_prof_n++;
  g_flops[i] = flops; g_exec[i] = flops; g_kind[i] = kind; g_bytes[i] = 0.0; g_tag[i] = 0;
  (void)hipEventRecord(g_ev0[i], st);
  return i;
}
// sg_conv2d_q's launches (the dominant kernel of the benchmarked step) also answer to bit 2 alone. An event pair is two barrier packets on the launch stream: around EVERY
// engine launch of a C3 step (~360) they cost ~3 ms of dispatch gaps per step (round 6, tools/kt_gaps.py on a traced bench run: 5-20 us of idle in front of every bracketed
// kernel), which is why bench.py's timed region brackets this one kernel only and takes the other families' figures from profiled steps after it.
extern "C" int sg_prof_begin_q(hipStream_t st, double flops, int kind) {
  if (!(g_prof_on & 5) || g_prof_n >= SG_PROF_MAX) return -1;
  const int i = g_prof_n++;
  g_flops[i] = flops; g_exec[i] = flops; g_kind[i] = kind; g_bytes[i] = 0.0; g_tag[i] = 0;
  (void)hipEventRecord(g_ev0[i], st);
  return i;
}
// a launch that reaches the algorithmic result with fewer operations (conv_q.h) reports what it executes
extern "C" void sg_prof_set_executed(int slot, double flops) {
  if (slot >= 0) g_exec[slot] = flops;
}
extern "C" void sg_prof_end(hipStream_t st, int slot) {
  if (slot >= 0) (void)hipEventRecord(g_ev1[slot], st);
}
// out[kind*3 + {0,1,2}] = {launches, total ms, total flops} for kind in 0..3 (0 conv fwd/dgrad, 1 conv wgrad, 2 gemm)
extern "C" int sg_prof_collect(double* out, int nkinds) {
  for (int k = 0; k < nkinds * 3; k++) out[k] = 0.0;
  for (int i = 0; i < g_prof_n; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_ev0[i], g_ev1[i]) != hipSuccess) { sg_set_error("sg_prof_collect: events not complete (synchronise first)"); return -2; }
    const int k = g_kind[i];
    if (k < 0 || k >= nkinds) continue;
    out[k * 3 + 0] += 1.0; out[k * 3 + 1] += ms; out[k * 3 + 2] += g_flops[i];
  }
  g_prof_n = 0;
  return 0;
}
extern "C" void sg_prof_tag(int slot, int engine, double alg_bytes) {
  if (slot >= 0) { g_tag[slot] = engine; g_bytes[slot] = alg_bytes; }
}
// per kernel family (SG_ENG_* of common.h): out[tag * 5 + {0..4}] = {launches, total ms, algorithmic flops, executed flops, algorithmic bytes}.
// Does NOT reset the log (call it before sg_prof_collect / sg_prof_collect_ex).
extern "C" int sg_prof_collect_tags(double* out, int ntags) {
  for (int k = 0; k < ntags * 5; k++) out[k] = 0.0;
  for (int i = 0; i < g_prof_n; i++) {
    if (g_kind[i] > 1) continue;            // convolution engine only
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_ev0[i], g_ev1[i]) != hipSuccess) { sg_set_error("sg_prof_collect_tags: events not complete (synchronise first)"); return -2; }
    const int k = g_tag[i];
    if (k < 0 || k >= ntags) continue;
    out[k * 5 + 0] += 1.0; out[k * 5 + 1] += ms; out[k * 5 + 2] += g_flops[i]; out[k * 5 + 3] += g_exec[i]; out[k * 5 + 4] += g_bytes[i];
  }
  return 0;
}
// same, four columns per kind: {launches, total ms, algorithmic flops, executed flops}
extern "C" int sg_prof_collect_ex(double* out, int nkinds) {
  for (int k = 0; k < nkinds * 4; k++) out[k] = 0.0;
  for (int i = 0; i < g_prof_n; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_ev0[i], g_ev1[i]) != hipSuccess) { sg_set_error("sg_prof_collect_ex: events not complete (synchronise first)"); return -2; }
    const int k = g_kind[i];
    if (k < 0 || k >= nkinds) continue;
    out[k * 4 + 0] += 1.0; out[k * 4 + 1] += ms; out[k * 4 + 2] += g_flops[i]; out[k * 4 + 3] += g_exec[i];
  }
  g_prof_n = 0;
  return 0;
}
