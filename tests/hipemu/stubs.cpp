// hipemu harness for ONE translation unit of csrc (TEST INFRASTRUCTURE): the few C-ABI symbols the kernel translation units expect from capi.hip
// (error string, launch profiler). The whole-library build (fullemu.py) links the real capi.hip instead.
#include <hip/hip_runtime.h>
#include <string>
static std::string g_err;
extern "C" void sg_set_error(const char* msg) { g_err = msg ? msg : ""; }
extern "C" const char* sg_last_error() { return g_err.c_str(); }
extern "C" int sg_prof_begin(hipStream_t, double, int) { return -1; }
extern "C" void sg_prof_end(hipStream_t, int) {}
extern "C" void sg_prof_set_executed(int, double) {}
extern "C" void sg_prof_tag(int, int, double) {}
