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
