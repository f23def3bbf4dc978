// Error plumbing and probes of the C ABI (include/gdm.h).
#include "gdm_common.h"

static thread_local char g_err[512] = "";

void gdm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* gdm_last_error(void) { return g_err; }
extern "C" int gdm_version(void) { return 1; }
extern "C" const char* gdm_arch(void) { return "gfx950"; }
#ifdef GDM_EXPERIMENT_BUILD
extern "C" int gdm_build_flavor(void) { return 1; }
#else
extern "C" int gdm_build_flavor(void) { return 0; }
#endif
