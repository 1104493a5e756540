// Error string + ABI version for the C-ABI library.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void gpt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* gpt_last_error(void) { return g_err; }
extern "C" int gpt_abi_version(void) { return 4; }
