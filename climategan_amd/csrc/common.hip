// Library-wide state: ABI version and the thread-local error string.
#include "cgan_common.h"

#include <string>

namespace {
thread_local std::string g_last_error;
}

void cgan_set_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

extern "C" int cgan_version(void) { return CGAN_ABI_VERSION; }
extern "C" const char* cgan_last_error(void) { return g_last_error.c_str(); }
