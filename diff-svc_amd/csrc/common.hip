// Error plumbing shared by every translation unit of libdsvc_hip.so.
#include <stdarg.h>

#include "../../include/dsvc.h"
#include "common.h"

namespace dsvc {
static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
}  // namespace dsvc

extern "C" const char* dsvc_last_error(void) { return dsvc::g_last_error.c_str(); }
