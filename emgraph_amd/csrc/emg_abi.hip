// emg_abi.hip — ABI plumbing: version, thread-local error string.
#include <string>

#include "emg_common.hpp"

namespace emg {

static thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

}  // namespace emg

extern "C" int emg_version(void) { return EMG_ABI_VERSION; }
extern "C" const char* emg_last_error(void) { return emg::g_last_error.c_str(); }
extern "C" const char* emg_target(void) { return "gfx950"; }
