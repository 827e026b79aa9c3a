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
// sha256 (first 16 hex digits) over the kernel sources this library was built from (csrc/build.sh): the committed rocprofv3
// tables and PMC passes under profiles/ carry the hash of the binary they measured, and bench.py quotes one only for the same hash
#ifndef EMG_SRC_HASH
#define EMG_SRC_HASH "unknown"
#endif
extern "C" const char* emg_source_hash(void) { return EMG_SRC_HASH; }
