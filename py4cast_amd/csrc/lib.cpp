// Library-wide state of libpy4cast_hip.so: thread-local error message, device queries.
#include <stdarg.h>

#include <mutex>

#include "common.hpp"

namespace p4c {

static thread_local char g_err[512] = "";

char* err_buf() { return g_err; }

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int num_cus() {
    static int cus = 0;
    static std::once_flag once;
    std::call_once(once, [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;  // MI355X
    });
    return cus;
}

}  // namespace p4c

extern "C" int p4c_version(void) { return P4C_VERSION; }
extern "C" const char* p4c_last_error(void) { return p4c::err_buf(); }
extern "C" int p4c_num_cus(void) { return p4c::num_cus(); }
