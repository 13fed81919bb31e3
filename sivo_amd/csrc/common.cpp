#include "common.hpp"

#include <mutex>
#include <stdexcept>

namespace sivo {

std::string &last_error_ref() {
    thread_local std::string msg;
    return msg;
}

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

bool first_use_on_device(int *flags) {
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    std::lock_guard<std::mutex> lock(mu);
    if (flags[dev]) return false;
    flags[dev] = 1;
    return true;
}

}  // namespace sivo

extern "C" const char *sivo_last_error(void) { return sivo::last_error_ref().c_str(); }
extern "C" int sivo_version(void) { return 100; }
extern "C" int sivo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
