#include "common.hpp"

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

}  // namespace sivo

extern "C" const char *sivo_last_error(void) { return sivo::last_error_ref().c_str(); }
extern "C" int sivo_version(void) { return 100; }
extern "C" int sivo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
