// common.hpp — error plumbing shared by every translation unit of libsivo_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/sivo_hip.h"

namespace sivo {

std::string &last_error_ref();
int fail(int code, const char *fmt, ...);

struct HipError {
    hipError_t e;
    const char *what;
    const char *file;
    int line;
};

#define SIVO_HIP(expr)                                                              \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) throw ::sivo::HipError{_e, #expr, __FILE__, __LINE__}; \
    } while (0)

// Run `body` (a lambda returning int) translating C++ exceptions to status codes.
template <class F>
int guarded(F &&body) {
    try {
        return body();
    } catch (const HipError &h) {
        return fail(SIVO_ERR_RUNTIME, "%s failed: %s (%s:%d)", h.what, hipGetErrorString(h.e), h.file, h.line);
    } catch (const std::invalid_argument &a) {
        return fail(SIVO_ERR_INVALID_ARGUMENT, "%s", a.what());
    } catch (const std::exception &x) {
        return fail(SIVO_ERR_RUNTIME, "%s", x.what());
    }
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        SIVO_HIP(hipGetDevice(&prev));
        if (prev != dev) SIVO_HIP(hipSetDevice(dev));
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

template <class T>
T *dev_alloc(size_t n) {
    T *p = nullptr;
    SIVO_HIP(hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(T)));
    return p;
}

// true the first time it is called on the current device (thread-safe): kernel attributes such as the dynamic-LDS opt-in are
// per device, and a process may drive several (sivo_segnet_create_multi)
bool first_use_on_device(int *flags /* 64 ints, zero-initialised, one array per call site */);

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace sivo
