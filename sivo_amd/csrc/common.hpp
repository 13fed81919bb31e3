// common.hpp — error plumbing shared by every translation unit of libsivo_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
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

// Restores the device that was current at construction, whatever the scope does with hipSetDevice in between (also
// when it leaves by an exception).
struct DeviceRestore {
    int prev = -1;
    DeviceRestore() { (void)hipGetDevice(&prev); }
    ~DeviceRestore() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceRestore(const DeviceRestore &) = delete;
    DeviceRestore &operator=(const DeviceRestore &) = delete;
};

template <class T>
T *dev_alloc(size_t n) {
    T *p = nullptr;
    SIVO_HIP(hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(T)));
    return p;
}

// `if (FirstUse once(flags); once) { … }` runs the block the first time it is reached on the current device: kernel
// attributes such as the dynamic-LDS opt-in are per device, and a process may drive several (sivo_segnet_create_multi).
// The guard keeps the lock until the block has finished, so a second thread that finds the flag set also finds the
// attributes set; afterwards the check is one atomic load.
class FirstUse {
 public:
    explicit FirstUse(int *flags /* 64 ints, zero-initialised, one array per call site */);
    ~FirstUse();
    FirstUse(const FirstUse &) = delete;
    FirstUse &operator=(const FirstUse &) = delete;
    explicit operator bool() const { return slot_ != nullptr || always_; }

 private:
    int *slot_ = nullptr;      // non-null: this guard holds the lock and publishes the flag on destruction
    bool always_ = false;      // device index unknown: run the block every time, unlocked
};

// Environment switches.  The product library reads NONE (handles take SivoSegnetOptions / setters).  The switches that
// select between kernel forms for A/B measurements, bit-identity tests and fault injection (SIVO_H3_BOOST, SIVO_MULTI_EMULATE ...) exist
// only in libsivo_hip_diag.so (`make diag`: every source compiled with -DSIVO_DIAG) — in the product build the macro is a null pointer
// and the name is not even in the binary.
#ifdef SIVO_DIAG
#define SIVO_DIAG_ENV(name) std::getenv(name)
#else
#define SIVO_DIAG_ENV(name) (static_cast<const char *>(nullptr))
#endif

#ifdef SIVO_DIAG
// Diagnostic build only: 64 words of pinned host memory the SIVO_W4_VERIFY comparison reports into (conv_wino4.hip; read through
// sivo_debug_words by tools/coresident_probe.py and tools/bridge_pair_repro.py): [4] M words that differ between two runs of a GEMM,
// [5] V' words that differ between two runs of its bridge, [6] layers compared, [16..19] geometry of the first differing layer,
// [20..] the first twelve differing V' words.
uint32_t *diag_words();
#endif

// The co-residency mitigation: a kernel that issues LDS-DMA in inline assembly leaves no LDS on its CU for a foreign
// workgroup (one workgroup claiming the CU's 160 KB, or — the classifier at 64 input channels — two of its own claiming 80 KB each).
// Its launcher notes the LDS it asked for PER CU; tests/test_gpu_coresidency.py reads the smallest note per kernel through
// sivo_debug_lds_claims and fails when a launch leaves room beside it.
enum { LDS_CLAIM_GEMM_H3 = 0, LDS_CLAIM_CONV3_H3, LDS_CLAIM_CLS_H3, LDS_CLAIM_CONV7_H3, LDS_CLAIM_KERNELS };
void lds_claim_note(int kernel, size_t bytes_per_cu);
void lds_claims(uint32_t out[LDS_CLAIM_KERNELS], bool reset);        // 0 = no launch since the last reset

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace sivo
