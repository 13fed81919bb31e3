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

static std::mutex &first_use_mutex() {
    static std::mutex mu;
    return mu;
}

FirstUse::FirstUse(int *flags) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        always_ = true;
        return;
    }
    if (__atomic_load_n(&flags[dev], __ATOMIC_ACQUIRE)) return;
    first_use_mutex().lock();
    if (__atomic_load_n(&flags[dev], __ATOMIC_RELAXED)) {
        first_use_mutex().unlock();
        return;
    }
    slot_ = &flags[dev];
}

FirstUse::~FirstUse() {
    if (!slot_) return;
    __atomic_store_n(slot_, 1, __ATOMIC_RELEASE);
    first_use_mutex().unlock();
}

static uint32_t g_lds_claim[LDS_CLAIM_KERNELS] = {0, 0, 0, 0};
void lds_claim_note(int kernel, size_t bytes_per_cu) {
    const uint32_t b = (uint32_t)bytes_per_cu;
    uint32_t cur = __atomic_load_n(&g_lds_claim[kernel], __ATOMIC_RELAXED);
    while ((cur == 0 || b < cur) && !__atomic_compare_exchange_n(&g_lds_claim[kernel], &cur, b, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
void lds_claims(uint32_t out[LDS_CLAIM_KERNELS], bool reset) {
    for (int k = 0; k < LDS_CLAIM_KERNELS; ++k) out[k] = reset ? __atomic_exchange_n(&g_lds_claim[k], 0u, __ATOMIC_RELAXED) : __atomic_load_n(&g_lds_claim[k], __ATOMIC_RELAXED);
}

#ifdef SIVO_DIAG
uint32_t *diag_words() {
    static uint32_t *w = [] {
        uint32_t *p = nullptr;
        if (hipHostMalloc((void **)&p, 64 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) throw std::runtime_error("diag_words: hipHostMalloc");
        for (int i = 0; i < 64; ++i) p[i] = 0;
        return p;
    }();
    return w;
}
#endif

}  // namespace sivo

extern "C" const char *sivo_last_error(void) { return sivo::last_error_ref().c_str(); }
extern "C" int sivo_version(void) { return 100; }
extern "C" int sivo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
