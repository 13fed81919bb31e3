// segnet_multi.cpp — the Monte-Carlo samples of one frame over several GPUs INSIDE one handle.
//
// The reference constructs ONE BayesianSegNet (reference src/orbslam/System.cc:94-95) and calls segmentImage once per
// frame (src/orbslam/Frame.cc:227-229); a drop-in multi-GPU path therefore has to live behind that object.  A handle made
// by sivo_segnet_create_multi owns one complete network per device (weights replicated, 118 MB), one RCCL communicator
// per device (single process, ncclCommInitAll) and one stream per device.  Per frame:
//   every device    its contiguous share of the T samples (dropout keyed by the GLOBAL sample index, so the maps do not
//                   depend on the number of devices beyond fp32 summation order), the sample-invariant prefix recomputed
//                   locally (no communication), softmax + sum over its samples written pixel-chunk-major
//                   [device chunk][class][pixel in chunk];
//   reduce-scatter  (sum, fp32) — device d receives the probability sums of ITS 1/ndev of the pixels, all classes:
//                   (ndev - 1) / ndev of 21.6 MB per device over xGMI instead of the all-reduce's 2 (ndev - 1) / ndev;
//   every device    mean / argmax / max / entropy (f64) of its pixel chunk;
//   all-gather      of the u8 class chunk and the two f64 chunks (17 B per pixel) -> full maps on every device;
//   device 0        copies the maps to the caller.
// RCCL is opened with dlopen(RTLD_LOCAL) on first use: libsivo_hip.so has no link-time dependency on it, and a process
// that also runs torch.distributed (bench.py at N > 1) keeps exactly one RCCL in its global symbol table.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "common.hpp"
#include "segnet_kernels.hpp"
#include "segnet_multi.hpp"

namespace sivo {
namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

// Opened once, by a throwing factory behind a function-local static (thread-safe, and a failed dlsym leaves nothing half
// filled: the next call tries again).  RTLD_NOLOAD first: a process that already runs RCCL (torch.distributed) keeps ONE.
Rccl load_rccl() {
    Rccl r;
    for (int flags : {RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD, RTLD_NOW | RTLD_LOCAL}) {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, flags);
            if (r.lib) break;
        }
        if (r.lib) break;
    }
    if (!r.lib) throw std::runtime_error(std::string("cannot open librccl.so: ") + dlerror());
    auto sym = [&](const char *n) {
        void *p = dlsym(r.lib, n);
        if (!p) {
            dlclose(r.lib);
            throw std::runtime_error(std::string("librccl.so lacks ") + n);
        }
        return p;
    };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.ReduceScatter = reinterpret_cast<decltype(r.ReduceScatter)>(sym("ncclReduceScatter"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    return r;
}
Rccl &rccl() {
    static Rccl r = load_rccl();
    return r;
}

void nccl_check(ncclResult_t rc, const char *what) {
    if (rc != ncclSuccess) throw std::runtime_error(std::string(what) + " failed: " + rccl().GetErrorString(rc));
}

}  // namespace

struct MultiDevice {
    int device = 0;
    sivo_segnet_t net = nullptr;        // a complete single-device handle (T = the largest share)
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int sample0 = 0, n_samples = 0;
    uint8_t *d_image = nullptr;
    double *d_sum = nullptr;            // [ndev][classes][chunk]: this device's f64 sums for every pixel chunk
    double *d_rs = nullptr;             // [classes][chunk]: all devices' sums for this device's pixel chunk
    unsigned char *d_slots = nullptr;   // [ndev][slot]: the prefix bands of all devices (own slot written here, the others gathered)
    uint8_t *d_cls_chunk = nullptr, *d_cls = nullptr;
    double *d_conf_chunk = nullptr, *d_conf = nullptr, *d_ent_chunk = nullptr, *d_ent = nullptr;
};

struct SegnetMulti {
    // emulate (SIVO_MULTI_EMULATE=1, tests): the "devices" may be one physical GPU several times; the two collectives are
    // then carried out by copies and f64 adds in device order instead of RCCL — everything else (sample shards, chunk-major
    // sums, per-chunk finalize, gathered maps) is the code the real path runs.
    bool emulate = false;
    int T = 0, H = 0, W = 0, classes = 0;
    int64_t hw = 0, chunk = 0;
    size_t slot_bytes = 0;              // > 0: the sample-invariant prefix is split into row bands over the devices (segnet.cpp PrefixBands)
    std::vector<MultiDevice> dev;
    ~SegnetMulti() {
        for (MultiDevice &d : dev) {
            (void)hipSetDevice(d.device);
            if (d.comm) (void)rccl().CommDestroy(d.comm);
            for (void *p : {(void *)d.d_slots, (void *)d.d_image, (void *)d.d_sum, (void *)d.d_rs, (void *)d.d_cls_chunk, (void *)d.d_cls, (void *)d.d_conf_chunk,
                            (void *)d.d_conf, (void *)d.d_ent_chunk, (void *)d.d_ent})
                if (p) (void)hipFree(p);
            if (d.stream) (void)hipStreamDestroy(d.stream);
            if (d.net) (void)sivo_segnet_destroy(d.net);
        }
    }
};

// Same contiguous split as sivo_amd/parallel.py: when T is not a multiple of ndev the LAST T % ndev devices take one more
// (device 0 also serves the host side of the frame).
static void shard(int T, int ndev, int d, int &sample0, int &n) {
    const int base = T / ndev, extra = T % ndev, first_heavy = ndev - extra;
    n = base + (d >= first_heavy ? 1 : 0);
    sample0 = d * base + (d > first_heavy ? d - first_heavy : 0);
}

SegnetMulti *segnet_multi_create(const char *text, size_t len, int t_total, const float *weights, size_t n_weights,
                                 const int *device_ids, int ndev, const SivoSegnetOptions *opts) {
    if (!device_ids || ndev < 1) throw std::invalid_argument("device_ids is empty");
    std::unique_ptr<SegnetMulti> M(new SegnetMulti);
    M->dev.resize((size_t)ndev);
    M->emulate = SIVO_DIAG_ENV("SIVO_MULTI_EMULATE") && std::atoi(SIVO_DIAG_ENV("SIVO_MULTI_EMULATE")) == 1;
    DeviceRestore restore;              // whatever happens below, the caller's device is current again on the way out
    // T from the prototxt unless overridden: build device 0 first to learn the shape
    for (int d = 0; d < ndev; ++d) {
        MultiDevice &D = M->dev[d];
        D.device = device_ids[d];
        for (int e = 0; e < d; ++e)
            if (device_ids[e] == D.device && !M->emulate) throw std::invalid_argument("device_ids holds a device twice (RCCL needs distinct devices)");
    }
    {
        // the reference's constructor checks apply to the TOTAL sample count (bayesian_segnet.cpp:67-70)
        sivo_segnet_t probe = nullptr;
        const int rc = sivo_segnet_create_opts(text, len, t_total, weights, n_weights, device_ids[0], opts, &probe);
        if (rc != SIVO_OK) throw std::invalid_argument(sivo_last_error());
        int32_t T, C, H, W, K;
        sivo_segnet_shape(probe, &T, &C, &H, &W, &K);
        M->T = T; M->H = H; M->W = W; M->classes = K;
        if (ndev == 1) M->dev[0].net = probe; else sivo_segnet_destroy(probe);
    }
    if (M->T < ndev) throw std::invalid_argument("fewer Monte-Carlo samples than devices");
    M->hw = (int64_t)M->H * M->W;
    if (M->hw % ndev) throw std::invalid_argument("H * W must be a multiple of the number of devices");
    M->chunk = M->hw / ndev;
    const int t_alloc = std::max(2, (M->T + ndev - 1) / ndev);
    for (int d = 0; d < ndev; ++d) {
        MultiDevice &D = M->dev[d];
        shard(M->T, ndev, d, D.sample0, D.n_samples);
        if (!D.net) {
            const int rc = sivo_segnet_create_opts(text, len, t_alloc, weights, n_weights, D.device, opts, &D.net);
            if (rc != SIVO_OK) throw std::runtime_error(sivo_last_error());
        }
        SIVO_HIP(hipSetDevice(D.device));
        SIVO_HIP(hipStreamCreateWithFlags(&D.stream, hipStreamNonBlocking));
        D.d_image = dev_alloc<uint8_t>((size_t)M->hw * 3);
        D.d_sum = dev_alloc<double>((size_t)M->classes * M->hw);
        D.d_rs = dev_alloc<double>((size_t)M->classes * M->chunk);
        D.d_cls_chunk = dev_alloc<uint8_t>((size_t)M->chunk); D.d_cls = dev_alloc<uint8_t>((size_t)M->hw);
        D.d_conf_chunk = dev_alloc<double>((size_t)M->chunk); D.d_conf = dev_alloc<double>((size_t)M->hw);
        D.d_ent_chunk = dev_alloc<double>((size_t)M->chunk); D.d_ent = dev_alloc<double>((size_t)M->hw);
    }
    // the prefix in row bands (one all-gather of ~2 MB per device) instead of ndev recomputations; SIVO_MULTI_BANDS=0 (diagnostic
    // build) keeps the recomputation, for the tests that compare the two
    const bool bands_off = SIVO_DIAG_ENV("SIVO_MULTI_BANDS") && std::atoi(SIVO_DIAG_ENV("SIVO_MULTI_BANDS")) == 0;
    if (ndev > 1 && !bands_off) {
        M->slot_bytes = segnet_prefix_slot_bytes(M->dev[0].net, ndev);
        for (int d = 0; d < ndev && M->slot_bytes; ++d) {
            MultiDevice &D = M->dev[d];
            if (segnet_prefix_slot_bytes(D.net, ndev) != M->slot_bytes) throw std::runtime_error("prefix bands: the devices plan differently");
            SIVO_HIP(hipSetDevice(D.device));
            D.d_slots = dev_alloc<unsigned char>((size_t)ndev * M->slot_bytes);
        }
    }
    if (!M->emulate) {
        std::vector<ncclComm_t> comms((size_t)ndev);
        nccl_check(rccl().CommInitAll(comms.data(), ndev, device_ids), "ncclCommInitAll");
        for (int d = 0; d < ndev; ++d) M->dev[d].comm = comms[d];
    }
    return M.release();
}

void segnet_multi_destroy(SegnetMulti *M) { delete M; }

void segnet_multi_shape(const SegnetMulti *M, int32_t *T, int32_t *H, int32_t *W, int32_t *classes, int32_t *ndev) {
    if (T) *T = M->T;
    if (H) *H = M->H;
    if (W) *W = M->W;
    if (classes) *classes = M->classes;
    if (ndev) *ndev = (int)M->dev.size();
}

static void multi_frame(SegnetMulti *M, const uint8_t *bgr, int rows, int cols, uint64_t seed, uint8_t *classes, double *confidence,
                        double *entropy) {
    const int ndev = (int)M->dev.size();
    const int H = M->H, W = M->W;
    const int x_tl = (rows == H && cols == W) ? 0 : cols / 2 - W / 2, y_tl = (rows == H && cols == W) ? 0 : rows / 2 - H / 2;
    const size_t csz = (size_t)M->classes * M->chunk;
    // 1. every device: image up, its samples, chunk-major f64 probability sums
    for (MultiDevice &D : M->dev) {
        SIVO_HIP(hipSetDevice(D.device));
        SIVO_HIP(hipMemcpy2DAsync(D.d_image, (size_t)W * 3, bgr + ((size_t)y_tl * cols + x_tl) * 3, (size_t)cols * 3, (size_t)W * 3, (size_t)H,
                                  hipMemcpyHostToDevice, D.stream));
        if (M->slot_bytes) segnet_prefix_band(D.net, D.d_image, (int)(&D - M->dev.data()), ndev, D.d_slots + (size_t)(&D - M->dev.data()) * M->slot_bytes, D.stream);
        else segnet_forward_chunked(D.net, D.d_image, D.n_samples, D.sample0, seed, D.d_sum, M->chunk, D.stream);
    }
    if (M->slot_bytes) {
        // 1b. all-gather of the prefix bands (in place: every device's own slot sits at its rank's position), then the per-sample part
        if (!M->emulate) {
            Rccl &R = rccl();
            nccl_check(R.GroupStart(), "ncclGroupStart");
            for (int d = 0; d < ndev; ++d) {
                MultiDevice &D = M->dev[d];
                nccl_check(R.AllGather(D.d_slots + (size_t)d * M->slot_bytes, D.d_slots, M->slot_bytes, ncclUint8, D.comm, D.stream), "ncclAllGather");
            }
            nccl_check(R.GroupEnd(), "ncclGroupEnd");
        } else {
            for (MultiDevice &D : M->dev) {
                SIVO_HIP(hipSetDevice(D.device));
                SIVO_HIP(hipStreamSynchronize(D.stream));
            }
            for (int d = 0; d < ndev; ++d) {
                MultiDevice &D = M->dev[d];
                SIVO_HIP(hipSetDevice(D.device));
                for (int e = 0; e < ndev; ++e)
                    if (e != d) SIVO_HIP(hipMemcpyAsync(D.d_slots + (size_t)e * M->slot_bytes, M->dev[e].d_slots + (size_t)e * M->slot_bytes, M->slot_bytes, hipMemcpyDeviceToDevice, D.stream));
            }
        }
        for (MultiDevice &D : M->dev) {
            SIVO_HIP(hipSetDevice(D.device));
            segnet_forward_chunked(D.net, D.d_image, D.n_samples, D.sample0, seed, D.d_sum, M->chunk, D.stream, D.d_slots, ndev);
        }
    }
    if (M->emulate)
        for (MultiDevice &D : M->dev) {
            SIVO_HIP(hipSetDevice(D.device));
            SIVO_HIP(hipStreamSynchronize(D.stream));
        }
    // 2. reduce-scatter over the pixel chunks (f64 sum: (ndev - 1) / ndev of 43 MB per device)
    if (!M->emulate) {
        Rccl &R = rccl();
        nccl_check(R.GroupStart(), "ncclGroupStart");
        for (MultiDevice &D : M->dev)
            nccl_check(R.ReduceScatter(D.d_sum, D.d_rs, csz, ncclDouble, ncclSum, D.comm, D.stream), "ncclReduceScatter");
        nccl_check(R.GroupEnd(), "ncclGroupEnd");
    } else {
        for (int d = 0; d < ndev; ++d) {
            MultiDevice &D = M->dev[d];
            SIVO_HIP(hipSetDevice(D.device));
            for (int e = 0; e < ndev; ++e) launch_add_f64(D.d_rs, M->dev[e].d_sum + (size_t)d * csz, (int64_t)csz, e == 0, D.stream);
        }
    }
    // 3. finalize the own chunk (f64 mean of ALL samples), 4. all-gather the three maps
    for (MultiDevice &D : M->dev) {
        SIVO_HIP(hipSetDevice(D.device));
        launch_mc_finalize64(D.d_rs, M->classes, M->chunk, M->T, D.d_cls_chunk, D.d_conf_chunk, D.d_ent_chunk, D.stream);
    }
    if (!M->emulate) {
        Rccl &R = rccl();
        nccl_check(R.GroupStart(), "ncclGroupStart");
        for (MultiDevice &D : M->dev) {
            nccl_check(R.AllGather(D.d_cls_chunk, D.d_cls, (size_t)M->chunk, ncclUint8, D.comm, D.stream), "ncclAllGather");
            nccl_check(R.AllGather(D.d_conf_chunk, D.d_conf, (size_t)M->chunk, ncclDouble, D.comm, D.stream), "ncclAllGather");
            nccl_check(R.AllGather(D.d_ent_chunk, D.d_ent, (size_t)M->chunk, ncclDouble, D.comm, D.stream), "ncclAllGather");
        }
        nccl_check(R.GroupEnd(), "ncclGroupEnd");
    } else {
        for (MultiDevice &D : M->dev) {
            SIVO_HIP(hipSetDevice(D.device));
            SIVO_HIP(hipStreamSynchronize(D.stream));
        }
        for (MultiDevice &D : M->dev) {
            SIVO_HIP(hipSetDevice(D.device));
            for (int e = 0; e < ndev; ++e) {
                const MultiDevice &E = M->dev[e];
                SIVO_HIP(hipMemcpyAsync(D.d_cls + (size_t)e * M->chunk, E.d_cls_chunk, (size_t)M->chunk, hipMemcpyDeviceToDevice, D.stream));
                SIVO_HIP(hipMemcpyAsync(D.d_conf + (size_t)e * M->chunk, E.d_conf_chunk, (size_t)M->chunk * sizeof(double), hipMemcpyDeviceToDevice, D.stream));
                SIVO_HIP(hipMemcpyAsync(D.d_ent + (size_t)e * M->chunk, E.d_ent_chunk, (size_t)M->chunk * sizeof(double), hipMemcpyDeviceToDevice, D.stream));
            }
        }
    }
    // 5. device 0 hands the maps over
    MultiDevice &D0 = M->dev[0];
    SIVO_HIP(hipSetDevice(D0.device));
    if (classes) SIVO_HIP(hipMemcpyAsync(classes, D0.d_cls, (size_t)M->hw, hipMemcpyDeviceToHost, D0.stream));
    if (confidence) SIVO_HIP(hipMemcpyAsync(confidence, D0.d_conf, (size_t)M->hw * sizeof(double), hipMemcpyDeviceToHost, D0.stream));
    if (entropy) SIVO_HIP(hipMemcpyAsync(entropy, D0.d_ent, (size_t)M->hw * sizeof(double), hipMemcpyDeviceToHost, D0.stream));
    for (MultiDevice &D : M->dev) {
        SIVO_HIP(hipSetDevice(D.device));
        SIVO_HIP(hipStreamSynchronize(D.stream));
    }
}

void segnet_multi_segment(SegnetMulti *M, const uint8_t *bgr, int rows, int cols, uint64_t seed, uint8_t *classes, double *confidence,
                          double *entropy) {
    DeviceRestore restore;              // also when an RCCL or HIP error unwinds mid-frame
    for (int attempt = 0; attempt < 2; ++attempt) {
        multi_frame(M, bgr, rows, cols, seed, classes, confidence, entropy);
        // a value left the fp16 range on some device: EVERY device's handle backs off the same way (the devices must run the
        // same arithmetic for the maps not to depend on the sharding) and the frame is computed once more, without f16x3
        bool tripped = false;
        std::vector<char> backed(M->dev.size(), 0);
        for (size_t d = 0; d < M->dev.size(); ++d) {
            bool b = false;
            tripped = segnet_fp16_overflowed(M->dev[d].net, &b) || tripped;
            backed[d] = b ? 1 : 0;
        }
        if (!tripped) break;
        for (size_t d = 0; d < M->dev.size(); ++d) segnet_fp16_back_off(M->dev[d].net, backed[d] != 0);
    }
}

}  // namespace sivo
