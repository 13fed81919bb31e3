// debug_entry.cpp — diagnostic / test entry points (sivo_debug_*): single kernels on caller-supplied or random data, with launch
// times.  NOT in the product library: libsivo_hip_dbg.so (`make dbg`: this file linked against libsivo_hip.so, whose kernels it
// calls) and the ablation build libsivo_hip_diag.so (`make diag`).  Declarations: include/sivo_hip_debug.h.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../include/sivo_hip_debug.h"
#include "common.hpp"
#include "segnet_kernels.hpp"

using namespace sivo;

// Diagnostic / test: the f16x3 GEMM alone.  V [36][C][Pp] and U [36][C][Kp] fp32 on the host (Pp = P rounded up to 128),
// M [36][Kp][Pp] out; V is packed with vscale, U with the scale wino4_h3_pack_weights chooses, M is scaled back.  iters > 0:
// mean launch time in *ms_out.
extern "C" int sivo_debug_h3_gemm(int C, int Kp, int P, const float *V, const float *U, float vscale, float *M, int iters, double *ms_out) {
    return guarded([&] {
        if (sivo_device_count() < 1) return fail(SIVO_ERR_RUNTIME, "no HIP device");
        if (!V || !U || !M || !wino4_h3_supported(C, Kp) || P < 1 || !(vscale > 0.f)) throw std::invalid_argument("bad argument");
        const int64_t Pp = ((int64_t)P + 127) / 128 * 128;
        const size_t nv = (size_t)36 * C * Pp, nm = (size_t)36 * Kp * Pp;
        std::vector<uint32_t> vp(nv);
        for (size_t i = 0; i < nv; ++i) vp[i] = wino4_h3_pack_value(V[i], vscale);
        std::vector<uint16_t> planes;
        const float uscale = wino4_h3_pack_weights(std::vector<float>(U, U + (size_t)36 * C * Kp), C, Kp, planes);
        uint32_t *dv = dev_alloc<uint32_t>(nv);
        uint16_t *du = dev_alloc<uint16_t>(planes.size());
        float *dm = dev_alloc<float>(nm);
        SIVO_HIP(hipMemcpy(dv, vp.data(), nv * 4, hipMemcpyHostToDevice));
        SIVO_HIP(hipMemcpy(du, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
        SIVO_HIP(hipMemset(dm, 0xff, nm * 4));
        launch_wino4_gemm_h3(dv, du, dm, C, Kp, P, (int)Pp, nullptr);
        SIVO_HIP(hipDeviceSynchronize());
        if (iters > 0 && ms_out) {
            hipEvent_t e0, e1;
            SIVO_HIP(hipEventCreate(&e0)); SIVO_HIP(hipEventCreate(&e1));
            SIVO_HIP(hipEventRecord(e0, nullptr));
            for (int i = 0; i < iters; ++i) launch_wino4_gemm_h3(dv, du, dm, C, Kp, P, (int)Pp, nullptr);
            SIVO_HIP(hipEventRecord(e1, nullptr));
            SIVO_HIP(hipEventSynchronize(e1));
            float ms = 0;
            SIVO_HIP(hipEventElapsedTime(&ms, e0, e1));
            *ms_out = ms / iters;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        std::vector<float> hm(nm);
        SIVO_HIP(hipMemcpy(hm.data(), dm, nm * 4, hipMemcpyDeviceToHost));
        const float inv = 1.f / (vscale * uscale);
        for (size_t i = 0; i < nm; ++i) M[i] = hm[i] * inv;
        (void)hipFree(dv); (void)hipFree(du); (void)hipFree(dm);
        return SIVO_OK;
    });
}

// Diagnostic / test: the direct f16x3 3x3 convolution (conv3_h3.hip) alone.  d_in / d_mask / d_out are device pointers
// (d_mask null: d_in is (N, Cin, H, W); else d_in is the pooled tensor (N, Cin, H/2, W/2) and d_mask its window codes), the
// weights (Caffe layout) and the per-channel affine are host arrays.
extern "C" int sivo_debug_conv3_h3_dev(int N, int Cin, int Cout, int H, int W, const float *d_in, const uint8_t *d_mask,
                                       const float *Wt, const float *scale, const float *shift, int relu, float vscale,
                                       float *d_out, int iters, double *ms_out, int *overflowed) {
    return guarded([&] {
        if (sivo_device_count() < 1) return fail(SIVO_ERR_RUNTIME, "no HIP device");
        if (!d_in || !Wt || !scale || !shift || !d_out || N < 1 || !(vscale > 0.f) || !conv3_h3_supported(3, Cin, Cout, H, W, d_mask != nullptr))
            throw std::invalid_argument("sivo_debug_conv3_h3_dev: bad argument / unsupported shape");
        std::vector<uint16_t> planes;
        const float uscale = conv3_h3_pack_weights(Wt, Cin, Cout, planes);
        uint16_t *du = dev_alloc<uint16_t>(planes.size());
        float *dsc = dev_alloc<float>(Cout), *dsh = dev_alloc<float>(Cout);
        uint32_t *flag = nullptr;
        SIVO_HIP(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
        *flag = 0;
        SIVO_HIP(hipMemcpy(du, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
        SIVO_HIP(hipMemcpy(dsc, scale, Cout * 4, hipMemcpyHostToDevice));
        SIVO_HIP(hipMemcpy(dsh, shift, Cout * 4, hipMemcpyHostToDevice));
        ConvArgs a{};
        const int64_t plane_in = d_mask ? (int64_t)(H / 2) * (W / 2) : (int64_t)H * W;
        a.in = d_in; a.in_sample_stride = (int64_t)Cin * plane_in; a.ep_scale = dsc; a.ep_shift = dsh; a.out = d_out;
        a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.CoutPad = Cout; a.relu = relu; a.drop_site = -1;
        a.unpool_mask = d_mask; a.unpool_mask_stride = d_mask ? (int64_t)Cin * plane_in : 0;
        a.wt_h3 = du; a.h3_vscale = vscale; a.h3_uscale = uscale; a.h3_flag = flag;
        uint32_t *stamps = nullptr;         // diagnostic build + SIVO_D3_STAMPS=1: cycle sums of the kernel's ABL & 64 form
        if (std::getenv("SIVO_D3_STAMPS")) {
            stamps = dev_alloc<uint32_t>(8 + 2 * 1024);      // [0 .. 5] sums, then per workgroup (cycles, stages)
            SIVO_HIP(hipMemset(stamps, 0, 32 + 8 * 1024));
            a.vmax = stamps;
        }
        launch_conv3_h3(a, nullptr);
        SIVO_HIP(hipDeviceSynchronize());
        auto report = [&](const char *what) {
            if (!stamps) return;
            uint32_t h[8];
            SIVO_HIP(hipMemcpy(h, stamps, 32, hipMemcpyDeviceToHost));
            if (h[4]) std::fprintf(stderr, "d3 stamps, %s (cycles per wave and stage): wait %.0f barrier %.0f output %.0f multiply %.0f; longest wave %u cycles; %u wave-stages\n",
                                   what, 16.0 * h[0] / h[4], 16.0 * h[1] / h[4], 16.0 * h[2] / h[4], 16.0 * h[3] / h[4], h[5], h[4]);
            SIVO_HIP(hipMemset(stamps, 0, 32));
        };
        report("first launch");
        if (iters > 0 && ms_out) {
            hipEvent_t e0, e1;
            SIVO_HIP(hipEventCreate(&e0)); SIVO_HIP(hipEventCreate(&e1));
            SIVO_HIP(hipEventRecord(e0, nullptr));
            for (int i = 0; i < iters; ++i) launch_conv3_h3(a, nullptr);
            SIVO_HIP(hipEventRecord(e1, nullptr));
            SIVO_HIP(hipEventSynchronize(e1));
            float ms = 0;
            SIVO_HIP(hipEventElapsedTime(&ms, e0, e1));
            *ms_out = ms / iters;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            report("timed launches");
        }
        if (overflowed) *overflowed = (int)*flag;
        (void)hipFree(du); (void)hipFree(dsc); (void)hipFree(dsh); (void)hipHostFree(flag); (void)hipFree(stamps);
        return SIVO_OK;
    });
}

// Diagnostic: time one convolution shape in isolation (random data), `variant` switches parts of
// the kernel off (see ConvArgs::variant).  Returns the mean launch time in ms.
extern "C" int sivo_debug_conv(int N, int Cin, int Cout, int H, int W, int ks, int iters, int variant, double *ms_out) {
    return guarded([&] {
        if (sivo_device_count() < 1) return fail(SIVO_ERR_RUNTIME, "no HIP device");
        const bool c7x6 = (variant & 65536) && conv7_x6_supported(ks, Cin, Cout, H, W);
        const bool wino4f = (variant & 1024) && wino4f_supported(ks, Cin, Cout, H, W);
        const bool wino4 = !wino4f && (variant & 512) && wino4_supported(ks, Cin, Cout, H, W);
        const bool wino = !wino4 && !wino4f && (variant & 64) && wino_supported(ks, Cin, Cout, H, W);
        const int wcfg = (variant & 128) ? ((variant & 32) ? 2 : 1) : 0;
        const bool v2 = !wino && (variant & 16) && conv2_supported(ks);
        const int KC = v2 ? 4 : conv_k_chunk(ks, Cin), BN = conv_cout_tile(ks, Cout);
        const int cout_pad = cdiv(Cout, BN) * BN, nchunks = cdiv(Cin, KC);
        const size_t nin = (size_t)N * Cin * H * W, nout = (size_t)N * Cout * H * W;
        const int w4group = wino4 ? wino4_group(N, Cin, Cout, H, W, (size_t)16384 << 20) : 0;
        const size_t nw = wino4f ? (size_t)((Cin + 3) / 4) * (Cout / 64) * wino4f_slab_floats() : wino4 ? (size_t)36 * Cin * wino4_cout_pad(Cout) : wino ? (size_t)wino_chunks(wcfg, Cin) * (Cout / wino_cout_tile(wcfg)) * wino_slab_floats(wcfg) : v2 ? (size_t)nchunks * (cout_pad / BN) * conv2_slab_floats(ks, Cout) : (size_t)nchunks * ks * ks * KC * cout_pad;
        std::vector<float> hin(nin), hw(nw), hs(Cout, 1.f);
        uint32_t st = 12345;
        auto rnd = [&] { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto &v : hin) v = rnd();
        for (auto &v : hw) v = rnd() * 0.05f;
        float *din = dev_alloc<float>(nin), *dout = dev_alloc<float>(nout), *dw = dev_alloc<float>(nw), *ds = dev_alloc<float>(Cout);
        SIVO_HIP(hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice));
        SIVO_HIP(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
        SIVO_HIP(hipMemcpy(ds, hs.data(), Cout * 4, hipMemcpyHostToDevice));
        ConvArgs a{};
        a.in = din; a.in_sample_stride = (int64_t)Cin * H * W; a.wt = dw; a.ep_scale = ds; a.ep_shift = ds; a.out = dout;
        a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.CoutPad = cout_pad; a.relu = 1; a.drop_site = -1; a.variant = variant;
        hipEvent_t e0, e1;
        SIVO_HIP(hipEventCreate(&e0)); SIVO_HIP(hipEventCreate(&e1));
        float *dws = wino4 ? dev_alloc<float>(wino4_workspace_floats(w4group, Cin, Cout, H, W)) : nullptr;
        if (wino4) a.CoutPad = wino4_cout_pad(Cout);
        void *dx6 = nullptr;
        if (wino4 && (variant & 2048) && wino4_x6_supported(Cin, a.CoutPad)) {      // bf16x6 GEMM on the same (random) U
            std::vector<uint16_t> planes;
            wino4_x6_pack_weights(hw, Cin, a.CoutPad, planes);
            dx6 = dev_alloc<uint16_t>(planes.size());
            SIVO_HIP(hipMemcpy(dx6, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
            a.wt_x6 = dx6;
        }
        if (wino4f) a.CoutPad = Cout;
        void *d7 = nullptr;
        if (c7x6) {
            std::vector<float> w7((size_t)Cout * Cin * 49);
            for (auto &v : w7) v = rnd() * 0.05f;
            std::vector<uint16_t> planes;
            conv7_x6_pack_weights(w7.data(), Cin, Cout, planes);
            d7 = dev_alloc<uint16_t>(planes.size());
            SIVO_HIP(hipMemcpy(d7, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
            a.wt_x6 = d7;
        }
        auto go = [&] { if (c7x6) launch_conv7_x6(a, nullptr); else if (wino4f) launch_conv_wino4f(a, nullptr); else if (wino4) launch_conv_wino4(a, dws, w4group, nullptr); else if (wino) launch_conv_wino(a, wcfg, nullptr); else if (v2) launch_conv2(a, ks, nullptr); else launch_conv(a, ks, nullptr); };
        if (wino) a.CoutPad = Cout;
        for (int i = 0; i < 2; ++i) go();
        SIVO_HIP(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) go();
        SIVO_HIP(hipEventRecord(e1, nullptr));
        SIVO_HIP(hipEventSynchronize(e1));
        float ms = 0;
        SIVO_HIP(hipEventElapsedTime(&ms, e0, e1));
        *ms_out = ms / iters;
        (void)hipFree(din); (void)hipFree(dout); (void)hipFree(dw); (void)hipFree(ds); (void)hipFree(dws); (void)hipFree(dx6); (void)hipFree(d7);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        return SIVO_OK;
    });
}

// The direct f16x3 kernel with packed input and / or output (see include/sivo_hip_debug.h).
extern "C" int sivo_debug_conv3_h3_pk_dev(int N, int Cin, int Cout, int H, int W, const float *d_in, const uint8_t *d_mask, const float *Wt,
                                          const float *scale, const float *shift, int relu, float vscale, float out_vscale, int mode,
                                          float *d_out, int iters, double *ms_out, int *overflowed) {
    return guarded([&] {
        if (sivo_device_count() < 1) return fail(SIVO_ERR_RUNTIME, "no HIP device");
        const bool pk_in = mode & 1, pk_out = mode & 2, unpool = d_mask != nullptr;
        if (!d_in || !Wt || !scale || !shift || !d_out || N < 1 || !(vscale > 0.f) || (pk_out && !(out_vscale > 0.f)) ||
            !conv3_h3_supported(3, Cin, Cout, H, W, unpool))
            throw std::invalid_argument("sivo_debug_conv3_h3_pk_dev: bad argument / unsupported shape");
        std::vector<uint16_t> planes;
        const float uscale = conv3_h3_pack_weights(Wt, Cin, Cout, planes);
        uint16_t *du = dev_alloc<uint16_t>(planes.size());
        float *dsc = dev_alloc<float>(Cout), *dsh = dev_alloc<float>(Cout);
        uint32_t *flag = nullptr;
        SIVO_HIP(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
        *flag = 0;
        SIVO_HIP(hipMemcpy(du, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
        SIVO_HIP(hipMemcpy(dsc, scale, Cout * 4, hipMemcpyHostToDevice));
        SIVO_HIP(hipMemcpy(dsh, shift, Cout * 4, hipMemcpyHostToDevice));
        const int tiles_x = (W + 63) / 64, tiles_y = (H + 7) / 8, extra_h = (mode & 4) ? 3 : 0, extra_w = (mode & 4) ? 5 : 0;
        const int h_in = unpool ? H / 2 : H, w_in = unpool ? W / 2 : W;
        ConvArgs a{};
        const int64_t plane_in = (int64_t)h_in * w_in;
        a.in = d_in; a.in_sample_stride = (int64_t)Cin * plane_in; a.ep_scale = dsc; a.ep_shift = dsh; a.out = d_out;
        a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.CoutPad = Cout; a.relu = relu; a.drop_site = -1;
        a.unpool_mask = d_mask; a.unpool_mask_stride = d_mask ? (int64_t)Cin * plane_in : 0;
        a.wt_h3 = du; a.h3_vscale = vscale; a.h3_uscale = uscale; a.h3_flag = flag;
        void *d_pin = nullptr, *d_pout = nullptr;
        uint32_t *d_bits = nullptr;
        if (pk_in) {
            a.in_Hp = (unpool ? tiles_y * 4 : tiles_y * 8) + 2 + extra_h;
            a.in_Wp = (unpool ? tiles_x * 32 : tiles_x * 64) + 2 + extra_w;
            const size_t nb = pk_bytes(N, Cin, a.in_Hp, a.in_Wp);
            SIVO_HIP(hipMalloc(&d_pin, nb));
            SIVO_HIP(hipMemset(d_pin, 0, nb));
            launch_pk_pack(d_in, a.in_sample_stride, d_pin, N, Cin, h_in, w_in, a.in_Hp, a.in_Wp, vscale, flag, nullptr);
            a.in_pk = d_pin; a.in_pk_sample_bytes = (int64_t)pk_bytes(1, Cin, a.in_Hp, a.in_Wp);
            a.in = nullptr; a.unpool_mask = nullptr;
            if (unpool) {
                const size_t nd = (size_t)N * (Cin / 8) * a.in_Hp * a.in_Wp;
                d_bits = dev_alloc<uint32_t>(nd);
                SIVO_HIP(hipMemset(d_bits, 0, nd * 4));
                launch_pool_bits(d_mask, d_bits, N, Cin, h_in, w_in, a.in_Hp, a.in_Wp, nullptr);
                a.unpool_bits = d_bits; a.unpool_bits_stride = (int64_t)(Cin / 8) * a.in_Hp * a.in_Wp;
            }
        }
        if (pk_out) {
            a.out_Hp = tiles_y * 8 + 2 + extra_h; a.out_Wp = tiles_x * 64 + 2 + extra_w;
            const size_t nb = pk_bytes(N, Cout, a.out_Hp, a.out_Wp);
            SIVO_HIP(hipMalloc(&d_pout, nb));
            SIVO_HIP(hipMemset(d_pout, 0, nb));
            a.out_pk = d_pout; a.out_vscale = out_vscale; a.out = nullptr;
        }
        uint32_t *stamps = nullptr;         // diagnostic build + SIVO_D3_STAMPS=1: cycle sums of the kernel's ABL & 64 form
        if (std::getenv("SIVO_D3_STAMPS")) {
            stamps = dev_alloc<uint32_t>(8 + 2 * 1024);
            SIVO_HIP(hipMemset(stamps, 0, 32 + 8 * 1024));
            a.vmax = stamps;
        }
        launch_conv3_h3(a, nullptr);
        SIVO_HIP(hipDeviceSynchronize());
        if (stamps) SIVO_HIP(hipMemset(stamps, 0, 32 + 8 * 1024));
        if (iters > 0 && ms_out) {
            hipEvent_t e0, e1;
            SIVO_HIP(hipEventCreate(&e0)); SIVO_HIP(hipEventCreate(&e1));
            SIVO_HIP(hipEventRecord(e0, nullptr));
            for (int i = 0; i < iters; ++i) launch_conv3_h3(a, nullptr);
            SIVO_HIP(hipEventRecord(e1, nullptr));
            SIVO_HIP(hipEventSynchronize(e1));
            float ms = 0;
            SIVO_HIP(hipEventElapsedTime(&ms, e0, e1));
            *ms_out = ms / iters;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            if (stamps) {
                uint32_t h[8];
                SIVO_HIP(hipMemcpy(h, stamps, 32, hipMemcpyDeviceToHost));
                if (h[4]) std::fprintf(stderr, "d3 stamps (cycles per wave and stage): wait %.0f barrier %.0f output %.0f multiply %.0f; longest wave %u cycles; %u wave-stages\n",
                                       16.0 * h[0] / h[4], 16.0 * h[1] / h[4], 16.0 * h[2] / h[4], 16.0 * h[3] / h[4], h[5], h[4]);
                // per workgroup of the LAST launch: cycles from start to end, stages — by XCD (workgroup b runs on XCD b % 8)
                std::vector<uint32_t> wg(2 * 1024);
                SIVO_HIP(hipMemcpy(wg.data(), stamps + 8, wg.size() * 4, hipMemcpyDeviceToHost));
                for (int x = 0; x < 8; ++x) {
                    double cmin = 1e30, cmax = 0, csum = 0, ssum = 0; int n = 0;
                    for (int b2 = x; b2 < 1024; b2 += 8) {
                        if (!wg[2 * b2 + 1]) continue;
                        const double c = wg[2 * b2], st = wg[2 * b2 + 1];
                        cmin = std::fmin(cmin, c / st); cmax = std::fmax(cmax, c / st); csum += c; ssum += st; ++n;
                    }
                    if (n) std::fprintf(stderr, "  xcd %d: %d workgroups, %.0f stages each on average, cycles per stage mean %.0f min %.0f max %.0f, longest workgroup %.0f cycles\n", x, n, ssum / n, csum / ssum, cmin, cmax, [&] { double m = 0; for (int b2 = x; b2 < 1024; b2 += 8) m = std::fmax(m, wg[2 * b2]); return m; }());
                }
            }
        }
        (void)hipFree(stamps);
        if (pk_out) {
            // the border of the packed output must still be zero: count what is not (returned through *overflowed bit 1)
            launch_pk_unpack(d_pout, d_out, N, Cout, H, W, a.out_Hp, a.out_Wp, out_vscale, nullptr);
            SIVO_HIP(hipDeviceSynchronize());
        }
        int border_dirty = 0;
        if (pk_out) {
            std::vector<uint16_t> host(pk_bytes(N, Cout, a.out_Hp, a.out_Wp) / 2);
            SIVO_HIP(hipMemcpy(host.data(), d_pout, host.size() * 2, hipMemcpyDeviceToHost));
            const size_t planes_n = (size_t)N * (Cout / 8) * 2;
            for (size_t pl = 0; pl < planes_n && !border_dirty; ++pl)
                for (int y = 0; y < a.out_Hp && !border_dirty; ++y)
                    for (int x = 0; x < a.out_Wp; ++x) {
                        if (y >= 1 && y <= H && x >= 1 && x <= W) { x = W; continue; }
                        const uint16_t *pc = host.data() + ((pl * a.out_Hp + y) * a.out_Wp + x) * 8;
                        for (int e = 0; e < 8; ++e) if (pc[e]) border_dirty = 1;
                    }
        }
        if (overflowed) *overflowed = (int)*flag | (border_dirty << 1);
        (void)hipFree(du); (void)hipFree(dsc); (void)hipFree(dsh); (void)hipHostFree(flag); (void)hipFree(d_pin); (void)hipFree(d_pout); (void)hipFree(d_bits);
        return SIVO_OK;
    });
}

// The f16x3 classifier + MC kernel alone (see include/sivo_hip_debug.h).
extern "C" int sivo_debug_conv_cls_h3_dev(int T, int Cin, int C, int H, int W, const float *d_in, const float *Wt, const float *scale,
                                          const float *shift, int relu, float vscale, float *d_logits, uint8_t *d_classes,
                                          double *d_confidence, double *d_entropy, int iters, double *ms_out) {
    return guarded([&] {
        if (sivo_device_count() < 1) return fail(SIVO_ERR_RUNTIME, "no HIP device");
        if (!d_in || !Wt || !scale || !shift || !d_logits || !d_classes || !d_confidence || !d_entropy || T < 1 || !(vscale > 0.f) ||
            !cls_h3_supported(3, Cin, C, H, W))
            throw std::invalid_argument("sivo_debug_conv_cls_h3_dev: bad argument / unsupported shape");
        std::vector<uint16_t> planes;
        const float uscale = cls_h3_pack_weights(Wt, Cin, C, planes);
        uint16_t *du = dev_alloc<uint16_t>(planes.size());
        float *dsc = dev_alloc<float>(16), *dsh = dev_alloc<float>(16);
        SIVO_HIP(hipMemcpy(du, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
        SIVO_HIP(hipMemset(dsc, 0, 64)); SIVO_HIP(hipMemset(dsh, 0, 64));
        SIVO_HIP(hipMemcpy(dsc, scale, C * 4, hipMemcpyHostToDevice));
        SIVO_HIP(hipMemcpy(dsh, shift, C * 4, hipMemcpyHostToDevice));
        int th, tw;
        cls_h3_tile(&th, &tw);
        ClsMcArgs a{};
        a.in_Hp = (H + th - 1) / th * th + 2; a.in_Wp = (W + tw - 1) / tw * tw + 2;
        const size_t nb = pk_bytes(T, Cin, a.in_Hp, a.in_Wp);
        void *d_pin = nullptr;
        SIVO_HIP(hipMalloc(&d_pin, nb));
        SIVO_HIP(hipMemset(d_pin, 0, nb));
        launch_pk_pack(d_in, (int64_t)Cin * H * W, d_pin, T, Cin, H, W, a.in_Hp, a.in_Wp, vscale, nullptr, nullptr);
        a.in_pk = d_pin; a.in_pk_sample_bytes = (int64_t)pk_bytes(1, Cin, a.in_Hp, a.in_Wp);
        a.wt_h3 = du; a.h3_vscale = vscale; a.h3_uscale = uscale;
        a.ep_scale = dsc; a.ep_shift = dsh;
        a.T = T; a.Cin = Cin; a.H = H; a.W = W; a.C = C; a.relu = relu;
        a.logits = d_logits; a.classes = d_classes; a.confidence = d_confidence; a.entropy = d_entropy;
        launch_conv_cls_h3(a, nullptr);
        SIVO_HIP(hipDeviceSynchronize());
        if (iters > 0 && ms_out) {
            a.logits = nullptr;
            hipEvent_t e0, e1;
            SIVO_HIP(hipEventCreate(&e0)); SIVO_HIP(hipEventCreate(&e1));
            SIVO_HIP(hipEventRecord(e0, nullptr));
            for (int i = 0; i < iters; ++i) launch_conv_cls_h3(a, nullptr);
            SIVO_HIP(hipEventRecord(e1, nullptr));
            SIVO_HIP(hipEventSynchronize(e1));
            float ms = 0;
            SIVO_HIP(hipEventElapsedTime(&ms, e0, e1));
            *ms_out = ms / iters;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        (void)hipFree(du); (void)hipFree(dsc); (void)hipFree(dsh); (void)hipFree(d_pin);
        return SIVO_OK;
    });
}

// What the launchers of the inline-asm LDS-DMA kernels asked for per CU since the last reset (common.hpp lds_claim_note): out[0] the
// f16x3 GEMM, [1] the direct f16x3 3x3 kernel, [2] the f16x3 classifier (per CU: both of its workgroups), [3] the f16x3 7x7 kernel;
// the smallest request of each, 0 = not launched.
extern "C" int sivo_debug_lds_claims(uint32_t out[4], int reset) {
    return sivo::guarded([&] {
        if (!out) throw std::invalid_argument("null argument");
        sivo::lds_claims(out, reset != 0);
        return SIVO_OK;
    });
}

#ifdef SIVO_DIAG
// diagnostic build: the two-kernel reproducer of DESIGN 3.3 — no network, no transforms but the bridge.  `lanes` streams, each with buffers
// of its own, run ONE bridged F(4x4) layer of n samples, C -> C channels at H x W over and over: f16x3 GEMM (V -> M), bridge (M -> V'),
// enqueued round-robin from this thread as the engine enqueues its lanes, so that one lane's bridge workgroups share CUs with another
// lane's GEMM.  With SIVO_W4_VERIFY=1 in the environment launch_conv_wino4 runs GEMM and bridge a second time into scratch buffers and
// compares word for word (diag words [4] M words, [5] V' words that differ, [6] layers compared).  V and the weights are random;
// SIVO_H3_LDS_ALL=0 gives the GEMM its exact LDS.  out: [0] layers run per lane, [1] 1 if an overflow flag was raised.
extern "C" int sivo_debug_bridge_pair(int lanes, int n, int C, int H, int W, int rounds, uint32_t out[2]) {
    return sivo::guarded([&] {
        if (sivo_device_count() < 1) return fail(SIVO_ERR_RUNTIME, "no HIP device");
        if (lanes < 1 || lanes > 4 || n < 1 || !wino4_h3_supported(C, C) || H < 4 || W % 4 || rounds < 1) throw std::invalid_argument("bad argument");
        const int th = (H + 3) / 4, tw = W / 4;
        const int64_t P = (int64_t)n * th * tw, Pp = (P + 127) / 128 * 128;
        const size_t nv = (size_t)36 * C * Pp;
        std::vector<uint32_t> vp(nv);
        uint64_t st = 0x9E3779B97F4A7C15ull;
        auto rnd = [&] { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((int64_t)(st >> 40) - (1 << 23)) * (1.f / (1 << 23)); };   // [-1, 1)
        for (size_t i = 0; i < nv; ++i) vp[i] = wino4_h3_pack_value(8.f * rnd(), 16.f);
        std::vector<float> U((size_t)36 * C * C);
        for (auto &u : U) u = 0.05f * rnd();
        std::vector<uint16_t> planes;
        const float uscale = wino4_h3_pack_weights(U, C, C, planes);
        std::vector<float> eps((size_t)C, 0.05f), sh((size_t)C, 0.01f);
        struct Lane { hipStream_t s; uint32_t *v; uint16_t *u; float *m, *vn, *sc, *sh; uint32_t *flag; };
        std::vector<Lane> L((size_t)lanes);
        for (auto &l : L) {
            SIVO_HIP(hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking));
            l.v = dev_alloc<uint32_t>(nv); l.u = dev_alloc<uint16_t>(planes.size()); l.m = dev_alloc<float>(nv); l.vn = dev_alloc<float>(nv);
            l.sc = dev_alloc<float>((size_t)C); l.sh = dev_alloc<float>((size_t)C); l.flag = dev_alloc<uint32_t>(1);
            SIVO_HIP(hipMemcpy(l.v, vp.data(), nv * 4, hipMemcpyHostToDevice));
            SIVO_HIP(hipMemcpy(l.u, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
            SIVO_HIP(hipMemcpy(l.sc, eps.data(), (size_t)C * 4, hipMemcpyHostToDevice));
            SIVO_HIP(hipMemcpy(l.sh, sh.data(), (size_t)C * 4, hipMemcpyHostToDevice));
            SIVO_HIP(hipMemset(l.flag, 0, 4));
            SIVO_HIP(hipMemset(l.m, 0, nv * 4)); SIVO_HIP(hipMemset(l.vn, 0, nv * 4));
        }
        SIVO_HIP(hipDeviceSynchronize());
        for (int r = 0; r < rounds; ++r)
            for (auto &l : L) {
                ConvArgs c{};
                c.in = nullptr; c.in_sample_stride = 0; c.wt = nullptr; c.ep_scale = l.sc; c.ep_shift = l.sh; c.out = nullptr;
                c.N = n; c.Cin = C; c.H = H; c.W = W; c.Cout = C; c.CoutPad = C; c.relu = 1; c.drop_site = -1; c.sample0 = 0; c.seed = 1;
                c.wt_h3 = l.u; c.h3_vscale = 16.f; c.h3_uscale = uscale; c.h3_flag = l.flag;
                Wino4Plan plan{};
                plan.V = reinterpret_cast<float *>(l.v); plan.M = l.m; plan.Vnext = l.vn; plan.skip_input = true; plan.bridge = true; plan.next_vscale = 16.f;
                launch_conv_wino4(c, nullptr, n, l.s, nullptr, false, &plan);
            }
        SIVO_HIP(hipDeviceSynchronize());
        uint32_t any = 0;
        for (auto &l : L) {
            uint32_t f = 0;
            SIVO_HIP(hipMemcpy(&f, l.flag, 4, hipMemcpyDeviceToHost));
            any |= f;
            (void)hipFree(l.v); (void)hipFree(l.u); (void)hipFree(l.m); (void)hipFree(l.vn); (void)hipFree(l.sc); (void)hipFree(l.sh); (void)hipFree(l.flag);
            (void)hipStreamDestroy(l.s);
        }
        if (out) { out[0] = (uint32_t)rounds; out[1] = any; }
        return SIVO_OK;
    });
}
// diagnostic build: the 64 report words of sivo::diag_words() (common.hpp); reset != 0 clears them after the read
extern "C" int sivo_debug_words(uint32_t out[64], int reset) {
    return sivo::guarded([&] {
        uint32_t *w = sivo::diag_words();
        for (int i = 0; i < 64; ++i) { if (out) out[i] = w[i]; if (reset) w[i] = 0; }
        return SIVO_OK;
    });
}
#endif
