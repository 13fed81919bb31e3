// segnet_kernels.hpp — launch interface of segnet_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace sivo {

struct ConvArgs {
    const float *in;           // (N or 1, Cin, H, W)
    int64_t in_sample_stride;  // Cin*H*W, or 0 when the input is shared by all samples
    const float *wt;           // [ceil(Cin/KC)][k*k][KC][CoutPad], zero padded
    const float *ep_scale;     // [Cout]  y = ep_scale*acc + ep_shift   (bias + BN folded)
    const float *ep_shift;     // [Cout]
    float *out;                // (N, Cout, H, W)
    int N, Cin, H, W, Cout, CoutPad;
    int tiles_x, tiles_y;      // filled by the launcher
    int relu;
    int drop_site;             // < 0: no dropout in the epilogue
    // F(4x4,3x3) three-kernel path only: >= 0: the INPUT is a sample-invariant tensor (in_sample_stride 0) that the network drops out
    // per sample in front of this layer (the fork pooling's Dropout): the input transform applies that dropout as it reads — the
    // counter-based word of (element, site, global sample) the pooling kernel would have used — and the T dropped copies never exist
    int in_drop_site = -1;
    int sample0;
    uint64_t seed;
    // F(4x4,3x3) path only: when set, `in` is the POOLED tensor (N or 1, Cin, H/2, W/2) and the kernel reads the
    // max-unpooled input through it (value at the recorded 2x2 window position, zero elsewhere): the Upsample
    // layer in front of the convolution is never materialised
    const uint8_t *unpool_mask = nullptr;
    int64_t unpool_mask_stride = 0;   // 0 when the mask is shared by all samples
    // F(4x4,3x3) three-kernel path only: when set, the output transform applies the MAX 2x2 pooling that follows (window
    // codes + the pooling layer's dropout) and writes these instead of `out`
    float *pool_out = nullptr;
    uint8_t *pool_mask = nullptr;
    int pool_drop_site = -1;
    const void *wt_x6 = nullptr;   // F(4x4,3x3) three-kernel path: the weights split into three bf16 planes (bf16x6 GEMM), or null
    // f16x3 GEMM (conv_wino4_h3.hip): the weights as fp16 hi / lo planes scaled by h3_uscale, the power of two this layer's
    // transformed input is multiplied with before it is split (0 = run the layer on the bf16x6 / fp32 GEMM), the flag an
    // out-of-range value raises, and (calibration passes) where the layer's largest |V| is recorded
    const void *wt_h3 = nullptr;
    float h3_vscale = 0.f, h3_uscale = 1.f;
    uint32_t *h3_flag = nullptr;
    uint32_t *vmax = nullptr;
    // conv3_h3.hip, packed activations: fp16 hi / lo planes [N][C / 8][plane][Hp][Wp][8], the image at rows / columns 1.. of a
    // zero-bordered (Hp, Wp) plane, values times the CONSUMER's power of two (pk_format.hip packs / unpacks).
    //   in_pk: this layer's input in that form (scaled by h3_vscale); with unpool_bits the POOLED tensor the layer reads through
    //          an Upsample, unpool_bits = [C / 8][Hp][Wp] dwords, byte k = channels whose maximum sat at window position k
    //   out_pk: write the output in that form for the next layer (times out_vscale) instead of `out`
    const void *in_pk = nullptr;
    int64_t in_pk_sample_bytes = 0;        // 0 when the input is shared by all samples
    int in_Hp = 0, in_Wp = 0;
    const uint32_t *unpool_bits = nullptr;
    int64_t unpool_bits_stride = 0;        // dwords per sample; 0 when shared
    void *out_pk = nullptr;
    int out_Hp = 0, out_Wp = 0;
    float out_vscale = 0.f;
    int variant;               // diagnostics only (sivo_debug_conv): bit0 no epilogue stores, bit1 no LDS commit, bit2 no global loads
};
int conv_cout_tile(int ks, int cout);  // BN the launcher will pick (CoutPad must be a multiple)
int conv_k_chunk(int ks, int cin);     // KC the launcher will pick (defines the weight layout)
void launch_conv(const ConvArgs &a, int ks, hipStream_t s);
// second-generation kernel (conv_v2.hip): KC = 4, double-buffered LDS, LDS-DMA weight slabs
bool conv2_supported(int ks);
int conv2_slab_floats(int ks, int cout);
void conv2_pack_weights(const float *W, int ks, int cin, int cout, std::vector<float> &out, int *cout_pad);
void launch_conv2(const ConvArgs &a, int ks, hipStream_t s);
// Winograd F(2x2,3x3) kernel (conv_wino.hip)
bool wino_supported(int ks, int cin, int cout, int H, int W);
void wino_pack_weights(const float *W, int cin, int cout, int cfg, std::vector<float> &out, int *cout_pad);
void launch_conv_wino(const ConvArgs &a, int cfg, hipStream_t s);
int wino_slab_floats(int cfg);
int wino_chunks(int cfg, int cin);
int wino_cout_tile(int cfg);

// Winograd F(4x4,3x3): input transform + batched GEMM + output transform (conv_wino4.hip)
bool wino4_supported(int ks, int cin, int cout, int H, int W);
int wino4_cout_pad(int cout);
void wino4_pack_weights(const float *W, int cin, int cout, std::vector<float> &out, int *cout_pad);
int wino4_group(int N, int cin, int cout, int H, int W, size_t budget_bytes);
// bf16x6 GEMM (fp32 operands split into three bf16 planes, six products, fp32 accumulate)
bool wino4_x6_supported(int cin, int cout_pad);
void wino4_x6_pack_weights(const std::vector<float> &U, int cin, int cout_pad, std::vector<uint16_t> &out);
size_t wino4_workspace_floats(int group, int cin, int cout, int H, int W);
// f16x3 GEMM (fp32 operands as fp16 hi + lo, three products, fp32 accumulate; conv_wino4_h3.hip)
bool wino4_h3_supported(int cin, int cout_pad);
float wino4_h3_pack_weights(const std::vector<float> &U, int cin, int cout_pad, std::vector<uint16_t> &out);   // returns the scale applied
uint32_t wino4_h3_pack_value(float x, float scale);
void launch_wino4_gemm_h3(const uint32_t *V, const void *U, float *M, int C, int Kp, int P, int Pp, hipStream_t s);
// direct 3x3 on the fp16 matrix cores, f16x3 (conv3_h3.hip): weights in ConvArgs::wt_h3, h3_vscale = the power of two the
// INPUT ACTIVATION is multiplied with before it is split (calibrated), h3_uscale the weights' power of two, h3_flag the overflow flag
bool conv3_h3_supported(int ks, int cin, int cout, int H, int W, bool unpool);
float conv3_h3_pack_weights(const float *W, int cin, int cout, std::vector<uint16_t> &out);      // returns the scale applied
void launch_conv3_h3(const ConvArgs &a, hipStream_t s);
void launch_absmax(const float *x, int64_t n, uint32_t *out_bits, hipStream_t s);                 // atomicMax of the bit pattern of |x|
// accuracy guard: out_bits[0] = max |a - b|, out_bits[1] = max |b| (bit patterns, atomicMax); out_sums[0] += sum (a - b)^2, [1] += sum b^2
void launch_absdiff_max(const float *a, const float *b, int64_t n, uint32_t *out_bits, double *out_sums, hipStream_t s);
// packed activation format of conv3_h3.hip (pk_format.hip).  Planes are (Hp, Wp) with the image at [1 .. H][1 .. W]; the
// border is never written (allocate zeroed).
size_t pk_bytes(int N, int C, int Hp, int Wp);
void launch_pk_pack(const float *in, int64_t in_sample_stride, void *out, int N, int C, int H, int W, int Hp, int Wp, float scale,
                    uint32_t *h3_flag, hipStream_t s);                                            // fp32 NCHW -> packed (x * scale split as fp16 hi + lo)
void launch_pk_unpack(const void *in, float *out, int N, int C, int H, int W, int Hp, int Wp, float scale, hipStream_t s);   // (hi + lo) / scale
// window codes [N][C][Hq][Wq] (u8, dy * 2 + dx) -> [N][C / 8][Hp][Wp] dwords: byte k, bit e = channel 8 o + e has code k
void launch_pool_bits(const uint8_t *codes, uint32_t *bits, int N, int C, int Hq, int Wq, int Hp, int Wp, hipStream_t s);
// direct 7x7, 64 -> 64, on the bf16 matrix cores with fp32 operands as three bf16 planes (conv7_x6.hip); weights in ConvArgs::wt_x6
bool conv7_x6_supported(int ks, int cin, int cout, int H, int W);
void conv7_x6_pack_weights(const float *W, int cin, int cout, std::vector<uint16_t> &out);
void launch_conv7_x6(const ConvArgs &a, hipStream_t s);
// the same layers on the fp16 matrix cores, f16x3 (conv7_h3.hip): weights in ConvArgs::wt_h3, h3_vscale = the calibrated power of two of
// the INPUT, h3_uscale the weights' (returned by the packer), h3_flag the overflow flag (a frame that raises it is recomputed on bf16x6)
bool conv7_h3_supported(int ks, int cin, int cout, int H, int W);
float conv7_h3_pack_weights(const float *W, int cin, int cout, std::vector<uint16_t> &out);
void launch_conv7_h3(const ConvArgs &a, hipStream_t s);
// fused Winograd F(4x4,3x3), 64 couts per workgroup (conv_wino4f.hip)
bool wino4f_supported(int ks, int cin, int cout, int H, int W);
int wino4f_slab_floats();
void wino4f_pack_weights(const float *W, int cin, int cout, std::vector<float> &out, int *cout_pad);
void launch_conv_wino4f(const ConvArgs &a, hipStream_t s);

// classifier convolution (3x3, <= 16 classes) fused with Softmax + the f64 mean over the samples + argmax / max / entropy
// (conv_cls_mc.hip): the logits never leave the chip
struct ClsMcArgs {
    const float *in;           // (T, Cin, H, W): the classifier's input for every sample
    int64_t in_sample_stride;
    const float *wt;           // cls_mc_pack_weights
    const float *ep_scale;     // [C]  logit = ep_scale*acc + ep_shift
    const float *ep_shift;     // [C]
    int T, Cin, H, W, C;       // T = samples in this pass (the mean divides by it), C = classes
    int relu;
    int tiles_x, tiles_y;      // filled by the launcher
    float *logits;             // optional (T, C, H, W): the logits the sums were formed from
    float *prob_sum;           // optional fp32 sums over the samples, layout as launch_mc_reduce (sum_chunk)
    double *prob_sum64 = nullptr;   // optional: the same sums as they are accumulated, in f64 (multi-device reduce-scatter)
    int64_t sum_chunk;         // 0 = [class][pixel]
    uint8_t *classes;          // optional maps (all three or none): argmax, max and entropy of the f64 mean
    double *confidence;
    double *entropy;
    // conv_cls_h3.hip: the same layer on the fp16 matrix cores, reading the PACKED input its producer wrote (conv3_h3.hip
    // OUT_PK: [T][Cin / 8][plane][in_Hp][in_Wp][8], times h3_vscale) — then `in` / `wt` are unused
    const void *in_pk = nullptr;
    int64_t in_pk_sample_bytes = 0;
    int in_Hp = 0, in_Wp = 0;
    const void *wt_h3 = nullptr;   // cls_h3_pack_weights
    float h3_vscale = 0.f, h3_uscale = 1.f;
};
bool cls_mc_supported(int ks, int cin, int cout, int H, int W);
// the f16x3 form (conv_cls_h3.hip): Cin % 32 == 0, Cin <= 96; tile = the workgroup's output pixels (its packed input plane must
// hold ceil(H / th) * th + 2 rows and ceil(W / tw) * tw + 2 columns)
bool cls_h3_supported(int ks, int cin, int cout, int H, int W);
void cls_h3_tile(int *th, int *tw);
float cls_h3_pack_weights(const float *W, int cin, int cout, std::vector<uint16_t> &out);      // returns the scale applied
void launch_conv_cls_h3(const ClsMcArgs &a, hipStream_t s);
void cls_mc_pack_weights(const float *W, int cin, int cout, std::vector<float> &out);
void launch_conv_cls_mc(const ClsMcArgs &a, hipStream_t s);

struct Wino4Plan {
    float *V, *M, *Vnext;    // disjoint buffers: this layer's transformed input, its GEMM output, the next layer's input
    bool skip_input;         // V was written by the previous layer's bridge
    bool bridge;             // fuse the output transform with the next layer's input transform (writes Vnext, not `out`)
    float next_vscale = 0.f; // bridge: > 0 when the next layer runs the f16x3 GEMM (Vnext is written as packed fp16 pairs)
    uint32_t *next_vmax = nullptr;   // bridge, calibration passes: where the next layer's largest |V| is recorded
};
size_t wino4_bridge_lds_bytes(int H, int W);
void launch_conv_wino4(const ConvArgs &a, float *workspace, int group, hipStream_t s, hipEvent_t *stage_events = nullptr,
                       bool gemm_only_events = false, const Wino4Plan *plan = nullptr);

struct PoolArgs {
    const float *in;
    int64_t in_sample_stride;
    float *out;
    uint8_t *mask;  // window code dy*2+dx
    int mask_N;     // how many samples' masks to write (1 when the input is shared)
    int N, C, H, W, Ho, Wo;
    int drop_site, sample0;
    uint64_t seed;
};
void launch_maxpool2(const PoolArgs &a, hipStream_t s);

struct UnpoolArgs {
    const float *in;
    const uint8_t *mask;
    int64_t mask_sample_stride;  // 0 when the mask is shared
    float *out;
    int N, C, H, W;              // input dims; output is (2H, 2W)
};
void launch_unpool2(const UnpoolArgs &a, hipStream_t s);

void launch_preprocess(const uint8_t *bgr, float *out, int64_t hw, hipStream_t s);
void launch_dropout(const float *in, int64_t in_sample_stride, float *out, int N, int64_t chw, int site,
                    int sample0, uint64_t seed, hipStream_t s);
void launch_lrn(const float *in, float *out, int N, int C, int64_t hw, int local_size, float alpha, float beta,
                hipStream_t s);
// chunk (0 = hw): pixel-chunk-major output [hw / chunk][C][chunk] for the multi-device reduce-scatter (segnet_multi.cpp)
// prob_sum64 (optional, instead of or beside prob_sum): the f64 sums themselves, same layout
int launch_mc_reduce(const float *logits, int n, int C, int64_t hw, float *prob_sum, float *prob, int accumulate,
                     hipStream_t s, int64_t chunk = 0, double *prob_sum64 = nullptr);
void launch_mc_reduce_finalize(const float *logits, int T, int C, int64_t hw, uint8_t *classes, double *confidence,
                               double *entropy, hipStream_t s);
void launch_mc_finalize(const float *prob_sum, int C, int64_t hw, int T, uint8_t *classes, double *confidence,
                        double *entropy, hipStream_t s);
void launch_mc_finalize64(const double *prob_sum, int C, int64_t hw, int T, uint8_t *classes, double *confidence,
                          double *entropy, hipStream_t s);
void launch_add_f64(double *dst, const double *src, int64_t n, bool init, hipStream_t s);      // dst = src (init) or dst += src
void launch_mc_variance(const float *prob, int T, int C, int64_t hw, const uint8_t *classes, double *variance,
                        hipStream_t s);
// row bands of the sample-invariant prefix (segnet_kernels.hip pack_bands_kernel / unpack_bands_kernel; segnet.cpp PrefixBands)
constexpr int BAND_ITEMS = 6, BAND_RANKS = 16;
struct BandPackItem { const unsigned char *src; int src_H, row0, n_rows, C, W, elt, rows_max; size_t off; int64_t vecs; };   // vecs = C * n_rows * W * elt / 16
struct BandPack { int n_items; BandPackItem item[BAND_ITEMS]; };
struct BandUnpackItem { unsigned char *dst; int C, H, W, elt, rows_max, drop; size_t off; int64_t vecs; int y0[BAND_RANKS + 1]; };      // vecs = C * H * W * elt / 16
struct BandUnpack { int n_items, world, n, site, sample0; uint64_t seed; size_t slot_bytes; BandUnpackItem item[BAND_ITEMS]; };
void launch_pack_bands(const BandPack &p, void *slot, hipStream_t s);
void launch_unpack_bands(const BandUnpack &u, const void *slots, hipStream_t s);
void launch_mask_to_index(const uint8_t *mask, float *out, int64_t total, int Ho, int Wo, int Win, hipStream_t s);

}  // namespace sivo
