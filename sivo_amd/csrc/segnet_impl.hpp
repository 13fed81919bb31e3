// segnet_impl.hpp — internal types of the Bayesian SegNet host runtime, shared by its four translation units:
//   segnet_plan.cpp    prototxt + weights -> fused launch plan (build)
//   segnet_guard.cpp   fp16 range guard (calibration, overflow protocol) and the load-time accuracy guard (build_guarded)
//   segnet_bands.cpp   the sample-invariant prefix in row bands over the ranks of a frame
//   segnet.cpp         per-frame runtime (run_ops, forward) and the C ABI
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "common.hpp"
#include "prototxt.hpp"
#include "segnet_kernels.hpp"
#include "segnet_multi.hpp"
#include <chrono>
#include <algorithm>


namespace sivo {
bool looks_like_caffemodel(const std::string &bytes);
std::vector<float> weights_from_caffemodel(const std::string &bytes, const ProtoNet &net);

struct Blob {
    std::string name;
    int C = 0, H = 0, W = 0;
    bool shared = true;    // independent of the MC sample (stored once)
    bool is_mask = false;  // pooling argmax codes (u8)
    int src_W = 0;         // masks: width of the pooled input plane (for index reconstruction)
    void *d = nullptr;
    bool fused_away = false;   // an Upsample output read straight through its pooled input by the next convolution
    int drop_pending = -1;     // >= 0: the blob holds the values IN FRONT of this dropout site (sample-invariant); the per-sample dropped tensor
                               // Caffe holds under this name only exists inside the consumer's input transform (sivo_segnet_blob re-creates it)
    // Packed form (conv3_h3.hip / pk_format.hip): the blob between two direct f16x3 layers as fp16 hi / lo pieces in zero-bordered
    // (pk_Hp, pk_Wp) planes, times the consumer's power of two.  pk_fresh: the last forward wrote ONLY this form (the fp32 array
    // is stale; sivo_segnet_blob unpacks).  Masks: d_bits = the window codes re-laid per channel octet for a consumer that
    // reads a packed pooled tensor through its Upsample (planes padded like that tensor's).
    void *d_pk = nullptr;
    int pk_Hp = 0, pk_Wp = 0;
    bool pk_fresh = false;
    float pk_scale = 0.f;
    uint32_t *d_bits = nullptr;
    int bits_Hp = 0, bits_Wp = 0;
    int64_t pk_sample_bytes() const { return (int64_t)(C / 8) * 2 * pk_Hp * pk_Wp * 16; }
    int64_t bits_sample_dwords() const { return (int64_t)(C / 8) * bits_Hp * bits_Wp; }
    int64_t chw() const { return (int64_t)C * H * W; }
};

enum OpKind { OP_CONV, OP_POOL, OP_UNPOOL, OP_DROPOUT, OP_LRN };

struct Op {
    OpKind kind;
    int in = -1, in2 = -1, out = -1, out2 = -1;
    // conv
    int ks = 0, cin = 0, cout = 0, cout_pad = 0;
    float *d_w = nullptr, *d_scale = nullptr, *d_shift = nullptr;
    size_t w_off = 0;          // offset of the layer's Caffe weights in the flat parameter array
    void *d_wx6 = nullptr;     // wino4: the transformed weights as three bf16 planes (bf16x6 GEMM); null = fp32 MFMA GEMM
    void *d_wh3 = nullptr;     // wino4: the transformed weights as fp16 hi / lo planes times h3_uscale (f16x3 GEMM, the default)
    float h3_uscale = 1.f;     // power of two
    float h3_vscale = 0.f;     // power of two the layer's transformed input is multiplied with (set by the calibration pass; 0 = not calibrated)
    float h3_vmax = 0.f;       // largest |V| of the calibration frame
    // conv3_h3.hip: the narrow 3x3 layers as a DIRECT convolution on the fp16 matrix cores (f16x3).  The flags below (wino4f /
    // wino / v2) then describe the layer's fp32 kernel, which runs the calibration frame and every frame after an fp16 overflow
    bool d3 = false;
    void *d_wd3 = nullptr;     // Caffe weights as fp16 hi / lo planes times d3_uscale, in the kernel's stage order
    float d3_uscale = 1.f, d3_vscale = 0.f, d3_vmax = 0.f;   // d3_vscale: power of two for the INPUT ACTIVATION (calibrated; 0 = not yet)
    // packed activations between direct f16x3 layers (decided at plan time, used while the handle runs f16x3):
    bool pk_in = false;        // this d3 layer reads its input (through its Upsample, if any) in the packed form
    int pk_to = -1;            // producer side: the d3 op that reads this layer's output in the packed form (its d3_vscale is the scale)
    bool make_bits = false;    // pooling: a packed consumer reads through this pooling's switches -> also write them per channel octet
    int bridge_to = -1;        // w4_bridge: the op whose transformed input this layer's bridge kernel writes
    float *d_w_mc = nullptr;   // classifier: second copy of the weights in the layout of conv_cls_mc.hip (fused with the MC post-processing)
    // classifier on the fp16 matrix cores (conv_cls_h3.hip), fed by its producer's packed output: weights in d_wd3 (cls_h3_pack_weights),
    // d3_uscale / d3_vscale / d3_vmax as for a direct f16x3 layer (input scale calibrated)
    bool c3 = false;
    bool cls_h3_last = false;  // profiling: the last fused launch was conv_cls_h3_kernel
    bool mc_fused_last = false;   // profiling: the last timed launch of this op was the fused kernel
    bool relu = false;
    bool v2 = false;           // conv_v2.hip kernel + weight layout
    bool wino = false;         // conv_wino.hip kernel + pre-transformed weights
    int wino_cfg = 0;
    bool wino4f = false;       // conv_wino4f.hip: fused F(4x4,3x3), 64 couts per workgroup (narrow layers)
    bool c7x6 = false;         // conv7_x6.hip: direct 7x7 on the bf16 matrix cores (bf16x6); weights in d_wx6
    bool c7h3 = false;         // conv7_h3.hip: the same layer on the fp16 matrix cores (f16x3: the default while the handle runs f16x3); weights
                               // in d_wd3, d3_uscale / d3_vscale / d3_vmax as for a direct f16x3 layer.  The bf16x6 form stays resident (fallback)
    bool wino4 = false;        // conv_wino4.hip: F(4x4,3x3) as input transform + batched GEMM + output transform
    int wino4_group = 1;       // samples per V/M workspace pass
    static constexpr int PROF_LANES = 4;           // (= sivo_segnet::MAX_LANES) profiling state is kept per lane: a profiled frame may keep its lanes
    std::vector<hipEvent_t> w4_ev[PROF_LANES];     // profiling: 4 events per group of the lane's last launch
    double w4_ms[3] = {0.0, 0.0, 0.0};             // input transform, GEMM, output transform
    int w4_groups_last[PROF_LANES] = {0, 0, 0, 0}, w4_launches = 0;
    unsigned timed_mask = 0;                       // lanes whose last launch of this op was bracketed
    bool w4_gemm_only_last = false;
    bool skip = false;             // Upsample fused into the following F(4x4,3x3) convolution
    bool w4_bridge = false;        // output transform fused with the next F(4x4) layer's input transform (no HBM round trip)
    bool w4_bridged_in = false;    // this layer's transformed input is written by its predecessor's bridge
    int pool_op = -1;              // F(4x4) conv: index of the MAX 2x2 pooling fused into its output transform
    int unpool_in = -1, unpool_mask = -1;   // that convolution: pooled blob and mask blob it reads through
    int drop_site = -1;
    // the fork pooling's Dropout moved into the F(4x4) input transform of its consumer (plan pass below): the pooling writes its values
    // once (sample-invariant), the consumer drops them out per sample as it reads (ConvArgs::in_drop_site)
    bool drop_moved = false;       // pooling: its dropout is applied by its consumer
    int in_drop_site = -1;         // convolution: dropout site applied to its (shared) input
    int guard_level = 0;           // accuracy guard (accuracy_guard): 0 as planned, 1 no F(4x4) (direct f16x3 at any width), 2 no f16x3 either (F(2x2) / direct fp32), 3 direct fp32 only
    // lrn
    int local_size = 5;
    float alpha = 0.f, beta = 0.f;
    double flops = 0.0;
    // profiling (sivo_segnet_profile): HIP events bracket the launch on the launch stream
    std::string name, kernel;
    hipEvent_t ev0[PROF_LANES] = {nullptr, nullptr, nullptr, nullptr}, ev1[PROF_LANES] = {nullptr, nullptr, nullptr, nullptr};
    double ms_total = 0.0, bytes = 0.0;
    int launches = 0, kernel_launches = 0, last_n = 0, lane_n[PROF_LANES] = {0, 0, 0, 0};
};


struct PrefixBands;
void free_bands(PrefixBands *);
}  // namespace sivo

struct sivo_segnet {
    SivoSegnetOptions opt{};              // how the caller wants this handle to run (normalised by segnet_options)
    sivo::SegnetMulti *multi = nullptr;   // set by sivo_segnet_create_multi: this handle only fronts the per-device ones
    // row bands of the sample-invariant prefix (PrefixBands below): what a band handle is built from, and the plan per world size
    sivo::ProtoNet proto;
    std::vector<float> prefix_weights;    // the Caffe parameters in front of the first test-time Dropout
    std::map<std::string, int> guard_levels_used;
    std::map<int, sivo::PrefixBands *> bands;
    bool owns_flag = true;                // (a band handle raises its owner's overflow flag)
    uint64_t last_seed = 0;               // of the last forward (sivo_segnet_blob re-creates a blob whose dropout moved downstream)
    int last_sample0 = 0;
    int device = 0;
    int T = 0, C = 3, H = 0, W = 0, classes = 0;
    std::vector<sivo::Blob> blobs;
    std::vector<sivo::Op> ops;
    std::map<std::string, int> blob_id;
    int input_blob = -1, logits_blob = -1;
    int cls_op = -1;               // the last op, when it is a classifier convolution conv_cls_mc.hip can fuse with the MC post-processing
    bool has_softmax = false;
    uint8_t *d_image = nullptr;     // H*W*3 staging for the host entry point
    float *d_prob_sum = nullptr;    // classes*H*W
    uint8_t *d_classes = nullptr;
    double *d_conf = nullptr, *d_ent = nullptr;
    hipStream_t stream = nullptr;   // for the host-level entry point
    int64_t sum_chunk = 0;          // layout of the probability sum the next forward writes (0 = [class][pixel])
    double *d_sum64 = nullptr;      // when set (segnet_forward_chunked): the next forward writes its f64 probability sums here
    double flops_shared = 0.0, flops_sample = 0.0;
    bool profile = false, pending = false;
    bool profile_mfma_only = false;   // bracket only the MFMA kernels (convolutions / the F(4x4) GEMM): fewer events in a timed run
    bool profile_keep_lanes = false;  // a profiled forward keeps its sample groups on their streams (events per lane; times are summed over the lanes)
    std::vector<void *> owned;
    // two-lane execution of the per-sample part: the MC samples are split in two halves that run on two streams, so the
    // tail of one lane's kernel (CUs running out of workgroups) and its launch bubbles are filled by the other lane
    static constexpr int MAX_LANES = 4;
    int ws_lanes = 1;               // workspace regions allocated
    hipStream_t lane_stream[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};      // [0] unused: lane 0 is the caller's stream
    hipEvent_t lane_fork = nullptr, lane_join[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    // f16x3 GEMM state (conv_wino4_h3.hip).  h3_on: the F(4x4) layers with fp16 weight planes and a calibrated V scale run
    // the f16x3 GEMM; cleared for good when a frame raised the overflow flag (a transformed value left the fp16 range: the
    // bf16x6 GEMM has fp32's range).  h3_flag: one word of pinned host memory the transform kernels store 1 into.
    bool h3_on = false, calibrating = false;
    bool pk_on = true;              // packed activations between direct f16x3 layers (SIVO_D3_PK=0 at construction: fp32 blobs everywhere)
    bool cls_pk_now = false;        // this forward hands the classifier its input packed (fused classifier + MC kernel on f16x3)
    volatile uint32_t *h3_flag = nullptr;
    uint32_t *d_h3_vmax = nullptr;  // calibration: one word per op (bit pattern of the largest |V|)
    int h3_overflow_frames = 0;     // frames that raised the flag (each was recomputed on the bf16x6 path when the entry point is synchronous)
    int h3_back_offs = 0;           // times the scales were lowered by 2^2 after such a frame (f16x3 is switched off at the fourth)
    bool h3_pause = false;          // the next forward runs without f16x3 (the recomputation of the frame that raised the flag)
    bool guard_over_budget = false; // build_guarded ran out of plans with the last verdict still over budget
    bool h3_unreported = false;     // a forward() / status query consumed the flag of an asynchronous frame nobody has asked about yet:
                                    // sivo_segnet_take_overflow still owes its caller a 1 (sticky until that call)
    // load-time accuracy guard (accuracy_guard below): one row per guarded layer, the budget it was held against, what it cost
    struct GuardRow { std::string layer, kernel; float rel_err = 0.f, rel_rms = 0.f, ref_max = 0.f, first_rel_err = 0.f; int level = 0; };
    std::vector<GuardRow> guard_rows;
    float guard_budget = 0.f, guard_logit_max = 0.f, guard_predicted = 0.f;
    double guard_ms = 0.0;
    int guard_builds = 0;
    float *d_wino4_ws = nullptr;    // V + M workspace shared by every F(4x4,3x3) layer (one region per lane)
    size_t wino4_ws_floats = 0;
    size_t wino4_slot_floats = 0;   // three rotating slots (V, M, next V) for layers that run all samples in one pass
    ~sivo_segnet() {
        if (multi) sivo::segnet_multi_destroy(multi);
        for (auto &kv : bands) sivo::free_bands(kv.second);
        for (sivo::Op &op : ops) {
            for (int l = 0; l < sivo::Op::PROF_LANES; ++l) {
                if (op.ev0[l]) (void)hipEventDestroy(op.ev0[l]);
                if (op.ev1[l]) (void)hipEventDestroy(op.ev1[l]);
                for (hipEvent_t e : op.w4_ev[l]) (void)hipEventDestroy(e);
            }
        }
        for (void *p : owned) (void)hipFree(p);
        if (h3_flag && owns_flag) (void)hipHostFree(const_cast<uint32_t *>(h3_flag));
        if (stream) (void)hipStreamDestroy(stream);
        for (int l = 0; l < MAX_LANES; ++l) {
            if (lane_stream[l]) (void)hipStreamDestroy(lane_stream[l]);
            if (lane_join[l]) (void)hipEventDestroy(lane_join[l]);
        }
        if (lane_fork) (void)hipEventDestroy(lane_fork);
    }
};

namespace sivo {

// ---- segnet_plan.cpp
// the caller's options (NULL / short struct: defaults) with every field in its valid range
SivoSegnetOptions segnet_options(const SivoSegnetOptions *opts);
size_t count_params(const ProtoNet &net);
// prefix_rows > 0: build only the SAMPLE-INVARIANT PREFIX of the net at a geometry of prefix_rows x W (a rank's row band, segnet_bands.cpp)
std::unique_ptr<sivo_segnet> build(const ProtoNet &net, int t_override, const float *weights, size_t n_weights, int device,
                                   const SivoSegnetOptions &opt, const std::map<std::string, int> &guard_levels = {}, int prefix_rows = 0);

// ---- segnet_guard.cpp
bool h3_flag_take(sivo_segnet &S);
void h3_back_off(sivo_segnet &S);
bool h3_tripped(sivo_segnet &S);
void h3_absorb(sivo_segnet &S);
std::vector<uint8_t> calibration_frame(int H, int W, int variant = 0);
void calibrate_h3(sivo_segnet &S);
// build + guard + (when a layer is over its budget) plan again with that layer one level down, until nothing moves
std::unique_ptr<sivo_segnet> build_guarded(const ProtoNet &net, int t_override, const float *weights, size_t n_weights, int device,
                                           const SivoSegnetOptions &opt);

// ---- segnet.cpp
// Where the Monte-Carlo post-processing of a whole frame goes (segmentImage): maps on the device, optionally the logits
// they were computed from.
struct McTargets {
    uint8_t *classes;
    double *conf, *ent;
    float *logits;     // optional (n, classes, H, W)
};
struct BandInput { const void *slots; int world; };      // the gathered prefix slots of all ranks (PrefixBands)
void harvest(sivo_segnet &S);
void run_ops(sivo_segnet &S, size_t first, size_t last, int n0, int n, int sample0, uint64_t seed, hipStream_t st, int lane);
void forward(sivo_segnet &S, const uint8_t *d_bgr, int n, int sample0, uint64_t seed, float *d_prob_sum, float *d_logits,
             float *d_prob, hipStream_t st, const McTargets *mc = nullptr, const BandInput *pre = nullptr);

// ---- segnet_bands.cpp
struct PrefixBands {
    int world = 0, fork = -1, pools = 0, rows_max = 0;
    struct Item { int blob; int shift; int elt; int C, H, W; size_t off; };     // rows of rank r at this blob: (y0[r] .. y0[r + 1]) << shift
    std::vector<Item> items;                // [0] = the fork pooling's output before dropout, then the pooling masks of the prefix
    size_t slot_bytes = 0;
    std::vector<int> y0;                    // [world + 1]: rows of the fork pooling's output per rank
    std::vector<int> in0, in1;              // [world]: input rows of each rank's band handle
    std::vector<sivo_segnet *> net;         // [world]: built on first use
    std::vector<std::vector<std::pair<int, int>>> op_map;      // [world]: (band op, owner op) pairs by layer name
    int device = 0;
};
PrefixBands &plan_bands(sivo_segnet &S, int world);
void bands_run(sivo_segnet &S, const uint8_t *d_bgr, int rank, int world, void *d_slot, hipStream_t st);
void bands_unpack(sivo_segnet &S, const BandInput &pre, int n, int sample0, uint64_t seed, hipStream_t st, size_t *suffix_begin);

}  // namespace sivo
