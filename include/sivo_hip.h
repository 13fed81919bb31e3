/*
 * sivo_hip.h — C ABI of libsivo_hip.so: the MI355X (gfx950) implementation of
 * navganti/SIVO's per-frame perception hot path.
 *
 * The reference has no FFI layer: its seam is the C++ class API of three
 * shared libraries (reference CMakeLists.txt:74-123).  Every entry point below
 * names the reference interface it stands behind; the C++ classes with the
 * reference's own signatures (sivo_amd/api/) are thin callers of this ABI, and
 * INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions: plain pointers and sizes only; `int` status (0 = ok); no
 * exceptions cross the boundary; `*_dev` entry points take DEVICE pointers and
 * a hipStream_t passed as void* (NULL = the default stream) and never
 * synchronise; the others take HOST pointers and return when the result is
 * ready.  Handles are not re-entrant (reference BayesianSegNet::segmentImage
 * mutates a shared input blob too); distinct handles may be used from
 * distinct host threads (reference Frame.cc:126-129 runs two ORBextractors
 * concurrently).
 */
#ifndef SIVO_HIP_H
#define SIVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIVO_OK 0
#define SIVO_ERR_INVALID_ARGUMENT 1 /* the C++ classes rethrow this as std::invalid_argument
                                       (reference bayesian_segnet.cpp:65-70,80-89) */
#define SIVO_ERR_RUNTIME 2          /* HIP error / no device */
#define SIVO_ERR_UNSUPPORTED 3      /* layer type or shape outside the two reference nets' vocabulary */
#define SIVO_ERR_IMAGE_TOO_SMALL 4  /* reference resizeImage returns an empty Mat (bayesian_segnet.cpp:142-162) */
#define SIVO_ERR_CAPACITY 5         /* caller buffer too small */

/* Thread-local text of the last failure on this thread. */
const char *sivo_last_error(void);
int sivo_version(void);
/* Number of visible HIP devices (0 when there is none; never fails). */
int sivo_device_count(void);

/* ===========================================================================
 * Bayesian SegNet — stands behind SIVO::BayesianSegNet
 * (reference include/bayesian_segnet/bayesian_segnet.hpp:108-170,
 *  src/bayesian_segnet/bayesian_segnet.cpp:46-78, 299-318).
 * ======================================================================== */
typedef struct sivo_segnet *sivo_segnet_t;

/* BayesianSegNet::BayesianSegNet (bayesian_segnet.cpp:46-78): parse the Caffe
 * prototxt text, take T,C,H,W from the input blob, upload the parameters.
 * `weights` is the flat fp32 parameter array in prototxt layer order
 * (Convolution: W[Cout][Cin][k][k] then bias[Cout]; BN: scale[C] then
 * shift[C]).  t_override > 0 replaces the prototxt batch size (the standard
 * prototxt ships with it blank).  Errors: empty text / C != 3 / T <= 1 ->
 * SIVO_ERR_INVALID_ARGUMENT, as the reference constructor throws. */
int sivo_segnet_create(const char *prototxt_text, size_t prototxt_len, int t_override,
                       const float *weights, size_t n_weights, int device, sivo_segnet_t *out);
/* How a handle runs, chosen by the caller at construction (the fields BayesianSegNetParams gains in sivo_amd/api/bayesian_segnet:
 * the reference's params struct, include/bayesian_segnet/bayesian_segnet.hpp:23-40, holds the two file names only).  The library reads
 * NO environment variable: every sivo_segnet_create* has an `_opts` twin taking this struct (NULL = the defaults below, which is also
 * what the plain functions use).  Zero-initialise, set struct_size = sizeof(SivoSegnetOptions), fill what differs from the default. */
typedef struct SivoSegnetOptions {
    uint32_t struct_size;        /* sizeof(SivoSegnetOptions) of the caller's header: fields beyond it take their defaults */
    int32_t lanes;               /* sample groups of the per-sample part on separate HIP streams: 1..4; 0 = default (2) */
    int32_t gemm;                /* arithmetic of the matrix-core layers: 0 = default f16x3 (fp32 operands as fp16 hi + lo, 3 products);
                                    1 = bf16x6; 2 = fp32 MFMA.  1 and 2 also turn the direct f16x3 kernels and the f16x3 classifier off */
    int32_t no_direct_f16x3;     /* != 0: the direct f16x3 3x3 / 7x7 kernels and the f16x3 classifier off (the fp32 fused kernels instead) */
    int32_t no_packed_activations; /* != 0: fp32 blobs between the direct f16x3 layers instead of packed fp16 hi | lo pieces */
    int32_t conv7_fp32;          /* != 0: SegNet-Basic's 7x7 layers on the fp32 matrix cores */
    int32_t wino4_workspace_mb;  /* workspace budget of the three-kernel F(4x4) path in MiB; 0 = default (16384) */
    int32_t debug_sync;          /* != 0: a device synchronisation behind every op of every lane (debugging) */
} SivoSegnetOptions;
int sivo_segnet_create_opts(const char *prototxt_text, size_t prototxt_len, int t_override, const float *weights, size_t n_weights,
                            int device, const SivoSegnetOptions *opts, sivo_segnet_t *out);
/* Caffe's Net::CopyTrainedLayersFrom (called at bayesian_segnet.cpp:61): read a
 * trained `.caffemodel` (binary protobuf NetParameter; both the `layer` and the
 * legacy `layers` encodings), match its layers to the prototxt BY NAME and write
 * the flat parameter array sivo_segnet_create takes.  Host-only (no GPU needed).
 * *n_weights receives the count; `out` may be NULL to query it; capacity <
 * count -> SIVO_ERR_CAPACITY.  A layer missing from the file or with blobs of the
 * wrong size -> SIVO_ERR_INVALID_ARGUMENT (Caffe CHECK-fails there). */
int sivo_caffemodel_weights(const char *prototxt_text, size_t prototxt_len, const void *model_bytes,
                            size_t model_len, float *out, size_t capacity, size_t *n_weights);
/* Same from files: model_file = prototxt, weights_file = the reference's
 * `.caffemodel` or a .sivow container (sivo_amd/weights.py).  Empty paths -> SIVO_ERR_INVALID_ARGUMENT
 * (bayesian_segnet.cpp:80-89, pinned by tests/test_bayesian_segnet.cpp:138-150). */
int sivo_segnet_create_from_files(const char *model_file, const char *weights_file, int t_override,
                                  int device, sivo_segnet_t *out);
int sivo_segnet_create_from_files_opts(const char *model_file, const char *weights_file, int t_override, int device,
                                       const SivoSegnetOptions *opts, sivo_segnet_t *out);
/* The same network with the T Monte-Carlo samples of a frame spread over several GPUs INSIDE the handle (the reference
 * constructs one BayesianSegNet, src/orbslam/System.cc:94-95, so a multi-GPU drop-in has to live behind that object).
 * One process; every device holds the full weights and takes a contiguous share of the samples (dropout keyed by the
 * global sample index); per frame: local softmax sums -> RCCL reduce-scatter over pixel ranges (fp32) -> every device
 * finalizes its 1/ndev of the pixels in f64 -> RCCL all-gather of the class / confidence / entropy chunks -> device
 * device_ids[0] hands the maps to the caller.  H*W must be a multiple of ndev, T >= ndev, devices distinct.
 * Only sivo_segnet_segment, _shape, _num_devices and _destroy take such a handle.  librccl.so is opened on first use. */
int sivo_segnet_create_multi(const char *prototxt_text, size_t prototxt_len, int t_override, const float *weights,
                             size_t n_weights, const int *device_ids, int ndev, sivo_segnet_t *out);
/* The same from the two files of BayesianSegNetParams (bayesian_segnet.hpp:23-40), like sivo_segnet_create_from_files. */
int sivo_segnet_create_multi_from_files(const char *model_file, const char *weights_file, int t_override,
                                        const int *device_ids, int ndev, sivo_segnet_t *out);
int sivo_segnet_create_multi_opts(const char *prototxt_text, size_t prototxt_len, int t_override, const float *weights, size_t n_weights,
                                  const int *device_ids, int ndev, const SivoSegnetOptions *opts, sivo_segnet_t *out);
int sivo_segnet_create_multi_from_files_opts(const char *model_file, const char *weights_file, int t_override, const int *device_ids,
                                             int ndev, const SivoSegnetOptions *opts, sivo_segnet_t *out);
int sivo_segnet_num_devices(sivo_segnet_t h, int *ndev);
int sivo_segnet_destroy(sivo_segnet_t h);
/* getInputGeometry (bayesian_segnet.hpp) and the blob shapes: T, C(=3), H, W, classes. */
int sivo_segnet_shape(sivo_segnet_t h, int32_t *T, int32_t *C, int32_t *H, int32_t *W, int32_t *classes);
/* Number of fp32 parameters the prototxt implies (size of `weights`). */
int sivo_segnet_num_params(const char *prototxt_text, size_t prototxt_len, size_t *n_params);

/* network->Forward() (bayesian_segnet.cpp:310) for `n_samples` Monte-Carlo
 * samples with global indices sample0 .. sample0+n_samples-1 (the T samples
 * of one frame are sharded over ranks this way; n_samples <= T).
 *   d_bgr       device, H*W*3 u8, BGR interleaved, already cropped to H x W
 *               (preprocessImage, :164-178: no scaling, no mean)
 *   d_prob_sum  device, classes*H*W fp32: sum over the n_samples of the
 *               per-pixel softmax (the tensor the all-reduce carries), accumulated
 *               in f64 and rounded once to fp32; the mean of extractMeanConfidence
 *               (:278-297) is this / T_total up to that rounding (6e-8 relative; the
 *               single-device entry points sivo_segnet_segment[_dev] keep f64 throughout)
 *   d_logits    device or NULL, n_samples*classes*H*W fp32 pre-softmax scores
 *   d_prob      device or NULL, n_samples*classes*H*W fp32 softmax ("prob" blob)
 * Dropout masks: Philox4x32-10 keyed on (seed; site, global sample, element). */
int sivo_segnet_forward_dev(sivo_segnet_t h, const uint8_t *d_bgr, int n_samples, int sample0,
                            uint64_t seed, float *d_prob_sum, float *d_logits, float *d_prob,
                            void *stream);

/* computeClasses / computeMaxConfidence / computeClassificationEntropy
 * (bayesian_segnet.cpp:180-203, 262-276) from the probability sum:
 * mean = sum / t_total in f64; argmax (first maximum wins), max, and
 * sum_c (p == 0 ? 0 : -p*log2 p). */
int sivo_mc_finalize_dev(const float *d_prob_sum, int classes, int64_t hw, int t_total,
                         uint8_t *d_classes, double *d_confidence, double *d_entropy, void *stream);

/* ---- The sample-invariant prefix split into row bands over the ranks that share a frame's samples (the reference has one
 * device and no such split: bayesian_segnet.cpp:174-177 copies the image T times and Caffe computes the encoder T times).
 * With the T samples sharded over `world` ranks (sivo_segnet_forward_dev with n_samples / sample0) every rank would recompute the
 * whole prefix (everything in front of the first test-time Dropout: conv1_1 .. pool3 in SegNet-Standard).  Instead rank r computes
 * the rows rows[r] .. rows[r + 1] of the prefix output from a band of the image (input_rows[2 r] .. input_rows[2 r + 1]: its rows
 * plus the receptive-field halo) and packs what the per-sample part reads of the prefix — the pooled values in front of the dropout
 * and the pooling masks — into a slot of *slot_bytes; ONE all-gather of the slots (rank order) replaces the recomputation:
 *     sivo_segnet_prefix_bands(h, world, &slot_bytes, rows, input_rows);            // once
 *     sivo_segnet_prefix_band_dev(h, d_bgr, rank, world, d_my_slot, stream);        // per frame: my band
 *     ... all-gather: d_slots = world slots in rank order (RCCL / torch.distributed) ...
 *     sivo_segnet_forward_banded_dev(h, d_slots, world, n_samples, sample0, seed, d_prob_sum, NULL, stream);
 * The results are bit-identical to sivo_segnet_forward_dev on the whole image (tests/test_gpu_prefix_bands.py): a band's valid
 * rows do not depend on the band.  The last H_out % world ranks take one row more (rank 0 the lighter share).
 * fp16 range guard across ranks: a band that leaves the fp16 range raises the flag of ITS rank's handle only
 * (sivo_segnet_take_overflow == 1 there), but every rank has consumed its rows.  The caller must OR-reduce the answers of
 * sivo_segnet_take_overflow over the ranks once per frame; when any rank says 1, EVERY rank reissues the frame (band, all-gather,
 * forward) — handles that said 0 have not backed off and keep their scales, which is the one situation in which the ranks'
 * arithmetic differs (each within tolerance, no longer bit-identical to the whole-image forward) until those handles are rebuilt;
 * bench.py --gpus N reduces the flag and refuses to report a rate containing such a frame.  The in-handle multi-device form
 * (sivo_segnet_create_multi) does all of this itself: every device backs off once per event and the frame is recomputed. */
int sivo_segnet_prefix_bands(sivo_segnet_t h, int world, size_t *slot_bytes, int32_t *rows, int32_t *input_rows);
int sivo_segnet_prefix_band_dev(sivo_segnet_t h, const uint8_t *d_bgr, int rank, int world, void *d_slot, void *stream);
int sivo_segnet_forward_banded_dev(sivo_segnet_t h, const void *d_slots, int world, int n_samples, int sample0, uint64_t seed,
                                   float *d_prob_sum, float *d_logits, void *stream);
/* Softmax over the class axis of (n, classes, hw) logits and sum over n
 * (Softmax layer + the sum half of extractMeanConfidence).  accumulate != 0
 * adds to d_prob_sum instead of overwriting it. */
int sivo_mc_reduce_dev(const float *d_logits, int n, int classes, int64_t hw, float *d_prob_sum,
                       float *d_prob, int accumulate, void *stream);

/* computeVariance (bayesian_segnet.cpp:205-260; private and unused in the
 * reference): sample variance over T of the winning class's probability. */
int sivo_mc_variance_dev(const float *d_prob, int T, int classes, int64_t hw, const uint8_t *d_classes,
                         double *d_variance, void *stream);

/* BayesianSegNet::segmentImage (bayesian_segnet.cpp:299-318), host buffers:
 * centre-crop (resizeImage :142-162), upload, T samples, finalize, download.
 * classes: H*W u8, confidence / entropy: H*W f64 (row-major, like MatXu/MatXd). */
int sivo_segnet_segment(sivo_segnet_t h, const uint8_t *bgr_hwc, int rows, int cols, uint64_t seed,
                        uint8_t *classes, double *confidence, double *entropy);

/* segmentImage with the frame already in HBM and the maps left there (no synchronisation): all T samples, then the
 * mean over the T float probabilities in f64 exactly as extractMeanConfidence (bayesian_segnet.cpp:278-297) — the
 * probability sum never goes through fp32 memory — and classes / confidence / entropy.  d_bgr: H*W*3 u8. */
int sivo_segnet_segment_dev(sivo_segnet_t h, const uint8_t *d_bgr, uint64_t seed, uint8_t *d_classes,
                            double *d_confidence, double *d_entropy, void *stream);

/* The same, and additionally the logits (T, classes, H, W fp32, the "conv1_1_D" blob of the reference's net) the maps
 * were computed from.  When the net ends in a 3x3 classifier convolution, segment / segment_dev / forward_dev (without
 * d_logits / d_prob) run that convolution, the Softmax layer and the reduction over the samples as ONE kernel
 * (conv_cls_mc.hip) and the logits never reach memory; this entry point makes that kernel also store them, so a test can
 * check (a) the logits against the reference net and (b) the maps against sivo_mc_segment_dev of exactly these logits.
 * Test/diagnostic entry point (adds T x 21.6 MB of stores per frame). */
int sivo_segnet_segment_logits_dev(sivo_segnet_t h, const uint8_t *d_bgr, uint64_t seed, uint8_t *d_classes,
                                   double *d_confidence, double *d_entropy, float *d_logits, void *stream);

/* The post-processing of segmentImage on given network outputs (bayesian_segnet.cpp:299-318 after Forward():
 * extractMeanConfidence :278-297, computeClasses :180-190, computeMaxConfidence :192-203,
 * computeClassificationEntropy :262-276): per pixel the Softmax of each of the T samples' logits (fp32), the mean of the T
 * float probabilities in f64, its argmax (first maximum wins), maximum and entropy in bits.  d_logits: (T, classes, hw). */
int sivo_mc_segment_dev(const float *d_logits, int T, int classes, int64_t hw, uint8_t *d_classes,
                        double *d_confidence, double *d_entropy, void *stream);

/* Copy a named blob of the last forward to the host (fp32; pooling masks are
 * returned as the flat input-plane index Caffe stores, as fp32).  shape =
 * {N, C, H, W}.  Test/diagnostic entry point.  Blobs that only exist on chip (fused away) are refused; the logits blob
 * (input of the Softmax layer) is written only by passes that ask for logits or per-sample probabilities
 * (sivo_segnet_forward_dev with d_logits / d_prob): segment / segment_dev / forward_dev without them run the classifier
 * fused with the Monte-Carlo post-processing and leave that blob as the last such pass wrote it. */
int sivo_segnet_blob(sivo_segnet_t h, const char *name, float *host_out, size_t capacity, int32_t shape[4]);

/* Algorithmic FLOPs of one forward (2*k*k*Cin*Cout*H*W per conv): shared =
 * the sample-invariant prefix, per_sample = the rest. */
int sivo_segnet_flops(sivo_segnet_t h, double *shared, double *per_sample);

/* Per-layer timing with HIP events recorded on the launch stream around every
 * kernel of the forward (enable: 0 off, 1 on, 2 on + reset).  Timings are
 * harvested lazily (at the next forward or at profile_read). */
typedef struct {
    char layer[64];          /* prototxt layer name of the (fused) op */
    char kernel[96];         /* kernel symbol as rocprofv3 --kernel-trace prints it (sivo:: prefix omitted) */
    int32_t samples;         /* batch of the last launch: 1 for the shared prefix, n_samples otherwise */
    int32_t launches;
    double flops_per_sample; /* algorithmic FLOPs of one sample (convolutions; 0 otherwise) */
    double bytes_per_sample; /* algorithmic HBM bytes of one sample */
    double ms_total;         /* sum of the event-timed durations of all harvested launches */
    int32_t kernel_launches; /* kernel launches behind `launches` forward passes (an F(4x4,3x3) layer runs its three
                              * kernels once per sample group and reports them as three rows) */
    int32_t pad_;
} SivoOpProfile;
/* enable: 0 off (does not wait for the events of the last profiled forward: they are harvested by the next sivo_segnet_profile_read or
 * before profiling is switched on again); 1 on; 2 on + reset the accumulators; 3 like 2 but only the MFMA kernels are bracketed (convolution
 * kernels and the F(4x4,3x3) GEMM): a handful of events per forward, for timing inside a throughput run; 4 like 3
 * without the reset (to profile a subset of the frames of a run); 5 / 6 like 3 / 4, but the profiled forward KEEPS its sample groups
 * on their streams (modes 1 - 4 run it in one lane, one launch per layer, so that a launch has the GPU to itself): events per lane,
 * ms_total is the sum over the lanes — the kernel's time while it shares the chip with the other lane. */
int sivo_segnet_profile(sivo_segnet_t h, int enable);
int sivo_segnet_profile_read(sivo_segnet_t h, SivoOpProfile *out, int capacity, int *n_out);

/* The arithmetic the matrix-core layers of the handle run: *mode = 2 f16x3 (fp32 operands as fp16 hi + lo planes, three
 * products: the default), 1 bf16x6 (three bf16 planes, six products: SIVO_GEMM=x6, and a handle whose fourth frame raised
 * the fp16 overflow flag), 0 fp32 MFMA (SIVO_GEMM=f32) or no such layer.  *overflow_frames = frames in which a value times
 * its layer's scale left the fp16 range since the handle was created.  Such a frame is wrong.  sivo_segnet_segment recomputes
 * it (without f16x3) before it returns; the asynchronous *_dev entry points cannot: once the caller has synchronised with the
 * frame it asks sivo_segnet_take_overflow, and if that says 1 it issues the same call again — that call runs without f16x3.
 * After every such frame the handle lowers its f16x3 scales by 2^2 (two more bits of headroom) and stays on f16x3; the
 * fourth switches it to bf16x6 / fp32 kernels for good.
 * per_layer (optional): one row per f16x3-capable layer with the calibration's largest |V| (three synthetic frames x MC
 * samples 0..11) and the powers of two in use for V and U; *n_layers = rows available. */
typedef struct SivoH3Layer {
    char layer[48];
    float vmax, vscale, uscale;
} SivoH3Layer;
int sivo_segnet_gemm_status(sivo_segnet_t h, int *mode, int *overflow_frames, SivoH3Layer *per_layer, int capacity, int *n_layers);
/* The load-time accuracy guard of the matrix-core layers.  At construction every 3x3 layer the plan runs on Winograd
 * F(4x4,3x3) or on the fp16 hi + lo split is evaluated on two built-in calibration frames x MC samples 0, 1 beside the direct fp32
 * kernel, on the same input: rel_err = max |layer - direct fp32| / max |direct fp32| (rel_rms the same in rms).  *predicted =
 * 0.5 sqrt(sum rel_err^2) estimates the error of the logits relative to their scale; *budget = 1e-3 / 30 (the tolerance at the
 * logit range of the reference configuration).  While the prediction was above the budget (a third of it, once a plan has needed
 * correction) the largest contributors were moved one level down and the handle planned again (*builds plans in all): level 0 as planned, 1 off F(4x4) (direct f16x3), 2 off
 * f16x3 as well (F(2x2) / direct fp32), 3 direct fp32 only.  Rows describe the FINAL plan (kernel = what the layer runs now);
 * first_rel_err = what the layer measured in the first plan.  *logit_max = largest |logit| of the guard's frames.
 * *guard_ms = wall time the guard added to construction; nothing runs per frame.  *predicted > *budget with *builds == 5: the guard
 * ran out of plans with the last one still over its budget (also written to stderr at construction).  No rows: nothing to guard, or the guard
 * was skipped (diagnostic build SIVO_GUARD=0; scales forced out of the fp16 range). */
typedef struct SivoGuardLayer {
    char layer[48];
    char kernel[24];
    float rel_err, rel_rms, ref_max, first_rel_err;
    int32_t level;
} SivoGuardLayer;
int sivo_segnet_guard_report(sivo_segnet_t h, SivoGuardLayer *rows, int capacity, int *n_rows, float *budget, float *predicted,
                             float *logit_max, double *guard_ms, int *builds);
/* *overflowed = 1 when a frame issued through an asynchronous entry point of this handle since the last call left the fp16
 * range: the handle has backed off as described above (once per event, however many frames in flight raised the flag) and its
 * NEXT forward runs without f16x3.  The answer is sticky: a later forward or status query that finds the flag first reacts
 * to it but leaves the report to this call, so with k frames in flight the caller that asks about frame i after frame i+1 was
 * issued still gets its 1.  The flag does not say WHICH frame: on 1, wait for every frame in flight, call this once more
 * (whatever they raised meanwhile belongs to the same event) and issue ALL of them again.
 * Reads one word of pinned host memory: free to call once per frame, after the frame's results were synchronised with. */
int sivo_segnet_take_overflow(sivo_segnet_t h, int *overflowed);

/* ===========================================================================
 * ORB extractor — stands behind SIVO::ORBextractor
 * (reference include/orbslam/ORBextractor.h:46-123, src/orbslam/ORBextractor.cc).
 * ======================================================================== */
typedef struct {  /* == cv::KeyPoint (28 bytes) */
    float x, y, size, angle, response;
    int32_t octave, class_id;
} SivoKeyPoint;

typedef struct sivo_orb *sivo_orb_t;

/* ORBextractor::ORBextractor (ORBextractor.cc:412-475). */
int sivo_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast,
                    int device, sivo_orb_t *out);
int sivo_orb_destroy(sivo_orb_t h);
/* The GaussianBlur(7x7, sigma 2) in front of the descriptors (ORBextractor.cc:1060-1062) is OpenCV's, and its 8-bit taps differ
 * between the OpenCV versions README.md:57 admits ("> 3.2"): variant 0 (default) 18 34 49 55 49 34 18 — OpenCV 3.2 - 3.4.12 and
 * 4.0 - 4.5.0; variant 1 18 34 48 56 48 34 18 — OpenCV >= 3.4.13 / >= 4.5.1 (error-diffused taps, sum 256).  Pick the one the
 * reference build you compare against was linked with; descriptors are bit-exact per variant (INTEGRATION.md "OpenCV contract"). */
int sivo_orb_set_gaussian(sivo_orb_t h, int variant);
/* How an extraction is issued (round 6).  mode bits, each "this group of kernels in ONE launch": 0 ComputePyramid (ORBextractor.cc:1085-1122: a
 * workgroup recomputes the footprint of its tile on the levels above it in LDS, instead of one copy + nlevels - 1 dependent resizes), 1 FAST +
 * the scan of the cell counts + the ordered emission (ORBextractor.cc:775-819: a cell waits for the cells before it, instead of three
 * launches), 2 GaussianBlur + the reflect-101 borders, 3 IC_Angle + computeOrbDescriptor (one wave per keypoint).  Default 15: four launches
 * and NO copy per image (the kernels read the kept keys from, and write candidates and [angle | descriptor] records to, pinned host memory);
 * 0 = the fifteen launches of rounds 1 - 5.  Inside a frame whose network keeps every CU busy 14 is 0.6 % faster than 15 (the level-by-level
 * resize kernels need no LDS); results are bit-identical in every mode (tests/test_gpu_orb.py). */
int sivo_orb_set_launch_mode(sivo_orb_t h, int mode);
/* GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares + mnFeaturesPerLevel; arrays of nlevels. */
int sivo_orb_tables(sivo_orb_t h, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                    int32_t *features_per_level);
/* ORBextractor::operator() (ORBextractor.cc:1019-1083): 8UC1 host image in,
 * keypoints (level 0 .. n-1 concatenated, pt scaled by the level factor) and
 * 32-byte descriptors out.  *n_out > capacity -> SIVO_ERR_CAPACITY. */
int sivo_orb_extract(sivo_orb_t h, const uint8_t *gray, int rows, int cols, int step,
                     SivoKeyPoint *keypoints, uint8_t *descriptors, int capacity, int *n_out);
/* Same with the image already resident in HBM (outputs still host). */
int sivo_orb_extract_dev(sivo_orb_t h, const uint8_t *d_gray, int rows, int cols, int step,
                         SivoKeyPoint *keypoints, uint8_t *descriptors, int capacity, int *n_out,
                         void *stream);
/* Both images of a stereo frame at once: Frame::Frame starts two ExtractORB threads and joins them (Frame.cc:126-131).  The right
 * extractor runs on a thread of the library's while the left one runs on the caller's; results as two sivo_orb_extract_dev calls. */
int sivo_orb_extract_pair_dev(sivo_orb_t left, sivo_orb_t right, const uint8_t *d_left, const uint8_t *d_right, int rows, int cols,
                              int step_left, int step_right, SivoKeyPoint *kp_left, uint8_t *desc_left, int capacity_left, int *n_left,
                              SivoKeyPoint *kp_right, uint8_t *desc_right, int capacity_right, int *n_right, void *stream);
/* Profiling of the extractor's kernels: enable != 0 brackets the kernel groups of every following extraction with HIP
 * events on the streams they run on and clears the accumulators.  sivo_orb_profile_read: mean ms per extraction of
 * ms5 = {pyramid (copy + resizes), blur + border, FAST cells + scan + compact, IC-angle, rBRIEF descriptors}, the number
 * of extractions measured and the mean number of keypoints. */
int sivo_orb_profile(sivo_orb_t h, int enable);
int sivo_orb_profile_read(sivo_orb_t h, double ms5[5], int *calls, double *mean_keys);

/* mvImagePyramid[level] (ORBextractor.h:83) of the last extraction: the level
 * image WITH its 19-pixel reflect-101 border, (rows+38) x (cols+38), tightly
 * packed; rows/cols report the interior size. */
int sivo_orb_level(sivo_orb_t h, int level, uint8_t *host_out, size_t capacity, int32_t *rows, int32_t *cols);
/* FAST candidates of one level of the last extraction (vToDistributeKeys,
 * ORBextractor.cc:765-819; coordinates relative to the 16-px border), in the
 * reference's emission order.  Test/diagnostic entry point. */
int sivo_orb_candidates(sivo_orb_t h, int level, SivoKeyPoint *out, int capacity, int *n_out);
/* DistributeOctTree (ORBextractor.cc:544-750), host only (sequential list surgery). */
int sivo_orb_distribute(const SivoKeyPoint *keys, int n, int min_x, int max_x, int min_y, int max_y,
                        int n_features, SivoKeyPoint *out, int capacity, int *n_out);

/* ===========================================================================
 * Hamming matching — stands behind SIVO::ORBmatcher
 * (reference include/orbslam/ORBmatcher.h:36-142, src/orbslam/ORBmatcher.cc).
 * Descriptors are rows of 32 bytes.
 * ======================================================================== */
/* ORBmatcher::DescriptorDistance (ORBmatcher.cc:1582-1596), dense nA x nB. */
int sivo_hamming_matrix_dev(const uint8_t *d_a, int n_a, const uint8_t *d_b, int n_b, int32_t *d_out,
                            void *stream);
int sivo_hamming_matrix(const uint8_t *a, int n_a, const uint8_t *b, int n_b, int32_t *out);
/* Candidate-list argmin with best / second best (the inner loop of every
 * Search* routine, e.g. ORBmatcher.cc:78-104): for query i the candidates are
 * rows cand_idx[cand_off[i] .. cand_off[i+1]) of b, visited in order; ties keep
 * the earlier candidate; empty list -> idx -1, dist 256.  second_idx (may be
 * NULL) = the row that holds the second-best distance as the reference's scan
 * leaves it (its octave is what the ratio rule of ORBmatcher.cc:117-119 compares), -1 if none. */
int sivo_hamming_argmin2_dev(const uint8_t *d_a, int n_a, const uint8_t *d_b, const int32_t *d_cand_off,
                             const int32_t *d_cand_idx, int32_t *d_best_idx, int32_t *d_best_dist,
                             int32_t *d_second_dist, int32_t *d_second_idx, void *stream);
int sivo_hamming_argmin2(const uint8_t *a, int n_a, const uint8_t *b, int n_b, const int32_t *cand_off,
                         const int32_t *cand_idx, int32_t *best_idx, int32_t *best_dist,
                         int32_t *second_dist, int32_t *second_idx);
/* Brute-force argmin over ALL rows of b for every row of a (best, second). */
int sivo_hamming_bruteforce_dev(const uint8_t *d_a, int n_a, const uint8_t *d_b, int n_b,
                                int32_t *d_best_idx, int32_t *d_best_dist, int32_t *d_second_dist,
                                void *stream);

/* ---------------------------------------------------------------------------
 * Guided matching — the Search* / Fuse members of SIVO::ORBmatcher
 * (reference include/orbslam/ORBmatcher.h:44-120) from the PROJECTED point on:
 * window query on the frame grid (Frame::GetFeaturesInArea, Frame.cc:326-390),
 * the per-candidate gates, best / second-best Hamming scan, acceptance rule,
 * the dependence of every iteration on the matches made by the earlier
 * iterations of the same call, and the 30-bin rotation histogram
 * (ComputeThreeMaxima, ORBmatcher.cc:1545-1577) — all on the GPU.  What needs
 * the SLAM object graph stays with the caller and arrives as arrays
 * (MapPoint::isBad / Observations / PredictScale / GetDescriptor, the cv::Mat
 * pose algebra, the DBoW2 node lists).
 *
 * The reference loops are sequential: iteration i skips the keypoints that
 * iterations < i matched.  The device reproduces exactly that result by
 * speculation + repair: every query matches in parallel against the state the
 * call started with; if two accepted queries chose one keypoint, rounds repeat
 * in which query i sees the picks of the queries < i of the previous round,
 * until nothing changes (query i is final after round i at the latest; one
 * round when there is no collision).
 * ------------------------------------------------------------------------ */
typedef struct sivo_mframe *sivo_mframe_t;
/* The part of Frame / KeyFrame the matcher reads: mvKeysSemantic, mvRight (NULL =
 * monocular, all -1), mDescriptorsSemantic (n x 32), image bounds mnMinX .. mnMaxY,
 * mvScaleFactors / mvLevelSigma2 / mvInvLevelSigma2.  Builds mGrid
 * (AssignFeaturesToGrid, Frame.cc:205-221; 64 x 48 cells) and uploads everything. */
int sivo_mframe_create(const SivoKeyPoint *keys, int n, const float *u_right, const uint8_t *descriptors,
                       float min_x, float max_x, float min_y, float max_y, const float *scale_factors,
                       const float *level_sigma2, const float *inv_level_sigma2, int nlevels, int device,
                       sivo_mframe_t *out);
int sivo_mframe_destroy(sivo_mframe_t h);
/* Frame::GetFeaturesInArea (Frame.cc:326-390) from that grid, host side, in the reference's order. */
int sivo_mframe_features_in_area(sivo_mframe_t h, float x, float y, float r, int min_level, int max_level,
                                 int32_t *out, int capacity, int *n_out);

#define SIVO_Q_VALID 1   /* the iteration is not skipped by the per-point tests in front of the window query */
#define SIVO_Q_BLOCKS 2  /* an accepted match of this query makes the keypoint unavailable to later queries */
#define SIVO_Q_STEREO 4  /* SearchForTriangulation: bStereo1 */
typedef struct {
    float u, v;            /* window centre (projected point; kp1.pt for SearchForTriangulation) */
    float radius;          /* GetFeaturesInArea r */
    int32_t lvl_lo, lvl_hi;/* admissible octaves lvl_lo <= octave <= lvl_hi */
    float ur;              /* projection into the right image (gate 1 and 2) */
    float gate;            /* gate 1: max |ur - mvRight[k]| */
    float angle;           /* keypoint angle of the query (rotation histogram) */
    int32_t flags;         /* SIVO_Q_* */
} SivoSearchQuery;

typedef struct {
    int32_t th_dist;           /* accept iff best <= th_dist (accept_lt: best < th_dist) */
    int32_t accept_lt;
    int32_t ratio_mode;        /* 0 none; 1 reject iff octave(best) == octave(second) && best > nn_ratio * second
                                  (ORBmatcher.cc:117-119); 2 accept only iff (float) best < nn_ratio * (float) second (:230, :582) */
    float nn_ratio;
    int32_t gate_mode;         /* 0 none; 1 skip k iff mvRight[k] > 0 && |ur - mvRight[k]| > gate (:93-97, :1353-1358);
                                  2 Fuse chi2: stereo 7.8 / mono 5.99 on the reprojection error (:880-902);
                                  3 SearchForTriangulation: dist <= th_dist, epipole distance, CheckDistEpipolarLine (:703-719) */
    int32_t check_orientation; /* mbCheckOrientation */
    int32_t dynamic;           /* later queries see earlier matches (sequential semantics) */
    int32_t tie_last;          /* 0: the first candidate wins ties (dist < best); 1: the last (:703 `dist > bestDist` continue) */
    float F12[9];              /* gate 3: fundamental matrix, row-major */
    float ex, ey;              /* gate 3: epipole in the train image */
} SivoSearchRule;

/* The engine.  Candidates of query q: the window query on the train frame's grid when cand_idx is NULL, else the
 * rows cand_idx[cand_begin[q] .. cand_end[q]) (BoW node lists; several queries may share a range).
 * blocked (n bytes, may be NULL): keypoints unavailable from the start.
 * Outputs (any may be NULL): match_query[q] = the keypoint query q is matched with after the rotation check, -1 none;
 * match_train[k] = the query whose match the call leaves in slot k (-1 untouched, -2 cleared by the rotation
 * check); best_dist / second_dist per query (256 = none); *n_matches as the reference routine counts it. */
int sivo_search(sivo_mframe_t train, const SivoSearchQuery *queries, const uint8_t *query_desc, int n_queries,
                const int32_t *cand_begin, const int32_t *cand_end, const int32_t *cand_idx, int n_cand,
                const SivoSearchRule *rule, const uint8_t *blocked, int32_t *match_query, int32_t *match_train,
                int32_t *best_dist, int32_t *second_dist, int *n_matches, int *rounds);

/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &, th)  (ORBmatcher.cc:44-127).
 * Per map point: track_in_view = mbTrackInView && !isBad(); proj_x / proj_y / proj_xr / level / view_cos = the
 * mTrack* fields; mp_desc = GetDescriptor(); mp_obs = Observations().  occ_obs[k] (in/out) = -1 where
 * F.mvpMapPoints[k] is NULL, else that point's Observations(); match[k] (out) = the map point stored into
 * F.mvpMapPoints[k], -1 where the call stored nothing. */
int sivo_search_by_projection_mappoints(sivo_mframe_t F, int n_mp, const uint8_t *track_in_view, const float *proj_x,
                                        const float *proj_y, const float *proj_xr, const int32_t *level,
                                        const float *view_cos, const uint8_t *mp_desc, const int32_t *mp_obs, float th,
                                        float nn_ratio, int32_t *occ_obs, int32_t *match, int *n_matches);
/* ORBmatcher::SearchByProjection(Frame &Current, const Frame &Last, th, bMono)  (ORBmatcher.cc:1278-1418).
 * Per last-frame key: valid = map point present && !mvbOutlier; u, v, inv_z = its projection with the current pose
 * (:1309-1322); last_octave / last_angle; mp_desc; mp_obs.  forward / backward = bForward / bBackward (:1299-1300).
 * match[k]: >= 0 last-frame key, -2 cleared by the rotation check, -1 untouched. */
int sivo_search_by_projection_frame(sivo_mframe_t current, int n_last, const uint8_t *valid, const float *u, const float *v,
                                    const float *inv_z, const int32_t *last_octave, const float *last_angle,
                                    const uint8_t *mp_desc, const int32_t *mp_obs, float th, int forward, int backward,
                                    float bf, int check_orientation, int32_t *occ_obs, int32_t *match, int *n_matches);
/* ORBmatcher::SearchByProjection(Frame &Current, KeyFrame*, sAlreadyFound, th, ORBdist)  (ORBmatcher.cc:1420-1543).
 * valid = pMP && !isBad && !sAlreadyFound && distance inside the scale-invariance range; occupied[k] (in/out). */
int sivo_search_by_projection_reloc(sivo_mframe_t current, int n_kf, const uint8_t *valid, const float *u, const float *v,
                                    const int32_t *pred_level, const float *kf_angle, const uint8_t *mp_desc, float th,
                                    int orb_dist, int check_orientation, uint8_t *occupied, int32_t *match, int *n_matches);
/* ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)  (ORBmatcher.cc:286-399). */
int sivo_search_by_projection_kf(sivo_mframe_t kf, int n_mp, const uint8_t *valid, const float *u, const float *v,
                                 const int32_t *pred_level, const uint8_t *mp_desc, int th, uint8_t *matched,
                                 int32_t *match, int *n_matches);
/* ORBmatcher::Fuse (ORBmatcher.cc:787-929; scw_variant != 0: :931-1053): best_idx[i] = the keypoint map point i is
 * fused with (-1: none / bestDist > TH_LOW); the Replace / AddObservation surgery (:909-923) is the caller's. */
int sivo_fuse(sivo_mframe_t kf, int n_mp, const uint8_t *valid, const float *u, const float *v, const float *ur,
              const int32_t *pred_level, const uint8_t *mp_desc, float th, int scw_variant, int32_t *best_idx,
              int32_t *best_dist, int *n_fused);
/* ORBmatcher::SearchBySim3, one direction (ORBmatcher.cc:1102-1176 / :1178-1252): match_out[i] = best key or -1. */
int sivo_search_by_sim3_dir(sivo_mframe_t kf, int n, const uint8_t *valid, const float *u, const float *v,
                            const int32_t *pred_level, const uint8_t *mp_desc, float th, int32_t *match_out);
/* The BoW-guided routines.  DBoW2 is outside this library: the vocabulary nodes both operands hold arrive as two CSR
 * lists (node k: keys idx1[off1[k] .. off1[k+1]) of the first operand, idx2[off2[k] .. off2[k+1]) of the second).
 * SearchByBoW(KeyFrame*, Frame&, ...) (ORBmatcher.cc:161-284): match_f[k] = keyframe key or -1. */
int sivo_search_by_bow_kf_frame(int n_nodes, const int32_t *off1, const int32_t *idx1, const int32_t *off2,
                                const int32_t *idx2, const uint8_t *kf_valid, const SivoKeyPoint *keys_kf,
                                const uint8_t *desc_kf, int n_kf, sivo_mframe_t frame, float nn_ratio,
                                int check_orientation, int32_t *match_f, int *n_matches);
/* SearchByBoW(KeyFrame*, KeyFrame*, ...) (ORBmatcher.cc:508-629): matches12[idx1] = idx2 or -1. */
int sivo_search_by_bow_kf_kf(int n_nodes, const int32_t *off1, const int32_t *idx1, const int32_t *off2,
                             const int32_t *idx2, const uint8_t *valid1, const SivoKeyPoint *keys1, const uint8_t *desc1,
                             int n1, const uint8_t *valid2, sivo_mframe_t kf2, float nn_ratio, int check_orientation,
                             int32_t *matches12, int *n_matches);
/* SearchForTriangulation (ORBmatcher.cc:631-785): F12 row-major, (ex, ey) the epipole in image 2. */
int sivo_search_for_triangulation(int n_nodes, const int32_t *off1, const int32_t *idx1, const int32_t *off2,
                                  const int32_t *idx2, const SivoKeyPoint *keys1, const float *u_right1,
                                  const uint8_t *has_mp1, const uint8_t *desc1, int n1, sivo_mframe_t kf2,
                                  const uint8_t *has_mp2, const float F12[9], float ex, float ey, int only_stereo,
                                  int check_orientation, int32_t *matches12, int *n_matches);

/* Frame::ComputeStereoMatches (reference src/orbslam/Frame.cc:444-629):
 * row-band candidates, octave +-1, disparity window, best Hamming < 100,
 * accept < 75, 11x11 SAD slide +-5 on the pyramid level, parabola, median
 * cull.  Left/right keypoints and descriptors are host arrays; the pyramids
 * are those of two sivo_orb handles' last extraction (still in HBM).
 * u_right / depth: n_left floats, -1 where unmatched. */
int sivo_stereo_match(sivo_orb_t left, sivo_orb_t right, const SivoKeyPoint *kp_left, const uint8_t *desc_left,
                      int n_left, const SivoKeyPoint *kp_right, const uint8_t *desc_right, int n_right,
                      float bf, float b, float *u_right, float *depth, int32_t *best_right);
/* The same in two steps, so that everything except the median cull (Frame.cc:616-628) can run while the network
 * is still computing the class map that SelectSemanticKeys (Frame.cc:177-203) needs: `begin` matches EVERY left
 * keypoint (each is independent of the other left keypoints) and returns the pre-cull u_right / depth plus the
 * SAD distance of each match (sad_dist, -1 where unmatched); `cull` then applies the median test over the
 * keypoints with keep[i] != 0 (NULL = all) and clears the others.  begin + cull(keep) == sivo_stereo_match on
 * the kept subset. */
int sivo_stereo_match_begin(sivo_orb_t left, sivo_orb_t right, const SivoKeyPoint *kp_left, const uint8_t *desc_left,
                            int n_left, const SivoKeyPoint *kp_right, const uint8_t *desc_right, int n_right,
                            float bf, float b, float *u_right, float *depth, int32_t *best_right, int32_t *sad_dist);
int sivo_stereo_match_cull(int n, const uint8_t *keep, const int32_t *sad_dist, float *u_right, float *depth);

/* ===========================================================================
 * Bundle-adjustment edges — stands behind the g2o edges SIVO::Optimizer
 * builds (reference src/orbslam/Optimizer.cc:318-409, 651-755):
 * EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ (+OnlyPose) computeError and
 * linearizeOplus, chi2 and the Huber kernel.
 * ======================================================================== */
typedef struct {
    int32_t pose;   /* index into poses: 12 doubles each, R row-major then t (world -> camera) */
    int32_t point;  /* index into points: 3 doubles each */
    int32_t stereo; /* 0 = (u,v), 1 = (u,v,uR) */
    int32_t pad_;
    double obs[3];
    double inv_sigma2; /* information = inv_sigma2 * I (Optimizer.cc:691-692, 730-733) */
} SivoEdge;

/* Per edge e: err[3e..] (err[2] = 0 for mono), Jx[9e..] 3x3 row-major
 * d err/d point, Jp[18e..] 3x6 row-major d err/d pose (rotation columns
 * first), chi2, rho (Huber-robustified chi2), w (Huber weight), depth_ok (z>0).
 * intr = {fx, fy, cx, cy, bf}. Any output pointer may be NULL. */
int sivo_ba_linearize_dev(const double *d_poses, const double *d_points, const SivoEdge *d_edges,
                          int64_t n_edges, const double intr[5], double delta_mono, double delta_stereo,
                          double *d_err, double *d_jx, double *d_jp, double *d_chi2, double *d_rho,
                          double *d_w, uint8_t *d_depth_ok, void *stream);
int sivo_ba_linearize(const double *poses, int n_poses, const double *points, int n_points,
                      const SivoEdge *edges, int64_t n_edges, const double intr[5], double delta_mono,
                      double delta_stereo, double *err, double *jx, double *jp, double *chi2, double *rho,
                      double *w, uint8_t *depth_ok);

/* ---------------------------------------------------------------------------
 * The optimisation loops g2o runs for SIVO::Optimizer, on arrays.
 * Levenberg-Marquardt as g2o::OptimizationAlgorithmLevenberg over BlockSolver_6_3:
 * robustified normal equations, landmark block marginalised by a Schur complement,
 * lambda_0 = 1e-5 max|diag H|, <= 10 trials per iteration, VertexSE3Expmap /
 * VertexSBAPointXYZ updates (restated in oracle/ba_solve_oracle.c).  Host arrays
 * in/out; every kernel runs on the GPU, the host only takes the accept/reject
 * decision of each trial (3 doubles back per trial).
 * ------------------------------------------------------------------------ */

/* One g2o `optimizer.optimize(iterations)` call — what Optimizer::BundleAdjustment
 * (Optimizer.cc:49-271) runs.  pose_fixed[i] != 0: keyframe i is held fixed
 * (setFixed, :73, :601-610).  level[e] != 0: edge excluded (setLevel(1));
 * robust[e] != 0: Huber kernel with delta_mono / delta_stereo; NULL = all
 * active / all robust.  At most one edge per (keyframe, map point) pair.
 * stop_flag (one byte, i.e. the reference's `bool *pbStopFlag`; may be NULL) is
 * polled between trials like g2o's forceStopFlag (:573-575).  poses and points are
 * updated in place; err_out (3 per edge, may be NULL) receives the error vectors
 * g2o would hold afterwards; hpp_last_out (36 per free pose, may be NULL) the
 * pose blocks of the last buildSystem. */
int sivo_ba_optimize(double *poses, const uint8_t *pose_fixed, int n_poses, double *points, int n_points,
                     const SivoEdge *edges, int64_t n_edges, const double intr[5], double delta_mono,
                     double delta_stereo, const uint8_t *level, const uint8_t *robust, int iterations,
                     const volatile uint8_t *stop_flag, double *err_out, double *hpp_last_out,
                     int *iterations_run, int *trials);

/* Optimizer::LocalBundleAdjustment from the point the graph is built
 * (Optimizer.cc:757-926): optimize(5) with Huber kernels; unless stopped, edges
 * with chi2 > 5.991 (mono) / 7.815 (stereo) or non-positive depth are excluded
 * and every kernel is dropped, optimize(10); outlier[e] = 1 for the observations
 * the caller must erase (:824-858).  cov (36, row-major) = the marginal block
 * of keyframe `cov_pose` as g2o's computeMarginals returns it (:900-907);
 * *cov_ok = 0 when that keyframe is fixed / out of range / its block is singular. */
int sivo_local_ba(double *poses, const uint8_t *pose_fixed, int n_poses, double *points, int n_points,
                  const SivoEdge *edges, int64_t n_edges, const double intr[5], const volatile uint8_t *stop_flag,
                  uint8_t *outlier, int cov_pose, double *cov, int *cov_ok, int *iterations, int *trials);

/* Optimizer::PoseOptimization from the point the edges are built
 * (Optimizer.cc:409-491): 4 rounds of optimize(10) from the initial pose, chi2
 * test (7.815) on the STEREO edges after each round — the reference never
 * re-classifies the mono edges (:432-467) — kernels of the stereo edges dropped
 * after the third round, pose + 6x6 covariance out.  edges[e].pose is ignored;
 * edges[e].point indexes `points` (map-point world positions, held fixed).
 * outlier (n_edges) = Frame::mvbOutlier; *n_inliers = nInitialCorrespondences -
 * nBad (0 and pose_out = pose0 when n_edges < 3, :409-411).  The whole schedule is
 * one launch of one persistent workgroup. */
int sivo_pose_optimize(const double pose0[12], const double *points, int n_points, const SivoEdge *edges,
                       int64_t n_edges, const double intr[5], uint8_t *outlier, double pose_out[12],
                       double cov[36], int *cov_ok, double *chi2, int *n_inliers, int *iterations,
                       int *trials);

/* ===========================================================================
 * Entropy feature-selection gate (the Tracking form; sivo_check_semantics below is the LocalMapping form) — stands behind SIVO's sivo_helpers
 * (reference src/sivo_helpers/sivo_helpers.cpp:64-88 computeStereoJacobianPose,
 * :160-180 computeStereoCovariance, :201-219 computeStereoMutualInformation) as
 * Tracking::CreateNewKeyFrame applies them per semantic keypoint
 * (reference src/orbslam/Tracking.cc:934-1023).  SURVEY.md 8f-1.
 * For keypoint i: skip unless depth[i] > 0; J = stereo pose Jacobian at
 * xyz[3i..] (the coordinates the caller passes — the reference passes WORLD
 * coordinates, Tracking.cc:963-977); S9 = [[Sx, Sx J'],[J Sx, J Sx J' + sigma2 I]];
 * MI = 0.5 log2(det Sx * det Sz / det S9); reduction = MI - entropy(row, col) at
 * the truncated keypoint position; accept iff reduction > th.
 * state_cov: 6x6 row-major (Frame::mSigmacw); level_sigma2: mvLevelSigma2.
 * ======================================================================== */
int sivo_entropy_gate_dev(int n, const SivoKeyPoint *d_kps, const float *d_depth, const double *d_xyz,
                          const double *d_entropy, int rows, int cols, const double state_cov[36], double fx,
                          double fy, double bl, const float *level_sigma2, int nlevels, double th,
                          double *d_mi, double *d_reduction, uint8_t *d_accept, void *stream);
/* Host key arrays against the DEVICE-resident entropy map (what sivo_segnet_segment_dev left in HBM): the form the per-frame
 * path uses — the keys come out of the semantic filter on the host, the map stays where the network wrote it.  Synchronous;
 * the caller has synchronised with the producer of d_entropy. */
int sivo_entropy_gate_map_dev(int n, const SivoKeyPoint *kps, const float *depth, const double *xyz,
                              const double *d_entropy, int rows, int cols, const double state_cov[36], double fx,
                              double fy, double bl, const float *level_sigma2, int nlevels, double th, double *mi,
                              double *reduction, uint8_t *accept);
int sivo_entropy_gate(int n, const SivoKeyPoint *kps, const float *depth, const double *xyz,
                      const double *entropy, int rows, int cols, const double state_cov[36], double fx,
                      double fy, double bl, const float *level_sigma2, int nlevels, double th, double *mi,
                      double *reduction, uint8_t *accept);

/* LocalMapping::CheckSemantics(pKF, idx, wP, compute_information = true) (reference src/orbslam/LocalMapping.cc:474-538)
 * over n keypoints: detected_class[i] = the class at the truncated keypoint position if depth > 0, the class is static
 * (<= TERRAIN = 8), confidence >= th_confidence and NOT (MI - entropy < th_entropy) — equality passes, unlike the
 * Tracking gate above — else VOID (255).  mi / reduction may be NULL. */
int sivo_check_semantics_dev(int n, const SivoKeyPoint *d_kps, const float *d_depth, const double *d_xyz,
                             const double *d_entropy, const double *d_confidence, const uint8_t *d_classes, int rows, int cols,
                             const double state_cov[36], double fx, double fy, double bl, const float *level_sigma2,
                             int nlevels, double th_entropy, double th_confidence, double *d_mi, double *d_reduction,
                             uint8_t *d_detected_class, void *stream);
int sivo_check_semantics(int n, const SivoKeyPoint *kps, const float *depth, const double *xyz, const double *entropy,
                         const double *confidence, const uint8_t *classes, int rows, int cols, const double state_cov[36],
                         double fx, double fy, double bl, const float *level_sigma2, int nlevels, double th_entropy,
                         double th_confidence, double *mi, double *reduction, uint8_t *detected_class);

#ifdef __cplusplus
}
#endif
#endif /* SIVO_HIP_H */
