// segnet_bands.cpp — the sample-invariant prefix of the net in row bands over the ranks that share a frame's samples.  The reference has
// one device and no such split (src/bayesian_segnet/bayesian_segnet.cpp:174-177 copies the image T times, Caffe computes the encoder T times).
#include "segnet_impl.hpp"

namespace sivo {

// ---------------------------------------------------------------------------------------------------------------------------
// Row bands of the sample-invariant prefix.  With the T samples sharded over N ranks every rank used to
// recompute the whole prefix (SegNet-Standard: conv1_1 .. pool3, 134 of 446 GFLOP per sample — 0.66 of the 1.82 ms the heaviest of 8
// ranks needs), which caps strong scaling at 3.8x.  The prefix is a chain of 3x3 / 7x7 convolutions and 2x2 poolings: the rows
// [y0, y1) of its output depend on the input rows [2^p y0 - halo, 2^p y1 + halo) only (halo = 18 rows for Standard, 21 for Basic), so
// rank r computes ITS band of output rows from a band of the image on a prefix-only handle of that height (build(prefix_rows));
// band edges inside the image see zero padding where the frame has pixels, which corrupts only halo rows that are discarded; true
// image edges coincide with band edges.  Every kernel of the prefix treats all output positions alike (direct convolutions: one fixed
// summation order per pixel), so a band's valid rows are BIT-identical to the full frame's (tests/test_gpu_prefix_bands.py).
// A rank packs the valid rows of what the per-sample part reads — the fork pooling's values before its dropout and every pooling
// mask of the prefix — into a fixed-size slot; one all-gather of the slots (SegNet-Standard, 8 ranks: 2.2 MB per rank) gives every
// rank the whole prefix; unpacking + the dropout of the fork pooling per sample (the same counter-based stream, keyed by element and
// global sample) replaces the prefix ops of the forward.
void free_bands(PrefixBands *B) {
    if (!B) return;
    (void)hipSetDevice(B->device);
    for (sivo_segnet *n : B->net) delete n;
    delete B;
}

PrefixBands &plan_bands(sivo_segnet &S, int world) {
    auto it = S.bands.find(world);
    if (it != S.bands.end()) return *it->second;
    if (world < 1 || world > BAND_RANKS) throw std::invalid_argument("prefix bands: 1 .. 16 ranks");
    if (S.prefix_weights.empty()) throw std::invalid_argument("prefix bands: the network has no sample-invariant prefix (no test-time dropout)");
    std::unique_ptr<PrefixBands, void (*)(PrefixBands *)> B(new PrefixBands, free_bands);
    B->world = world; B->device = S.device;
    size_t fork = 0;
    while (fork < S.ops.size() && (S.ops[fork].skip || S.blobs[S.ops[fork].out].shared)) ++fork;
    // the prefix ends in the pooling whose in-place Dropout makes the blobs per-sample: either that pooling is the first per-sample op
    // itself, or its dropout moved into the input transform of the convolution behind it (drop_moved) and that convolution is
    if (fork < S.ops.size() && S.ops[fork].kind == OP_CONV && S.ops[fork].in_drop_site >= 0 && fork > 0 && S.ops[fork - 1].out == S.ops[fork].in &&
        S.ops[fork - 1].kind == OP_POOL && S.ops[fork - 1].drop_moved)
        --fork;
    else if (fork >= S.ops.size() || S.ops[fork].kind != OP_POOL || S.ops[fork].drop_site < 0 || !S.blobs[S.ops[fork].in].shared)
        throw std::invalid_argument("prefix bands: the sample-invariant prefix must end in a pooling with test-time dropout");
    B->fork = (int)fork;
    for (size_t i = 0; i <= fork; ++i) {
        const Op &op = S.ops[i];
        if (op.skip || op.kind == OP_UNPOOL || op.kind == OP_DROPOUT || (i > 0 && op.in != S.ops[i - 1].out))
            throw std::invalid_argument("prefix bands: the prefix must be a plain chain of convolutions, LRN and poolings");
        if (op.kind == OP_POOL) ++B->pools;
    }
    const int align = 1 << B->pools;
    const Blob &bo = S.blobs[S.ops[fork].out];
    if (S.H % align || bo.H != S.H >> B->pools) throw std::invalid_argument("prefix bands: the image height must be a multiple of 2^poolings");
    if (bo.H < world) throw std::invalid_argument("prefix bands: more ranks than rows of the prefix output");
    // rows of the prefix output per rank: the LAST H % world ranks take one more (rank 0, which also runs ORB and the host side, the light share)
    B->y0.resize((size_t)world + 1);
    const int base = bo.H / world, extra = bo.H % world;
    for (int r = 0; r <= world; ++r) B->y0[(size_t)r] = r * base + std::max(0, r - (world - extra));
    B->rows_max = base + (extra ? 1 : 0);
    // input rows each band needs: walk the chain backwards (pooling: x2; k x k convolution: +- k / 2), align to 2^poolings
    B->in0.resize((size_t)world); B->in1.resize((size_t)world);
    for (int r = 0; r < world; ++r) {
        int lo = B->y0[(size_t)r], hi = B->y0[(size_t)r + 1];
        for (int i = (int)fork; i >= 0; --i) {
            const Op &op = S.ops[(size_t)i];
            if (op.kind == OP_POOL) { lo *= 2; hi *= 2; }
            else if (op.kind == OP_CONV) { lo -= op.ks / 2; hi += op.ks / 2; }
            lo = std::max(lo, 0); hi = std::min(hi, S.blobs[op.in].H);
        }
        B->in0[(size_t)r] = lo / align * align;
        B->in1[(size_t)r] = std::min(S.H, (hi + align - 1) / align * align);
    }
    // what the per-sample part reads of the prefix: the fork pooling's values and every pooling mask
    auto add = [&](int blob, int level, int elt) {
        const Blob &b = S.blobs[blob];
        PrefixBands::Item it2{blob, B->pools - level, elt, b.C, b.H, b.W, B->slot_bytes};
        if ((b.W * elt) % 16) throw std::invalid_argument("prefix bands: rows of the exchanged blobs must be multiples of 16 bytes");
        if ((int)B->items.size() >= BAND_ITEMS) throw std::invalid_argument("prefix bands: more poolings in the prefix than the exchange holds");
        B->slot_bytes += ((size_t)b.C * ((size_t)B->rows_max << it2.shift) * b.W * elt + 255) / 256 * 256;
        B->items.push_back(it2);
    };
    add(S.ops[fork].out, B->pools, 4);
    int level = 0;
    for (size_t i = 0; i <= fork; ++i)
        if (S.ops[i].kind == OP_POOL) add(S.ops[i].out2, ++level, 1);
    B->net.assign((size_t)world, nullptr);
    B->op_map.resize((size_t)world);
    PrefixBands *raw = B.release();
    S.bands[world] = raw;
    return *raw;
}

sivo_segnet &band_net(sivo_segnet &S, PrefixBands &B, int rank) {
    if (rank < 0 || rank >= B.world) throw std::invalid_argument("prefix bands: rank out of range");
    if (!B.net[(size_t)rank]) {
        std::unique_ptr<sivo_segnet> N = build(S.proto, 2, S.prefix_weights.data(), S.prefix_weights.size(), S.device, S.opt, S.guard_levels_used,
                                               B.in1[(size_t)rank] - B.in0[(size_t)rank]);
        if ((int)N->ops.size() != B.fork + 1) throw std::runtime_error("prefix bands: the band handle's plan does not match the prefix");
        N->h3_flag = S.h3_flag; N->owns_flag = false;
        for (size_t i = 0; i < N->ops.size(); ++i)
            for (size_t k = 0; k < S.ops.size(); ++k)
                if (S.ops[k].name == N->ops[i].name && S.ops[k].kind == N->ops[i].kind) { B.op_map[(size_t)rank].push_back({(int)i, (int)k}); break; }
        B.net[(size_t)rank] = N.release();
    }
    return *B.net[(size_t)rank];
}

// rank's band of the prefix on stream st -> its slot
void bands_enqueue(sivo_segnet &S, PrefixBands &B, sivo_segnet &N, const uint8_t *d_bgr, int rank, void *d_slot, hipStream_t st) {
    const int in0 = B.in0[(size_t)rank], rows = B.in1[(size_t)rank] - in0;
    launch_preprocess(d_bgr + (size_t)in0 * S.W * 3, (float *)N.blobs[N.input_blob].d, (int64_t)rows * S.W, st);
    run_ops(N, 0, N.ops.size(), 0, 1, 0, 0, st, 0);
    BandPack pk{};
    for (const PrefixBands::Item &it : B.items) {
        const Blob &full = S.blobs[it.blob];
        const auto bid = N.blob_id.find(full.name);
        if (bid == N.blob_id.end()) throw std::runtime_error("prefix bands: blob '" + full.name + "' is missing in the band handle");
        const Blob &bb = N.blobs[bid->second];
        const int level = B.pools - it.shift;
        BandPackItem &q = pk.item[pk.n_items++];
        q.src = static_cast<const unsigned char *>(bb.d); q.src_H = bb.H;
        q.row0 = (B.y0[(size_t)rank] << it.shift) - (in0 >> level);
        q.n_rows = (B.y0[(size_t)rank + 1] - B.y0[(size_t)rank]) << it.shift;
        q.C = it.C; q.W = it.W; q.elt = it.elt; q.rows_max = B.rows_max << it.shift; q.off = it.off;
        q.vecs = (int64_t)q.C * q.n_rows * (q.W * q.elt / 16);
    }
    launch_pack_bands(pk, d_slot, st);
}

void bands_run(sivo_segnet &S, const uint8_t *d_bgr, int rank, int world, void *d_slot, hipStream_t st) {
    h3_absorb(S);               // (as forward(): a flag from an earlier asynchronous frame is acted on before this band reads the scales; the
                                //  pause it sets covers this band AND the forward that consumes it — one frame, one arithmetic)
    PrefixBands &B = plan_bands(S, world);
    sivo_segnet &N = band_net(S, B, rank);
    // the owner's arithmetic: its calibrated (and possibly backed-off) scales; a frame that is being recomputed runs without f16x3
    for (const auto &[bi, oi] : B.op_map[(size_t)rank]) {
        N.ops[(size_t)bi].d3_vscale = S.ops[(size_t)oi].d3_vscale; N.ops[(size_t)bi].h3_vscale = S.ops[(size_t)oi].h3_vscale;
    }
    N.h3_on = S.h3_on && !S.h3_pause;
    // (Replaying the band's ~16 launches from a HIP graph was measured: 0.274 ms either way on one MI355X — the band is bound by its
    // kernels' own floor, one work item per CU, not by launch overhead — and removed.)
    bands_enqueue(S, B, N, d_bgr, rank, d_slot, st);
    SIVO_HIP(hipGetLastError());
}

void bands_unpack(sivo_segnet &S, const BandInput &pre, int n, int sample0, uint64_t seed, hipStream_t st, size_t *suffix_begin) {
    PrefixBands &B = plan_bands(S, pre.world);
    const Op &P = S.ops[(size_t)B.fork];
    BandUnpack u{};
    u.world = B.world; u.n = n; u.site = P.drop_site; u.sample0 = sample0; u.seed = seed; u.slot_bytes = B.slot_bytes;
    for (const PrefixBands::Item &it : B.items) {
        BandUnpackItem &q = u.item[u.n_items++];
        // the fork pooling's values: straight into the per-sample blob, through its dropout — unless that dropout moved into the
        // consumer's input transform (drop_moved): then the blob is the sample-invariant one and the values go in as they are
        q.drop = (&it == &B.items[0] && !P.drop_moved) ? 1 : 0;
        q.dst = static_cast<unsigned char *>(S.blobs[it.blob].d);
        q.C = it.C; q.H = it.H; q.W = it.W; q.elt = it.elt; q.rows_max = B.rows_max << it.shift; q.off = it.off;
        q.vecs = (int64_t)q.C * q.H * (q.W * q.elt / 16);
        for (int r = 0; r <= B.world; ++r) q.y0[r] = B.y0[(size_t)r] << it.shift;
    }
    launch_unpack_bands(u, pre.slots, st);
    // the switches re-laid per channel octet for the decoder layers that read packed tensors through an Upsample (run_ops does this
    // behind the pooling kernel)
    if (S.pk_on && S.h3_on && !S.calibrating)
        for (int i = 0; i <= B.fork; ++i) {
            const Op &op = S.ops[(size_t)i];
            if (op.kind != OP_POOL || !op.make_bits) continue;
            const Blob &bm = S.blobs[op.out2], &bi = S.blobs[op.in], &bp = S.blobs[op.out];
            launch_pool_bits((const uint8_t *)bm.d, bm.d_bits, 1, bi.C, bp.H, bp.W, bm.bits_Hp, bm.bits_Wp, st);
        }
    *suffix_begin = (size_t)B.fork + 1;
}

}  // namespace sivo
