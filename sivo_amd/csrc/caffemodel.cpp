// caffemodel.cpp — reader for the trained-weights file the reference loads with
// caffe::Net::CopyTrainedLayersFrom (reference src/bayesian_segnet/bayesian_segnet.cpp:61): a binary
// protobuf `NetParameter` (caffe.proto of BVLC Caffe / the caffe-segnet fork).  Only the wire format
// is needed: NetParameter.layer = 100 (LayerParameter: name = 1, type = 2, blobs = 7)
// or the legacy NetParameter.layers = 2 (V1LayerParameter: name = 4, blobs = 6); BlobProto.data = 5
// (repeated float, packed or not), .double_data = 8, .shape = 7 (BlobShape.dim = 1), legacy
// num/channels/height/width = 1..4.  Layers are matched to the prototxt BY NAME, as Caffe does; the
// result is the flat parameter array of sivo_segnet_create (conv: W then bias; BN: scale then shift).
#include <cstdint>
#include <cstring>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "prototxt.hpp"

namespace sivo {
namespace {

struct Reader {
    const uint8_t *p, *end;
    bool eof() const { return p >= end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (p >= end) throw std::invalid_argument("caffemodel: truncated varint");
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
        }
        throw std::invalid_argument("caffemodel: varint too long");
    }
    Reader sub() {
        const uint64_t n = varint();
        if ((uint64_t)(end - p) < n) throw std::invalid_argument("caffemodel: truncated field");
        Reader r{p, p + n};
        p += n;
        return r;
    }
    void skip(int wire) {
        switch (wire) {
            case 0: varint(); break;
            case 1: if (end - p < 8) throw std::invalid_argument("caffemodel: truncated"); p += 8; break;
            case 2: sub(); break;
            case 5: if (end - p < 4) throw std::invalid_argument("caffemodel: truncated"); p += 4; break;
            default: throw std::invalid_argument("caffemodel: unsupported wire type");
        }
    }
};

struct Blob { std::vector<float> data; std::vector<int64_t> shape; };

Blob read_blob(Reader r) {
    Blob b;
    int64_t legacy[4] = {0, 0, 0, 0};
    bool has_legacy = false;
    while (!r.eof()) {
        const uint64_t key = r.varint();
        const int field = (int)(key >> 3), wire = (int)(key & 7);
        if (field == 5 && wire == 2) {               // packed floats
            Reader d = r.sub();
            const size_t n = (size_t)(d.end - d.p) / 4;
            const size_t old = b.data.size();
            b.data.resize(old + n);
            std::memcpy(b.data.data() + old, d.p, n * 4);
        } else if (field == 5 && wire == 5) {        // unpacked float
            float f;
            if (r.end - r.p < 4) throw std::invalid_argument("caffemodel: truncated float");
            std::memcpy(&f, r.p, 4); r.p += 4;
            b.data.push_back(f);
        } else if (field == 8 && wire == 2) {        // packed doubles
            Reader d = r.sub();
            for (; d.end - d.p >= 8; d.p += 8) { double v; std::memcpy(&v, d.p, 8); b.data.push_back((float)v); }
        } else if (field == 7 && wire == 2) {        // BlobShape
            Reader sh = r.sub();
            while (!sh.eof()) {
                const uint64_t k = sh.varint();
                if ((k >> 3) == 1 && (k & 7) == 2) { Reader d = sh.sub(); while (!d.eof()) b.shape.push_back((int64_t)d.varint()); }
                else if ((k >> 3) == 1 && (k & 7) == 0) b.shape.push_back((int64_t)sh.varint());
                else sh.skip((int)(k & 7));
            }
        } else if (field >= 1 && field <= 4 && wire == 0) {
            legacy[field - 1] = (int64_t)r.varint(); has_legacy = true;
        } else {
            r.skip(wire);
        }
    }
    if (b.shape.empty() && has_legacy) b.shape.assign(legacy, legacy + 4);
    return b;
}

void read_layer(Reader r, bool v1, std::map<std::string, std::vector<Blob>> &out) {
    std::string name;
    std::vector<Blob> blobs;
    const int f_name = v1 ? 4 : 1, f_blobs = v1 ? 6 : 7;
    while (!r.eof()) {
        const uint64_t key = r.varint();
        const int field = (int)(key >> 3), wire = (int)(key & 7);
        if (field == f_name && wire == 2) { Reader s = r.sub(); name.assign((const char *)s.p, (size_t)(s.end - s.p)); }
        else if (field == f_blobs && wire == 2) blobs.push_back(read_blob(r.sub()));
        else r.skip(wire);
    }
    if (!blobs.empty()) out[name] = std::move(blobs);
}

}  // namespace

bool looks_like_caffemodel(const std::string &bytes) {
    // a NetParameter starts with field 1 (name, key 0x0A), 2 (layers, 0x12) or 100 (layer, keys 0xA2 0x06)
    if (bytes.size() < 2) return false;
    const uint8_t b0 = (uint8_t)bytes[0];
    return b0 == 0x0A || b0 == 0x12 || b0 == 0xA2;
}

// Throws std::invalid_argument when a parametrised prototxt layer has no (or wrongly sized) blobs in the file.
std::vector<float> weights_from_caffemodel(const std::string &bytes, const ProtoNet &net) {
    std::map<std::string, std::vector<Blob>> layers;
    Reader r{(const uint8_t *)bytes.data(), (const uint8_t *)bytes.data() + bytes.size()};
    while (!r.eof()) {
        const uint64_t key = r.varint();
        const int field = (int)(key >> 3), wire = (int)(key & 7);
        if (field == 100 && wire == 2) read_layer(r.sub(), false, layers);
        else if (field == 2 && wire == 2) read_layer(r.sub(), true, layers);
        else r.skip(wire);
    }
    std::vector<float> flat;
    std::map<std::string, int> ch;
    ch[net.input] = net.shape[1];
    for (const ProtoLayer &L : net.layers) {
        const int cin = L.bottom.empty() ? net.shape[1] : ch[L.bottom[0]];
        size_t want[2] = {0, 0};
        if (L.type == "Convolution") {
            want[0] = (size_t)L.num_output * cin * L.kernel_size * L.kernel_size; want[1] = (size_t)L.num_output;
            ch[L.top[0]] = L.num_output;
        } else if (L.type == "BN") {
            want[0] = want[1] = (size_t)cin;
            ch[L.top[0]] = cin;
        } else {
            for (auto &t : L.top) ch[t] = cin;
            continue;
        }
        auto it = layers.find(L.name);
        if (it == layers.end() || it->second.size() < 2) throw std::invalid_argument("caffemodel: no trained blobs for layer '" + L.name + "'");
        for (int k = 0; k < 2; ++k) {
            const Blob &b = it->second[k];
            if (b.data.size() != want[k]) {
                std::ostringstream m;
                m << "caffemodel: layer '" << L.name << "' blob " << k << " holds " << b.data.size() << " values, the prototxt implies " << want[k];
                throw std::invalid_argument(m.str());
            }
            flat.insert(flat.end(), b.data.begin(), b.data.end());
        }
    }
    return flat;
}

}  // namespace sivo
