// prototxt.cpp — tokenizer + recursive-descent reader of Caffe text-format
// messages, generic over nesting; only the fields named in prototxt.hpp are
// interpreted, everything else (param{}, weight_filler{}, ...) is skipped.
#include "prototxt.hpp"

#include <cctype>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>

namespace sivo {
namespace {

struct Msg;
struct Field {
    std::string key;
    std::string scalar;             // valid when !msg
    std::unique_ptr<Msg> msg;       // nested message
};
struct Msg {
    std::vector<Field> fields;
    const std::string *first(const std::string &k) const {
        for (auto &f : fields)
            if (!f.msg && f.key == k) return &f.scalar;
        return nullptr;
    }
    const Msg *sub(const std::string &k) const {
        for (auto &f : fields)
            if (f.msg && f.key == k) return f.msg.get();
        return nullptr;
    }
    std::vector<std::string> all(const std::string &k) const {
        std::vector<std::string> v;
        for (auto &f : fields)
            if (!f.msg && f.key == k) v.push_back(f.scalar);
        return v;
    }
};

struct Lexer {
    const std::string &s;
    size_t i = 0;
    explicit Lexer(const std::string &t) : s(t) {}
    void skip() {
        for (;;) {
            while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
            if (i < s.size() && s[i] == '#') {
                while (i < s.size() && s[i] != '\n') ++i;
                continue;
            }
            break;
        }
    }
    bool eof() { skip(); return i >= s.size(); }
    char peek() { skip(); return i < s.size() ? s[i] : '\0'; }
    std::string token() {
        skip();
        if (i >= s.size()) throw std::invalid_argument("prototxt: unexpected end of text");
        if (s[i] == '"' || s[i] == '\'') {
            const char q = s[i++];
            std::string out;
            while (i < s.size() && s[i] != q) out.push_back(s[i++]);
            if (i >= s.size()) throw std::invalid_argument("prototxt: unterminated string");
            ++i;
            return out;
        }
        if (s[i] == '{' || s[i] == '}' || s[i] == ':') return std::string(1, s[i++]);
        std::string out;
        while (i < s.size() && !std::isspace((unsigned char)s[i]) && s[i] != '{' && s[i] != '}' && s[i] != ':' &&
               s[i] != '#')
            out.push_back(s[i++]);
        return out;
    }
};

void parse_msg(Lexer &lx, Msg &m, bool top) {
    for (;;) {
        if (lx.eof()) {
            if (top) return;
            throw std::invalid_argument("prototxt: missing '}'");
        }
        if (lx.peek() == '}') {
            if (top) throw std::invalid_argument("prototxt: stray '}'");
            lx.token();
            return;
        }
        Field f;
        f.key = lx.token();
        char c = lx.peek();
        if (c == ':') {
            lx.token();
            c = lx.peek();
            if (c == '{') {
                lx.token();
                f.msg.reset(new Msg);
                parse_msg(lx, *f.msg, false);
            } else if (c == '}' || lx.eof()) {
                f.scalar.clear();  // "dim: # SET SAMPLE SIZE HERE" — value left blank in the reference file
            } else {
                // a blank value followed by the next "key:" on a later line: detect by look-ahead
                const size_t save = lx.i;
                std::string v = lx.token();
                if (lx.peek() == ':' || lx.peek() == '{') {  // v was actually the next key
                    lx.i = save;
                    f.scalar.clear();
                } else {
                    f.scalar = v;
                }
            }
        } else if (c == '{') {
            lx.token();
            f.msg.reset(new Msg);
            parse_msg(lx, *f.msg, false);
        } else {
            throw std::invalid_argument("prototxt: expected ':' or '{' after '" + f.key + "'");
        }
        m.fields.push_back(std::move(f));
    }
}

int to_int(const std::string *s, int dflt) { return s && !s->empty() ? std::atoi(s->c_str()) : dflt; }
float to_float(const std::string *s, float dflt) { return s && !s->empty() ? (float)std::atof(s->c_str()) : dflt; }

}  // namespace

ProtoNet parse_prototxt(const std::string &text) {
    Lexer lx(text);
    Msg root;
    parse_msg(lx, root, true);
    ProtoNet net;
    if (auto *n = root.first("name")) net.name = *n;
    if (auto *n = root.first("input")) net.input = *n;
    std::vector<std::string> dims = root.all("input_dim");
    if (dims.empty())
        if (const Msg *sh = root.sub("input_shape")) dims = sh->all("dim");
    std::vector<int> d;
    for (auto &s : dims)
        if (!s.empty()) d.push_back(std::atoi(s.c_str()));
    if (d.size() == 3) d.insert(d.begin(), 0);  // sample size left blank
    if (d.size() != 4) throw std::invalid_argument("prototxt: input shape needs 4 dims");
    for (int i = 0; i < 4; ++i) net.shape[i] = d[i];

    for (auto &f : root.fields) {
        if (!f.msg || (f.key != "layer" && f.key != "layers")) continue;
        const Msg &lm = *f.msg;
        ProtoLayer L;
        if (auto *s = lm.first("name")) L.name = *s;
        if (auto *s = lm.first("type")) L.type = *s;
        L.bottom = lm.all("bottom");
        L.top = lm.all("top");
        if (const Msg *p = lm.sub("convolution_param")) {
            L.num_output = to_int(p->first("num_output"), 0);
            L.pad = to_int(p->first("pad"), 0);
            L.kernel_size = to_int(p->first("kernel_size"), 0);
            L.stride = to_int(p->first("stride"), 1);
        }
        if (const Msg *p = lm.sub("pooling_param")) {
            if (auto *s = p->first("pool")) L.pool = *s;
            L.kernel_size = to_int(p->first("kernel_size"), 0);
            L.stride = to_int(p->first("stride"), 1);
            L.pad = to_int(p->first("pad"), 0);
        }
        if (const Msg *p = lm.sub("upsample_param")) L.scale = to_int(p->first("scale"), 2);
        if (const Msg *p = lm.sub("dropout_param")) {
            L.dropout_ratio = to_float(p->first("dropout_ratio"), 0.5f);
            if (auto *s = p->first("sample_weights_test")) L.sample_weights_test = (*s == "true");
        }
        if (const Msg *p = lm.sub("lrn_param")) {
            L.local_size = to_int(p->first("local_size"), 5);
            L.alpha = to_float(p->first("alpha"), 1.f);
            L.beta = to_float(p->first("beta"), 0.75f);
        }
        if (const Msg *p = lm.sub("bn_param"))
            if (auto *s = p->first("bn_mode")) L.bn_mode = *s;
        net.layers.push_back(std::move(L));
    }
    return net;
}

}  // namespace sivo
