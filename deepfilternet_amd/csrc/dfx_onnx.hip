// The reference's shipped model artefact, read directly: `<model>_onnx.tar.gz` = enc.onnx + erb_dec.onnx + df_dec.onnx + config.ini
// (+ version.txt), written by DeepFilterNet/df/scripts/export.py:133-337 and opened by libDF/src/tract.rs:29-70 (DfParams::from_targz).
//
// The reference hands the three graphs to tract, which executes them.  Here the graphs are *read*: their weight-bearing nodes are
// matched, in execution order and with shape checks, against the module structure of DeepFilterNet3 (deepfilternet3.py:100-331), and
// what comes out is the same (dfx_model_cfg, packed state-dict blob) pair dfx_model_create() takes from a checkpoint:
//   * DSP parameters: the config.ini keys DfTract::new reads (tract.rs:264-285: sr, hop_size, fft_size, min_nb_erb_freqs, nb_erb, nb_df,
//     df_order, conv_lookahead, df_lookahead, norm_alpha | norm_tau; [train] model must be deepfilternet3, :308-315);
//   * network structure (conv_ch, hidden sizes, layer counts, linear groups, skip kinds, pathway kernel, enc_concat, lsnr range): from the
//     graphs themselves, like tract — the ini's [deepfilternet] section is not trusted for them;
//   * BatchNorm: the TorchScript exporter folds eval-mode BatchNorm into the preceding Conv (weight' = w*g/sqrt(v+eps), bias); such a
//     conv is stored as weight' with an identity BatchNorm carrying the bias; an un-fused BatchNormalization node is read as it is;
//   * GRU: ONNX gate order z,r,h and one [1,6H] bias -> PyTorch's r,z,n and bias_ih / bias_hh; linear_before_reset must be 1.
// No protobuf / onnx library: a ~100-line wire-format reader (ModelProto.graph -> node / initializer; TensorProto raw_data / float_data).
// gzip through zlib; tar: ustar / pax / GNU long names.
#include <zlib.h>

#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "dfx_common.h"
#include "dfx_manifest.h"

namespace {

typedef std::vector<unsigned char> Bytes;

struct Fail {
    std::string msg;
};
[[noreturn]] void fail(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Fail{buf};
}

// ------------------------------------------------------------------------------------------------------------ gzip + tar
Bytes read_file(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) fail("Could not open model tar file '%s'", path);  // tract.rs:31
    Bytes b;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) b.insert(b.end(), buf, buf + n);
    fclose(f);
    return b;
}

Bytes gunzip(const Bytes &in) {
    z_stream s;
    memset(&s, 0, sizeof(s));
    if (inflateInit2(&s, 16 + MAX_WBITS) != Z_OK) fail("zlib: inflateInit2 failed");
    s.next_in = const_cast<unsigned char *>(in.data());
    s.avail_in = (uInt)in.size();
    Bytes out;
    std::vector<unsigned char> buf(1 << 20);
    int rc;
    do {
        s.next_out = buf.data();
        s.avail_out = (uInt)buf.size();
        rc = inflate(&s, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) {
            inflateEnd(&s);
            fail("Could not extract models from tar file: not a gzip stream (zlib %d)", rc);  // tract.rs:44
        }
        out.insert(out.end(), buf.data(), buf.data() + (buf.size() - s.avail_out));
    } while (rc != Z_STREAM_END);
    inflateEnd(&s);
    return out;
}

struct TarEntry {
    std::string path;
    const unsigned char *data;
    size_t size;
};

std::vector<TarEntry> untar(const Bytes &t) {
    std::vector<TarEntry> out;
    std::string longname;
    size_t off = 0;
    while (off + 512 <= t.size()) {
        const unsigned char *h = t.data() + off;
        bool zero = true;
        for (int i = 0; i < 512 && zero; ++i) zero = h[i] == 0;
        if (zero) break;
        size_t size = 0;
        if (h[124] & 0x80) {  // GNU base-256
            for (int i = 125; i < 136; ++i) size = (size << 8) | h[i];
        } else {
            for (int i = 124; i < 136 && h[i]; ++i)
                if (h[i] >= '0' && h[i] <= '7') size = size * 8 + (size_t)(h[i] - '0');
        }
        const char type = (char)h[156];
        const unsigned char *data = h + 512;
        if (size > t.size() - off - 512) fail("Could not open model tar entry: truncated archive");   // (no sum: a base-256 size near 2^64 must not wrap)
        if (type == 'L') {  // GNU long name for the next entry
            longname.assign(reinterpret_cast<const char *>(data), strnlen(reinterpret_cast<const char *>(data), size));
        } else if (type == 'x' || type == 'g') {  // pax header: "len path=value\n" records
            std::string rec(reinterpret_cast<const char *>(data), size);
            size_t p = 0;
            while (p < rec.size()) {
                size_t sp = rec.find(' ', p);
                if (sp == std::string::npos) break;
                const size_t len = (size_t)atol(rec.substr(p, sp - p).c_str());
                if (len == 0 || p + len > rec.size()) break;
                const std::string kv = rec.substr(sp + 1, p + len - sp - 2);
                if (type == 'x' && kv.compare(0, 5, "path=") == 0) longname = kv.substr(5);
                p += len;
            }
        } else {
            std::string name(reinterpret_cast<const char *>(h), strnlen(reinterpret_cast<const char *>(h), 100));
            if (memcmp(h + 257, "ustar", 5) == 0 && h[345]) {
                std::string prefix(reinterpret_cast<const char *>(h + 345), strnlen(reinterpret_cast<const char *>(h + 345), 155));
                name = prefix + "/" + name;
            }
            if (!longname.empty()) name = longname;
            longname.clear();
            if (type == '0' || type == 0) out.push_back({name, data, size});
        }
        off += 512 + ((size + 511) / 512) * 512;
    }
    return out;
}

std::string basename_of(const std::string &p) {
    const size_t s = p.find_last_of('/');
    return s == std::string::npos ? p : p.substr(s + 1);
}

// ------------------------------------------------------------------------------------------------------------ config.ini
typedef std::map<std::string, std::map<std::string, std::string>> Ini;

std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}
std::string lower(std::string s) {
    for (auto &c : s) c = (char)tolower((unsigned char)c);
    return s;
}

Ini parse_ini(const unsigned char *d, size_t n) {
    Ini ini;
    std::string text(reinterpret_cast<const char *>(d), n), sec;
    size_t p = 0;
    while (p <= text.size()) {
        size_t e = text.find('\n', p);
        if (e == std::string::npos) e = text.size();
        std::string line = trim(text.substr(p, e - p));
        p = e + 1;
        if (line.empty() || line[0] == ';' || line[0] == '#') continue;
        if (line[0] == '[') {
            const size_t r = line.find(']');
            if (r == std::string::npos) fail("Could not load config from tar file: bad section header '%s'", line.c_str());
            sec = lower(trim(line.substr(1, r - 1)));
            ini[sec];
            continue;
        }
        size_t q = line.find_first_of("=:");
        if (q == std::string::npos) continue;
        ini[sec][lower(trim(line.substr(0, q)))] = trim(line.substr(q + 1));
    }
    return ini;
}

const std::string *ini_get(const Ini &ini, const char *sec, const char *key) {
    auto s = ini.find(sec);
    if (s == ini.end()) return nullptr;
    auto k = s->second.find(key);
    return k == s->second.end() ? nullptr : &k->second;
}
long ini_int(const Ini &ini, const char *sec, const char *key, const char *sec2 = nullptr) {
    const std::string *v = ini_get(ini, sec, key);
    if (!v && sec2) v = ini_get(ini, sec2, key);
    if (!v) fail("config.ini: option '%s' missing from section [%s]", key, sec);  // the reference: .unwrap() on None
    char *end = nullptr;
    const long x = strtol(v->c_str(), &end, 10);
    if (end == v->c_str() || *end) fail("config.ini: option '%s' = '%s' is not an integer", key, v->c_str());
    return x;
}

// tract.rs:989-999 calc_norm_alpha (f32 arithmetic; round-half-away like f32::round)
float calc_norm_alpha(long sr, long hop, float tau) {
    const float dt = (float)hop / (float)sr;
    const float alpha = expf(-dt / tau);
    float a = 1.f;
    int precision = 3;
    while (a >= 1.f) {
        float pw = 1.f;
        for (int i = 0; i < precision; ++i) pw *= 10.f;
        a = roundf(alpha * pw) / pw;
        ++precision;
    }
    return a;
}

// ------------------------------------------------------------------------------------------------------------ protobuf wire format
struct Rd {
    const unsigned char *p, *e;
    bool more() const { return p < e; }
    uint64_t varint() {
        uint64_t r = 0;
        int s = 0;
        while (true) {
            if (p >= e || s > 63) fail("onnx: truncated varint");
            const unsigned char c = *p++;
            r |= (uint64_t)(c & 0x7f) << s;
            s += 7;
            if (c < 0x80) return r;
        }
    }
    // returns the field number; the value is in (u | sub)
    int field(int *wire, uint64_t *u, Rd *sub) {
        const uint64_t k = varint();
        *wire = (int)(k & 7);
        switch (*wire) {
        case 0: *u = varint(); break;
        case 1:
            if (e - p < 8) fail("onnx: truncated fixed64");
            memcpy(u, p, 8);
            p += 8;
            break;
        case 2: {
            const uint64_t l = varint();
            if ((uint64_t)(e - p) < l) fail("onnx: truncated field");
            *sub = Rd{p, p + l};
            p += l;
            break;
        }
        case 5: {
            if (e - p < 4) fail("onnx: truncated fixed32");
            uint32_t v;
            memcpy(&v, p, 4);
            *u = v;
            p += 4;
            break;
        }
        default: fail("onnx: unsupported wire type %d", *wire);
        }
        return (int)(k >> 3);
    }
    std::string str() const { return std::string(reinterpret_cast<const char *>(p), (size_t)(e - p)); }
};

struct Tensor {
    std::vector<int64_t> dims;
    int dtype = 0;  // TensorProto.DataType: 1 = FLOAT, 7 = INT64
    std::vector<float> f;
    int64_t numel() const {   // -1: a negative dimension or more than 2^31 elements (no tensor of these models comes near; a malformed file might)
        int64_t n = 1;
        for (int64_t d : dims) {
            if (d < 0 || d > ((int64_t)1 << 31)) return -1;
            n *= d;
            if (n > ((int64_t)1 << 31)) return -1;
        }
        return n;
    }
};

void ints_field(int wire, uint64_t u, Rd sub, std::vector<int64_t> *out) {
    if (wire == 0) out->push_back((int64_t)u);
    else
        while (sub.more()) out->push_back((int64_t)sub.varint());
}

// TensorProto: dims = 1, data_type = 2, float_data = 4 (packed), name = 8, raw_data = 9
Tensor parse_tensor(Rd r, std::string *name) {
    Tensor t;
    Rd raw{nullptr, nullptr}, fdata{nullptr, nullptr};
    std::vector<float> unpacked;
    while (r.more()) {
        int w;
        uint64_t u = 0;
        Rd s{nullptr, nullptr};
        switch (r.field(&w, &u, &s)) {
        case 1: ints_field(w, u, s, &t.dims); break;
        case 2: t.dtype = (int)u; break;
        case 4:
            if (w == 2) fdata = s;
            else {
                float v;
                uint32_t b = (uint32_t)u;
                memcpy(&v, &b, 4);
                unpacked.push_back(v);
            }
            break;
        case 8:
            if (name) *name = s.str();
            break;
        case 9: raw = s; break;
        default: break;
        }
    }
    if (t.numel() < 0) fail("onnx: tensor '%s' has a negative or absurd dimension", name ? name->c_str() : "?");
    if (t.dtype == 1) {
        const int64_t n = t.numel();
        if (raw.p && (int64_t)(raw.e - raw.p) == 4 * n) {
            t.f.resize((size_t)n);
            memcpy(t.f.data(), raw.p, (size_t)(4 * n));
        } else if (fdata.p && (int64_t)(fdata.e - fdata.p) == 4 * n) {
            t.f.resize((size_t)n);
            memcpy(t.f.data(), fdata.p, (size_t)(4 * n));
        } else if ((int64_t)unpacked.size() == n) {
            t.f = unpacked;
        } else if (n != 0) {
            fail("onnx: float tensor '%s' has no usable data (external data is not supported)", name ? name->c_str() : "?");
        }
    }
    return t;
}

struct Node {
    std::string op, name;
    std::vector<std::string> in, out;
    std::map<std::string, std::vector<int64_t>> ints;  // attributes: ints, or a single i
    std::map<std::string, Tensor> tensors;              // attribute t
    int64_t attr_i(const char *k, int64_t dflt) const {
        auto it = ints.find(k);
        return it == ints.end() || it->second.empty() ? dflt : it->second[0];
    }
};

// AttributeProto: name = 1, i = 3, t = 5, ints = 8
void parse_attr(Rd r, Node *n) {
    std::string name;
    std::vector<int64_t> ints;
    bool has_i = false, has_t = false;
    Tensor t;
    while (r.more()) {
        int w;
        uint64_t u = 0;
        Rd s{nullptr, nullptr};
        switch (r.field(&w, &u, &s)) {
        case 1: name = s.str(); break;
        case 3:
            ints.assign(1, (int64_t)u);
            has_i = true;
            break;
        case 5:
            t = parse_tensor(s, nullptr);
            has_t = true;
            break;
        case 8: ints_field(w, u, s, &ints); break;
        default: break;
        }
    }
    (void)has_i;
    if (has_t) n->tensors[name] = t;
    else n->ints[name] = ints;
}

// NodeProto: input = 1, output = 2, name = 3, op_type = 4, attribute = 5
Node parse_node(Rd r) {
    Node n;
    while (r.more()) {
        int w;
        uint64_t u = 0;
        Rd s{nullptr, nullptr};
        switch (r.field(&w, &u, &s)) {
        case 1: n.in.push_back(s.str()); break;
        case 2: n.out.push_back(s.str()); break;
        case 3: n.name = s.str(); break;
        case 4: n.op = s.str(); break;
        case 5: parse_attr(s, &n); break;
        default: break;
        }
    }
    return n;
}

struct Graph {
    std::string file;
    std::vector<Node> nodes;
    std::map<std::string, Tensor> consts;  // initializers and the values of Constant nodes
    const Tensor *cst(const std::string &name) const {
        auto it = consts.find(name);
        return it == consts.end() ? nullptr : &it->second;
    }
};

// ModelProto.graph = 7; GraphProto: node = 1, initializer = 5
Graph parse_model(const char *file, const unsigned char *d, size_t n) {
    Graph g;
    g.file = file;
    Rd m{d, d + n}, gr{nullptr, nullptr};
    while (m.more()) {
        int w;
        uint64_t u = 0;
        Rd s{nullptr, nullptr};
        if (m.field(&w, &u, &s) == 7 && w == 2) gr = s;
    }
    if (!gr.p) fail("%s: no graph in the ONNX model", file);
    while (gr.more()) {
        int w;
        uint64_t u = 0;
        Rd s{nullptr, nullptr};
        const int f = gr.field(&w, &u, &s);
        if (f == 1 && w == 2) {
            Node nd = parse_node(s);
            if (nd.op == "Constant" && !nd.out.empty() && nd.tensors.count("value")) g.consts[nd.out[0]] = nd.tensors["value"];
            else g.nodes.push_back(std::move(nd));
        } else if (f == 5 && w == 2) {
            std::string name;
            Tensor t = parse_tensor(s, &name);
            g.consts[name] = std::move(t);
        }
    }
    return g;
}

// ------------------------------------------------------------------------------------------------------------ matching
bool is_param_op(const Graph &g, const Node &n) {
    static const char *ops[] = {"Conv", "ConvTranspose", "Einsum", "GRU", "MatMul", "Gemm", "BatchNormalization"};
    bool known = false;
    for (const char *o : ops) known = known || n.op == o;
    if (!known) return false;
    for (size_t i = 0; i < n.in.size(); ++i) {
        const Tensor *t = g.cst(n.in[i]);
        if (t && t->dtype == 1 && t->numel() > 1) return true;
        if (t && t->dtype == 1 && (n.op == "MatMul" || n.op == "Gemm")) return true;
    }
    return false;
}

// The weight-bearing nodes of one graph, in execution order, consumed front to back.
struct Cursor {
    const Graph &g;
    std::vector<const Node *> seq;
    size_t pos = 0;
    explicit Cursor(const Graph &g_) : g(g_) {
        for (const Node &n : g.nodes)
            if (is_param_op(g, n)) seq.push_back(&n);
    }
    const Node *peek(size_t ahead = 0) const { return pos + ahead < seq.size() ? seq[pos + ahead] : nullptr; }
    const Node *take(const char *op, const char *what) {
        const Node *n = peek();
        if (!n) fail("%s: graph ends where %s (%s) was expected", g.file.c_str(), what, op);
        if (n->op != op) fail("%s: expected %s for %s, found %s '%s'", g.file.c_str(), op, what, n->op.c_str(), n->name.c_str());
        ++pos;
        return n;
    }
    const Tensor &weight(const Node *n, size_t i, const char *what) const {
        const Tensor *t = i < n->in.size() ? g.cst(n->in[i]) : nullptr;
        if (!t || t->dtype != 1) fail("%s: input %zu of %s '%s' (%s) is not a constant float tensor", g.file.c_str(), i, n->op.c_str(), n->name.c_str(), what);
        return *t;
    }
    // consumers of a tensor's DATA among all nodes (Shape only looks at the dimensions: the exporter's dynamic reshapes hang one on
    // almost every tensor)
    std::vector<const Node *> consumers(const std::string &tensor) const {
        std::vector<const Node *> out;
        for (const Node &n : g.nodes) {
            if (n.op == "Shape") continue;
            for (const std::string &i : n.in)
                if (i == tensor) {
                    out.push_back(&n);
                    break;
                }
        }
        return out;
    }
};

bool dims_are(const Tensor &t, std::initializer_list<int64_t> d) {
    if (t.dims.size() != d.size()) return false;
    size_t i = 0;
    for (int64_t v : d)
        if (t.dims[i++] != v) return false;
    return true;
}
std::string dims_str(const Tensor &t) {
    std::string s = "[";
    for (size_t i = 0; i < t.dims.size(); ++i) s += (i ? "," : "") + std::to_string(t.dims[i]);
    return s + "]";
}

struct Builder {
    std::map<std::string, std::vector<float>> sd;  // reference state-dict name -> data
    void put(const std::string &name, const std::vector<float> &v) { sd[name] = v; }
    void put(const std::string &name, size_t n, float v) { sd[name] = std::vector<float>(n, v); }
};

// One Conv2dNormAct / ConvTranspose2dNormAct (modules.py:18-126) = [depthwise or full conv] (+ 1x1 conv when separable) + BatchNorm,
// as the exporter leaves it: BatchNorm folded into the last conv (weight', bias) or kept as a BatchNormalization node.
void take_conv_block(Cursor &c, Builder &b, const std::string &p, int in_ch, int out_ch, int kt, int kf, bool transposed) {
    int idx = kt > 1 ? 1 : 0;
    const int groups = dfx_gcd(in_ch, out_ch);
    bool sep = groups > 1;
    if (!transposed && (kt > kf ? kt : kf) == 1) sep = false;
    const Node *n = c.take(transposed ? "ConvTranspose" : "Conv", p.c_str());
    const Tensor &w = c.weight(n, 1, p.c_str());
    const bool shape_ok = transposed ? dims_are(w, {in_ch, out_ch / groups, kt, kf}) : dims_are(w, {out_ch, in_ch / groups, kt, kf});
    if (!shape_ok || n->attr_i("group", 1) != groups)
        fail("%s: %s: weight %s / group %lld does not fit a %d -> %d channel (%d,%d) %sconvolution", c.g.file.c_str(), p.c_str(),
             dims_str(w).c_str(), (long long)n->attr_i("group", 1), in_ch, out_ch, kt, kf, transposed ? "transposed " : "");
    b.put(p + "." + std::to_string(idx) + ".weight", w.f);
    ++idx;
    const Node *last = n;
    if (sep) {
        const Node *pw = c.take("Conv", (p + " (pointwise)").c_str());
        const Tensor &w1 = c.weight(pw, 1, p.c_str());
        if (!dims_are(w1, {out_ch, out_ch, 1, 1})) fail("%s: %s: pointwise weight %s, expected [%d,%d,1,1]", c.g.file.c_str(), p.c_str(), dims_str(w1).c_str(), out_ch, out_ch);
        b.put(p + "." + std::to_string(idx) + ".weight", w1.f);
        ++idx;
        last = pw;
    }
    const std::string bn = p + "." + std::to_string(idx);
    const Node *nx = c.peek();
    if (nx && nx->op == "BatchNormalization" && !last->out.empty() && !nx->in.empty() && nx->in[0] == last->out[0]) {
        ++c.pos;
        const char *leaf[4] = {".weight", ".bias", ".running_mean", ".running_var"};
        for (int i = 0; i < 4; ++i) {
            const Tensor &t = c.weight(nx, (size_t)i + 1, bn.c_str());
            if (t.numel() != out_ch) fail("%s: %s: BatchNormalization input %d has %lld elements, expected %d", c.g.file.c_str(), bn.c_str(), i + 1, (long long)t.numel(), out_ch);
            b.put(bn + leaf[i], t.f);
        }
        // (epsilon is a float attribute this reader does not decode: the reference's BatchNorm2d layers all use the default 1e-5)
        if (last->in.size() > 2 && !last->in[2].empty()) fail("%s: %s: convolution bias together with BatchNormalization", c.g.file.c_str(), p.c_str());
        return;
    }
    // folded: y = conv(x; w') + bias  ==  BatchNorm(gamma = 1, beta = bias, mean = 0, var + eps = 1) after conv(x; w')
    if (last->in.size() < 3 || last->in[2].empty()) fail("%s: %s: neither a folded bias nor a BatchNormalization node after the convolution", c.g.file.c_str(), p.c_str());
    const Tensor &bias = c.weight(last, 2, (p + " bias").c_str());
    if (bias.numel() != out_ch) fail("%s: %s: bias has %lld elements, expected %d", c.g.file.c_str(), p.c_str(), (long long)bias.numel(), out_ch);
    const float eps = 1e-5f;
    volatile float var = 1.0f - eps;  // the engine folds with g / sqrtf(v + 1e-5f): this v makes that exactly 1
    if ((float)(var + eps) != 1.0f) fail("internal: identity BatchNorm variance is not exact");
    b.put(bn + ".weight", (size_t)out_ch, 1.0f);
    b.put(bn + ".bias", bias.f);
    b.put(bn + ".running_mean", (size_t)out_ch, 0.0f);
    b.put(bn + ".running_var", (size_t)out_ch, (float)var);
}

// GroupedLinearEinsum (modules.py:741-780): Einsum("btgi,gih->btgh") with the weight [G, I/G, H/G]
const Tensor &take_einsum(Cursor &c, Builder &b, const std::string &name) {
    const Node *n = c.take("Einsum", name.c_str());
    const Tensor &w = c.weight(n, 1, name.c_str());
    if (w.dims.size() != 3) fail("%s: %s: grouped-linear weight %s is not [G, I/G, H/G]", c.g.file.c_str(), name.c_str(), dims_str(w).c_str());
    b.put(name, w.f);
    return w;
}

// nn.GRU layers (modules.py:721): one ONNX GRU node per layer.  Returns (layers, hidden size).
int take_grus(Cursor &c, Builder &b, const std::string &p, int *hidden) {
    int layers = 0;
    while (c.peek() && c.peek()->op == "GRU") {
        const Node *n = c.take("GRU", p.c_str());
        const Tensor &W = c.weight(n, 1, p.c_str()), &R = c.weight(n, 2, p.c_str());
        const int64_t H = n->attr_i("hidden_size", 0);
        if (H <= 0 || H > 65536 || W.dims.size() != 3 || W.dims[2] <= 0) fail("%s: %s layer %d: GRU with hidden_size %lld / W %s", c.g.file.c_str(), p.c_str(), layers, (long long)H, dims_str(W).c_str());
        if (W.dims.size() != 3 || W.dims[0] != 1 || W.dims[1] != 3 * H || !dims_are(R, {1, 3 * H, H}))
            fail("%s: %s layer %d: W %s / R %s do not fit a unidirectional GRU of hidden size %lld", c.g.file.c_str(), p.c_str(), layers, dims_str(W).c_str(), dims_str(R).c_str(), (long long)H);
        if (n->attr_i("linear_before_reset", 0) != 1) fail("%s: %s: GRU without linear_before_reset (not PyTorch's GRU)", c.g.file.c_str(), p.c_str());
        const int64_t I = W.dims[2];
        // ONNX gate order z, r, h  ->  PyTorch r, z, n
        auto regate = [&](const float *src, int64_t cols) {
            std::vector<float> out((size_t)(3 * H * cols));
            const int64_t from[3] = {1, 0, 2};
            for (int g = 0; g < 3; ++g) memcpy(out.data() + g * H * cols, src + from[g] * H * cols, (size_t)(H * cols) * sizeof(float));
            return out;
        };
        const std::string s = std::to_string(layers);
        b.put(p + ".weight_ih_l" + s, regate(W.f.data(), I));
        b.put(p + ".weight_hh_l" + s, regate(R.f.data(), H));
        if (n->in.size() > 3 && !n->in[3].empty()) {
            const Tensor &B = c.weight(n, 3, p.c_str());
            if (B.numel() != 6 * H) fail("%s: %s: GRU bias %s, expected [1,%lld]", c.g.file.c_str(), p.c_str(), dims_str(B).c_str(), (long long)(6 * H));
            b.put(p + ".bias_ih_l" + s, regate(B.f.data(), 1));
            b.put(p + ".bias_hh_l" + s, regate(B.f.data() + 3 * H, 1));
        } else {
            b.put(p + ".bias_ih_l" + s, (size_t)(3 * H), 0.f);
            b.put(p + ".bias_hh_l" + s, (size_t)(3 * H), 0.f);
        }
        if (layers == 0) *hidden = (int)H;
        else if (*hidden != (int)H) fail("%s: %s: GRU layers of different hidden sizes", c.g.file.c_str(), p.c_str());
        ++layers;
    }
    return layers;
}

// Is the output of `linear_out` (Einsum -> Reshape -> Relu) added to something right away?  That Add is SqueezedGRU_S's skip
// connection (modules.py:731-737); the decoders' own additions happen after a reshape / transpose of that tensor.
bool relu_feeds_add(const Cursor &c, const Node *einsum) {
    std::string t = einsum->out.empty() ? "" : einsum->out[0];
    for (int hop = 0; hop < 4 && !t.empty(); ++hop) {
        const auto cons = c.consumers(t);
        if (cons.size() != 1) return false;
        if (cons[0]->op == "Relu") {
            for (const Node *n : c.consumers(cons[0]->out[0]))
                if (n->op == "Add") return true;
            return false;
        }
        t = cons[0]->out.empty() ? "" : cons[0]->out[0];
    }
    return false;
}

// SqueezedGRU_S (modules.py:677-738): linear_in -> GRU x layers -> [linear_out] (+ skip(input)).  Returns the skip kind.
int take_squeezed_gru(Cursor &c, Builder &b, const std::string &p, const std::string &gru_name, bool has_linear_out, int *layers, int *hidden,
                      const Tensor **w_in) {
    *w_in = &take_einsum(c, b, p + ".linear_in.0.weight");
    *layers = take_grus(c, b, gru_name, hidden);
    if (*layers == 0) fail("%s: %s: no GRU node after linear_in", c.g.file.c_str(), p.c_str());
    if (!has_linear_out) return DFX_SKIP_NONE;
    const Node *lo = c.peek();
    take_einsum(c, b, p + ".linear_out.0.weight");
    if (!relu_feeds_add(c, lo)) return DFX_SKIP_NONE;
    if (c.peek() && c.peek()->op == "Einsum") {
        take_einsum(c, b, p + ".gru_skip.weight");
        return DFX_SKIP_GROUPEDLINEAR;
    }
    return DFX_SKIP_IDENTITY;
}

const TarEntry *find_entry(const std::vector<TarEntry> &es, const char *base) {
    for (const TarEntry &e : es)
        if (basename_of(e.path) == base) return &e;  // tract.rs:47-62: path.ends_with(<file name>) compares whole components
    return nullptr;
}

// A scalar constant operand of the first `op` node downstream of tensor `t` (within a few element-wise hops)
bool scalar_after(const Cursor &c, std::string t, const char *op, float *val, std::string *out) {
    for (int hop = 0; hop < 3; ++hop) {
        for (const Node *n : c.consumers(t)) {
            if (n->op != op) continue;
            for (const std::string &i : n->in) {
                const Tensor *k = c.g.cst(i);
                if (k && k->dtype == 1 && k->numel() == 1 && !k->f.empty()) {
                    *val = k->f[0];
                    *out = n->out[0];
                    return true;
                }
            }
        }
        const auto cons = c.consumers(t);
        if (cons.size() != 1 || cons[0]->out.empty()) return false;
        t = cons[0]->out[0];
    }
    return false;
}

void read_targz(const char *path, dfx_model_cfg *cfg, std::vector<float> *blob, std::string *version) {
    const Bytes tar = gunzip(read_file(path));
    const std::vector<TarEntry> es = untar(tar);
    const TarEntry *e_enc = find_entry(es, "enc.onnx"), *e_erb = find_entry(es, "erb_dec.onnx"), *e_df = find_entry(es, "df_dec.onnx"),
                   *e_ini = find_entry(es, "config.ini"), *e_ver = find_entry(es, "version.txt");
    if (!e_enc || !e_erb || !e_df) fail("'%s': enc.onnx / erb_dec.onnx / df_dec.onnx not all present in the model tar file", path);
    if (!e_ini) fail("'%s': Could not load config from tar file (no config.ini)", path);
    if (e_ver && version) *version = trim(std::string(reinterpret_cast<const char *>(e_ver->data), e_ver->size));
    const Ini ini = parse_ini(e_ini->data, e_ini->size);

    dfx_model_cfg c;
    memset(&c, 0, sizeof(c));
    // tract.rs:245-246: both sections must exist
    if (!ini.count("deepfilternet") || !ini.count("df")) fail("config.ini: sections [df] and [deepfilternet] are required");
    c.sr = (int32_t)ini_int(ini, "df", "sr");
    c.hop_size = (int32_t)ini_int(ini, "df", "hop_size");
    c.fft_size = (int32_t)ini_int(ini, "df", "fft_size");
    c.min_nb_freqs = (int32_t)ini_int(ini, "df", "min_nb_erb_freqs");
    c.nb_erb = (int32_t)ini_int(ini, "df", "nb_erb");
    c.nb_df = (int32_t)ini_int(ini, "df", "nb_df");
    c.df_order = (int32_t)ini_int(ini, "df", "df_order", "deepfilternet");
    c.conv_lookahead = (int32_t)ini_int(ini, "deepfilternet", "conv_lookahead");
    c.df_lookahead = (int32_t)ini_int(ini, "df", "df_lookahead", "deepfilternet");
    if (const std::string *a = ini_get(ini, "df", "norm_alpha")) c.norm_alpha = strtof(a->c_str(), nullptr);
    else {
        const std::string *tau = ini_get(ini, "df", "norm_tau");
        if (!tau) fail("config.ini: neither norm_alpha nor norm_tau in section [df]");
        c.norm_alpha = calc_norm_alpha(c.sr, c.hop_size, strtof(tau->c_str(), nullptr));
    }
    const std::string *mt = ini_get(ini, "train", "model");
    if (!mt) fail("config.ini: option 'model' missing from section [train]");
    if (*mt == "deepfilternet2") fail("DeepFilterNet2 models are deprecated. Please use version v0.3.1 for these models.");  // tract.rs:310-312
    if (*mt != "deepfilternet3") fail("Unsupported model type %s", mt->c_str());                                            // tract.rs:314
    c.mask_pf = 0;  // DfNet-level options (deepfilternet3.py:373-378) are not part of the exported graphs; the runtime's post filter is
    c.pf_beta = 0.02f;  // RuntimeParams::post_filter_beta (tract.rs:150-167)
    c.lsnr_min = -15;
    c.lsnr_max = 35;

    const Graph g_enc = parse_model("enc.onnx", e_enc->data, e_enc->size);
    const Graph g_erb = parse_model("erb_dec.onnx", e_erb->data, e_erb->size);
    const Graph g_df = parse_model("df_dec.onnx", e_df->data, e_df->size);
    Builder b;

    // ---- encoder (deepfilternet3.py:100-185; forward :159-185)
    {
        Cursor k(g_enc);
        const Node *first = k.peek();
        if (!first || first->op != "Conv") fail("enc.onnx: does not start with erb_conv0");
        const Tensor &w0 = k.weight(first, 1, "enc.erb_conv0");
        if (w0.dims.size() != 4) fail("enc.onnx: erb_conv0 weight %s", dims_str(w0).c_str());
        const int C = (int)w0.dims[0];
        c.conv_ch = C;
        take_conv_block(k, b, "enc.erb_conv0", 1, C, 3, 3, false);
        take_conv_block(k, b, "enc.erb_conv1", C, C, 1, 3, false);
        take_conv_block(k, b, "enc.erb_conv2", C, C, 1, 3, false);
        take_conv_block(k, b, "enc.erb_conv3", C, C, 1, 3, false);
        take_conv_block(k, b, "enc.df_conv0", 2, C, 3, 3, false);
        take_conv_block(k, b, "enc.df_conv1", C, C, 1, 3, false);
        const Tensor &wfc = take_einsum(k, b, "enc.df_fc_emb.0.weight");
        const int emb = C * c.nb_erb / 4;
        c.enc_lin_groups = (int32_t)wfc.dims[0];
        if (wfc.dims[0] * wfc.dims[1] != (int64_t)C * c.nb_df / 2 || wfc.dims[0] * wfc.dims[2] != emb)
            fail("enc.onnx: df_fc_emb weight %s does not map %d -> %d features (conv_ch %d, nb_df %d, nb_erb %d from config.ini)", dims_str(wfc).c_str(), C * c.nb_df / 2, emb, C, c.nb_df, c.nb_erb);
        int layers = 0, hidden = 0;
        const Tensor *w_in = nullptr;
        c.emb_gru_skip_enc = take_squeezed_gru(k, b, "enc.emb_gru", "enc.emb_gru.gru", true, &layers, &hidden, &w_in);
        if (layers != 1) fail("enc.onnx: %d GRU layers in the encoder (DeepFilterNet3 has one)", layers);
        c.emb_hidden_dim = hidden;
        c.lin_groups = (int32_t)w_in->dims[0];
        const int64_t in_dim = w_in->dims[0] * w_in->dims[1];
        if (in_dim == 2 * (int64_t)emb) c.enc_concat = 1;
        else if (in_dim != emb) fail("enc.onnx: emb_gru.linear_in takes %lld features, the embedding has %d", (long long)in_dim, emb);
        // lsnr_fc: Linear(emb, 1) as MatMul + Add (or Gemm), then sigmoid * (max - min) + min
        const Node *fc = k.peek();
        if (!fc || (fc->op != "MatMul" && fc->op != "Gemm")) fail("enc.onnx: lsnr_fc not found after the embedding GRU");
        ++k.pos;
        const Tensor &wl = k.weight(fc, 1, "enc.lsnr_fc");
        if (wl.numel() != emb) fail("enc.onnx: lsnr_fc weight %s, expected %d elements", dims_str(wl).c_str(), emb);
        b.put("enc.lsnr_fc.0.weight", wl.f);
        std::vector<float> bias(1, 0.f);
        if (fc->op == "Gemm" && fc->in.size() > 2) bias = k.weight(fc, 2, "enc.lsnr_fc bias").f;
        else
            for (const Node *n : k.consumers(fc->out[0]))
                if (n->op == "Add")
                    for (const std::string &i : n->in) {
                        const Tensor *t = g_enc.cst(i);
                        if (t && t->dtype == 1 && t->numel() == 1) bias = t->f;
                    }
        b.put("enc.lsnr_fc.0.bias", bias);
        // fc -> (Add bias) -> Sigmoid -> Mul(max - min) -> Add(min)   (deepfilternet3.py:149-151,183-184)
        float scale = 0.f, offset = 0.f;
        std::string t = fc->out[0], t1, t2;
        for (int hop = 0; hop < 3; ++hop) {
            const auto cons = k.consumers(t);
            if (cons.empty() || cons[0]->out.empty()) break;
            t = cons[0]->out[0];
            if (cons[0]->op == "Sigmoid") break;
        }
        if (scalar_after(k, t, "Mul", &scale, &t1) && scalar_after(k, t1, "Add", &offset, &t2)) {
            c.lsnr_min = (int32_t)lrintf(offset);
            c.lsnr_max = (int32_t)lrintf(offset + scale);
        }
        if (k.peek()) fail("enc.onnx: unexpected %s '%s' after lsnr_fc", k.peek()->op.c_str(), k.peek()->name.c_str());
    }
    const int C = c.conv_ch, emb = C * c.nb_erb / 4;
    // ---- ERB decoder (deepfilternet3.py:188-275)
    {
        Cursor k(g_erb);
        int layers = 0, hidden = 0;
        const Tensor *w_in = nullptr;
        c.emb_gru_skip = take_squeezed_gru(k, b, "erb_dec.emb_gru", "erb_dec.emb_gru.gru", true, &layers, &hidden, &w_in);
        c.emb_num_layers = layers + 1;
        if (hidden != c.emb_hidden_dim) fail("erb_dec.onnx: GRU hidden size %d, the encoder's is %d", hidden, c.emb_hidden_dim);
        if ((int32_t)w_in->dims[0] != c.lin_groups) fail("erb_dec.onnx: %lld linear groups, the encoder has %d", (long long)w_in->dims[0], c.lin_groups);
        take_conv_block(k, b, "erb_dec.conv3p", C, C, 1, 1, false);
        take_conv_block(k, b, "erb_dec.convt3", C, C, 1, 3, false);
        take_conv_block(k, b, "erb_dec.conv2p", C, C, 1, 1, false);
        take_conv_block(k, b, "erb_dec.convt2", C, C, 1, 3, true);
        take_conv_block(k, b, "erb_dec.conv1p", C, C, 1, 1, false);
        take_conv_block(k, b, "erb_dec.convt1", C, C, 1, 3, true);
        take_conv_block(k, b, "erb_dec.conv0p", C, C, 1, 1, false);
        take_conv_block(k, b, "erb_dec.conv0_out", C, 1, 1, 3, false);
        if (k.peek()) fail("erb_dec.onnx: unexpected %s '%s' after conv0_out", k.peek()->op.c_str(), k.peek()->name.c_str());
    }
    // ---- DF decoder (deepfilternet3.py:278-331)
    {
        Cursor k(g_df);
        int layers = 0, hidden = 0;
        const Tensor *w_in = nullptr;
        take_squeezed_gru(k, b, "df_dec.df_gru", "df_dec.df_gru.gru", false, &layers, &hidden, &w_in);
        c.df_num_layers = layers;
        c.df_hidden_dim = hidden;
        if (w_in->dims[0] * w_in->dims[1] != emb) fail("df_dec.onnx: df_gru.linear_in takes %lld features, the embedding has %d", (long long)(w_in->dims[0] * w_in->dims[1]), emb);
        c.df_gru_skip = DFX_SKIP_NONE;
        if (k.peek() && k.peek()->op == "Einsum" && k.peek(1) && k.peek(1)->op == "Conv") {
            take_einsum(k, b, "df_dec.df_skip.weight");
            c.df_gru_skip = DFX_SKIP_GROUPEDLINEAR;
        } else {
            // identity: c = df_gru(emb) + emb — the only Add with the graph input `emb` as a direct operand
            for (const Node &n : g_df.nodes)
                if (n.op == "Add")
                    for (const std::string &i : n.in)
                        if (i == "emb") c.df_gru_skip = DFX_SKIP_IDENTITY;
        }
        const Node *cp = k.peek();
        if (!cp || cp->op != "Conv") fail("df_dec.onnx: df_convp not found after the GRU");
        const Tensor &wp = k.weight(cp, 1, "df_dec.df_convp");
        if (wp.dims.size() != 4) fail("df_dec.onnx: df_convp weight %s", dims_str(wp).c_str());
        c.df_pathway_kernel_size_t = (int32_t)wp.dims[2];
        if (wp.dims[0] != 2 * (int64_t)c.df_order) fail("df_dec.onnx: df_convp has %lld output channels, df_order %d (config.ini) needs %d", (long long)wp.dims[0], c.df_order, 2 * c.df_order);
        take_conv_block(k, b, "df_dec.df_convp", C, 2 * c.df_order, c.df_pathway_kernel_size_t, 1, false);
        const Tensor &wo = take_einsum(k, b, "df_dec.df_out.0.weight");
        if (wo.dims[0] * wo.dims[2] != (int64_t)c.nb_df * 2 * c.df_order || wo.dims[0] * wo.dims[1] != c.df_hidden_dim)
            fail("df_dec.onnx: df_out weight %s does not map %d -> nb_df * 2 * df_order = %d", dims_str(wo).c_str(), c.df_hidden_dim, c.nb_df * 2 * c.df_order);
        if ((int32_t)wo.dims[0] != c.lin_groups) fail("df_dec.onnx: df_out has %lld groups, the encoder's linear layers %d", (long long)wo.dims[0], c.lin_groups);
        if (k.peek()) fail("df_dec.onnx: unexpected %s '%s' after df_out", k.peek()->op.c_str(), k.peek()->name.c_str());
    }

    // ---- pack per the engine's manifest
    const DfxManifest man = dfx_build_manifest(c);
    blob->assign((size_t)man.total, 0.f);
    for (const DfxTensor &t : man.t) {
        auto it = b.sd.find(t.name);
        if (it == b.sd.end()) fail("'%s': the graphs do not provide '%s'", path, t.name.c_str());
        if ((int64_t)it->second.size() != t.numel())
            fail("'%s': '%s' has %zu elements in the graphs, the engine expects %lld", path, t.name.c_str(), it->second.size(), (long long)t.numel());
        memcpy(blob->data() + t.offset, it->second.data(), (size_t)t.numel() * sizeof(float));
    }
    *cfg = c;
}

}  // namespace

// Internal (dfx_capi.hip): also hands back version.txt (tract.rs:56-59 logs it)
int dfx_read_onnx_targz(const char *path, dfx_model_cfg *cfg, std::vector<float> *blob, std::string *version) {
    try {
        read_targz(path, cfg, blob, version);
    } catch (const Fail &f) {
        DFX_FAIL(DFX_ERR_INVALID_ARG, "%s", f.msg.c_str());
    } catch (const std::exception &e) {
        DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_onnx: %s", e.what());
    }
    return DFX_OK;
}

extern "C" int dfx_onnx_targz_read(const char *path, dfx_model_cfg *cfg_out, float *blob_out, int64_t blob_cap, int64_t *blob_floats) {
    if (!path || !cfg_out) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_onnx_targz_read: null argument");
    dfx_model_cfg c;
    std::vector<float> blob;
    if (int rc = dfx_read_onnx_targz(path, &c, &blob, nullptr)) return rc;
    *cfg_out = c;
    if (blob_floats) *blob_floats = (int64_t)blob.size();
    if (blob_out) {
        if (blob_cap < (int64_t)blob.size()) DFX_FAIL(DFX_ERR_INVALID_ARG, "dfx_onnx_targz_read: blob_out holds %lld floats, %zu needed", (long long)blob_cap, blob.size());
        memcpy(blob_out, blob.data(), blob.size() * sizeof(float));
    }
    return DFX_OK;
}
