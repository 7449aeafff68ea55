// onnx_reader.cpp — the weights fastembed actually caches: FastEmbedder::with_cache_dir
// (/root/reference/src/embed/embedder.rs:218-245) points fastembed at a cache directory in which hf-hub
// leaves the model's ONNX export (onnx/model.onnx for Xenova/bge-small-en-v1.5, model.onnx or
// model_optimized.onnx for other registry entries) beside config.json and tokenizer.json — not
// model.safetensors.  This file reads the BERT parameters out of such an ONNX file into the flat f32 block
// of include/cs_bert_params.h.  Host-only C++; protobuf wire format decoded by hand (no protobuf library,
// no onnxruntime): only the handful of messages and fields below are interpreted, the rest is skipped.
//
//   ModelProto   { graph = 7 }
//   GraphProto   { node = 1 (repeated NodeProto), initializer = 5 (repeated TensorProto) }
//   NodeProto    { input = 1, output = 2, name = 3, op_type = 4, attribute = 5 }
//   AttributeProto { name = 1, i = 3 }
//   TensorProto  { dims = 1, data_type = 2 (1 FLOAT, 10 FLOAT16, 16 BFLOAT16; 2 UINT8 / 3 INT8 with a scale), float_data = 4,
//                  name = 8, raw_data = 9, external_data = 13, data_location = 14 }
//
// Where the tensors are.  An exporter (torch.onnx / optimum) keeps the state-dict name of every parameter an
// op consumes as it is — embeddings, LayerNorm weights, all biases — but folds the transpose of a Linear
// weight into a constant, so `encoder.layer.N.….dense.weight` [out, in] arrives as an anonymous
// initializer (`onnx::MatMul_1234`) of shape [in, out].  Those are found through the graph: the Add that
// consumes the layer's bias, the MatMul feeding that Add, the MatMul's second input.  Gemm nodes (weight as
// input 1, transB) and what onnxruntime's transformer optimiser leaves (model_optimized.onnx) are understood as well: the
// fused com.microsoft Attention / QAttention node (packed [H, 3H] QKV weight as input 1, packed [3H] bias as input 2)
// and the nodes that swallow a bias Add — SkipLayerNormalization(input, skip, gamma, beta, bias), BiasGelu(x, bias).
//
// Quantised exports (the reference's DEFAULT model is one: ModelType::AllMiniLML6V2Q, embedder.rs:12-13,367-372 — fastembed
// downloads its model_quantized.onnx, written by onnxruntime's dynamic quantiser): every Linear weight W is stored as
// W_quantized (INT8 / UINT8, [in, out]) + W_scale + W_zero_point — per tensor or per channel — and consumed by
// DynamicQuantizeLinear -> MatMulInteger -> Cast -> Mul(scales) -> Add(bias); Gather tables may be quantised in place the
// same way.  They are read as w = (q - zero_point) * scale into the f32 block, and cs_bert_params_from_onnx_q also hands out
// every weight column's scale: with those the embedder runs each Linear as the file's graph does (activations re-quantised
// to 8 bits per call, integer product: csrc/gemm_q8.hip, CS_GEMM_Q8_DYNAMIC).  Without them (CS_ENCODER_QUANT=0, or a file
// only part of whose Linears are quantised) it runs the f32 GRAPH OF THE QUANTISED WEIGHTS, whose embeddings differ from
// the reference's by the activation-rounding noise.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/cs_bert_params.h"
#include "common.hpp"

namespace {

using cs::fail;

struct Span { const uint8_t* p = nullptr; const uint8_t* end = nullptr; };

bool varint(Span& s, uint64_t& v) {
    v = 0;
    for (int shift = 0; shift < 64 && s.p < s.end; shift += 7) {
        const uint8_t b = *s.p++;
        v |= (uint64_t)(b & 0x7f) << shift;
        if (!(b & 0x80)) return true;
    }
    return false;
}

// One field of a message: key, then (for wire type 2) the payload span, (for 0) the value.
struct Field { uint32_t num = 0, wire = 0; uint64_t val = 0; Span sub; };

bool next_field(Span& s, Field& f) {
    uint64_t key;
    if (!varint(s, key)) return false;
    f.num = (uint32_t)(key >> 3);
    f.wire = (uint32_t)(key & 7);
    switch (f.wire) {
        case 0: return varint(s, f.val);
        case 1: if (s.end - s.p < 8) return false; std::memcpy(&f.val, s.p, 8); s.p += 8; return true;
        case 5: if (s.end - s.p < 4) return false; f.val = 0; std::memcpy(&f.val, s.p, 4); s.p += 4; return true;
        case 2: {
            uint64_t n;
            if (!varint(s, n) || (uint64_t)(s.end - s.p) < n) return false;
            f.sub = Span{s.p, s.p + n};
            s.p += n;
            return true;
        }
        default: return false;  // groups (3, 4) do not occur in ONNX files
    }
}

std::string str_of(const Span& s) { return std::string(reinterpret_cast<const char*>(s.p), (size_t)(s.end - s.p)); }

struct Tensor {
    std::string name;
    std::vector<uint64_t> dims;
    int dtype = 0;
    Span raw;          // raw_data
    Span float_data;   // packed float_data (dtype 1 only)
    bool external = false;
    // product of the dims, saturating: dims come from the file (a 2^40 x 2^40 "tensor" must not wrap to a small count)
    uint64_t count() const {
        uint64_t c = 1;
        for (uint64_t d : dims) {
            if (d != 0 && c > UINT64_MAX / d) return UINT64_MAX;
            c *= d;
        }
        return c;
    }
};

struct Node {
    std::string op;
    std::vector<std::string> in, out;
    int64_t transB = 0;
};

bool parse_tensor(Span s, Tensor& t) {
    Field f;
    while (s.p < s.end) {
        if (!next_field(s, f)) return false;
        if (f.num == 1 && f.wire == 0) t.dims.push_back(f.val);
        else if (f.num == 1 && f.wire == 2) {  // packed dims
            Span d = f.sub;
            uint64_t v;
            while (d.p < d.end) { if (!varint(d, v)) return false; t.dims.push_back(v); }
        } else if (f.num == 2 && f.wire == 0) t.dtype = (int)f.val;
        else if (f.num == 4 && f.wire == 2) t.float_data = f.sub;
        else if (f.num == 4 && f.wire == 5) return false;  // unpacked float_data: not produced by any exporter we know
        else if (f.num == 8 && f.wire == 2) t.name = str_of(f.sub);
        else if (f.num == 9 && f.wire == 2) t.raw = f.sub;
        else if (f.num == 13) t.external = true;
        else if (f.num == 14 && f.wire == 0 && f.val == 1) t.external = true;
    }
    return true;
}

bool parse_node(Span s, Node& n) {
    Field f;
    while (s.p < s.end) {
        if (!next_field(s, f)) return false;
        if (f.num == 1 && f.wire == 2) n.in.push_back(str_of(f.sub));
        else if (f.num == 2 && f.wire == 2) n.out.push_back(str_of(f.sub));
        else if (f.num == 4 && f.wire == 2) n.op = str_of(f.sub);
        else if (f.num == 5 && f.wire == 2) {  // attribute: only transB matters
            Span a = f.sub;
            Field g;
            std::string an;
            int64_t iv = 0;
            while (a.p < a.end) {
                if (!next_field(a, g)) return false;
                if (g.num == 1 && g.wire == 2) an = str_of(g.sub);
                else if (g.num == 3 && g.wire == 0) iv = (int64_t)g.val;
            }
            if (an == "transB") n.transB = iv;
        }
    }
    return true;
}

float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// How an INT8 / UINT8 initialiser turns back into numbers: w = (q - zero_point) * scale, one pair for the whole tensor
// or one per slice along `axis` (onnxruntime's quantize_dynamic / optimum's ORTQuantizer write both forms).
struct Quant {
    const Tensor* scale = nullptr;
    const Tensor* zp = nullptr;   // may be null: zero point 0
};

// element i of a tensor as f32 (FLOAT, FLOAT16, BFLOAT16; INT8 / UINT8 through a Quant)
struct Reader {
    const Tensor* t;
    const uint8_t* base;
    int esz;
    bool ok;
    // quantised payloads: per-element scale index = (i / q_inner) % q_n  (q_n == 1: per tensor)
    std::vector<float> q_scale;
    std::vector<int32_t> q_zp;
    uint64_t q_inner = 1, q_n = 1;
    // rows x cols matrix quantised per tensor, per column (scale count == cols) or per row (== rows)
    Reader(const Tensor& tt, const Quant& q, uint64_t rows, uint64_t cols) : t(&tt), base(nullptr), esz(0), ok(false) {
        if ((tt.dtype != 2 && tt.dtype != 3) || !q.scale || tt.count() != (uint64_t)(tt.raw.end - tt.raw.p)) return;
        Reader rs(*q.scale);
        if (!rs.ok) return;
        const uint64_t ns = q.scale->count();
        if (ns == 1) { q_n = 1; q_inner = 1; }
        else if (ns == cols) { q_n = cols; q_inner = 1; }
        else if (ns == rows) { q_n = rows; q_inner = cols; }
        else return;
        q_scale.resize(ns);
        for (uint64_t i = 0; i < ns; ++i) q_scale[i] = rs.at(i);
        q_zp.assign(ns, 0);
        if (q.zp) {
            const uint64_t nz = q.zp->count();
            if ((q.zp->dtype != 2 && q.zp->dtype != 3) || (nz != 1 && nz != ns) || nz != (uint64_t)(q.zp->raw.end - q.zp->raw.p)) return;
            for (uint64_t i = 0; i < ns; ++i) {
                const uint8_t b = q.zp->raw.p[nz == 1 ? 0 : i];
                q_zp[i] = q.zp->dtype == 3 ? (int32_t)(int8_t)b : (int32_t)b;
            }
        }
        base = tt.raw.p;
        esz = 1;
        ok = true;
    }
    explicit Reader(const Tensor& tt) : t(&tt), base(nullptr), esz(0), ok(false) {
        const uint64_t n = tt.count();
        // byte lengths are compared by division: n * esz could wrap for a count taken from the file
        auto holds = [&](const Span& sp, uint64_t e) {
            const uint64_t bytes = (uint64_t)(sp.end - sp.p);
            return bytes % e == 0 && bytes / e == n;
        };
        if (tt.dtype == 1) {
            esz = 4;
            if (holds(tt.raw, 4)) base = tt.raw.p;
            else if (holds(tt.float_data, 4)) base = tt.float_data.p;
        } else if (tt.dtype == 10 || tt.dtype == 16) {
            esz = 2;
            if (holds(tt.raw, 2)) base = tt.raw.p;
        }
        ok = base != nullptr;
    }
    float at(uint64_t i) const {
        if (esz == 1) {
            const uint64_t c = q_n == 1 ? 0 : (i / q_inner) % q_n;
            const int32_t v = t->dtype == 3 ? (int32_t)(int8_t)base[i] : (int32_t)base[i];
            return (float)(v - q_zp[c]) * q_scale[c];
        }
        if (esz == 4) { float f; std::memcpy(&f, base + i * 4, 4); return f; }
        uint16_t v;
        std::memcpy(&v, base + i * 2, 2);
        if (t->dtype == 16) { const uint32_t b = (uint32_t)v << 16; float f; std::memcpy(&f, &b, 4); return f; }
        return half_to_float(v);
    }
};

struct Model {
    std::map<std::string, Tensor> init;
    std::vector<Node> nodes;
    std::map<std::string, size_t> producer;                 // output name -> node
    std::multimap<std::string, size_t> consumers;           // input name -> nodes

    const Tensor* tensor(const std::string& name) const {
        auto it = init.find(name);
        return it == init.end() ? nullptr : &it->second;
    }
    // initializers sometimes reach an op through Identity / Cast nodes
    const Tensor* resolve(const std::string& name, int depth = 0) const {
        if (const Tensor* t = tensor(name)) return t;
        if (depth > 4) return nullptr;
        auto it = producer.find(name);
        if (it == producer.end()) return nullptr;
        const Node& n = nodes[it->second];
        if ((n.op == "Identity" || n.op == "Cast") && n.in.size() == 1) return resolve(n.in[0], depth + 1);
        return nullptr;
    }
};

int32_t parse_model(Span file, Model& m, const char* path) {
    Field f;
    Span graph;
    while (file.p < file.end) {
        if (!next_field(file, f)) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s is not an ONNX file (bad protobuf)", path);
        if (f.num == 7 && f.wire == 2) graph = f.sub;
    }
    if (!graph.p) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s holds no graph", path);
    while (graph.p < graph.end) {
        if (!next_field(graph, f)) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s has a malformed graph", path);
        if (f.num == 5 && f.wire == 2) {
            Tensor t;
            if (!parse_tensor(f.sub, t)) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s has a malformed initializer", path);
            std::string nm = t.name;
            m.init.emplace(std::move(nm), std::move(t));
        } else if (f.num == 1 && f.wire == 2) {
            Node n;
            if (!parse_node(f.sub, n)) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s has a malformed node", path);
            m.nodes.push_back(std::move(n));
        }
    }
    for (size_t i = 0; i < m.nodes.size(); ++i) {
        for (const auto& o : m.nodes[i].out) m.producer[o] = i;
        for (const auto& in : m.nodes[i].in) m.consumers.emplace(in, i);
    }
    return CS_OK;
}

// dst[r * cols + c] = src, where src is [rows, cols] (direct) or [cols, rows] (transposed)
// `q` (optional): the tensor is INT8 / UINT8 and dequantised while it is copied; its stored shape is [rows, cols]
// (direct) or [cols, rows] (transposed).
int32_t copy_matrix(const Tensor& t, uint64_t rows, uint64_t cols, bool transposed, float* dst, const char* what,
                    const Quant* q = nullptr) {
    if (t.external)
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: %s keeps its data in an external file", what);
    Reader r = q ? (transposed ? Reader(t, *q, cols, rows) : Reader(t, *q, rows, cols)) : Reader(t);
    if (!r.ok)
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: %s has data type %d, a truncated payload or "
                    "(quantised) no usable scale / zero point (FLOAT, FLOAT16, BFLOAT16; INT8 / UINT8 with a scale)", what, t.dtype);
    if (!transposed) {
        for (uint64_t i = 0; i < rows * cols; ++i) dst[i] = r.at(i);
    } else {
        for (uint64_t c = 0; c < cols; ++c)
            for (uint64_t rr = 0; rr < rows; ++rr) dst[rr * cols + c] = r.at(c * rows + rr);
    }
    return CS_OK;
}

bool shape_is(const Tensor& t, std::initializer_list<uint64_t> want) {
    return t.dims == std::vector<uint64_t>(want);
}

// onnxruntime's quantizer names the pieces of a quantised initialiser W as W_quantized, W_scale, W_zero_point.
bool quant_of(const Model& m, const Tensor& wq, Quant& q) {
    static const std::string suf = "_quantized";
    if (wq.name.size() <= suf.size() || wq.name.compare(wq.name.size() - suf.size(), suf.size(), suf) != 0) return false;
    const std::string stem = wq.name.substr(0, wq.name.size() - suf.size());
    q.scale = m.tensor(stem + "_scale");
    q.zp = m.tensor(stem + "_zero_point");
    return q.scale != nullptr;
}

// The weight behind the tensor `y` = x W (+ nothing yet): y's producer is the MatMul itself or, in a dynamically quantised
// file, the Mul(Cast(MatMulInteger(..)), Mul(x_scale, W_scale)) that follows it — walk back to the product.
const Tensor* weight_behind(const Model& m, const std::string& y, uint64_t out, uint64_t in, bool& transposed, Quant* quant) {
    auto p = m.producer.find(y);
    if (p == m.producer.end()) return nullptr;
    const Node* mm = &m.nodes[p->second];
    for (int hop = 0; hop < 4 && mm && (mm->op == "Mul" || mm->op == "Cast"); ++hop) {
        const Node* next = nullptr;
        for (const std::string& inp : mm->in) {
            auto pp = m.producer.find(inp);
            if (pp == m.producer.end()) continue;
            const Node& cand = m.nodes[pp->second];
            if (cand.op == "MatMulInteger" || cand.op == "Cast") { next = &cand; break; }
            if (cand.op == "Mul" && !next) next = &cand;  // (the scale product is a Mul too: only if nothing better)
        }
        mm = next;
    }
    if (!mm) return nullptr;
    if (mm->op == "MatMulInteger" && mm->in.size() >= 2 && quant) {
        const Tensor* w = m.tensor(mm->in[1]);
        if (w && shape_is(*w, {in, out}) && quant_of(m, *w, *quant)) { transposed = true; return w; }
        return nullptr;
    }
    if (mm->op != "MatMul" || mm->in.size() != 2) return nullptr;
    const Tensor* w = m.resolve(mm->in[1]);
    if (w && shape_is(*w, {in, out})) { transposed = true; return w; }
    return nullptr;
}

// The [out, in] weight that belongs to `bias_name` (see the header comment).  -> tensor + whether it is [in, out].
// The bias is consumed by Gemm (with the weight), by the Add behind a MatMul, or — in files that went through
// onnxruntime's transformer optimiser (model_optimized.onnx) — by the fused node that swallowed that Add:
// com.microsoft SkipLayerNormalization(input, skip, gamma, beta, bias) or BiasGelu / FastGelu(x, bias).
// Dynamic quantisation (the *Q models of the registry, e.g. Xenova/all-MiniLM-L6-v2's model_quantized.onnx) replaces
// MatMul by DynamicQuantizeLinear -> MatMulInteger(x_q, W_quantized, x_zp, W_zero_point) -> Cast -> Mul(scales): the weight
// is then the INT8 / UINT8 second input of that MatMulInteger and *quant says how to read it.
const Tensor* weight_of_bias(const Model& m, const std::string& bias_name, uint64_t out, uint64_t in, bool& transposed,
                             Quant* quant = nullptr) {
    auto range = m.consumers.equal_range(bias_name);
    for (auto it = range.first; it != range.second; ++it) {
        const Node& n = m.nodes[it->second];
        if (n.op == "Gemm" && n.in.size() >= 3 && n.in[2] == bias_name) {
            const Tensor* w = m.resolve(n.in[1]);
            if (!w) continue;
            if (n.transB && shape_is(*w, {out, in})) { transposed = false; return w; }
            if (!n.transB && shape_is(*w, {in, out})) { transposed = true; return w; }
        } else if (n.op == "Add" && n.in.size() == 2) {
            const std::string& other = n.in[0] == bias_name ? n.in[1] : n.in[0];
            if (const Tensor* w = weight_behind(m, other, out, in, transposed, quant)) return w;
        } else if (n.op == "SkipLayerNormalization" && n.in.size() >= 5 && n.in[4] == bias_name) {
            for (int k = 0; k < 2; ++k)  // input or skip: whichever the product feeds
                if (const Tensor* w = weight_behind(m, n.in[k], out, in, transposed, quant)) return w;
        } else if ((n.op == "BiasGelu" || n.op == "FastGelu") && n.in.size() >= 2 && n.in[1] == bias_name) {
            if (const Tensor* w = weight_behind(m, n.in[0], out, in, transposed, quant)) return w;
        }
    }
    return nullptr;
}

// ---- NomicBert exports (the registry's nomic-embed-text entries: what fastembed caches is onnx/model.onnx, or
// onnx/model_quantized.onnx for the *Q entry) ------------------------------------------------------------------------------
// The Linear layers of the published checkpoints have NO bias, so there is no named bias whose Add leads to the weight (the
// anchor of the BERT reader above): a torch export leaves every such weight as an anonymous transposed initialiser
// (`onnx::MatMul_N`, [in, out]) on the second input of a MatMul — MatMulInteger behind DynamicQuantizeLinear in the
// quantised file.  They are therefore taken by STRUCTURE: the MatMul / MatMulInteger nodes whose second input is a 2-D
// initialiser, in graph (= execution) order, are per layer  Wqkv [H, 3H] | out_proj [H, H] | fc11, fc12 [H, I] | fc2 [I, H]
// (the products between activations — Q K^T, P V — have no initialiser and drop out); of the two [H, I] products the gate
// (fc12) is the one whose result reaches a Sigmoid (silu(g) = g * sigmoid(g)); a file in which neither does (a fused
// activation op) is refused rather than guessed at.  LayerNorm parameters and the embedding tables keep their module names (emb_ln, encoder.layers.N.norm1 /
// norm2).  A quantised file is read as (q - zero_point) * scale into the f32 block: the Nomic encoder runs the f32 graph of
// those weights (the dynamic-quantisation mode is BERT's, embedder.hip).  Restated from how torch.onnx lays such a module
// out and pinned on a file torch's own exporter wrote (tests/golden/nomic_tiny_export.onnx); the hub's own export is not
// on disk here.
struct WeightProduct { const Node* node; const Tensor* w; Quant q; bool quantised; };
std::vector<WeightProduct> weight_products(const Model& m);

// does the float result of product node `n` reach a Sigmoid through at most `depth` Cast / Mul nodes?
bool reaches_sigmoid(const Model& m, const std::string& name, int depth) {
    auto range = m.consumers.equal_range(name);
    for (auto it = range.first; it != range.second; ++it) {
        const Node& c = m.nodes[it->second];
        if (c.op == "Sigmoid") return true;
    }
    if (depth <= 0) return false;
    for (auto it = range.first; it != range.second; ++it) {
        const Node& c = m.nodes[it->second];
        if ((c.op == "Cast" || c.op == "Mul") && !c.out.empty() && reaches_sigmoid(m, c.out[0], depth - 1)) return true;
    }
    return false;
}

int32_t nomic_params_from_onnx(const Model& m, const cs_bert_config* cfg, const cs_bert_offsets& o, float* params, const char* path) {
    const uint64_t H = cfg->hidden, I = cfg->intermediate;
    std::string mod_prefix;
    {
        static const std::string anchor = "embeddings.word_embeddings.weight";
        for (const auto& kv : m.init) {
            const std::string& nm = kv.first;
            if (nm.size() >= anchor.size() && nm.compare(nm.size() - anchor.size(), anchor.size(), anchor) == 0) {
                mod_prefix = nm.substr(0, nm.size() - anchor.size());
                break;
            }
        }
    }
    auto named = [&](const std::string& name) -> const Tensor* { return m.tensor(mod_prefix + name); };
    auto vec = [&](const std::string& name, uint64_t n, float* dst, bool optional) -> int32_t {
        const Tensor* t = named(name);
        if (!t) {
            if (optional) { std::memset(dst, 0, n * sizeof(float)); return CS_OK; }
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: tensor %s is missing from %s", name.c_str(), path);
        }
        if (t->count() != n || t->dims.size() > 2)
            return fail(CS_ERR_DIM_MISMATCH, "Failed to initialize embedding model: %s has %llu elements, config.json implies %llu",
                        name.c_str(), (unsigned long long)t->count(), (unsigned long long)n);
        return copy_matrix(*t, 1, n, false, dst, name.c_str());
    };
    auto table = [&](const std::string& name, uint64_t rows, float* dst) -> int32_t {
        const Tensor* t = named(name);
        if (!t) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: tensor %s is missing from %s", name.c_str(), path);
        if (!shape_is(*t, {rows, H}))
            return fail(CS_ERR_DIM_MISMATCH, "Failed to initialize embedding model: %s does not have shape [%llu, %llu]",
                        name.c_str(), (unsigned long long)rows, (unsigned long long)H);
        return copy_matrix(*t, rows, H, false, dst, name.c_str());
    };
    CS_TRY(table("embeddings.word_embeddings.weight", cfg->vocab_size, params + o.word));
    CS_TRY(table("embeddings.token_type_embeddings.weight", cfg->type_vocab_size, params + o.type));
    CS_TRY(vec("emb_ln.weight", H, params + o.emb_ln_g, false));
    CS_TRY(vec("emb_ln.bias", H, params + o.emb_ln_b, false));

    const std::vector<WeightProduct> prods = weight_products(m);
    if (prods.size() != (size_t)5 * cfg->layers)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s holds %zu weight products (MatMul / MatMulInteger with a "
                    "2-D initialiser), a nomic_bert export of %u layers holds %u", path, prods.size(), cfg->layers, 5 * cfg->layers);
    std::vector<float> packed((size_t)3 * H * H);
    for (uint32_t l = 0; l < cfg->layers; ++l) {
        cs_bert_layer_offsets lo;
        cs_bert_layer_layout(cfg, &o, l, &lo);
        const WeightProduct* e = &prods[(size_t)5 * l];
        auto want = [&](int i, uint64_t in, uint64_t out, const char* what) -> int32_t {
            if (!shape_is(*e[i].w, {in, out}))
                return fail(CS_ERR_DIM_MISMATCH, "Failed to initialize embedding model: layer %u's %s product does not hold a [%llu, %llu] "
                            "weight (%s)", l, what, (unsigned long long)in, (unsigned long long)out, e[i].w->name.c_str());
            return CS_OK;
        };
        CS_TRY(want(0, H, 3 * H, "Wqkv"));
        CS_TRY(want(1, H, H, "out_proj"));
        CS_TRY(want(2, H, I, "fc11 / fc12"));
        CS_TRY(want(3, H, I, "fc11 / fc12"));
        CS_TRY(want(4, I, H, "fc2"));
        auto copy = [&](int i, uint64_t out, uint64_t in, float* dst, const char* what) -> int32_t {
            return copy_matrix(*e[i].w, out, in, true, dst, what, e[i].quantised ? &e[i].q : nullptr);
        };
        CS_TRY(copy(0, 3 * H, H, packed.data(), "Wqkv"));
        std::memcpy(params + lo.q_w, packed.data(), H * H * sizeof(float));
        std::memcpy(params + lo.k_w, packed.data() + H * H, H * H * sizeof(float));
        std::memcpy(params + lo.v_w, packed.data() + 2 * H * H, H * H * sizeof(float));
        CS_TRY(copy(1, H, H, params + lo.ao_w, "out_proj"));
        const bool g2 = !e[2].node->out.empty() && reaches_sigmoid(m, e[2].node->out[0], 3);
        const bool g3 = !e[3].node->out.empty() && reaches_sigmoid(m, e[3].node->out[0], 3);
        if (g2 && g3)
            return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: both feed-forward up projections of layer %u reach a "
                        "Sigmoid in %s (not the swiglu arrangement)", l, path);
        if (!g2 && !g3)  // (a fused or contrib activation, a deeper Cast chain: guessing would swap value and gate silently — ADVICE r5)
            return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: neither feed-forward up projection of layer %u reaches a "
                        "Sigmoid in %s: which one is the SwiGLU gate cannot be read off this export", l, path);
        const int gate = g2 ? 2 : 3, value = g2 ? 3 : 2;
        CS_TRY(copy(value, I, H, params + lo.up_w, "fc11"));
        CS_TRY(copy(gate, I, H, params + lo.gate_w, "fc12"));
        CS_TRY(copy(4, H, I, params + lo.down_w, "fc2"));
        const std::string p = "encoder.layers." + std::to_string(l) + ".";
        // Linear biases: none in the published files (zero slots); taken by name where an export kept one
        std::vector<float> b3((size_t)3 * H);
        CS_TRY(vec(p + "attn.Wqkv.bias", 3 * H, b3.data(), true));
        std::memcpy(params + lo.q_b, b3.data(), H * sizeof(float));
        std::memcpy(params + lo.k_b, b3.data() + H, H * sizeof(float));
        std::memcpy(params + lo.v_b, b3.data() + 2 * H, H * sizeof(float));
        CS_TRY(vec(p + "attn.out_proj.bias", H, params + lo.ao_b, true));
        CS_TRY(vec(p + "mlp.fc11.bias", I, params + lo.up_b, true));
        CS_TRY(vec(p + "mlp.fc12.bias", I, params + lo.gate_b, true));
        CS_TRY(vec(p + "mlp.fc2.bias", H, params + lo.down_b, true));
        CS_TRY(vec(p + "norm1.weight", H, params + lo.ao_ln_g, false));
        CS_TRY(vec(p + "norm1.bias", H, params + lo.ao_ln_b, false));
        CS_TRY(vec(p + "norm2.weight", H, params + lo.out_ln_g, false));
        CS_TRY(vec(p + "norm2.bias", H, params + lo.out_ln_b, false));
    }
    return CS_OK;
}

// the weight products of a graph, in node order: MatMul with a 2-D initialiser as its second input, or MatMulInteger with an INT8 /
// UINT8 one that has a scale — what a bias-free Linear leaves behind in an export (the weight transposed and anonymous)
std::vector<WeightProduct> weight_products(const Model& m) {
    std::vector<WeightProduct> prods;
    for (const Node& n : m.nodes) {
        if (n.op == "MatMul" && n.in.size() == 2) {
            const Tensor* w = m.resolve(n.in[1]);
            if (w && w->dims.size() == 2) prods.push_back({&n, w, Quant{}, false});
        } else if (n.op == "MatMulInteger" && n.in.size() >= 2) {
            const Tensor* w = m.tensor(n.in[1]);
            Quant q;
            if (w && w->dims.size() == 2 && (w->dtype == 2 || w->dtype == 3) && quant_of(m, *w, q)) prods.push_back({&n, w, q, true});
        }
    }
    return prods;
}

// does `name` reach an `op` node through at most `depth` element-wise nodes (Cast / Mul / Div / Add)?
bool reaches_op(const Model& m, const std::string& name, const char* op, int depth) {
    auto range = m.consumers.equal_range(name);
    for (auto it = range.first; it != range.second; ++it)
        if (m.nodes[it->second].op == op) return true;
    if (depth <= 0) return false;
    for (auto it = range.first; it != range.second; ++it) {
        const Node& c = m.nodes[it->second];
        if ((c.op == "Cast" || c.op == "Mul" || c.op == "Div" || c.op == "Add") && !c.out.empty() && reaches_op(m, c.out[0], op, depth - 1))
            return true;
    }
    return false;
}

// ---- ModernBERT exports (CS_ARCH_MODERN; the registry's modernbert-embed-large entry, /root/reference/src/embed/embedder.rs:47,
// :72: fastembed caches onnx/model.onnx) ------------------------------------------------------------------------------------
// The modelling code's Linears carry no bias in the published configuration (attention_bias / mlp_bias / norm_bias false), so
// an export holds, per layer and in this order, four weight products with anonymous transposed initialisers — Wqkv [H, 3H],
// attn.Wo [H, H], mlp.Wi [H, 2 I_f] (columns [0, I_f) go through the GELU: our gate; the rest multiply it: our value) and
// mlp.Wo [I_f, H] — while the LayerNorm weights keep their state-dict names (embeddings.norm, layers.N.attn_norm for N > 0,
// layers.N.mlp_norm, final_norm; a .bias is taken where one exists).  I_f is the file's width (2,624); the parameter block's
// (cfg->intermediate, 2,688) is zero-padded exactly as the safetensors reader pads it (checkpoint.cpp).  Pinned on a file
// written by torch.onnx's exporter from transformers' own ModernBertModel (tests/golden/make_modern_onnx_fixture.py).
int32_t modern_params_from_onnx(const Model& m, const cs_bert_config* cfg, const cs_bert_offsets& o, float* params, const char* path) {
    const uint64_t H = cfg->hidden, I = cfg->intermediate;
    std::memset(params, 0, o.total * sizeof(float));
    std::string mod_prefix;
    bool anchored = false;
    {
        static const std::string anchor = "embeddings.tok_embeddings.weight";
        for (const auto& kv : m.init) {
            const std::string& nm = kv.first;
            if (nm.size() >= anchor.size() && nm.compare(nm.size() - anchor.size(), anchor.size(), anchor) == 0) {
                mod_prefix = nm.substr(0, nm.size() - anchor.size());
                anchored = true;
                break;
            }
        }
    }
    if (!anchored)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: tensor embeddings.tok_embeddings.weight is missing from %s", path);
    auto named = [&](const std::string& name) -> const Tensor* { return m.tensor(mod_prefix + name); };
    auto vec = [&](const std::string& name, uint64_t n, float* dst, bool optional) -> int32_t {
        const Tensor* t = named(name);
        if (!t) {
            if (optional) return CS_OK;  // (the block is zeroed)
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: tensor %s is missing from %s", name.c_str(), path);
        }
        if (t->count() != n || t->dims.size() > 2)
            return fail(CS_ERR_DIM_MISMATCH, "Failed to initialize embedding model: %s has %llu elements, config.json implies %llu",
                        name.c_str(), (unsigned long long)t->count(), (unsigned long long)n);
        return copy_matrix(*t, 1, n, false, dst, name.c_str());
    };
    {
        const Tensor* t = named("embeddings.tok_embeddings.weight");
        if (!shape_is(*t, {cfg->vocab_size, H}))
            return fail(CS_ERR_DIM_MISMATCH, "Failed to initialize embedding model: embeddings.tok_embeddings.weight does not have shape "
                        "[%u, %llu]", cfg->vocab_size, (unsigned long long)H);
        CS_TRY(copy_matrix(*t, cfg->vocab_size, H, false, params + o.word, "embeddings.tok_embeddings.weight"));
    }
    CS_TRY(vec("embeddings.norm.weight", H, params + o.emb_ln_g, false));
    CS_TRY(vec("embeddings.norm.bias", H, params + o.emb_ln_b, true));
    CS_TRY(vec("final_norm.weight", H, params + o.final_ln_g, false));
    CS_TRY(vec("final_norm.bias", H, params + o.final_ln_b, true));

    const std::vector<WeightProduct> prods = weight_products(m);
    if (prods.size() != (size_t)4 * cfg->layers)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s holds %zu weight products (MatMul / MatMulInteger with a "
                    "2-D initialiser), a modernbert export of %u layers holds %u", path, prods.size(), cfg->layers, 4 * cfg->layers);
    std::vector<float> packed;
    for (uint32_t l = 0; l < cfg->layers; ++l) {
        cs_bert_layer_offsets lo;
        cs_bert_layer_layout(cfg, &o, l, &lo);
        const WeightProduct* e = &prods[(size_t)4 * l];
        const std::string p = "layers." + std::to_string(l) + ".";
        auto bad = [&](const char* what, const Tensor& w) {
            return fail(CS_ERR_DIM_MISMATCH, "Failed to initialize embedding model: layer %u's %s product does not hold the weight "
                        "config.json implies (%s)", l, what, w.name.c_str());
        };
        auto copy = [&](int i, uint64_t out, uint64_t in, float* dst, const char* what) -> int32_t {
            return copy_matrix(*e[i].w, out, in, true, dst, what, e[i].quantised ? &e[i].q : nullptr);
        };
        if (!shape_is(*e[0].w, {H, 3 * H})) return bad("attn.Wqkv", *e[0].w);
        if (!shape_is(*e[1].w, {H, H})) return bad("attn.Wo", *e[1].w);
        const uint64_t wi = e[2].w->dims.size() == 2 ? e[2].w->dims[1] : 0;
        if (e[2].w->dims[0] != H || wi == 0 || wi % 2 || wi / 2 > I) return bad("mlp.Wi", *e[2].w);
        const uint64_t If = wi / 2;
        if (!shape_is(*e[3].w, {If, H})) return bad("mlp.Wo", *e[3].w);

        if (l) {
            CS_TRY(vec(p + "attn_norm.weight", H, params + lo.ao_ln_g, false));
            CS_TRY(vec(p + "attn_norm.bias", H, params + lo.ao_ln_b, true));
        } else {
            for (uint64_t i = 0; i < H; ++i) params[lo.ao_ln_g + i] = 1.0f;  // (never read: layer 0's attn_norm is the identity)
        }
        CS_TRY(vec(p + "mlp_norm.weight", H, params + lo.out_ln_g, false));
        CS_TRY(vec(p + "mlp_norm.bias", H, params + lo.out_ln_b, true));

        packed.resize((size_t)3 * H * H);
        CS_TRY(copy(0, 3 * H, H, packed.data(), "attn.Wqkv"));
        std::memcpy(params + lo.q_w, packed.data(), H * H * sizeof(float));
        std::memcpy(params + lo.k_w, packed.data() + H * H, H * H * sizeof(float));
        std::memcpy(params + lo.v_w, packed.data() + 2 * H * H, H * H * sizeof(float));
        CS_TRY(copy(1, H, H, params + lo.ao_w, "attn.Wo"));

        // which half of Wi's output meets the activation: the module's order is [through GELU | multiplier]; an export that
        // splits the product is checked (the half whose Split output reaches the Erf of an exact GELU is the gate)
        bool gate_first = true;
        if (!e[2].node->out.empty()) {
            auto range = m.consumers.equal_range(e[2].node->out[0]);
            for (auto it = range.first; it != range.second; ++it) {
                const Node& c = m.nodes[it->second];
                if (c.op == "Split" && c.out.size() == 2) {
                    const bool g0 = reaches_op(m, c.out[0], "Erf", 3), g1 = reaches_op(m, c.out[1], "Erf", 3);
                    if (g0 && g1)
                        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: both halves of layer %u's mlp.Wi reach an "
                                    "Erf in %s (not the gated arrangement)", l, path);
                    if (g1) gate_first = false;
                }
            }
        }
        packed.resize((size_t)2 * If * H);
        CS_TRY(copy(2, 2 * If, H, packed.data(), "mlp.Wi"));
        std::memcpy(params + (gate_first ? lo.gate_w : lo.up_w), packed.data(), If * H * sizeof(float));
        std::memcpy(params + (gate_first ? lo.up_w : lo.gate_w), packed.data() + If * H, If * H * sizeof(float));
        packed.resize((size_t)H * If);
        CS_TRY(copy(3, H, If, packed.data(), "mlp.Wo"));
        for (uint64_t r = 0; r < H; ++r) std::memcpy(params + lo.down_w + r * I, packed.data() + r * If, If * sizeof(float));

        // Linear biases: none in the published files (zero slots); taken by name where an export kept one
        std::vector<float> b3((size_t)3 * H, 0.0f);
        CS_TRY(vec(p + "attn.Wqkv.bias", 3 * H, b3.data(), true));
        std::memcpy(params + lo.q_b, b3.data(), H * sizeof(float));
        std::memcpy(params + lo.k_b, b3.data() + H, H * sizeof(float));
        std::memcpy(params + lo.v_b, b3.data() + 2 * H, H * sizeof(float));
        CS_TRY(vec(p + "attn.Wo.bias", H, params + lo.ao_b, true));
        if (const Tensor* wb = named(p + "mlp.Wi.bias")) {
            if (wb->count() != 2 * If) return bad("mlp.Wi.bias", *wb);
            std::vector<float> b2((size_t)2 * If);
            CS_TRY(copy_matrix(*wb, 1, 2 * If, false, b2.data(), "mlp.Wi.bias"));
            std::memcpy(params + (gate_first ? lo.gate_b : lo.up_b), b2.data(), If * sizeof(float));
            std::memcpy(params + (gate_first ? lo.up_b : lo.gate_b), b2.data() + If, If * sizeof(float));
        }
        CS_TRY(vec(p + "mlp.Wo.bias", H, params + lo.down_b, true));
    }
    return CS_OK;
}

}  // namespace

namespace cs {

// 1 when some initialiser's name contains `needle`, 0 when none does, -1 when the file cannot be read (checkpoint.cpp: which
// JinaBert variant wrote an export)
int onnx_initializer_mentions(const char* path, const char* needle) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return -1;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size <= 0) { close(fd); return -1; }
    void* map = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return -1;
    struct Unmap { void* p; size_t n; ~Unmap() { munmap(p, n); } } unmap{map, (size_t)sb.st_size};
    Model m;
    if (parse_model(Span{(const uint8_t*)map, (const uint8_t*)map + sb.st_size}, m, path) != CS_OK) return -1;
    for (const auto& kv : m.init)
        if (kv.first.find(needle) != std::string::npos) return 1;
    return 0;
}

}  // namespace cs

extern "C" {

int32_t cs_bert_params_from_onnx(const char* path, const cs_bert_config* cfg, float* params, uint64_t n_params) {
    return cs_bert_params_from_onnx_q(path, cfg, params, n_params, nullptr, 0, nullptr);
}

int32_t cs_bert_params_from_onnx_q(const char* path, const cs_bert_config* cfg, float* params, uint64_t n_params,
                                   float* wscale, uint64_t n_wscale, int32_t* quantized) {
    if (quantized) *quantized = 0;
    if (!path || !cfg || !params) return fail(CS_ERR_BAD_ARG, "null argument");
    if (cfg->arch != CS_ARCH_BERT && cfg->arch != CS_ARCH_NOMIC && cfg->arch != CS_ARCH_MODERN && !cs_arch_alibi(cfg->arch))
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: only BERT, NomicBert, JinaBert and ModernBERT exports are read from ONNX files (%s)", path);
    const uint64_t qcols = 5 * (uint64_t)cfg->hidden + cfg->intermediate;
    if (wscale && n_wscale != (uint64_t)cfg->layers * qcols)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: room for %llu column scales, %llu needed",
                    (unsigned long long)n_wscale, (unsigned long long)((uint64_t)cfg->layers * qcols));
    bool all_quantized = true;  // every Linear of every layer an INT8 / UINT8 initialiser behind MatMulInteger
    cs_bert_offsets o;
    cs_bert_layout(cfg, &o);
    if (n_params != o.total)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: expected %llu parameters, got %llu",
                    (unsigned long long)o.total, (unsigned long long)n_params);
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: cannot open %s", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size <= 0) {
        close(fd);
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s is empty", path);
    }
    void* map = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return fail(CS_ERR_OOM, "Failed to initialize embedding model: cannot map %s", path);
    struct Unmap { void* p; size_t n; ~Unmap() { munmap(p, n); } } unmap{map, (size_t)sb.st_size};

    Model m;
    CS_TRY(parse_model(Span{(const uint8_t*)map, (const uint8_t*)map + sb.st_size}, m, path));
    if (cfg->arch == CS_ARCH_NOMIC) return nomic_params_from_onnx(m, cfg, o, params, path);  // (*quantized stays 0: f32 graph)
    if (cfg->arch == CS_ARCH_MODERN) return modern_params_from_onnx(m, cfg, o, params, path);
    const uint64_t H = cfg->hidden, I = cfg->intermediate;

    // a tensor whose state-dict name survived the export, under whatever module prefix the exported model was
    // wrapped in ("", "bert.", "0.auto_model.", ...): the prefix is what precedes the word-embedding table's name
    std::string mod_prefix;
    {
        static const std::string anchors[2] = {"embeddings.word_embeddings.weight", "embeddings.word_embeddings.weight_quantized"};
        for (const std::string& anchor : anchors) {
            bool found = false;
            for (const auto& kv : m.init) {
                const std::string& nm = kv.first;
                if (nm.size() >= anchor.size() && nm.compare(nm.size() - anchor.size(), anchor.size(), anchor) == 0) {
                    mod_prefix = nm.substr(0, nm.size() - anchor.size());
                    found = true;
                    break;
                }
            }
            if (found) break;
        }
    }
    auto named = [&](const std::string& name) -> const Tensor* { return m.tensor(mod_prefix + name); };
    auto vec = [&](const std::string& name, uint64_t n, float* dst) -> int32_t {
        const Tensor* t = named(name);
        if (!t) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: tensor %s is missing from %s", name.c_str(), path);
        if (t->count() != n || t->dims.size() > 2)
            return fail(CS_ERR_DIM_MISMATCH, "Failed to initialize embedding model: %s has %llu elements, config.json implies %llu",
                        name.c_str(), (unsigned long long)t->count(), (unsigned long long)n);
        return copy_matrix(*t, 1, n, false, dst, name.c_str());
    };
    auto table = [&](const std::string& name, uint64_t rows, float* dst) -> int32_t {
        const Tensor* t = named(name);
        if (!t) {  // a Gather table quantised in place: name_quantized + name_scale (+ name_zero_point)
            if (const Tensor* tq = named(name + "_quantized")) {
                Quant q;
                if (!shape_is(*tq, {rows, H}) || !quant_of(m, *tq, q))
                    return fail(CS_ERR_DIM_MISMATCH, "Failed to initialize embedding model: %s_quantized does not have shape [%llu, %llu] "
                                "with a scale", name.c_str(), (unsigned long long)rows, (unsigned long long)H);
                return copy_matrix(*tq, rows, H, false, dst, name.c_str(), &q);
            }
        }
        if (!t) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: tensor %s is missing from %s", name.c_str(), path);
        if (!shape_is(*t, {rows, H}))
            return fail(CS_ERR_DIM_MISMATCH, "Failed to initialize embedding model: %s does not have shape [%llu, %llu]",
                        name.c_str(), (unsigned long long)rows, (unsigned long long)H);
        return copy_matrix(*t, rows, H, false, dst, name.c_str());
    };
    // a Linear layer: weight [out, in] + bias [out]
    // sc_dst (optional): the scale of each of the `out` columns when the weight is quantised per tensor or per output
    // channel — the two forms MatMulInteger's b_zero_point allows; anything else clears all_quantized
    auto linear = [&](const std::string& prefix, uint64_t out, uint64_t in, float* w_dst, float* b_dst, float* sc_dst) -> int32_t {
        const std::string bname = prefix + ".bias", wname = prefix + ".weight";
        CS_TRY(vec(bname, out, b_dst));
        if (const Tensor* w = named(wname)) {  // name kept (Gemm exports, hand-built files)
            all_quantized = false;
            if (shape_is(*w, {out, in})) return copy_matrix(*w, out, in, false, w_dst, wname.c_str());
            return fail(CS_ERR_DIM_MISMATCH, "Failed to initialize embedding model: %s does not have shape [%llu, %llu]",
                        wname.c_str(), (unsigned long long)out, (unsigned long long)in);
        }
        bool tr = false;
        Quant q;
        const Tensor* w = weight_of_bias(m, mod_prefix + bname, out, in, tr, &q);
        if (!w)
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: no MatMul/Gemm/MatMulInteger weight of shape [%llu, %llu] "
                        "feeds the Add of %s in %s", (unsigned long long)in, (unsigned long long)out, bname.c_str(), path);
        bool per_column = false;
        if (q.scale && tr && (w->dtype == 2 || w->dtype == 3)) {
            Reader rs(*q.scale);
            const uint64_t ns = q.scale->count();
            if (rs.ok && (ns == 1 || ns == out)) {
                per_column = true;
                if (sc_dst)
                    for (uint64_t c = 0; c < out; ++c) sc_dst[c] = rs.at(ns == 1 ? 0 : c);
            }
        }
        if (!per_column) all_quantized = false;
        return copy_matrix(*w, out, in, tr, w_dst, wname.c_str(), q.scale ? &q : nullptr);
    };

    CS_TRY(table("embeddings.word_embeddings.weight", cfg->vocab_size, params + o.word));
    const bool jina = cs_arch_alibi(cfg->arch);  // JinaBert: BERT's names for the attention block, no position table (ALiBi)
    if (!jina) CS_TRY(table("embeddings.position_embeddings.weight", cfg->max_position, params + o.pos));
    CS_TRY(table("embeddings.token_type_embeddings.weight", cfg->type_vocab_size, params + o.type));
    CS_TRY(vec("embeddings.LayerNorm.weight", H, params + o.emb_ln_g));
    CS_TRY(vec("embeddings.LayerNorm.bias", H, params + o.emb_ln_b));

    // ORT-optimised files: one fused Attention node per layer, in layer order; their dynamically quantised form is
    // com.microsoft QAttention(input_q, weight_q [H, 3H], bias [3H], input_scale, weight_scale, mask_index,
    // input_zero_point, weight_zero_point): the same packed projection as integers (weight_scale / weight_zero_point
    // per tensor or per output column)
    std::vector<const Node*> fused;
    for (const Node& n : m.nodes)
        if ((n.op == "Attention" || n.op == "QAttention") && n.in.size() >= 3) fused.push_back(&n);
    // JinaBert's gated up projection carries no bias (mlp.gated_layers in jinaai/jina-bert-implementation, mlp.up_gated_layer in
    // jina-bert-v2-qk-post-norm): an anonymous transposed [H, 2I] initialiser behind a MatMul — the only weight products of that
    // shape in the graph, one per layer in layer order
    std::vector<WeightProduct> gated;
    if (jina) {
        all_quantized = false;
        for (const WeightProduct& wp : weight_products(m))
            if (shape_is(*wp.w, {H, 2 * I})) gated.push_back(wp);
        const bool has_qln = named("encoder.layer.0.attention.self.layer_norm_q.weight") != nullptr;
        if (has_qln != (cfg->arch == CS_ARCH_JINA_QKNORM))
            return fail(CS_ERR_DIM_MISMATCH, "Failed to initialize embedding model: %s %s query / key LayerNorm weights, the configuration "
                        "says the opposite", path, has_qln ? "holds" : "holds no");
    }
    for (uint32_t l = 0; l < cfg->layers; ++l) {
        cs_bert_layer_offsets lo;
        cs_bert_layer_layout(cfg, &o, l, &lo);
        const std::string p = "encoder.layer." + std::to_string(l) + ".";
        float* sc = wscale ? wscale + (size_t)l * qcols : nullptr;  // query | key | value | attention.output | intermediate | output
        if (named(p + "attention.self.query.bias")) {
            CS_TRY(linear(p + "attention.self.query", H, H, params + lo.q_w, params + lo.q_b, sc));
            CS_TRY(linear(p + "attention.self.key", H, H, params + lo.k_w, params + lo.k_b, sc ? sc + H : nullptr));
            CS_TRY(linear(p + "attention.self.value", H, H, params + lo.v_w, params + lo.v_b, sc ? sc + 2 * H : nullptr));
        } else if (fused.size() == cfg->layers) {
            const Node& an = *fused[l];
            const bool qatt = an.op == "QAttention";
            const Tensor* w = qatt ? m.tensor(an.in[1]) : m.resolve(an.in[1]);
            const Tensor* b = m.resolve(an.in[2]);
            if (!w || !b || !shape_is(*w, {H, 3 * H}) || b->count() != 3 * H)
                return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: layer %u's fused %s node lacks a "
                            "[%llu, %llu] weight and [%llu] bias", l, an.op.c_str(), (unsigned long long)H, (unsigned long long)(3 * H),
                            (unsigned long long)(3 * H));
            Quant aq;
            if (qatt) {
                aq.scale = an.in.size() > 4 ? m.resolve(an.in[4]) : nullptr;
                aq.zp = an.in.size() > 7 && !an.in[7].empty() ? m.tensor(an.in[7]) : nullptr;
            }
            Reader rw = qatt ? Reader(*w, aq, H, 3 * H) : Reader(*w);
            Reader rb(*b);
            if (!rw.ok || !rb.ok || w->external || b->external)
                return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: layer %u's fused QKV weight has an "
                            "unsupported data type or (QAttention) no usable weight_scale / weight_zero_point", l);
            bool per_column = false;
            if (qatt) {
                Reader rs(*aq.scale);
                const uint64_t ns = aq.scale->count();
                if (rs.ok && (ns == 1 || ns == 3 * H)) {
                    per_column = true;
                    if (sc)
                        for (uint64_t c = 0; c < 3 * H; ++c) sc[c] = rs.at(ns == 1 ? 0 : c);  // query | key | value columns
                }
            }
            if (!per_column) all_quantized = false;
            float* wd[3] = {params + lo.q_w, params + lo.k_w, params + lo.v_w};
            float* bd[3] = {params + lo.q_b, params + lo.k_b, params + lo.v_b};
            for (int part = 0; part < 3; ++part) {
                for (uint64_t out = 0; out < H; ++out) {
                    bd[part][out] = rb.at(part * H + out);
                    for (uint64_t in = 0; in < H; ++in) wd[part][out * H + in] = rw.at(in * 3 * H + part * H + out);
                }
            }
        } else {
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s has neither %sattention.self.query.bias nor "
                        "one fused Attention node per layer", path, p.c_str());
        }
        CS_TRY(linear(p + "attention.output.dense", H, H, params + lo.ao_w, params + lo.ao_b, sc ? sc + 3 * H : nullptr));
        CS_TRY(vec(p + "attention.output.LayerNorm.weight", H, params + lo.ao_ln_g));
        CS_TRY(vec(p + "attention.output.LayerNorm.bias", H, params + lo.ao_ln_b));
        if (jina) {
            if (cfg->arch == CS_ARCH_JINA_QKNORM) {
                CS_TRY(vec(p + "attention.self.layer_norm_q.weight", H, params + lo.qln_g));
                CS_TRY(vec(p + "attention.self.layer_norm_q.bias", H, params + lo.qln_b));
                CS_TRY(vec(p + "attention.self.layer_norm_k.weight", H, params + lo.kln_g));
                CS_TRY(vec(p + "attention.self.layer_norm_k.bias", H, params + lo.kln_b));
            }
            // which modelling file wrote the graph: mlp.gated_layers + mlp.wo ([through GELU | multiplier]) or mlp.up_gated_layer +
            // mlp.down_layer ([multiplier | through GELU]) — told apart by the down projection's bias name
            const bool first_file = named(p + "mlp.wo.bias") != nullptr;
            const std::string up = p + (first_file ? "mlp.gated_layers" : "mlp.up_gated_layer"), down = p + (first_file ? "mlp.wo" : "mlp.down_layer");
            std::vector<float> packed((size_t)2 * I * H), b2((size_t)2 * I, 0.0f);
            if (named(up + ".bias")) {  // an export of a variant that kept the bias: the weight is the one behind its Add
                CS_TRY(linear(up, 2 * I, H, packed.data(), b2.data(), nullptr));
            } else {
                if (gated.size() != cfg->layers)
                    return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s holds %zu bias-free [%llu, %llu] weight products, "
                                "a JinaBert export of %u layers holds %u (the gated up projections)", path, gated.size(),
                                (unsigned long long)H, (unsigned long long)(2 * I), cfg->layers, cfg->layers);
                CS_TRY(copy_matrix(*gated[l].w, 2 * I, H, true, packed.data(), up.c_str(), gated[l].quantised ? &gated[l].q : nullptr));
            }
            const size_t gate0 = first_file ? 0 : I, up0 = first_file ? I : 0;
            std::memcpy(params + lo.gate_w, packed.data() + gate0 * H, I * H * sizeof(float));
            std::memcpy(params + lo.up_w, packed.data() + up0 * H, I * H * sizeof(float));
            std::memcpy(params + lo.gate_b, b2.data() + gate0, I * sizeof(float));
            std::memcpy(params + lo.up_b, b2.data() + up0, I * sizeof(float));
            CS_TRY(linear(down, H, I, params + lo.down_w, params + lo.down_b, nullptr));
            CS_TRY(vec(p + "mlp.layernorm.weight", H, params + lo.out_ln_g));
            CS_TRY(vec(p + "mlp.layernorm.bias", H, params + lo.out_ln_b));
            all_quantized = false;
            continue;
        }
        CS_TRY(linear(p + "intermediate.dense", I, H, params + lo.up_w, params + lo.up_b, sc ? sc + 4 * H : nullptr));
        CS_TRY(linear(p + "output.dense", H, I, params + lo.down_w, params + lo.down_b, sc ? sc + 4 * H + I : nullptr));
        CS_TRY(vec(p + "output.LayerNorm.weight", H, params + lo.out_ln_g));
        CS_TRY(vec(p + "output.LayerNorm.bias", H, params + lo.out_ln_b));
    }
    if (quantized) *quantized = all_quantized ? 1 : 0;
    return CS_OK;
}

}  // extern "C"
