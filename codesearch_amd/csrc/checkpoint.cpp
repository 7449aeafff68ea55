// checkpoint.cpp — real-weight loading behind the C ABI: a HF model directory
// (config.json + model.safetensors, what `hf-hub` leaves in fastembed's cache dir for
// BAAI/bge-small-en-v1.5 and its BERT siblings) -> cs_bert_config + the flat f32 parameter block of
// include/cs_bert_params.h.  Stands in for the model-loading half of FastEmbedder::with_cache_dir
// (/root/reference/src/embed/embedder.rs:218-245).  fastembed's own cache holds the ONNX export of the same
// tensors and a tokenizer.json instead: cs_embedder_create_from_dir takes either (ONNX: onnx_reader.cpp), and
// cs_tokenizer_create_from_json / _from_dir below read tokenizer.json.  Host-only C++; no GPU needed for the
// loaders.
//
// safetensors file = u64 LE header length N | N bytes of JSON {"name": {"dtype": "F32"|"F16"|"BF16",
// "shape": [...], "data_offsets": [begin, end]}, ..., "__metadata__": {...}} | tensor bytes.
// Tensor names are HF BertModel state-dict names, optionally prefixed "bert."; pooler, position_ids
// and any other extra tensors are ignored.

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/cs_bert_params.h"
#include "common.hpp"
#include "bpe.hpp"
#include "unigram.hpp"

namespace {

using cs::fail;

// ---- a small JSON reader: just enough for config.json and a safetensors header --------------------
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0.0;
    bool b = false;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json* get(const char* key) const {
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct JsonParser {
    const char* p;
    const char* end;
    bool ok = true;

    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    bool lit(const char* s) {
        const size_t n = std::strlen(s);
        if ((size_t)(end - p) >= n && std::memcmp(p, s, n) == 0) { p += n; return true; }
        return false;
    }
    std::string string() {
        std::string out;
        if (p >= end || *p != '"') { ok = false; return out; }
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                    case 'n': out.push_back('\n'); break;
                    case 't': out.push_back('\t'); break;
                    case 'r': out.push_back('\r'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'u': {  // \uXXXX (and surrogate pairs) -> UTF-8
                        if (end - p < 5) { ok = false; return out; }
                        unsigned cp = (unsigned)std::strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                        p += 4;
                        if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 7 && p[1] == '\\' && p[2] == 'u') {
                            const unsigned lo = (unsigned)std::strtoul(std::string(p + 3, p + 7).c_str(), nullptr, 16);
                            if (lo >= 0xDC00 && lo < 0xE000) {
                                cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                                p += 6;
                            }
                        }
                        if (cp < 0x80) out.push_back((char)cp);
                        else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                        else if (cp < 0x10000) { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                        else { out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                        break;
                    }
                    default: out.push_back(*p);
                }
                ++p;
            } else {
                out.push_back(*p++);
            }
        }
        if (p >= end) { ok = false; return out; }
        ++p;
        return out;
    }
    Json value(int depth = 0) {
        Json j;
        ws();
        if (p >= end || depth > 64) { ok = false; return j; }
        if (*p == '{') {
            j.kind = Json::Obj;
            ++p;
            ws();
            if (p < end && *p == '}') { ++p; return j; }
            while (ok) {
                ws();
                std::string k = string();
                ws();
                if (!ok || p >= end || *p != ':') { ok = false; break; }
                ++p;
                j.obj.emplace_back(std::move(k), value(depth + 1));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                ok = false;
            }
        } else if (*p == '[') {
            j.kind = Json::Arr;
            ++p;
            ws();
            if (p < end && *p == ']') { ++p; return j; }
            while (ok) {
                j.arr.push_back(value(depth + 1));
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                ok = false;
            }
        } else if (*p == '"') {
            j.kind = Json::Str;
            j.str = string();
        } else if (lit("true")) { j.kind = Json::Bool; j.b = true; }
        else if (lit("false")) { j.kind = Json::Bool; }
        else if (lit("null")) { j.kind = Json::Null; }
        else {
            char* e = nullptr;
            const std::string tmp(p, (size_t)std::min<ptrdiff_t>(end - p, 64));
            j.num = std::strtod(tmp.c_str(), &e);
            if (e == tmp.c_str()) { ok = false; return j; }
            j.kind = Json::Num;
            p += e - tmp.c_str();
        }
        return j;
    }
};

bool read_file(const std::string& path, std::string& out, uint64_t max_bytes = ~0ull) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char tmp[1 << 16];
    size_t got;
    while (out.size() < max_bytes && (got = std::fread(tmp, 1, sizeof(tmp), f)) > 0) out.append(tmp, got);
    std::fclose(f);
    return true;
}

bool file_exists(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fclose(f);
    return true;
}

float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {  // subnormal
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

struct TensorRef { std::string dtype; std::vector<uint64_t> shape; uint64_t begin = 0, end = 0; };

// name -> (shape, offset in the flat block) for every tensor of the layout, in layout order
// src_rows != 0: the file's tensor is [src_rows, shape[1]] and rows [row0, row0 + shape[0]) of it are wanted (a fused
// projection); optional: a tensor the file may lack (its slot stays zero)
struct Want { std::string name; std::vector<uint64_t> shape; uint64_t off; uint64_t src_rows = 0, row0 = 0; bool optional = false; };

// NomicBert checkpoints (nomic-ai/nomic-embed-text-v1 / v1.5, the model repository's modeling file): emb_ln,
// encoder.layers.N.{attn.Wqkv, attn.out_proj, norm1, mlp.fc11, mlp.fc12, mlp.fc2, norm2}; no Linear biases in the
// published files (taken when present)
std::vector<Want> nomic_layout_table(const cs_bert_config& c) {
    cs_bert_offsets o;
    cs_bert_layout(&c, &o);
    const uint64_t H = c.hidden, I = c.intermediate;
    std::vector<Want> t = {
        {"embeddings.word_embeddings.weight", {c.vocab_size, H}, o.word},
        {"embeddings.token_type_embeddings.weight", {c.type_vocab_size, H}, o.type},
        {"emb_ln.weight", {H}, o.emb_ln_g},
        {"emb_ln.bias", {H}, o.emb_ln_b},
    };
    for (uint32_t l = 0; l < c.layers; ++l) {
        cs_bert_layer_offsets lo;
        cs_bert_layer_layout(&c, &o, l, &lo);
        const std::string p = "encoder.layers." + std::to_string(l) + ".";
        const uint64_t qw[3] = {lo.q_w, lo.k_w, lo.v_w}, qb[3] = {lo.q_b, lo.k_b, lo.v_b};
        for (uint64_t i = 0; i < 3; ++i) {
            t.push_back({p + "attn.Wqkv.weight", {H, H}, qw[i], 3 * H, i * H, false});
            t.push_back({p + "attn.Wqkv.bias", {H}, qb[i], 3 * H, i * H, true});
        }
        t.push_back({p + "attn.out_proj.weight", {H, H}, lo.ao_w});
        t.push_back({p + "attn.out_proj.bias", {H}, lo.ao_b, 0, 0, true});
        t.push_back({p + "norm1.weight", {H}, lo.ao_ln_g});
        t.push_back({p + "norm1.bias", {H}, lo.ao_ln_b});
        t.push_back({p + "mlp.fc11.weight", {I, H}, lo.up_w});
        t.push_back({p + "mlp.fc11.bias", {I}, lo.up_b, 0, 0, true});
        t.push_back({p + "mlp.fc12.weight", {I, H}, lo.gate_w});
        t.push_back({p + "mlp.fc12.bias", {I}, lo.gate_b, 0, 0, true});
        t.push_back({p + "mlp.fc2.weight", {H, I}, lo.down_w});
        t.push_back({p + "mlp.fc2.bias", {H}, lo.down_b, 0, 0, true});
        t.push_back({p + "norm2.weight", {H}, lo.out_ln_g});
        t.push_back({p + "norm2.bias", {H}, lo.out_ln_b});
    }
    return t;
}

// JinaBert checkpoints (jinaai/jina-embeddings-v2-base-code; the two modelling files of the family): BERT names for the
// embeddings and the attention block (+ attention.self.layer_norm_q / layer_norm_k with CS_ARCH_JINA_QKNORM), the
// feed-forward as mlp.up_gated_layer [2I, H] (rows [0, I) the value, rows [I, 2I) through GELU), mlp.down_layer,
// mlp.layernorm — or, `first_file`, mlp.gated_layers (rows [0, I) through GELU, rows [I, 2I) the value), mlp.wo.  The up
// projection has no bias (zero slots).
std::vector<Want> jina_layout_table(const cs_bert_config& c, bool first_file) {
    cs_bert_offsets o;
    cs_bert_layout(&c, &o);
    const uint64_t H = c.hidden, I = c.intermediate;
    std::vector<Want> t = {
        {"embeddings.word_embeddings.weight", {c.vocab_size, H}, o.word},
        {"embeddings.token_type_embeddings.weight", {c.type_vocab_size, H}, o.type},
        {"embeddings.LayerNorm.weight", {H}, o.emb_ln_g},
        {"embeddings.LayerNorm.bias", {H}, o.emb_ln_b},
    };
    for (uint32_t l = 0; l < c.layers; ++l) {
        cs_bert_layer_offsets lo;
        cs_bert_layer_layout(&c, &o, l, &lo);
        const std::string p = "encoder.layer." + std::to_string(l) + ".";
        t.push_back({p + "attention.self.query.weight", {H, H}, lo.q_w});
        t.push_back({p + "attention.self.query.bias", {H}, lo.q_b});
        t.push_back({p + "attention.self.key.weight", {H, H}, lo.k_w});
        t.push_back({p + "attention.self.key.bias", {H}, lo.k_b});
        t.push_back({p + "attention.self.value.weight", {H, H}, lo.v_w});
        t.push_back({p + "attention.self.value.bias", {H}, lo.v_b});
        if (c.arch == CS_ARCH_JINA_QKNORM) {
            t.push_back({p + "attention.self.layer_norm_q.weight", {H}, lo.qln_g});
            t.push_back({p + "attention.self.layer_norm_q.bias", {H}, lo.qln_b});
            t.push_back({p + "attention.self.layer_norm_k.weight", {H}, lo.kln_g});
            t.push_back({p + "attention.self.layer_norm_k.bias", {H}, lo.kln_b});
        }
        t.push_back({p + "attention.output.dense.weight", {H, H}, lo.ao_w});
        t.push_back({p + "attention.output.dense.bias", {H}, lo.ao_b});
        t.push_back({p + "attention.output.LayerNorm.weight", {H}, lo.ao_ln_g});
        t.push_back({p + "attention.output.LayerNorm.bias", {H}, lo.ao_ln_b});
        const std::string up = p + (first_file ? "mlp.gated_layers" : "mlp.up_gated_layer");
        const std::string down = p + (first_file ? "mlp.wo" : "mlp.down_layer");
        t.push_back({up + ".weight", {I, H}, lo.up_w, 2 * I, first_file ? I : 0, false});
        t.push_back({up + ".weight", {I, H}, lo.gate_w, 2 * I, first_file ? 0 : I, false});
        t.push_back({up + ".bias", {I}, lo.up_b, 2 * I, first_file ? I : 0, true});
        t.push_back({up + ".bias", {I}, lo.gate_b, 2 * I, first_file ? 0 : I, true});
        t.push_back({down + ".weight", {H, I}, lo.down_w});
        t.push_back({down + ".bias", {H}, lo.down_b});
        t.push_back({p + "mlp.layernorm.weight", {H}, lo.out_ln_g});
        t.push_back({p + "mlp.layernorm.bias", {H}, lo.out_ln_b});
    }
    return t;
}

// ModernBERT checkpoints (lightonai/modernbert-embed-large; HF ModernBertModel's names, optional "model." prefix):
// embeddings.tok_embeddings / norm, layers.N.{attn_norm (N >= 1), attn.Wqkv, attn.Wo, mlp_norm, mlp.Wi [2 I_f, H], mlp.Wo
// [H, I_f]}, final_norm; biases optional everywhere (the published files have none: zero slots).  The file's intermediate
// size I_f may be smaller than the config's (2,624 against the 2,688 = 21 x 128 the kernels tile): the extra rows of Wi and
// columns of Wo are zero — the same function.
int32_t modern_params_from_safetensors(FILE* f, const std::map<std::string, TensorRef>& have, uint64_t data0, const cs_bert_config& c,
                                       float* params, const char* path) {
    cs_bert_offsets o;
    cs_bert_layout(&c, &o);
    const uint64_t H = c.hidden, I = c.intermediate;
    std::memset(params, 0, o.total * sizeof(float));
    std::vector<unsigned char> raw;
    // the tensor `name` as f32 (empty + CS_OK when it is absent and optional)
    auto fetch = [&](const std::string& name, bool optional, std::vector<float>& out, std::vector<uint64_t>& shape) -> int32_t {
        out.clear();
        auto it = have.find(name);
        if (it == have.end()) it = have.find("model." + name);
        if (it == have.end()) {
            if (optional) return CS_OK;
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: tensor %s is missing from %s", name.c_str(), path);
        }
        const TensorRef& t = it->second;
        uint64_t count = 1;
        for (uint64_t d : t.shape) {
            if (d != 0 && count > (1ull << 40) / d) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s has an implausible shape", name.c_str());
            count *= d;
        }
        const uint32_t esz = t.dtype == "F32" ? 4 : (t.dtype == "F16" || t.dtype == "BF16") ? 2 : 0;
        if (!esz) return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: %s has dtype %s (F32, F16, BF16 only)", name.c_str(), t.dtype.c_str());
        if (t.end < t.begin || t.end - t.begin != count * esz || count > o.total)
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s has inconsistent data_offsets", name.c_str());
        if (fseeko(f, (off_t)(data0 + t.begin), SEEK_SET) != 0) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s is truncated", path);
        out.resize(count);
        if (esz == 4) {
            if (std::fread(out.data(), 4, count, f) != count) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s is truncated", path);
        } else {
            raw.resize(count * 2);
            if (std::fread(raw.data(), 2, count, f) != count) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s is truncated", path);
            const bool bf = t.dtype == "BF16";
            for (uint64_t i = 0; i < count; ++i) {
                const uint16_t v = (uint16_t)(raw[2 * i] | (raw[2 * i + 1] << 8));
                if (bf) { const uint32_t bits = (uint32_t)v << 16; std::memcpy(&out[i], &bits, 4); }
                else out[i] = half_to_float(v);
            }
        }
        shape = t.shape;
        return CS_OK;
    };
    auto bad_shape = [&](const std::string& name) {
        return fail(CS_ERR_DIM_MISMATCH, "Failed to initialize embedding model: %s does not have the shape config.json implies", name.c_str());
    };
    std::vector<float> v;
    std::vector<uint64_t> sh;
    auto vec = [&](const std::string& name, uint64_t n, uint64_t off, bool optional) -> int32_t {
        CS_TRY(fetch(name, optional, v, sh));
        if (v.empty()) return CS_OK;
        if (v.size() != n) return bad_shape(name);
        std::memcpy(params + off, v.data(), n * sizeof(float));
        return CS_OK;
    };
    CS_TRY(fetch("embeddings.tok_embeddings.weight", false, v, sh));
    if (sh != std::vector<uint64_t>{c.vocab_size, H}) return bad_shape("embeddings.tok_embeddings.weight");
    std::memcpy(params + o.word, v.data(), v.size() * sizeof(float));
    CS_TRY(vec("embeddings.norm.weight", H, o.emb_ln_g, false));
    CS_TRY(vec("embeddings.norm.bias", H, o.emb_ln_b, true));
    CS_TRY(vec("final_norm.weight", H, o.final_ln_g, false));
    CS_TRY(vec("final_norm.bias", H, o.final_ln_b, true));
    for (uint32_t l = 0; l < c.layers; ++l) {
        cs_bert_layer_offsets lo;
        cs_bert_layer_layout(&c, &o, l, &lo);
        const std::string p = "layers." + std::to_string(l) + ".";
        if (l) {
            CS_TRY(vec(p + "attn_norm.weight", H, lo.ao_ln_g, false));
            CS_TRY(vec(p + "attn_norm.bias", H, lo.ao_ln_b, true));
        } else {
            for (uint64_t i = 0; i < H; ++i) params[lo.ao_ln_g + i] = 1.0f;  // (never read: layer 0's attn_norm is the identity)
        }
        CS_TRY(fetch(p + "attn.Wqkv.weight", false, v, sh));
        if (sh != std::vector<uint64_t>{3 * H, H}) return bad_shape(p + "attn.Wqkv.weight");
        std::memcpy(params + lo.q_w, v.data(), H * H * sizeof(float));
        std::memcpy(params + lo.k_w, v.data() + H * H, H * H * sizeof(float));
        std::memcpy(params + lo.v_w, v.data() + 2 * H * H, H * H * sizeof(float));
        CS_TRY(fetch(p + "attn.Wqkv.bias", true, v, sh));
        if (!v.empty()) {
            if (v.size() != 3 * H) return bad_shape(p + "attn.Wqkv.bias");
            std::memcpy(params + lo.q_b, v.data(), H * sizeof(float));
            std::memcpy(params + lo.k_b, v.data() + H, H * sizeof(float));
            std::memcpy(params + lo.v_b, v.data() + 2 * H, H * sizeof(float));
        }
        CS_TRY(fetch(p + "attn.Wo.weight", false, v, sh));
        if (sh != std::vector<uint64_t>{H, H}) return bad_shape(p + "attn.Wo.weight");
        std::memcpy(params + lo.ao_w, v.data(), H * H * sizeof(float));
        CS_TRY(vec(p + "attn.Wo.bias", H, lo.ao_b, true));
        CS_TRY(vec(p + "mlp_norm.weight", H, lo.out_ln_g, false));
        CS_TRY(vec(p + "mlp_norm.bias", H, lo.out_ln_b, true));
        CS_TRY(fetch(p + "mlp.Wi.weight", false, v, sh));
        if (sh.size() != 2 || sh[1] != H || sh[0] % 2 || sh[0] / 2 > I || sh[0] == 0) return bad_shape(p + "mlp.Wi.weight");
        const uint64_t If = sh[0] / 2;  // rows [0, I_f): through the activation (our gate); rows [I_f, 2 I_f): our value
        std::memcpy(params + lo.gate_w, v.data(), If * H * sizeof(float));
        std::memcpy(params + lo.up_w, v.data() + If * H, If * H * sizeof(float));
        CS_TRY(fetch(p + "mlp.Wi.bias", true, v, sh));
        if (!v.empty()) {
            if (v.size() != 2 * If) return bad_shape(p + "mlp.Wi.bias");
            std::memcpy(params + lo.gate_b, v.data(), If * sizeof(float));
            std::memcpy(params + lo.up_b, v.data() + If, If * sizeof(float));
        }
        CS_TRY(fetch(p + "mlp.Wo.weight", false, v, sh));
        if (sh != std::vector<uint64_t>{H, If}) return bad_shape(p + "mlp.Wo.weight");
        for (uint64_t r = 0; r < H; ++r) std::memcpy(params + lo.down_w + r * I, v.data() + r * If, If * sizeof(float));
        CS_TRY(vec(p + "mlp.Wo.bias", H, lo.down_b, true));
    }
    return CS_OK;
}

std::vector<Want> layout_table(const cs_bert_config& c, bool jina_first_file = false) {
    if (c.arch == CS_ARCH_NOMIC) return nomic_layout_table(c);
    if (cs_arch_alibi(c.arch)) return jina_layout_table(c, jina_first_file);
    cs_bert_offsets o;
    cs_bert_layout(&c, &o);
    const uint64_t H = c.hidden, I = c.intermediate;
    std::vector<Want> t = {
        {"embeddings.word_embeddings.weight", {c.vocab_size, H}, o.word},
        {"embeddings.position_embeddings.weight", {c.max_position, H}, o.pos},
        {"embeddings.token_type_embeddings.weight", {c.type_vocab_size, H}, o.type},
        {"embeddings.LayerNorm.weight", {H}, o.emb_ln_g},
        {"embeddings.LayerNorm.bias", {H}, o.emb_ln_b},
    };
    for (uint32_t l = 0; l < c.layers; ++l) {
        cs_bert_layer_offsets lo;
        cs_bert_layer_layout(&c, &o, l, &lo);
        const std::string p = "encoder.layer." + std::to_string(l) + ".";
        t.push_back({p + "attention.self.query.weight", {H, H}, lo.q_w});
        t.push_back({p + "attention.self.query.bias", {H}, lo.q_b});
        t.push_back({p + "attention.self.key.weight", {H, H}, lo.k_w});
        t.push_back({p + "attention.self.key.bias", {H}, lo.k_b});
        t.push_back({p + "attention.self.value.weight", {H, H}, lo.v_w});
        t.push_back({p + "attention.self.value.bias", {H}, lo.v_b});
        t.push_back({p + "attention.output.dense.weight", {H, H}, lo.ao_w});
        t.push_back({p + "attention.output.dense.bias", {H}, lo.ao_b});
        t.push_back({p + "attention.output.LayerNorm.weight", {H}, lo.ao_ln_g});
        t.push_back({p + "attention.output.LayerNorm.bias", {H}, lo.ao_ln_b});
        t.push_back({p + "intermediate.dense.weight", {I, H}, lo.up_w});
        t.push_back({p + "intermediate.dense.bias", {I}, lo.up_b});
        t.push_back({p + "output.dense.weight", {H, I}, lo.down_w});
        t.push_back({p + "output.dense.bias", {H}, lo.down_b});
        t.push_back({p + "output.LayerNorm.weight", {H}, lo.out_ln_g});
        t.push_back({p + "output.LayerNorm.bias", {H}, lo.out_ln_b});
    }
    return t;
}

bool json_u32(const Json& root, const char* key, uint32_t& out) {
    const Json* j = root.get(key);
    if (!j || j->kind != Json::Num || !(j->num >= 0 && j->num <= 4294967295.0)) return false;
    out = (uint32_t)j->num;
    return true;
}

}  // namespace

extern "C" {

static int safetensors_header_mentions(const std::string& path, const char* needle);

// the weight files cs_embedder_create_from_dir reads, in its order of preference ("" when the directory holds none)
static std::string model_file_in(const std::string& dir, bool& safetensors) {
    safetensors = file_exists(dir + "/model.safetensors");
    if (safetensors) return dir + "/model.safetensors";
    for (const char* rel : {"/onnx/model.onnx", "/model.onnx", "/model_optimized.onnx", "/onnx/model_optimized.onnx",
                            "/onnx/model_quantized.onnx", "/model_quantized.onnx"})
        if (file_exists(dir + rel)) return dir + rel;
    return "";
}

// JinaBert: 1 when the directory's weights hold query / key LayerNorm tensors, 0 when they do not, -1 when there is no readable file
static int jina_variant_from_files(const char* model_dir) {
    bool st = false;
    const std::string f = model_file_in(model_dir, st);
    if (f.empty()) return -1;
    return st ? safetensors_header_mentions(f, "attention.self.layer_norm_q.") : cs::onnx_initializer_mentions(f.c_str(), "attention.self.layer_norm_q.");
}

int32_t cs_bert_config_from_dir(const char* model_dir, int32_t pooling, cs_bert_config* cfg) {
    if (!model_dir || !cfg) return fail(CS_ERR_BAD_ARG, "null argument");
    const bool pooling_given = pooling != -1;
    bool pooling_file = false;
    if (pooling == -1) {  // auto: the sentence-transformers pooling module of the snapshot, CLS when there is none
        pooling = CS_POOL_CLS;
        std::string ptext;
        if (read_file(std::string(model_dir) + "/1_Pooling/config.json", ptext, 1 << 20)) {
            pooling_file = true;
            JsonParser pj{ptext.data(), ptext.data() + ptext.size()};
            const Json proot = pj.value();
            if (pj.ok && proot.kind == Json::Obj) {
                const Json* mean = proot.get("pooling_mode_mean_tokens");
                const Json* cls = proot.get("pooling_mode_cls_token");
                if (mean && mean->kind == Json::Bool && mean->b && !(cls && cls->kind == Json::Bool && cls->b))
                    pooling = CS_POOL_MEAN;
            }
        }
    }
    if (pooling != CS_POOL_CLS && pooling != CS_POOL_MEAN) return fail(CS_ERR_BAD_ARG, "unknown pooling %d", pooling);
    const std::string path = std::string(model_dir) + "/config.json";
    std::string text;
    if (!read_file(path, text, 1 << 24))
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: cannot open %s", path.c_str());
    JsonParser jp{text.data(), text.data() + text.size()};
    const Json root = jp.value();
    if (!jp.ok || root.kind != Json::Obj)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s is not a JSON object", path.c_str());
    if (const Json* mt = root.get("model_type"))
        if (mt->kind == Json::Str && mt->str == "nomic_bert") {
            // NomicBert (the registry's nomic-embed-text entries): GPT-2 style keys.  Only the published configuration is
            // built — full rotary fraction, non-interleaved, no scale base / scaling factor, swiglu, post-norm.
            auto is = [&](const char* key, auto pred) { const Json* j = root.get(key); return !j || pred(*j); };
            if (!is("rotary_emb_fraction", [](const Json& j) { return j.kind == Json::Num && j.num == 1.0; }) ||
                !is("rotary_emb_interleaved", [](const Json& j) { return j.kind == Json::Bool && !j.b; }) ||
                !is("rotary_emb_scale_base", [](const Json& j) { return j.kind == Json::Null; }) ||
                !is("rotary_scaling_factor", [](const Json& j) { return j.kind == Json::Null || (j.kind == Json::Num && j.num >= 1.0); }) ||
                !is("activation_function", [](const Json& j) { return j.kind == Json::Str && j.str == "swiglu"; }) ||
                !is("prenorm", [](const Json& j) { return j.kind == Json::Bool && !j.b; }))
                return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: this nomic_bert configuration is not built "
                            "(only full non-interleaved rotary positions without scaling, swiglu, post-norm)");
            cs_bert_config c{};
            if (!json_u32(root, "vocab_size", c.vocab_size) || !json_u32(root, "n_embd", c.hidden) ||
                !json_u32(root, "n_layer", c.layers) || !json_u32(root, "n_head", c.heads))
                return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s lacks a nomic_bert size field", path.c_str());
            if (!json_u32(root, "n_inner", c.intermediate)) c.intermediate = 4 * c.hidden;
            // no position table to size: the bound is what fastembed's default InitOptions truncate to (embedder.rs:238)
            uint32_t npos = 0;
            c.max_position = (json_u32(root, "n_positions", npos) && npos && npos < 512) ? npos : 512;
            // rotary_scaling_factor (dynamic NTK): the factor changes the rotary base only for sequences LONGER than
            // max_trained_positions (2,048 by default); this loader never runs more than 512 positions, so the table is the
            // unscaled one whatever the factor says (the 8k-context checkpoints carry a factor of 2) — refused
            // only where the scaling would actually apply
            {
                uint32_t trained = 0;
                if (!json_u32(root, "max_trained_positions", trained) || trained == 0) trained = 2048;
                const Json* rf = root.get("rotary_scaling_factor");
                if (rf && rf->kind == Json::Num && rf->num != 1.0 && c.max_position > trained)
                    return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: this nomic_bert configuration is not built "
                                "(rotary scaling factor %g applies from %u positions on, %u are run)", rf->num, trained, c.max_position);
            }
            if (!json_u32(root, "type_vocab_size", c.type_vocab_size)) c.type_vocab_size = 2;
            const Json* eps = root.get("layer_norm_epsilon");
            c.layer_norm_eps = (eps && eps->kind == Json::Num) ? (float)eps->num : 1e-12f;
            const Json* rb = root.get("rotary_emb_base");
            c.rotary_base = (rb && rb->kind == Json::Num) ? (float)rb->num : 10000.0f;
            c.arch = CS_ARCH_NOMIC;
            c.pooling = pooling_given ? pooling : (pooling_file ? pooling : CS_POOL_MEAN);  // fastembed pools the family by mean
            *cfg = c;
            return CS_OK;
        }
    if (const Json* mt = root.get("model_type"))
        if (mt->kind == Json::Str && mt->str == "modernbert") {
            // ModernBERT (the registry's modernbert-embed-large): only the published arrangement — erf-GELU gate, default rotary
            // positions (no scaling), full / sliding layers by global_attn_every_n_layers
            if (const Json* act = root.get("hidden_activation"))
                if (act->kind == Json::Str && act->str != "gelu")
                    return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: this ModernBERT configuration is not built "
                                "(hidden_activation \"%s\"; only \"gelu\")", act->str.c_str());
            cs_bert_config c{};
            if (!json_u32(root, "vocab_size", c.vocab_size) || !json_u32(root, "hidden_size", c.hidden) ||
                !json_u32(root, "num_hidden_layers", c.layers) || !json_u32(root, "num_attention_heads", c.heads) ||
                !json_u32(root, "intermediate_size", c.intermediate))
                return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s lacks a ModernBERT size field", path.c_str());
            c.intermediate = (c.intermediate + 127) / 128 * 128;  // the kernels' tile: zero rows / columns fill the difference
            uint32_t npos = 0;
            c.max_position = (json_u32(root, "max_position_embeddings", npos) && npos && npos < 512) ? npos : 512;  // (fastembed truncates to 512)
            c.type_vocab_size = 1;  // (no token-type table in this family; the field only has to be non-zero)
            const Json* eps = root.get("norm_eps");
            c.layer_norm_eps = (eps && eps->kind == Json::Num) ? (float)eps->num : 1e-5f;
            if (!json_u32(root, "global_attn_every_n_layers", c.global_every) || c.global_every == 0) c.global_every = 3;
            uint32_t local = 0;
            if (!json_u32(root, "local_attention", local) || local < 2) local = 128;
            c.local_window = local / 2;
            auto num = [&](const Json* j, float dflt) { return (j && j->kind == Json::Num && j->num > 1.0) ? (float)j->num : dflt; };
            c.rotary_base = num(root.get("global_rope_theta"), 160000.0f);
            c.rotary_base_local = num(root.get("local_rope_theta"), 10000.0f);
            if (const Json* rp = root.get("rope_parameters"))  // the newer serialisation: {full_attention: {rope_theta, rope_type}, sliding_attention: {...}}
                if (rp->kind == Json::Obj)
                    for (const char* which : {"full_attention", "sliding_attention"}) {
                        const Json* e = rp->get(which);
                        if (!e || e->kind != Json::Obj) continue;
                        if (const Json* ty = e->get("rope_type"))
                            if (ty->kind == Json::Str && ty->str != "default")
                                return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: this ModernBERT configuration is not built "
                                            "(rope_type \"%s\")", ty->str.c_str());
                        float& dst = which[0] == 'f' ? c.rotary_base : c.rotary_base_local;
                        dst = num(e->get("rope_theta"), dst);
                    }
            c.arch = CS_ARCH_MODERN;
            c.pooling = pooling_given ? pooling : (pooling_file ? pooling : CS_POOL_MEAN);  // fastembed pools the model by mean
            *cfg = c;
            return CS_OK;
        }
    if (const Json* mt = root.get("model_type"))
        if (mt->kind == Json::Str && mt->str != "bert")
            return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: model_type \"%s\" is not a BERT encoder",
                        mt->str.c_str());
    if (const Json* act = root.get("hidden_act"))
        if (act->kind == Json::Str && act->str != "gelu")
            return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: hidden_act \"%s\" (only erf-GELU)",
                        act->str.c_str());
    bool jina = false;
    if (const Json* pe = root.get("position_embedding_type")) {
        if (pe->kind == Json::Str && pe->str == "alibi") {
            // JinaBert (the registry's jina-embeddings-v2-base-code): only the published arrangement — GELU-gated feed-forward
            const Json* ff = root.get("feed_forward_type");
            if (!ff || ff->kind != Json::Str || ff->str != "geglu")
                return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: this JinaBert configuration is not built "
                            "(feed_forward_type \"%s\"; only \"geglu\")", (ff && ff->kind == Json::Str) ? ff->str.c_str() : "original");
            jina = true;
        } else if (pe->kind == Json::Str && pe->str != "absolute") {
            return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: position_embedding_type \"%s\"",
                        pe->str.c_str());
        }
    }
    cs_bert_config c{};
    if (!json_u32(root, "vocab_size", c.vocab_size) || !json_u32(root, "hidden_size", c.hidden) ||
        !json_u32(root, "num_hidden_layers", c.layers) || !json_u32(root, "num_attention_heads", c.heads) ||
        !json_u32(root, "intermediate_size", c.intermediate) ||
        !json_u32(root, "max_position_embeddings", c.max_position))
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s lacks a BERT size field", path.c_str());
    if (!json_u32(root, "type_vocab_size", c.type_vocab_size)) c.type_vocab_size = 2;
    const Json* eps = root.get("layer_norm_eps");
    c.layer_norm_eps = (eps && eps->kind == Json::Num) ? (float)eps->num : 1e-12f;
    c.pooling = pooling;
    if (jina) {
        // no position table to size: the bound is what fastembed's default InitOptions truncate to (embedder.rs:238)
        if (c.max_position > 512) c.max_position = 512;
        // the modelling file the config names decides whether Q and K rows are LayerNorm'ed ("...qk-post-norm...")
        bool qkn = true;
        if (const Json* am = root.get("auto_map"))
            if (am->kind == Json::Obj && !am->obj.empty()) {
                qkn = false;
                for (const auto& kv : am->obj)
                    if (kv.second.kind == Json::Str && kv.second.str.find("qk-post-norm") != std::string::npos) qkn = true;
            }
        // ... and the weights file next to it has the last word: its own tensors say whether the query / key LayerNorms exist
        const int in_file = jina_variant_from_files(model_dir);
        if (in_file >= 0) qkn = in_file != 0;
        c.arch = qkn ? CS_ARCH_JINA_QKNORM : CS_ARCH_JINA;
        c.pooling = pooling_given ? pooling : (pooling_file ? pooling : CS_POOL_MEAN);  // fastembed pools the model by mean
    }
    *cfg = c;
    return CS_OK;
}

int32_t cs_bert_params_from_safetensors(const char* path, const cs_bert_config* cfg, float* params,
                                        uint64_t n_params) {
    if (!path || !cfg || !params) return fail(CS_ERR_BAD_ARG, "null argument");
    cs_bert_offsets o;
    cs_bert_layout(cfg, &o);
    if (n_params != o.total)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: expected %llu parameters, got %llu",
                    (unsigned long long)o.total, (unsigned long long)n_params);
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: cannot open %s", path);
    struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{f};
    unsigned char lenb[8];
    if (std::fread(lenb, 1, 8, f) != 8)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s is not a safetensors file", path);
    uint64_t hlen = 0;
    for (int i = 7; i >= 0; --i) hlen = (hlen << 8) | lenb[i];
    if (hlen < 2 || hlen > (100ull << 20))
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s has an implausible header length", path);
    std::string header(hlen, '\0');
    if (std::fread(&header[0], 1, hlen, f) != hlen)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s is truncated", path);
    JsonParser jp{header.data(), header.data() + header.size()};
    const Json root = jp.value();
    if (!jp.ok || root.kind != Json::Obj)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s has a malformed header", path);
    std::map<std::string, TensorRef> have;
    for (const auto& kv : root.obj) {
        if (kv.first == "__metadata__" || kv.second.kind != Json::Obj) continue;
        const Json *dt = kv.second.get("dtype"), *sh = kv.second.get("shape"), *off = kv.second.get("data_offsets");
        if (!dt || dt->kind != Json::Str || !sh || sh->kind != Json::Arr || !off || off->kind != Json::Arr ||
            off->arr.size() != 2)
            continue;
        TensorRef t;
        t.dtype = dt->str;
        // numbers of the header become integers only inside [0, 2^53]: a negative, huge or non-numeric entry makes the
        // tensor unusable (it then counts as missing) instead of an undefined double -> integer conversion
        bool sane = off->arr[0].kind == Json::Num && off->arr[1].kind == Json::Num;
        auto as_u64 = [&](const Json& j) -> uint64_t {
            if (j.kind != Json::Num || !(j.num >= 0.0 && j.num <= 9007199254740992.0)) { sane = false; return 0; }
            return (uint64_t)j.num;
        };
        for (const Json& d : sh->arr) t.shape.push_back(as_u64(d));
        t.begin = as_u64(off->arr[0]);
        t.end = as_u64(off->arr[1]);
        if (!sane) continue;
        have[kv.first] = std::move(t);
    }
    const uint64_t data0 = 8 + hlen;
    if (cfg->arch == CS_ARCH_MODERN) return modern_params_from_safetensors(f, have, data0, *cfg, params, path);
    std::vector<unsigned char> raw;
    bool jina_first_file = false;
    if (cs_arch_alibi(cfg->arch))
        for (const auto& kv : have)
            if (kv.first.find("mlp.gated_layers.weight") != std::string::npos) { jina_first_file = true; break; }
    for (const Want& w : layout_table(*cfg, jina_first_file)) {
        auto it = have.find(w.name);
        if (it == have.end()) it = have.find("bert." + w.name);
        if (it == have.end()) it = have.find("model." + w.name);
        uint64_t count = 1;
        for (uint64_t d : w.shape) count *= d;
        if (it == have.end()) {
            if (w.optional) {
                std::memset(params + w.off, 0, count * sizeof(float));
                continue;
            }
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: tensor %s is missing from %s",
                        w.name.c_str(), path);
        }
        const TensorRef& t = it->second;
        std::vector<uint64_t> file_shape = w.shape;
        if (w.src_rows) file_shape[0] = w.src_rows;
        const uint64_t skip = w.src_rows ? w.row0 * (count / w.shape[0]) : 0;  // elements in front of the wanted rows
        uint64_t file_count = 1;
        for (uint64_t d : file_shape) file_count *= d;
        if (t.shape != file_shape) {
            std::string got, exp;
            for (uint64_t d : t.shape) got += (got.empty() ? "" : ", ") + std::to_string(d);
            for (uint64_t d : file_shape) exp += (exp.empty() ? "" : ", ") + std::to_string(d);
            return fail(CS_ERR_DIM_MISMATCH, "Failed to initialize embedding model: %s has shape [%s], config.json implies [%s]",
                        w.name.c_str(), got.c_str(), exp.c_str());
        }
        const uint32_t esz = t.dtype == "F32" ? 4 : (t.dtype == "F16" || t.dtype == "BF16") ? 2 : 0;
        if (!esz)
            return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: %s has dtype %s (F32, F16, BF16 only)",
                        w.name.c_str(), t.dtype.c_str());
        if (t.end < t.begin || t.end - t.begin != file_count * esz)
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s has inconsistent data_offsets", w.name.c_str());
        if (fseeko(f, (off_t)(data0 + t.begin + skip * esz), SEEK_SET) != 0)
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s is truncated", path);
        float* dst = params + w.off;
        if (esz == 4) {
            if (std::fread(dst, 4, count, f) != count)
                return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s is truncated", path);
        } else {
            raw.resize(count * 2);
            if (std::fread(raw.data(), 2, count, f) != count)
                return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s is truncated", path);
            const bool bf = t.dtype == "BF16";
            for (uint64_t i = 0; i < count; ++i) {
                const uint16_t v = (uint16_t)(raw[2 * i] | (raw[2 * i + 1] << 8));
                if (bf) {
                    const uint32_t bits = (uint32_t)v << 16;
                    std::memcpy(dst + i, &bits, 4);
                } else {
                    dst[i] = half_to_float(v);
                }
            }
        }
    }
    return CS_OK;
}

// Does the JSON header of a safetensors file mention `needle` (a tensor-name fragment)?  -1: the file cannot be read.
static int safetensors_header_mentions(const std::string& path, const char* needle) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return -1;
    unsigned char lenb[8];
    uint64_t hlen = 0;
    int r = -1;
    if (std::fread(lenb, 1, 8, f) == 8) {
        for (int i = 7; i >= 0; --i) hlen = (hlen << 8) | lenb[i];
        if (hlen >= 2 && hlen <= (100ull << 20)) {
            std::string header(hlen, '\0');
            if (std::fread(&header[0], 1, hlen, f) == hlen) r = header.find(needle) != std::string::npos ? 1 : 0;
        }
    }
    std::fclose(f);
    return r;
}

int32_t cs_embedder_create_from_dir(const char* model_dir, int32_t pooling, int32_t device, cs_embedder** out) {
    if (!out) return fail(CS_ERR_BAD_ARG, "null out pointer");
    *out = nullptr;
    cs_bert_config cfg;
    CS_TRY(cs_bert_config_from_dir(model_dir, pooling, &cfg));
    // hf-hub snapshot of the PyTorch model (model.safetensors) or fastembed's cache of the ONNX export
    // (onnx/model.onnx for Xenova/bge-small-en-v1.5; model.onnx / model_optimized.onnx for other entries; the *Q entries of the
    // registry fetch onnx/model_quantized.onnx — Xenova/all-MiniLM-L6-v2, the reference's default model — or a
    // model_optimized.onnx that holds quantised weights; hf-hub leaves only the file asked for)
    bool have_st = false;
    const std::string file = model_file_in(model_dir, have_st);
    if (file.empty())
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s holds neither model.safetensors nor "
                    "onnx/model.onnx, model.onnx, model_optimized.onnx or model_quantized.onnx", model_dir);
    const std::string& st_path = file;
    const std::string& onnx = file;
    const uint64_t n = cs_bert_param_count(&cfg);
    std::vector<float> params;
    try {
        params.resize(n);
    } catch (const std::bad_alloc&) {
        return fail(CS_ERR_OOM, "out of host memory for %llu parameters", (unsigned long long)n);
    }
    if (have_st) {
        CS_TRY(cs_bert_params_from_safetensors(st_path.c_str(), &cfg, params.data(), n));
    } else {
        std::vector<float> wscale((size_t)cfg.layers * cs_bert_quant_columns(&cfg));
        int32_t quantized = 0;
        CS_TRY(cs_bert_params_from_onnx_q(onnx.c_str(), &cfg, params.data(), n, wscale.data(), wscale.size(), &quantized));
        if (quantized)  // every Linear behind MatMulInteger: run them as the graph does (CS_GEMM_Q8_DYNAMIC)
            return cs_embedder_create_quantized(&cfg, params.data(), wscale.data(), wscale.size(), device, out);
    }
    return cs_embedder_create(&cfg, params.data(), 0, device, out);
}

// ---- tokenizer.json with a SentencePiece-unigram model (the XLM-R vocabulary of the registry's multilingual entries) ----
// Read into a cs::UnigramSpec (unigram.hpp) — every component must be one unigram.cpp restates, anything else is refused:
//   model        type Unigram, unk_id, vocab [[piece, score], ...], byte_fallback false
//   normalizer   null | Precompiled | Replace | Strip | Sequence of those; Replace patterns: Regex " {2,}" or a String
//   pre_tokenizer  WhitespaceSplit | Metaspace | Sequence of those
//   post_processor TemplateProcessing whose `single` is <bos> $A <eos>
//   added_tokens   special, not normalized, not single_word (matched in the raw text)
static bool base64_decode(const std::string& in, std::string& out) {
    out.clear();
    uint32_t acc = 0;
    int bits = 0;
    for (unsigned char c : in) {
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else if (c == '=' || c == '\n' || c == '\r') continue;
        else return false;
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back((char)((acc >> bits) & 0xFF));
        }
    }
    return true;
}

static int32_t unigram_norm(const Json& nz, std::vector<cs::UnigramSpec::Norm>& out) {
    if (nz.kind == Json::Null) return CS_OK;
    if (nz.kind != Json::Obj) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: tokenizer.json normalizer is not an object");
    const Json* ty = nz.get("type");
    const std::string t = ty && ty->kind == Json::Str ? ty->str : "";
    cs::UnigramSpec::Norm n;
    if (t == "Sequence") {
        const Json* list = nz.get("normalizers");
        if (!list || list->kind != Json::Arr) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: normalizer Sequence without a list");
        for (const Json& e : list->arr) CS_TRY(unigram_norm(e, out));
        return CS_OK;
    }
    if (t == "Precompiled") {
        const Json* b = nz.get("precompiled_charsmap");
        n.kind = cs::UnigramSpec::Norm::PRECOMPILED;
        if (b && b->kind == Json::Str && !base64_decode(b->str, n.blob))
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: precompiled_charsmap is not base64");
        out.push_back(std::move(n));
        return CS_OK;
    }
    if (t == "Replace") {
        const Json* pat = nz.get("pattern");
        const Json* content = nz.get("content");
        if (!pat || pat->kind != Json::Obj || !content || content->kind != Json::Str)
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: normalizer Replace without pattern / content");
        n.content = content->str;
        if (const Json* re = pat->get("Regex")) {
            if (re->kind != Json::Str || re->str != " {2,}")
                return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: normalizer Replace with the regular expression \"%s\" "
                            "(only \" {2,}\" is built)", re->kind == Json::Str ? re->str.c_str() : "?");
            n.kind = cs::UnigramSpec::Norm::REPLACE_SPACES;
        } else if (const Json* st = pat->get("String")) {
            if (st->kind != Json::Str) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: normalizer Replace pattern is not a string");
            n.kind = cs::UnigramSpec::Norm::REPLACE_STRING;
            n.pattern = st->str;
        } else {
            return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: normalizer Replace pattern kind is not built");
        }
        out.push_back(std::move(n));
        return CS_OK;
    }
    if (t == "Strip") {
        n.kind = cs::UnigramSpec::Norm::STRIP;
        const Json* l = nz.get("strip_left");
        const Json* r = nz.get("strip_right");
        n.left = l && l->kind == Json::Bool && l->b;
        n.right = r && r->kind == Json::Bool && r->b;
        out.push_back(std::move(n));
        return CS_OK;
    }
    return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: normalizer \"%s\" is not built for unigram tokenizers", t.c_str());
}

static int32_t unigram_pre(const Json& pj, std::vector<cs::UnigramSpec::Pre>& out) {
    if (pj.kind == Json::Null) return CS_OK;
    if (pj.kind != Json::Obj) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: tokenizer.json pre_tokenizer is not an object");
    const Json* ty = pj.get("type");
    const std::string t = ty && ty->kind == Json::Str ? ty->str : "";
    cs::UnigramSpec::Pre p;
    if (t == "Sequence") {
        const Json* list = pj.get("pretokenizers");
        if (!list || list->kind != Json::Arr) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: pre_tokenizer Sequence without a list");
        for (const Json& e : list->arr) CS_TRY(unigram_pre(e, out));
        return CS_OK;
    }
    if (t == "WhitespaceSplit") {
        p.kind = cs::UnigramSpec::Pre::WHITESPACE_SPLIT;
        out.push_back(p);
        return CS_OK;
    }
    if (t == "Metaspace") {
        p.kind = cs::UnigramSpec::Pre::METASPACE;
        const Json* rep = pj.get("replacement");
        p.replacement = rep && rep->kind == Json::Str ? rep->str : "\xE2\x96\x81";
        if (p.replacement.empty() || p.replacement == " ")
            return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: Metaspace replacement must be one non-space character");
        p.prepend = 1;
        if (const Json* ps = pj.get("prepend_scheme")) {
            if (ps->kind == Json::Str) {
                if (ps->str == "always") p.prepend = 1;
                else if (ps->str == "never") p.prepend = 0;
                else if (ps->str == "first") p.prepend = 2;
                else return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: Metaspace prepend_scheme \"%s\"", ps->str.c_str());
            }
        } else if (const Json* aps = pj.get("add_prefix_space")) {  // the older serialisation
            if (aps->kind == Json::Bool) p.prepend = aps->b ? 1 : 0;
        }
        const Json* sp = pj.get("split");
        p.split = !(sp && sp->kind == Json::Bool && !sp->b);
        out.push_back(p);
        return CS_OK;
    }
    return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: pre_tokenizer \"%s\" is not built for unigram tokenizers", t.c_str());
}

static int32_t unigram_from_json(const Json& root, const Json& model, const char* json_path, uint32_t max_length, cs_tokenizer** out) {
    cs::UnigramSpec spec;
    const Json* vocab = model.get("vocab");
    if (!vocab || vocab->kind != Json::Arr || vocab->arr.empty())
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s has no unigram vocabulary", json_path);
    spec.vocab.reserve(vocab->arr.size());
    for (const Json& e : vocab->arr) {
        if (e.kind != Json::Arr || e.arr.size() != 2 || e.arr[0].kind != Json::Str || e.arr[1].kind != Json::Num)
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: a unigram vocabulary entry is not [piece, score]");
        spec.vocab.emplace_back(e.arr[0].str, e.arr[1].num);
    }
    const Json* unk = model.get("unk_id");
    if (!unk || unk->kind != Json::Num || !(unk->num >= 0 && unk->num < (double)spec.vocab.size()))
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: unigram model without a valid unk_id");
    spec.unk_id = (int32_t)unk->num;
    if (const Json* bf = model.get("byte_fallback"))
        if (bf->kind == Json::Bool && bf->b) return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: unigram byte_fallback is not built");
    if (const Json* nz = root.get("normalizer")) CS_TRY(unigram_norm(*nz, spec.norms));
    if (const Json* pj = root.get("pre_tokenizer")) CS_TRY(unigram_pre(*pj, spec.pres));
    auto id_of = [&](const std::string& piece) -> int32_t {
        int32_t id = -1;
        for (size_t i = 0; i < spec.vocab.size(); ++i)
            if (spec.vocab[i].first == piece) id = (int32_t)i;
        return id;
    };
    if (const Json* at = root.get("added_tokens")) {
        if (at->kind == Json::Arr)
            for (const Json& e : at->arr) {
                if (e.kind != Json::Obj) continue;
                const Json* content = e.get("content");
                const Json* id = e.get("id");
                if (!content || content->kind != Json::Str || !id || id->kind != Json::Num || content->str.empty()) continue;
                auto flag = [&](const char* k) { const Json* v = e.get(k); return v && v->kind == Json::Bool && v->b; };
                if (flag("single_word"))
                    return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: added token \"%s\" is single_word (not built)",
                                content->str.c_str());
                if (!(id->num >= 0 && id->num < (double)spec.vocab.size()))
                    return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: added token \"%s\" lies outside the unigram vocabulary",
                                content->str.c_str());
                cs::UnigramSpec::Added a;
                a.text = content->str;
                a.id = (int32_t)id->num;
                a.lstrip = flag("lstrip");
                a.rstrip = flag("rstrip");
                spec.added.push_back(std::move(a));
            }
    }
    // TemplateProcessing: single = [SpecialToken bos, Sequence A, SpecialToken eos]
    std::string bos = "<s>", eos = "</s>";
    if (const Json* pp = root.get("post_processor")) {
        if (pp->kind == Json::Obj) {
            const Json* ty = pp->get("type");
            if (!ty || ty->kind != Json::Str || ty->str != "TemplateProcessing")
                return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: post_processor \"%s\" is not built for unigram tokenizers",
                            ty && ty->kind == Json::Str ? ty->str.c_str() : "?");
            const Json* single = pp->get("single");
            if (!single || single->kind != Json::Arr || single->arr.size() != 3)
                return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: TemplateProcessing `single` is not <bos> $A <eos>");
            auto special = [&](const Json& e, std::string& name) {
                const Json* st = e.kind == Json::Obj ? e.get("SpecialToken") : nullptr;
                const Json* id = st && st->kind == Json::Obj ? st->get("id") : nullptr;
                if (!id || id->kind != Json::Str) return false;
                name = id->str;
                return true;
            };
            const Json* seq = single->arr[1].kind == Json::Obj ? single->arr[1].get("Sequence") : nullptr;
            if (!special(single->arr[0], bos) || !special(single->arr[2], eos) || !seq)
                return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: TemplateProcessing `single` is not <bos> $A <eos>");
        }
    } else {
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: unigram tokenizer.json without a post_processor");
    }
    spec.bos = id_of(bos);
    spec.eos = id_of(eos);
    spec.pad = id_of("<pad>");
    if (const Json* pad = root.get("padding"))
        if (pad->kind == Json::Obj)
            if (const Json* pid = pad->get("pad_id"))
                if (pid->kind == Json::Num && pid->num >= 0 && pid->num < (double)spec.vocab.size()) spec.pad = (int32_t)pid->num;
    if (max_length == 0) {
        max_length = 512;  // fastembed's default truncation length
        if (const Json* tr = root.get("truncation"))
            if (tr->kind == Json::Obj)
                if (const Json* ml = tr->get("max_length"))
                    if (ml->kind == Json::Num && ml->num >= 2 && ml->num <= 1e6) max_length = (uint32_t)ml->num;
    }
    return cs::tokenizer_from_unigram(std::move(spec), max_length, out);
}

// ---- tokenizer.json with a byte-level BPE model (the registry's JinaEmbeddingsV2BaseCode) -----------------------------------
// Read into a cs::BpeSpec (bpe.hpp) — every component must be one bpe.cpp restates, anything else is refused:
//   model          type BPE, vocab {token: id}, merges ["a b", ...] or [["a", "b"], ...], unk_token, fuse_unk, ignore_merges;
//                  no dropout, no continuing_subword_prefix / end_of_word_suffix, no byte_fallback
//   normalizer     null
//   pre_tokenizer  ByteLevel | Digits | Sequence of those ending in ByteLevel
//   post_processor RobertaProcessing (cls, sep) | TemplateProcessing whose `single` is <bos> $A <eos> | ByteLevel (no template)
//                  | Sequence of those
//   added_tokens   special, not normalized, not single_word (matched in the raw text)
static int32_t bpe_pre(const Json& pj, std::vector<cs::BpeSpec::Pre>& out) {
    if (pj.kind == Json::Null) return CS_OK;
    if (pj.kind != Json::Obj) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: tokenizer.json pre_tokenizer is not an object");
    const Json* ty = pj.get("type");
    const std::string t = ty && ty->kind == Json::Str ? ty->str : "";
    auto flag = [&](const char* k, bool dflt) { const Json* v = pj.get(k); return v && v->kind == Json::Bool ? v->b : dflt; };
    if (t == "Sequence") {
        const Json* list = pj.get("pretokenizers");
        if (!list || list->kind != Json::Arr) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: pre_tokenizer Sequence without a list");
        for (const Json& e : list->arr) CS_TRY(bpe_pre(e, out));
        return CS_OK;
    }
    cs::BpeSpec::Pre p;
    if (t == "ByteLevel") {
        p.kind = cs::BpeSpec::Pre::BYTE_LEVEL;
        p.add_prefix_space = flag("add_prefix_space", true);
        p.use_regex = flag("use_regex", true);
    } else if (t == "Digits") {
        p.kind = cs::BpeSpec::Pre::DIGITS;
        p.individual_digits = flag("individual_digits", false);
    } else {
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: pre_tokenizer \"%s\" is not built for BPE tokenizers (ByteLevel, Digits)", t.c_str());
    }
    out.push_back(p);
    return CS_OK;
}

static int32_t bpe_post(const Json& pp, std::string& bos, std::string& eos) {
    if (pp.kind == Json::Null) return CS_OK;
    if (pp.kind != Json::Obj) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: tokenizer.json post_processor is not an object");
    const Json* ty = pp.get("type");
    const std::string t = ty && ty->kind == Json::Str ? ty->str : "";
    if (t == "ByteLevel") return CS_OK;  // offsets only
    if (t == "Sequence") {
        const Json* list = pp.get("processors");
        if (!list || list->kind != Json::Arr) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: post_processor Sequence without a list");
        for (const Json& e : list->arr) CS_TRY(bpe_post(e, bos, eos));
        return CS_OK;
    }
    if (t == "RobertaProcessing" || t == "BertProcessing") {
        auto first = [&](const char* k, std::string& name) {
            const Json* v = pp.get(k);
            if (!v || v->kind != Json::Arr || v->arr.size() != 2 || v->arr[0].kind != Json::Str) return false;
            name = v->arr[0].str;
            return true;
        };
        if (!first("cls", bos) || !first("sep", eos))
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s without cls / sep", t.c_str());
        return CS_OK;
    }
    if (t == "TemplateProcessing") {
        const Json* single = pp.get("single");
        if (!single || single->kind != Json::Arr)
            return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: TemplateProcessing without `single`");
        auto special = [&](const Json& e, std::string& name) {
            const Json* st = e.kind == Json::Obj ? e.get("SpecialToken") : nullptr;
            const Json* id = st && st->kind == Json::Obj ? st->get("id") : nullptr;
            if (!id || id->kind != Json::Str) return false;
            name = id->str;
            return true;
        };
        auto is_seq = [&](const Json& e) { return e.kind == Json::Obj && e.get("Sequence") != nullptr; };
        const auto& a = single->arr;
        bool ok = false;
        if (a.size() == 1 && is_seq(a[0])) ok = true;                                                  // $A
        else if (a.size() == 2 && special(a[0], bos) && is_seq(a[1])) ok = true;                       // <bos> $A
        else if (a.size() == 2 && is_seq(a[0]) && special(a[1], eos)) ok = true;                       // $A <eos>
        else if (a.size() == 3 && special(a[0], bos) && is_seq(a[1]) && special(a[2], eos)) ok = true; // <bos> $A <eos>
        if (!ok) return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: TemplateProcessing `single` is not [<bos>] $A [<eos>]");
        return CS_OK;
    }
    return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: post_processor \"%s\" is not built for BPE tokenizers", t.c_str());
}

static int32_t bpe_from_json(const Json& root, const Json& model, const char* json_path, uint32_t max_length, cs_tokenizer** out) {
    cs::BpeSpec spec;
    const Json* vocab = model.get("vocab");
    if (!vocab || vocab->kind != Json::Obj || vocab->obj.empty())
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s has no BPE vocabulary", json_path);
    for (const auto& kv : vocab->obj) {
        if (kv.second.kind != Json::Num || !(kv.second.num >= 0 && kv.second.num < 2147483647.0))
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: BPE vocabulary entry \"%s\" has no valid id", kv.first.c_str());
        spec.vocab.emplace_back(kv.first, (int32_t)kv.second.num);
    }
    const Json* merges = model.get("merges");
    if (!merges || merges->kind != Json::Arr)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s has no BPE merges", json_path);
    for (const Json& m : merges->arr) {
        if (m.kind == Json::Str) {  // "left right"
            const size_t sp = m.str.find(' ');
            if (sp == std::string::npos || m.str.find(' ', sp + 1) != std::string::npos)
                return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: merge \"%s\" is not two tokens", m.str.c_str());
            spec.merges.emplace_back(m.str.substr(0, sp), m.str.substr(sp + 1));
        } else if (m.kind == Json::Arr && m.arr.size() == 2 && m.arr[0].kind == Json::Str && m.arr[1].kind == Json::Str) {
            spec.merges.emplace_back(m.arr[0].str, m.arr[1].str);
        } else {
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: a BPE merge is neither \"a b\" nor [a, b]");
        }
    }
    auto refuse_set = [&](const char* key, const char* what) -> int32_t {
        const Json* v = model.get(key);
        if (!v || v->kind == Json::Null) return CS_OK;
        if (v->kind == Json::Str && v->str.empty()) return CS_OK;
        if (v->kind == Json::Bool && !v->b) return CS_OK;
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: BPE %s is not built", what);
    };
    CS_TRY(refuse_set("dropout", "dropout"));
    CS_TRY(refuse_set("continuing_subword_prefix", "continuing_subword_prefix"));
    CS_TRY(refuse_set("end_of_word_suffix", "end_of_word_suffix"));
    CS_TRY(refuse_set("byte_fallback", "byte_fallback"));
    if (const Json* u = model.get("unk_token"))
        if (u->kind == Json::Str) spec.unk_token = u->str;
    auto mflag = [&](const char* k) { const Json* v = model.get(k); return v && v->kind == Json::Bool && v->b; };
    spec.fuse_unk = mflag("fuse_unk");
    spec.ignore_merges = mflag("ignore_merges");
    if (const Json* nz = root.get("normalizer"))
        if (nz->kind != Json::Null) {
            // NFC (ModernBERT's file), alone or as the one member of a Sequence; anything else is refused
            const Json* one = nz;
            if (nz->kind == Json::Obj)
                if (const Json* ty = nz->get("type"))
                    if (ty->kind == Json::Str && ty->str == "Sequence")
                        if (const Json* list = nz->get("normalizers"))
                            if (list->kind == Json::Arr && list->arr.size() == 1) one = &list->arr[0];
            const Json* ty = one->kind == Json::Obj ? one->get("type") : nullptr;
            if (!ty || ty->kind != Json::Str || ty->str != "NFC")
                return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: normalizer \"%s\" in front of a BPE model is not built (NFC is)",
                            ty && ty->kind == Json::Str ? ty->str.c_str() : "?");
            spec.nfc = true;
        }
    if (const Json* pj = root.get("pre_tokenizer")) CS_TRY(bpe_pre(*pj, spec.pres));
    std::map<std::string, int32_t> added_ids;
    if (const Json* at = root.get("added_tokens"))
        if (at->kind == Json::Arr)
            for (const Json& e : at->arr) {
                if (e.kind != Json::Obj) continue;
                const Json* content = e.get("content");
                const Json* id = e.get("id");
                if (!content || content->kind != Json::Str || !id || id->kind != Json::Num || content->str.empty()) continue;
                auto flag = [&](const char* k) { const Json* v = e.get(k); return v && v->kind == Json::Bool && v->b; };
                if (flag("single_word"))
                    return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: added token \"%s\" is single_word (not built)",
                                content->str.c_str());
                if (!(id->num >= 0 && id->num < 2147483647.0))
                    return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: added token \"%s\" has no valid id", content->str.c_str());
                cs::BpeSpec::Added a;
                a.text = content->str;
                a.id = (int32_t)id->num;
                a.lstrip = flag("lstrip");
                a.rstrip = flag("rstrip");
                a.normalized = flag("normalized");  // matched in the normalised text, behind the verbatim ones (bpe.hpp)
                added_ids[a.text] = a.id;
                spec.added.push_back(std::move(a));
            }
    std::string bos, eos;
    if (const Json* pp = root.get("post_processor")) CS_TRY(bpe_post(*pp, bos, eos));
    auto id_of = [&](const std::string& tok) -> int32_t {
        if (tok.empty()) return -1;
        auto it = added_ids.find(tok);
        if (it != added_ids.end()) return it->second;
        for (const auto& kv : spec.vocab)
            if (kv.first == tok) return kv.second;
        return -2;
    };
    spec.bos = id_of(bos);
    spec.eos = id_of(eos);
    if (spec.bos == -2 || spec.eos == -2)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: the post_processor's special tokens are not in the vocabulary");
    spec.pad = id_of("<pad>");
    if (spec.pad == -2) spec.pad = -1;
    if (const Json* pad = root.get("padding"))
        if (pad->kind == Json::Obj)
            if (const Json* pid = pad->get("pad_id"))
                if (pid->kind == Json::Num && pid->num >= 0 && pid->num < 2147483647.0) spec.pad = (int32_t)pid->num;
    if (max_length == 0) {
        max_length = 512;  // fastembed's default truncation length
        if (const Json* tr = root.get("truncation"))
            if (tr->kind == Json::Obj)
                if (const Json* ml = tr->get("max_length"))
                    if (ml->kind == Json::Num && ml->num >= 2 && ml->num <= 1e6) max_length = (uint32_t)ml->num;
    }
    return cs::tokenizer_from_bpe(std::move(spec), max_length, out);
}

// ---- tokenizer.json (the `tokenizers` crate's serialisation; what fastembed loads) ------------------
// Read: model.type == "WordPiece", model.vocab {token: id}, model.unk_token / continuing_subword_prefix /
// max_input_chars_per_word (must be the BERT values cs_tokenizer implements), normalizer BertNormalizer
// .lowercase (strip_accents null or equal to it), truncation.max_length.  The vocabulary is handed to
// cs_tokenizer_create in vocab.txt form (one token per line, id = line number).
int32_t cs_tokenizer_create_from_json(const char* json_path, uint32_t max_length, cs_tokenizer** out) {
    if (!out) return fail(CS_ERR_BAD_ARG, "null out pointer");
    *out = nullptr;
    if (!json_path) return fail(CS_ERR_BAD_ARG, "null tokenizer.json path");
    std::string text;
    if (!read_file(json_path, text, 1ull << 30))
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: cannot open %s", json_path);
    JsonParser jp{text.data(), text.data() + text.size()};
    const Json root = jp.value();
    if (!jp.ok || root.kind != Json::Obj)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s is not a JSON object", json_path);
    const Json* model = root.get("model");
    if (!model || model->kind != Json::Obj)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s has no \"model\" object", json_path);
    const Json* type = model->get("type");
    if (type && type->kind == Json::Str && type->str == "Unigram") return unigram_from_json(root, *model, json_path, max_length, out);
    if (type && type->kind == Json::Str && type->str == "BPE") return bpe_from_json(root, *model, json_path, max_length, out);
    if (type && type->kind == Json::Str && type->str != "WordPiece")
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: tokenizer model \"%s\" (WordPiece, Unigram and BPE are built)",
                    type->str.c_str());
    const Json* vocab = model->get("vocab");
    if (!vocab || vocab->kind != Json::Obj || vocab->obj.empty())
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s has no WordPiece vocabulary", json_path);
    if (const Json* j = model->get("unk_token"))
        if (j->kind == Json::Str && j->str != "[UNK]")
            return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: unk_token \"%s\" (expected [UNK])", j->str.c_str());
    if (const Json* j = model->get("continuing_subword_prefix"))
        if (j->kind == Json::Str && j->str != "##")
            return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: continuing_subword_prefix \"%s\" (expected ##)",
                        j->str.c_str());
    if (const Json* j = model->get("max_input_chars_per_word"))
        if (j->kind == Json::Num && j->num != 100.0)
            return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: max_input_chars_per_word %g (expected 100)", j->num);
    int lowercase = 1;
    if (const Json* nz = root.get("normalizer")) {
        if (nz->kind == Json::Obj) {
            const Json* nt = nz->get("type");
            if (nt && nt->kind == Json::Str && nt->str != "BertNormalizer")
                return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: normalizer \"%s\" (only BertNormalizer)",
                            nt->str.c_str());
            const Json* lc = nz->get("lowercase");
            if (lc && lc->kind == Json::Bool) lowercase = lc->b ? 1 : 0;
            const Json* sa = nz->get("strip_accents");
            if (sa && sa->kind == Json::Bool && (int)sa->b != lowercase)
                return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: strip_accents differs from lowercase");
            for (const char* key : {"clean_text", "handle_chinese_chars"}) {
                const Json* v = nz->get(key);
                if (v && v->kind == Json::Bool && !v->b)
                    return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: normalizer.%s = false is not built", key);
            }
        }
    }
    if (max_length == 0) {
        max_length = 512;  // fastembed's default truncation length
        if (const Json* tr = root.get("truncation"))
            if (tr->kind == Json::Obj)
                if (const Json* ml = tr->get("max_length"))
                    if (ml->kind == Json::Num && ml->num >= 2 && ml->num <= 1e6) max_length = (uint32_t)ml->num;
    }
    uint64_t max_id = 0;
    for (const auto& kv : vocab->obj) {
        if (kv.second.kind != Json::Num || !(kv.second.num >= 0 && kv.second.num <= 16777216.0))
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: vocabulary id of \"%s\" is not a small integer",
                        kv.first.c_str());
        max_id = std::max<uint64_t>(max_id, (uint64_t)kv.second.num);
    }
    std::vector<const std::string*> by_id(max_id + 1, nullptr);
    for (const auto& kv : vocab->obj) {
        if (kv.first.find('\n') != std::string::npos)
            return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: a vocabulary entry contains a line break");
        by_id[(size_t)kv.second.num] = &kv.first;
    }
    std::string lines;
    for (size_t i = 0; i <= max_id; ++i) {
        if (by_id[i]) lines += *by_id[i];
        else lines += "[unused-id-" + std::to_string(i) + "]";  // a hole in the id space: never produced
        lines.push_back('\n');
    }
    return cs_tokenizer_create(lines.data(), lines.size(), lowercase, max_length, out);
}

// What fastembed builds its tokenizer from, in a model directory: tokenizer.json when present (truncation at
// min(max_length or 512, tokenizer_config.json's model_max_length)), otherwise vocab.txt with
// tokenizer_config.json's do_lower_case (default true).
int32_t cs_tokenizer_create_from_dir(const char* model_dir, uint32_t max_length, cs_tokenizer** out) {
    if (!out) return fail(CS_ERR_BAD_ARG, "null out pointer");
    *out = nullptr;
    if (!model_dir) return fail(CS_ERR_BAD_ARG, "null model directory");
    const std::string dir(model_dir);
    int lowercase = 1;
    uint32_t cap = 0;
    std::string ctext;
    if (read_file(dir + "/tokenizer_config.json", ctext, 1 << 24)) {
        JsonParser jp{ctext.data(), ctext.data() + ctext.size()};
        const Json root = jp.value();
        if (jp.ok && root.kind == Json::Obj) {
            const Json* lc = root.get("do_lower_case");
            if (lc && lc->kind == Json::Bool) lowercase = lc->b ? 1 : 0;
            const Json* mm = root.get("model_max_length");
            if (mm && mm->kind == Json::Num && mm->num >= 2 && mm->num <= 1e6) cap = (uint32_t)mm->num;
        }
    }
    uint32_t want = max_length ? max_length : 512;
    if (cap && cap < want) want = cap;
    if (file_exists(dir + "/tokenizer.json")) return cs_tokenizer_create_from_json((dir + "/tokenizer.json").c_str(), want, out);
    if (file_exists(dir + "/vocab.txt")) return cs_tokenizer_create_from_file((dir + "/vocab.txt").c_str(), lowercase, want, out);
    return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s holds neither tokenizer.json nor vocab.txt", model_dir);
}

}  // extern "C"
