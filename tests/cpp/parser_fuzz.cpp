// parser_fuzz.cpp — CPU-only hardening driver for the hand-written readers of UNTRUSTED model files
// (codesearch_amd/csrc/onnx_reader.cpp: protobuf wire format, checkpoint.cpp: config.json / safetensors /
// tokenizer.json, tokenizer.cpp: vocab.txt).  Built by `make -C tests/cpp asan` with
// -fsanitize=address,undefined on the host compiler; the three sources are compiled as they are, the few symbols they
// take from the HIP side of the library (error plumbing, cs_embedder_create, the parameter layout) are stubbed below.
//
//   parser_fuzz <kind> <file> <seed> <flips> [<aux>]
//     kind = onnx | safetensors | tokenizer_json | vocab | config_dir
//     Feeds the loader (a) every prefix truncation of the file (all lengths up to 4 KiB, then 257 evenly spaced ones)
//     and (b) `flips` seeded mutations (1-8 random byte flips, or a random 8-byte little-endian length field
//     overwritten with a huge value).  Every call must RETURN — CS_OK or a CS_ERR_* with a message — never crash, trip a
//     sanitizer, or allocate without bound.  Prints a one-line summary; exit code 0 = survived.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "../../codesearch_amd/csrc/common.hpp"
#include "../../include/cs_bert_params.h"

namespace cs {
std::string& last_error_ref() {
    static thread_local std::string msg;
    return msg;
}
int32_t fail(int32_t code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}
}  // namespace cs

extern "C" {
const char* cs_last_error(void) { return cs::last_error_ref().c_str(); }
uint64_t cs_bert_param_count(const cs_bert_config* cfg) {
    cs_bert_offsets off;
    cs_bert_layout(cfg, &off);
    return off.total;
}
// the device half is out of reach here: a directory that parses cleanly ends in this stub
int32_t cs_embedder_create(const cs_bert_config*, const float*, uint64_t, int32_t, cs_embedder** out) {
    if (out) *out = nullptr;
    return cs::fail(CS_ERR_HIP, "stub: no device in the parser hardening driver");
}
uint64_t cs_bert_quant_columns(const cs_bert_config* cfg) { return cfg ? 5 * (uint64_t)cfg->hidden + cfg->intermediate : 0; }
int32_t cs_embedder_create_quantized(const cs_bert_config*, const float*, const float*, uint64_t, int32_t, cs_embedder** out) {
    if (out) *out = nullptr;
    return cs::fail(CS_ERR_HIP, "stub: no device in the parser hardening driver");
}
}

static std::vector<uint8_t> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string& path, const uint8_t* p, size_t n) {
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    f.write(reinterpret_cast<const char*>(p), (std::streamsize)n);
}

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: parser_fuzz <onnx|safetensors|tokenizer_json|vocab|config_dir> <file> <seed> <flips> [\"vocab hidden layers heads intermediate maxpos [arch]\"]\n");
        return 2;
    }
    const std::string kind = argv[1], src = argv[2];
    const uint64_t seed = strtoull(argv[3], nullptr, 10);
    const int flips = atoi(argv[4]);
    const std::vector<uint8_t> good = slurp(src);
    if (good.empty()) { fprintf(stderr, "cannot read %s\n", src.c_str()); return 2; }
    char tmpl[] = "/tmp/cs_parser_fuzz_XXXXXX";
    const char* dir = mkdtemp(tmpl);
    if (!dir) return 2;
    const std::string d = dir;
    // the config the model files are read against: argv[5] = "<vocab> <hidden> <layers> <heads> <intermediate> <maxpos>"
    cs_bert_config cfg{};
    cfg.vocab_size = 64; cfg.hidden = 384; cfg.layers = 1; cfg.heads = 12; cfg.intermediate = 1536; cfg.max_position = 32;
    cfg.type_vocab_size = 2; cfg.layer_norm_eps = 1e-12f; cfg.pooling = 0;
    if (argc > 5) {
        unsigned v[7] = {0, 0, 0, 0, 0, 0, 0};  // an optional seventh field: the encoder family (1 = NomicBert, rotary base 1000)
        const int got = sscanf(argv[5], "%u %u %u %u %u %u %u", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6]);
        if (got >= 6) {
            cfg.vocab_size = v[0]; cfg.hidden = v[1]; cfg.layers = v[2]; cfg.heads = v[3]; cfg.intermediate = v[4]; cfg.max_position = v[5];
            if (got == 7 && v[6] == 1) { cfg.arch = CS_ARCH_NOMIC; cfg.rotary_base = 1000.0f; }
            if (got == 7 && (v[6] == 2 || v[6] == 3)) cfg.arch = v[6] == 2 ? CS_ARCH_JINA : CS_ARCH_JINA_QKNORM;
            if (got == 7 && v[6] == 4) {
                cfg.arch = CS_ARCH_MODERN; cfg.rotary_base = 160000.0f; cfg.rotary_base_local = 10000.0f; cfg.local_window = 64;
                cfg.global_every = 3; cfg.type_vocab_size = 1;
            }
        }
    }
    const uint64_t n_params = cs_bert_param_count(&cfg);
    std::vector<float> params(n_params);
    std::string target = d + "/model.bin";
    if (kind == "config_dir") {  // the mutated file is config.json of a directory that holds nothing else
        target = d + "/config.json";
    } else if (kind == "tokenizer_json") {
        target = d + "/tokenizer.json";
    } else if (kind == "vocab") {
        target = d + "/vocab.txt";
    }
    unsigned long ok = 0, refused = 0, calls = 0;
    auto run = [&](const std::vector<uint8_t>& bytes, size_t n) {
        spit(target, bytes.data(), n);
        int32_t st = CS_OK;
        if (kind == "onnx") {
            std::vector<float> wscale((size_t)cfg.layers * cs_bert_quant_columns(&cfg));
            int32_t quantized = 0;
            st = cs_bert_params_from_onnx_q(target.c_str(), &cfg, params.data(), n_params, wscale.data(), wscale.size(), &quantized);
        } else if (kind == "safetensors") {
            st = cs_bert_params_from_safetensors(target.c_str(), &cfg, params.data(), n_params);
        } else if (kind == "tokenizer_json") {
            cs_tokenizer* t = nullptr;
            st = cs_tokenizer_create_from_json(target.c_str(), 0, &t);
            if (st == CS_OK && t) {  // a tokenizer that loads must also tokenise
                // (accents composed and combining, CJK, an emoji ZWJ sequence, width variants, special tokens of both kinds)
                const char text[] = "fn main() { let caf\xc3\xa9 = [SEP] 42; } e\xcc\x81\xcc\x82 \xe4\xb8\x96\xe7\x95\x8c  <mask> </s>"
                                    " \xf0\x9f\x91\xa8\xe2\x80\x8d\xf0\x9f\x91\xa9 \xef\xbd\x86\xef\xbd\x95 \xd8\x80\xd9\xa1 \r\n\t end";
                const uint64_t off[2] = {0, sizeof text - 1};
                uint32_t L = 0;
                (void)cs_tokenizer_encode_batch(t, text, off, 1, 0, nullptr, nullptr, 0, &L);
                std::vector<int32_t> ids(L ? L : 1), mask(L ? L : 1);
                (void)cs_tokenizer_encode_batch(t, text, off, 1, 0, ids.data(), mask.data(), L, &L);
            }
            cs_tokenizer_destroy(t);
        } else if (kind == "vocab") {
            cs_tokenizer* t = nullptr;
            st = cs_tokenizer_create_from_file(target.c_str(), 1, 64, &t);
            if (st == CS_OK && t) {
                const char text[] = "hello wor\xe4\xb8\x96ld [UNK] \xff\xfe";
                const uint64_t off[2] = {0, sizeof text - 1};
                uint32_t L = 0;
                (void)cs_tokenizer_encode_batch(t, text, off, 1, 0, nullptr, nullptr, 0, &L);
            }
            cs_tokenizer_destroy(t);
        } else if (kind == "config_dir") {
            cs_bert_config c2{};
            st = cs_bert_config_from_dir(d.c_str(), -1, &c2);
            cs_embedder* e = nullptr;
            (void)cs_embedder_create_from_dir(d.c_str(), -1, 0, &e);
            cs_tokenizer* t = nullptr;
            (void)cs_tokenizer_create_from_dir(d.c_str(), 0, &t);
            cs_tokenizer_destroy(t);
        } else {
            fprintf(stderr, "unknown kind %s\n", kind.c_str());
            exit(2);
        }
        ++calls;
        if (st == CS_OK) ++ok; else { ++refused; if (!*cs_last_error()) { fprintf(stderr, "status %d without a message\n", st); exit(1); } }
    };
    run(good, good.size());
    const unsigned long ok_intact = ok;
    // (a) truncations
    for (size_t n = 0; n < good.size() && n < 4096; ++n) run(good, n);
    for (int i = 0; i < 257; ++i) run(good, (size_t)((double)good.size() * i / 257.0));
    // (b) seeded mutations
    std::mt19937_64 rng(seed);
    for (int i = 0; i < flips; ++i) {
        std::vector<uint8_t> m = good;
        if (rng() % 4 == 0 && m.size() >= 16) {  // a length-looking field becomes huge
            const size_t at = rng() % (m.size() - 8);
            const uint64_t huge = (rng() % 2) ? 0xffffffffffffffffull : (1ull << (20 + rng() % 40));
            memcpy(&m[at], &huge, rng() % 2 ? 8 : 4);
        } else {
            const int k = 1 + (int)(rng() % 8);
            for (int j = 0; j < k; ++j) m[rng() % m.size()] = (uint8_t)rng();
        }
        run(m, m.size());
    }
    unlink(target.c_str());
    rmdir(d.c_str());
    printf("%s %s: %lu loader calls, %lu accepted (intact file: %s), %lu refused with a message, 0 crashes\n", kind.c_str(),
           src.c_str(), calls, ok, ok_intact ? "accepted" : "REFUSED", refused);
    return 0;
}
