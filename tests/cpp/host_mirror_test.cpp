// Exercises the C++ host mirror (codesearch_amd/host/codesearch_gpu.hpp) the way the
// reference's own unit test does (/root/reference/src/vectordb/store.rs:846-893), then a tiny
// encoder call.  Built by tests/test_cpp_host.py with g++ against libcsgpu.so; run on the GPU.
#include <cmath>
#include <cstdio>

#include "../../codesearch_amd/host/codesearch_gpu.hpp"

#define REQUIRE(c)                                                          \
    do {                                                                    \
        if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } \
    } while (0)

int main() {
    using namespace cs;
    if (cs_device_count() < 1) { std::printf("no HIP device\n"); return 77; }
    VectorStore store("test.db", 4);
    REQUIRE(store.dimensions() == 4 && !store.is_indexed());
    EmbeddedChunk a, b;
    a.chunk.content = "fn authenticate() {}"; a.chunk.kind = "Function"; a.chunk.path = "auth.rs";
    a.embedding = {1.f, 0.f, 0.f, 0.f};
    b.chunk.content = "fn calculate() {}"; b.chunk.kind = "Function"; b.chunk.path = "math.rs";
    b.embedding = {0.f, 1.f, 0.f, 0.f};
    REQUIRE(store.insert_chunks({a, b}) == 2);
    bool threw = false;
    try { store.search({0.9f, 0.1f, 0.f, 0.f}, 2); } catch (const Error& e) {
        threw = std::string(e.what()) == "Index not built. Call build_index() after inserting chunks.";
    }
    REQUIRE(threw);
    store.build_index();
    REQUIRE(store.is_indexed());
    auto res = store.search({0.9f, 0.1f, 0.f, 0.f}, 2);
    REQUIRE(res.size() == 2);
    REQUIRE(res[0].meta.content.find("authenticate") != std::string::npos);
    REQUIRE(res[0].score > res[1].score);
    REQUIRE(std::fabs((1.f - 2.f * res[0].distance) - 0.993884f) < 1e-5f);
    EmbeddedChunk bad = a;
    bad.embedding = {1.f, 2.f};
    threw = false;
    try { store.insert_chunks_with_ids({bad}); } catch (const Error& e) {
        threw = std::string(e.what()) == "Embedding dimension mismatch: expected 4, got 2" && e.code == CS_ERR_DIM_MISMATCH;
    }
    REQUIRE(threw);
    REQUIRE(store.delete_chunks({0}) == 1 && !store.is_indexed());
    store.build_index();
    REQUIRE(store.search({0.9f, 0.1f, 0.f, 0.f}, 2).size() == 1);
    REQUIRE(store.stats().total_chunks == 1);

    cs_bert_config cfg;
    cs_bert_config_bge_small(&cfg);
    cfg.vocab_size = 512; cfg.layers = 1;
    FastEmbedder emb(cfg, nullptr, 7);
    std::vector<int32_t> ids = {101, 300, 301, 102, 101, 400, 102, 0}, mask = {1, 1, 1, 1, 1, 1, 1, 0};
    auto e = emb.embed_batch(ids, mask, 2, 4);
    REQUIRE(e.size() == 2 && e[0].size() == 384);
    double n = 0;
    for (float v : e[1]) n += (double)v * v;
    REQUIRE(std::fabs(std::sqrt(n) - 1.0) < 1e-5);
    // embed_batch(Vec<String>): vocab.txt built in memory, ids 0..: [PAD] [UNK] [CLS] [SEP] fn main ( ) ##s
    const std::string vocab = "[PAD]\n[UNK]\n[CLS]\n[SEP]\nfn\nmain\n(\n)\n##s\n";
    Tokenizer tok(vocab.data(), vocab.size());
    std::vector<int32_t> tid, tmask;
    REQUIRE(tok.encode_batch({"fn mains()", "FN"}, tid, tmask) == 7);
    REQUIRE((tid == std::vector<int32_t>{2, 4, 5, 8, 6, 7, 3, 2, 4, 3, 0, 0, 0, 0}));
    emb.attach_tokenizer(&tok);
    auto te = emb.embed_batch(std::vector<std::string>{"fn mains()", "FN"});
    auto ti = emb.embed_batch(tid, tmask, 2, 7);
    REQUIRE(te.size() == 2);
    for (size_t i = 0; i < 2; ++i)
        for (size_t j = 0; j < 384; ++j) REQUIRE(te[i][j] == ti[i][j]);
    auto one = emb.embed_one("FN");
    for (size_t j = 0; j < 384; ++j) REQUIRE(std::fabs(one[j] - te[1][j]) < 1e-5);  // padding-invariant
    std::printf("host mirror ok\n");
    return 0;
}
