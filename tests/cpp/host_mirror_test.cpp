// Exercises the C++ host mirror (codesearch_amd/host/codesearch_gpu.hpp) the way the
// reference's own unit test does (/root/reference/src/vectordb/store.rs:846-893), then a tiny
// encoder call.  Built by tests/test_cpp_host.py with g++ against libcsgpu.so; run on the GPU.
#include <cmath>
#include <cstdio>

#include <cstring>

#include "../../codesearch_amd/host/codesearch_callers.hpp"
#include "../../codesearch_amd/host/codesearch_gpu.hpp"

#define REQUIRE(c)                                                          \
    do {                                                                    \
        if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } \
    } while (0)

// The reference's own known answers for the callers either side of the hot path (no GPU involved):
// batch.rs:238-314 (stats, clean_docstring, prepare_text), rerank/mod.rs:273-337 (fusion), search/mod.rs:494-611.
static int cpu_checks() {
    using namespace cs;
    EmbeddingStats st;  // batch.rs:238-251
    st.total_chunks = 100; st.embedded_chunks = 80; st.cached_chunks = 20; st.total_time_ms = 1000;
    REQUIRE(st.cache_hit_rate() == 0.2 && st.success_rate() == 0.8 && st.chunks_per_second() == 80.0);
    // batch.rs:253-274
    REQUIRE(clean_docstring("/// This is a doc comment\n/// with multiple lines") == "This is a doc comment with multiple lines");
    REQUIRE(clean_docstring("\"\"\"This is a Python docstring\"\"\"") == "\"\"This is a Python docstring\"\"");
    REQUIRE(clean_docstring("/**\n * JSDoc comment\n * with multiple lines\n */") == "JSDoc comment with multiple lines");
    REQUIRE(clean_docstring("\"This is a quoted docstring\"") == "This is a quoted docstring");
    REQUIRE(clean_docstring("") == "" && clean_docstring("//! inner\r\n// plain") == "inner plain");
    REQUIRE(clean_docstring("  \xc2\xa0/// nbsp-indented\xe3\x80\x80") == "nbsp-indented");  // Unicode trim
    // batch.rs:276-314
    Chunk c;
    c.content = "fn test() { println!(\"test\"); }"; c.kind = "Function"; c.path = "test.rs";
    c.context = {"File: test.rs", "Function: test"};
    c.signature = "fn test()"; c.docstring = "/// Test function";
    REQUIRE(prepare_text(c) == "Context: File: test.rs > Function: test\nSignature: fn test()\nName: test\n"
                               "Documentation: Test function\nCode:\nfn test() { println!(\"test\"); }");
    Chunk g2;
    g2.content = "x"; g2.signature = "fn sort<T: Ord>(items: Vec<T>) -> Vec<T>";
    REQUIRE(prepare_text(g2).find("Name: sort\n") != std::string::npos);
    Chunk bare;
    bare.content = "body";
    REQUIRE(prepare_text(bare) == "Code:\nbody");
    bare.signature = "lonely";
    REQUIRE(prepare_text(bare).find("Name:") == std::string::npos);
    // BatchEmbedder slices of 32 (batch.rs:94) over any embedder with embed_batch(vector<string>)
    struct Fake {
        std::vector<size_t> calls;
        std::vector<std::vector<float>> embed_batch(const std::vector<std::string>& t) {
            calls.push_back(t.size());
            std::vector<std::vector<float>> o;
            for (const auto& x : t) o.push_back({(float)x.size()});
            return o;
        }
        std::vector<float> embed_one(const std::string& t) { return embed_batch({t})[0]; }
        size_t dimensions() const { return 1; }
    } fake;
    std::vector<Chunk> chunks(70);
    for (size_t i = 0; i < chunks.size(); ++i) chunks[i].content = "content " + std::to_string(i);
    BatchEmbedder<Fake> be(fake);
    auto ecs = be.embed_chunks(chunks);
    REQUIRE((fake.calls == std::vector<size_t>{32, 32, 6}) && ecs.size() == 70);
    REQUIRE(ecs[69].embedding[0] == (float)std::strlen("Code:\ncontent 69") && ecs[5].chunk.content == "content 5");

    auto vres = [](uint32_t id, float score) { SearchResult r; r.id = id; r.score = score; r.distance = 1.0f - score; return r; };
    // search/mod.rs:494-611
    REQUIRE(retrieval_limit(25, true, false) == 25 && retrieval_limit(25, false, true) == 100);
    REQUIRE(retrieval_limit(50, false, true) == 150 && retrieval_limit(25, false, false) == 200 && retrieval_limit(60, false, false) == 300);
    auto merged = merge_variant_results({{vres(1, 0.9f), vres(2, 0.8f), vres(3, 0.7f)}, {vres(2, 0.95f), vres(4, 0.6f), vres(1, 0.85f)}}, 3);
    REQUIRE(merged.size() == 3 && merged[0].id == 2 && merged[0].score == 0.95f && merged[1].id == 1 && merged[2].id == 3);
    std::vector<SearchResult> hi;
    for (uint32_t i = 0; i < 6; ++i) hi.push_back(vres(i, 0.9f));
    REQUIRE(should_use_vector_only(hi, false) && !should_use_vector_only(hi, true) && !should_use_vector_only({}, false));
    hi[4] = vres(9, 0.8f);  // distance 0.2 inside the top five
    REQUIRE(!should_use_vector_only(hi, false));
    // rerank/mod.rs:273-306
    auto fused = rrf_fusion({vres(1, 0.9f), vres(2, 0.8f), vres(3, 0.7f)}, {{2, 10.0f}, {1, 8.0f}, {4, 6.0f}}, 20.0f);
    REQUIRE(fused.size() == 4);
    std::map<uint32_t, FusedResult> by;
    for (const auto& f : fused) by[f.chunk_id] = f;
    REQUIRE(by[1].vector_rank && by[1].fts_rank && by[2].vector_rank && by[2].fts_rank);
    REQUIRE(!by[4].vector_rank && by[4].fts_rank);
    REQUIRE((fused[0].chunk_id == 1 || fused[0].chunk_id == 2) && (fused[1].chunk_id == 1 || fused[1].chunk_id == 2));
    for (size_t i = 0; i + 1 < fused.size(); ++i) REQUIRE(fused[i].rrf_score >= fused[i + 1].rrf_score);
    // rerank/mod.rs:308-324 (f32 arithmetic)
    auto one = rrf_fusion({vres(1, 0.9f)}, {{1, 10.0f}}, 20.0f);
    REQUIRE(one.size() == 1 && one[0].rrf_score == 1.0f / 21.0f + 1.0f / 21.0f);
    // rerank/mod.rs:326-336
    auto vo = vector_only({vres(1, 0.9f), vres(2, 0.8f)});
    REQUIRE(vo.size() == 2 && vo[0].chunk_id == 1 && vo[0].rrf_score == 0.9f && !vo[0].fts_score);
    // three-way (rerank/mod.rs:139-241): exact rank 1 with k = 5 outweighs a rank-1 vector hit
    auto tw = rrf_fusion_with_exact({vres(1, 0.9f), vres(2, 0.8f)}, {{2, 4.0f}, {3, 2.0f}}, {{3, 9.0f}});
    REQUIRE(tw[0].chunk_id == 3 && std::fabs(tw[0].rrf_score - (1.0f / 6 + 1.0f / 22)) < 1e-6f);
    REQUIRE(*tw[0].fts_score == 5.5f && *tw[0].fts_rank == 2);
    std::printf("host callers ok\n");
    return 0;
}

int main(int argc, char** argv) {
    using namespace cs;
    if (int rc = cpu_checks()) return rc;
    if (argc > 1 && std::string(argv[1]) == "cpu") return 0;
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

    {   // the same store over three shards inside this process (cs_shards_*), and the variant merge on the device
        VectorStore one("one.db", 8), three("three.db", 8, std::vector<int32_t>{0, 0, 0}, /*rows_per_stripe=*/2);
        std::vector<EmbeddedChunk> cs;
        for (int i = 0; i < 11; ++i) {
            EmbeddedChunk c;
            c.chunk.content = "fn f" + std::to_string(i) + "() {}"; c.chunk.kind = "Function"; c.chunk.path = "m" + std::to_string(i % 3) + ".rs";
            c.embedding.assign(8, 0.05f * (float)((i * 7) % 5));
            c.embedding[i % 8] = 1.0f; c.embedding[(i * 3 + 1) % 8] += 0.4f;
            cs.push_back(c);
        }
        auto ids1 = one.insert_chunks_with_ids(cs), ids3 = three.insert_chunks_with_ids(cs);
        REQUIRE(ids1 == ids3 && ids3.front() == 0 && ids3.back() == 10 && three.sharded());
        REQUIRE(one.delete_chunks({4}) == 1 && three.delete_chunks({4}) == 1);
        one.build_index(); three.build_index();
        std::vector<std::vector<float>> variants = {cs[2].embedding, cs[9].embedding, cs[2].embedding};
        variants[2][5] += 0.3f;
        auto r1 = one.search_batch(variants, 5), r3 = three.search_batch(variants, 5);
        REQUIRE(r1.size() == 3 && r3.size() == 3);
        for (size_t v = 0; v < 3; ++v) {
            REQUIRE(r1[v].size() == r3[v].size() && r1[v].size() == 5);
            for (size_t j = 0; j < r1[v].size(); ++j) REQUIRE(r1[v][j].id == r3[v][j].id && r1[v][j].score == r3[v][j].score);
        }
        REQUIRE(r1[0][0].id == 2 && r1[1][0].id == 9);
        bool conf1 = true, conf3 = true;
        auto m1 = one.search_variants(variants, 5, &conf1), m3 = three.search_variants(variants, 5, &conf3);
        auto want = merge_variant_results(r1, 5);  // codesearch_callers.hpp: the host statement of mod.rs:513-590
        REQUIRE(m1.size() == want.size() && m3.size() == want.size() && conf1 == conf3);
        REQUIRE(conf1 == should_use_vector_only(want, false));
        for (size_t j = 0; j < want.size(); ++j) {
            REQUIRE(m1[j].score == want[j].score && m3[j].score == want[j].score);
            REQUIRE(m1[j].id == m3[j].id);
        }
    }

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
    {   // BatchEmbedder on the queue (batch.rs:84-115 with its slices of 32): same vectors as one embed_batch call
        std::vector<Chunk> chunks;
        for (int i = 0; i < 75; ++i) {
            Chunk c;
            c.content = (i % 3 ? "fn mains() fn" : "main ( )"); c.kind = "Function"; c.path = "q.rs";
            for (int r = 0; r < i % 5; ++r) c.content += " fn main";
            chunks.push_back(c);
        }
        BatchEmbedder<FastEmbedder> be(emb);
        auto got = be.embed_chunks(chunks);
        std::vector<std::string> texts;
        for (const auto& c : chunks) texts.push_back(prepare_text(c));
        auto want = emb.embed_batch(texts);
        REQUIRE(got.size() == 75 && want.size() == 75);
        for (size_t i = 0; i < 75; ++i) {
            REQUIRE(got[i].chunk.content == chunks[i].content);
            for (size_t j = 0; j < 384; ++j) REQUIRE(std::fabs(got[i].embedding[j] - want[i][j]) < 2e-6);
        }
        // replicas + sharded store: embed and append in one call, ids contiguous, rows searchable
        EmbedderReplicas reps(cfg, nullptr, 7, std::vector<int32_t>{0, 0});
        reps.attach_tokenizer(&tok);
        VectorStore sharded("s.db", 384, std::vector<int32_t>{0, 0, 0, 0}, /*rows_per_stripe=*/8);
        auto ids_a = reps.index_texts(sharded.shards_handle(), std::vector<std::string>(texts.begin(), texts.begin() + 30));
        auto ids_b = reps.index_texts(sharded.shards_handle(), std::vector<std::string>(texts.begin() + 30, texts.end()));
        REQUIRE(ids_a.size() == 30 && ids_a.front() == 0 && ids_a.back() == 29 && ids_b.front() == 30 && ids_b.back() == 74);
        ids_a.insert(ids_a.end(), ids_b.begin(), ids_b.end());
        sharded.put_metadata(ids_a, chunks);
        sharded.build_index();
        auto hit = sharded.search(want[41], 3);
        REQUIRE(!hit.empty() && hit[0].score > 0.9999f);
        REQUIRE(hit[0].meta.content == chunks[hit[0].id].content);   // several chunks share a text: any of them may lead
        auto both = reps.embed_batch(texts);
        for (size_t j = 0; j < 384; ++j) REQUIRE(std::fabs(both[41][j] - want[41][j]) < 2e-6);
    }
    std::printf("host mirror ok\n");
    return 0;
}
