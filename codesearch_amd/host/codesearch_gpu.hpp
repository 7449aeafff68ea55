// codesearch_gpu.hpp — header-only C++ mirror of the reference's VectorStore / FastEmbedder
// over the C ABI (include/codesearch_gpu.h), for compiled callers.  Same method names,
// argument meaning and error texts as /root/reference/src/vectordb/store.rs:94-750 and
// src/embed/embedder.rs:201-322; `anyhow::Result<T>` becomes a thrown cs::Error carrying
// the cs_status and the reference-worded message.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/codesearch_gpu.h"

namespace cs {

struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string& m) : std::runtime_error(m), code(c) {}
};

inline void check(int32_t status) {
    if (status != CS_OK) throw Error(status, cs_last_error());
}

// src/chunker/mod.rs:22-62 (fields that cross into the hot path / its metadata)
struct Chunk {
    std::string content;
    size_t start_line = 0, end_line = 0;
    std::string kind;  // ChunkKind Debug name
    std::string path;
    std::vector<std::string> context;
    std::optional<std::string> signature, docstring;
    std::string hash;
    std::optional<std::string> context_prev, context_next;
};

// src/embed/batch.rs:47-57
struct EmbeddedChunk {
    Chunk chunk;
    std::vector<float> embedding;
};

// store.rs:19-85 (searchable_text omitted: FTS is out of scope)
struct ChunkMetadata {
    std::string content, path;
    size_t start_line = 0, end_line = 0;
    std::string kind;
    std::optional<std::string> signature, docstring, context;
    std::string hash;
    std::optional<std::string> context_prev, context_next;

    static ChunkMetadata from_embedded_chunk(const EmbeddedChunk& ec) {
        ChunkMetadata m;
        const Chunk& c = ec.chunk;
        m.content = c.content; m.path = c.path; m.start_line = c.start_line; m.end_line = c.end_line;
        m.kind = c.kind; m.signature = c.signature; m.docstring = c.docstring; m.hash = c.hash;
        m.context_prev = c.context_prev; m.context_next = c.context_next;
        if (!c.context.empty()) {
            std::string j;
            for (size_t i = 0; i < c.context.size(); ++i) j += (i ? " > " : "") + c.context[i];
            m.context = j;
        }
        return m;
    }
};

// store.rs:753-772
struct SearchResult {
    uint32_t id = 0;
    ChunkMetadata meta;
    float distance = 0.f;  // arroy Cosine: (1 - cos) / 2
    float score = 0.f;     // 1 - distance (store.rs:478)
};

// store.rs:784-792
struct StoreStats {
    size_t total_chunks = 0, total_files = 0;
    bool indexed = false;
    size_t dimensions = 0;
    uint32_t max_chunk_id = 0;
};

class VectorStore {
  public:
    // VectorStore::new(db_path, dimensions) — store.rs:110-176 (db_path unused: nothing persists)
    VectorStore(const std::string& /*db_path*/, size_t dimensions, int device = 0, uint64_t capacity = 0,
                uint32_t id_base = 0)
        : dimensions_(dimensions) {
        check(cs_index_create((uint32_t)dimensions, capacity, device, id_base, &h_));
    }
    // The same store row-sharded over several GPUs of the node inside this process (cs_shards_*):
    // ids stay contiguous, rows are dealt in stripes of rows_per_stripe ids, a search scans every shard and merges
    // on devices[0].
    VectorStore(const std::string& /*db_path*/, size_t dimensions, const std::vector<int32_t>& devices,
                uint64_t rows_per_stripe = 65536, uint64_t capacity = 0)
        : dimensions_(dimensions) {
        check(cs_shards_create((uint32_t)dimensions, (uint32_t)devices.size(), devices.data(), rows_per_stripe, capacity, &sh_));
    }
    ~VectorStore() { if (sh_) cs_shards_destroy(sh_); else cs_index_destroy(h_); }
    VectorStore(const VectorStore&) = delete;
    VectorStore& operator=(const VectorStore&) = delete;
    bool sharded() const { return sh_ != nullptr; }
    cs_shards* shards_handle() const { return sh_; }
    // metadata of chunks whose vectors were appended by EmbedderReplicas::index_texts (ids assigned there)
    void put_metadata(const std::vector<uint32_t>& ids, const std::vector<Chunk>& chunks) {
        for (size_t i = 0; i < ids.size() && i < chunks.size(); ++i) {
            EmbeddedChunk ec;
            ec.chunk = chunks[i];
            meta_[ids[i]] = ChunkMetadata::from_embedded_chunk(ec);
        }
    }

    // store.rs:618-686
    std::vector<uint32_t> insert_chunks_with_ids(const std::vector<EmbeddedChunk>& chunks) {
        if (chunks.empty()) return {};
        std::vector<float> rows;
        rows.reserve(chunks.size() * dimensions_);
        for (const auto& c : chunks) {
            if (c.embedding.size() != dimensions_)  // store.rs:666-672
                throw Error(CS_ERR_DIM_MISMATCH, "Embedding dimension mismatch: expected " +
                                                     std::to_string(dimensions_) + ", got " +
                                                     std::to_string(c.embedding.size()));
            rows.insert(rows.end(), c.embedding.begin(), c.embedding.end());
        }
        std::vector<uint32_t> ids(chunks.size());
        check(sh_ ? cs_shards_add(sh_, rows.data(), chunks.size(), (uint32_t)dimensions_, ids.data())
                  : cs_index_add(h_, rows.data(), chunks.size(), (uint32_t)dimensions_, ids.data()));
        for (size_t i = 0; i < chunks.size(); ++i) meta_[ids[i]] = ChunkMetadata::from_embedded_chunk(chunks[i]);
        return ids;
    }
    // store.rs:334-379
    size_t insert_chunks(const std::vector<EmbeddedChunk>& chunks) { return insert_chunks_with_ids(chunks).size(); }
    // store.rs:386-430
    void build_index() { check(sh_ ? cs_shards_build(sh_) : cs_index_build(h_)); }
    // store.rs:548-610
    size_t delete_chunks(const std::vector<uint32_t>& ids) {
        uint64_t removed = 0;
        check(sh_ ? cs_shards_remove(sh_, ids.data(), ids.size(), &removed) : cs_index_remove(h_, ids.data(), ids.size(), &removed));
        for (uint32_t id : ids) meta_.erase(id);
        return (size_t)removed;
    }
    // store.rs:690-707
    void clear() { check(sh_ ? cs_shards_clear(sh_) : cs_index_clear(h_)); meta_.clear(); }
    // store.rs:745
    bool is_indexed() const { return (sh_ ? cs_shards_is_built(sh_) : cs_index_is_built(h_)) != 0; }

    // store.rs:431-486
    std::vector<SearchResult> search(const std::vector<float>& query_embedding, size_t limit) const {
        auto all = search_batch({query_embedding}, limit);
        return all.empty() ? std::vector<SearchResult>{} : all[0];
    }
    // all query variants in one call (src/search/mod.rs:508-511)
    std::vector<std::vector<SearchResult>> search_batch(const std::vector<std::vector<float>>& queries,
                                                        size_t limit) const {
        const size_t nq = queries.size();
        if (nq == 0) return {};
        const size_t qdim = queries[0].size();
        std::vector<float> q;
        for (const auto& v : queries) {
            if (v.size() != qdim) throw Error(CS_ERR_BAD_ARG, "queries of unequal length");
            q.insert(q.end(), v.begin(), v.end());
        }
        std::vector<float> cos(nq * limit);
        std::vector<uint32_t> ids(nq * limit), counts(nq);
        check(sh_ ? cs_shards_search(sh_, q.data(), (uint32_t)nq, (uint32_t)qdim, (uint32_t)limit, cos.data(), ids.data(),
                                     counts.data())
                  : cs_index_search(h_, q.data(), (uint32_t)nq, (uint32_t)qdim, (uint32_t)limit, cos.data(), ids.data(),
                                    counts.data()));
        std::vector<std::vector<SearchResult>> out(nq);
        for (size_t i = 0; i < nq; ++i)
            for (uint32_t j = 0; j < counts[i]; ++j) {
                auto it = meta_.find(ids[i * limit + j]);
                if (it == meta_.end()) continue;  // store.rs:465
                SearchResult r;
                r.id = ids[i * limit + j];
                r.meta = it->second;
                r.distance = cs_cos_to_distance(cos[i * limit + j]);
                r.score = 1.0f - r.distance;
                out[i].push_back(std::move(r));
            }
        return out;
    }
    // search::search's vector leg in one call (src/search/mod.rs:508-611): every variant searched for `limit` rows, a
    // chunk found by several variants keeps its best score, the best `limit` distinct chunks best-first — merged on the
    // device; *high_confidence = the top five all have distance < 0.15 (the reference then skips its FTS leg).
    std::vector<SearchResult> search_variants(const std::vector<std::vector<float>>& variants, size_t limit,
                                              bool* high_confidence = nullptr) const {
        const size_t nq = variants.size();
        if (nq == 0) return {};
        const size_t qdim = variants[0].size();
        std::vector<float> q;
        for (const auto& v : variants) {
            if (v.size() != qdim) throw Error(CS_ERR_BAD_ARG, "queries of unequal length");
            q.insert(q.end(), v.begin(), v.end());
        }
        std::vector<float> cos(limit);
        std::vector<uint32_t> ids(limit);
        uint32_t count = 0;
        int32_t flag = 0;
        check(sh_ ? cs_shards_search_variants(sh_, q.data(), (uint32_t)nq, (uint32_t)qdim, (uint32_t)limit, cos.data(),
                                              ids.data(), &count, &flag)
                  : cs_index_search_variants(h_, q.data(), (uint32_t)nq, (uint32_t)qdim, (uint32_t)limit, cos.data(),
                                             ids.data(), &count, &flag));
        if (high_confidence) *high_confidence = flag != 0;
        std::vector<SearchResult> out;
        for (uint32_t j = 0; j < count; ++j) {
            auto it = meta_.find(ids[j]);
            if (it == meta_.end()) continue;
            SearchResult r;
            r.id = ids[j];
            r.meta = it->second;
            r.distance = cs_cos_to_distance(cos[j]);
            r.score = 1.0f - r.distance;
            out.push_back(std::move(r));
        }
        return out;
    }
    std::optional<ChunkMetadata> get_chunk(uint32_t id) const {  // store.rs:709-713
        auto it = meta_.find(id);
        return it == meta_.end() ? std::nullopt : std::optional<ChunkMetadata>(it->second);
    }
    std::map<std::string, std::vector<uint32_t>> get_chunks_by_file() const {  // store.rs:529-543
        std::map<std::string, std::vector<uint32_t>> m;
        for (const auto& kv : meta_) m[kv.second.path].push_back(kv.first);
        return m;
    }
    StoreStats stats() const {  // store.rs:501-523
        StoreStats s;
        s.total_chunks = meta_.size();
        s.total_files = get_chunks_by_file().size();
        s.indexed = is_indexed();
        s.dimensions = dimensions_;
        s.max_chunk_id = meta_.empty() ? 0 : meta_.rbegin()->first;
        return s;
    }
    size_t dimensions() const { return dimensions_; }
    cs_index* handle() const { return h_; }
    // searches of >= n queries take the f16 filter + exact f32 refine path (default 2; 1 = always)
    void set_filter_min_queries(uint32_t n) { if (h_) check(cs_index_set_filter_min_queries(h_, n)); }
    void set_single_query_route(int32_t route) { if (h_) check(cs_index_set_single_query_route(h_, route)); }

  private:
    cs_index* h_ = nullptr;
    cs_shards* sh_ = nullptr;
    size_t dimensions_;
    std::map<uint32_t, ChunkMetadata> meta_;
};

// The tokenizer fastembed builds from the model's tokenizer.json (tokenizers 0.22.2): vocab.txt in,
// [CLS] ... [SEP] ids out (csrc/tokenizer.cpp).
class Tokenizer {
  public:
    explicit Tokenizer(const std::string& vocab_path, bool lowercase = true, uint32_t max_length = 512) {
        check(cs_tokenizer_create_from_file(vocab_path.c_str(), lowercase ? 1 : 0, max_length, &h_));
    }
    Tokenizer(const char* vocab_txt, size_t bytes, bool lowercase = true, uint32_t max_length = 512) {
        check(cs_tokenizer_create(vocab_txt, bytes, lowercase ? 1 : 0, max_length, &h_));
    }
    ~Tokenizer() { cs_tokenizer_destroy(h_); }
    Tokenizer(const Tokenizer&) = delete;
    Tokenizer& operator=(const Tokenizer&) = delete;

    // encode_batch: ids/mask [n, L] row-major, L = the batch's longest sequence
    size_t encode_batch(const std::vector<std::string>& texts, std::vector<int32_t>& ids, std::vector<int32_t>& mask,
                        uint32_t max_length = 0) const {
        std::string blob;
        std::vector<uint64_t> off;
        pack(texts, blob, off);
        uint32_t L = 0;
        check(cs_tokenizer_encode_batch(h_, blob.data(), off.data(), (uint32_t)texts.size(), max_length, nullptr,
                                        nullptr, 0, &L));
        ids.assign(texts.size() * L, 0);
        mask.assign(texts.size() * L, 0);
        if (!texts.empty())
            check(cs_tokenizer_encode_batch(h_, blob.data(), off.data(), (uint32_t)texts.size(), max_length,
                                            ids.data(), mask.data(), L, nullptr));
        return L;
    }
    int32_t token_to_id(const std::string& token) const { return cs_tokenizer_token_to_id(h_, token.c_str()); }
    size_t vocab_size() const { return cs_tokenizer_vocab_size(h_); }
    const cs_tokenizer* handle() const { return h_; }

    static void pack(const std::vector<std::string>& texts, std::string& blob, std::vector<uint64_t>& off) {
        off.assign(1, 0);
        for (const auto& t : texts) {
            blob += t;
            off.push_back(blob.size());
        }
    }

  private:
    cs_tokenizer* h_ = nullptr;
};

// FastEmbedder: from strings when a Tokenizer is attached (embed_batch(Vec<String>)), or from token ids.
class FastEmbedder {
  public:
    // with_cache_dir — embedder.rs:218-245; params == nullptr => synthetic weights from seed
    FastEmbedder(const cs_bert_config& cfg, const float* params, uint64_t seed = 0, int device = 0) {
        check(cs_embedder_create(&cfg, params, seed, device, &h_));
    }
    // with_cache_dir from a HF snapshot directory (config.json + model.safetensors)
    explicit FastEmbedder(const std::string& model_dir, cs_pooling pooling = CS_POOL_CLS, int device = 0) {
        check(cs_embedder_create_from_dir(model_dir.c_str(), (int32_t)pooling, device, &h_));
    }
    ~FastEmbedder() { cs_embedder_destroy(h_); }
    FastEmbedder(const FastEmbedder&) = delete;
    FastEmbedder& operator=(const FastEmbedder&) = delete;

    // embed_batch — embedder.rs:249-263 (batch 0 = CODESEARCH_BATCH_SIZE or 256/128/64)
    std::vector<std::vector<float>> embed_batch(const std::vector<int32_t>& ids, const std::vector<int32_t>& mask,
                                                size_t n, size_t seq_len,
                                                const volatile int32_t* shutdown = nullptr) {
        return embed_batch_chunked(ids, mask, n, seq_len, 0, shutdown);
    }
    // embed_batch_chunked — embedder.rs:266-295
    std::vector<std::vector<float>> embed_batch_chunked(const std::vector<int32_t>& ids,
                                                        const std::vector<int32_t>& mask, size_t n,
                                                        size_t seq_len, size_t batch_size,
                                                        const volatile int32_t* shutdown = nullptr) {
        const size_t d = dimensions();
        std::vector<float> flat(n * d);
        check(cs_embedder_embed_ids(h_, ids.data(), mask.data(), n, (uint32_t)seq_len, (uint32_t)batch_size,
                                    flat.data(), shutdown));
        std::vector<std::vector<float>> out(n);
        for (size_t i = 0; i < n; ++i) out[i].assign(flat.begin() + i * d, flat.begin() + (i + 1) * d);
        return out;
    }
    // embed_batch(texts) — embedder.rs:249-263, from strings
    void attach_tokenizer(const Tokenizer* t) { tok_ = t; }
    std::vector<std::vector<float>> embed_batch(const std::vector<std::string>& texts,
                                                const volatile int32_t* shutdown = nullptr) {
        return embed_batch_chunked(texts, 0, shutdown);
    }
    std::vector<std::vector<float>> embed_batch_chunked(const std::vector<std::string>& texts, size_t batch_size,
                                                        const volatile int32_t* shutdown = nullptr) {
        if (!tok_) throw Error(CS_ERR_UNSUPPORTED, "Failed to generate embeddings: no tokenizer attached");
        std::string blob;
        std::vector<uint64_t> off;
        Tokenizer::pack(texts, blob, off);
        const size_t d = dimensions(), n = texts.size();
        std::vector<float> flat(n * d);
        check(cs_embedder_embed_texts(h_, tok_->handle(), blob.data(), off.data(), n, (uint32_t)batch_size,
                                      flat.data(), shutdown));
        std::vector<std::vector<float>> out(n);
        for (size_t i = 0; i < n; ++i) out[i].assign(flat.begin() + i * d, flat.begin() + (i + 1) * d);
        return out;
    }
    // embed_one — embedder.rs:298-304
    std::vector<float> embed_one(const std::string& text) {
        auto r = embed_batch(std::vector<std::string>{text});
        if (r.empty()) throw Error(CS_ERR_BAD_ARG, "No embedding generated");
        return r[0];
    }
    // The reference's call shape on a queue (src/embed/batch.rs:84-115: slices of 32 under a mutex): submit() tokenises
    // and queues a slice, wait() returns its rows; the first wait embeds everything queued as full device batches.
    // Safe from several threads on one embedder.
    uint64_t submit(const std::vector<std::string>& texts) {
        if (!tok_) throw Error(CS_ERR_UNSUPPORTED, "Failed to generate embeddings: no tokenizer attached");
        std::string blob;
        std::vector<uint64_t> off;
        Tokenizer::pack(texts, blob, off);
        uint64_t ticket = 0;
        check(cs_embedder_submit_texts(h_, tok_->handle(), blob.data(), off.data(), texts.size(), &ticket));
        return ticket;
    }
    std::vector<std::vector<float>> wait(uint64_t ticket, size_t n, const volatile int32_t* shutdown = nullptr) {
        const size_t d = dimensions();
        std::vector<float> flat(n * d + 1);
        check(cs_embedder_wait(h_, ticket, flat.data(), shutdown));
        std::vector<std::vector<float>> out(n);
        for (size_t i = 0; i < n; ++i) out[i].assign(flat.begin() + i * d, flat.begin() + (i + 1) * d);
        return out;
    }
    void discard(uint64_t ticket) { check(cs_embedder_discard(h_, ticket)); }
    size_t dimensions() const { return cs_embedder_dim(h_); }  // embedder.rs:307
    cs_embedder* handle() const { return h_; }
    // CS_GEMM_SPLIT_F16 (default), CS_GEMM_F32 (exact-f32 MFMA) or, for a dynamically quantised model (a *Q registry
    // entry's directory: what it comes up in), CS_GEMM_Q8_DYNAMIC — include/codesearch_gpu.h
    void set_gemm_mode(cs_gemm_mode mode) { check(cs_embedder_set_gemm_mode(h_, (int32_t)mode)); }
    cs_gemm_mode gemm_mode() const { return (cs_gemm_mode)cs_embedder_gemm_mode(h_); }

  private:
    cs_embedder* h_ = nullptr;
    const Tokenizer* tok_ = nullptr;
};

// One encoder replica per GPU inside this process and the index loop of src/index/mod.rs:626-762 over a sharded
// VectorStore: every replica embeds the chunks whose ids fall on the shards of its own GPU, rows are appended without
// leaving HBM, no collective (cs_embedders_*).
class EmbedderReplicas {
  public:
    EmbedderReplicas(const cs_bert_config& cfg, const float* params, uint64_t seed, const std::vector<int32_t>& devices) {
        check(cs_embedders_create(&cfg, params, seed, devices.data(), (uint32_t)devices.size(), &h_));
    }
    EmbedderReplicas(const std::string& model_dir, cs_pooling pooling, const std::vector<int32_t>& devices) {
        check(cs_embedders_create_from_dir(model_dir.c_str(), (int32_t)pooling, devices.data(), (uint32_t)devices.size(), &h_));
    }
    ~EmbedderReplicas() { cs_embedders_destroy(h_); }
    EmbedderReplicas(const EmbedderReplicas&) = delete;
    EmbedderReplicas& operator=(const EmbedderReplicas&) = delete;
    void attach_tokenizer(const Tokenizer* t) { tok_ = t; }
    size_t dimensions() const { return cs_embedders_dim(h_); }
    size_t size() const { return cs_embedders_count(h_); }
    // embed_batch over all replicas: row i = text i
    std::vector<std::vector<float>> embed_batch(const std::vector<std::string>& texts, const volatile int32_t* shutdown = nullptr) {
        if (!tok_) throw Error(CS_ERR_UNSUPPORTED, "Failed to generate embeddings: no tokenizer attached");
        std::string blob;
        std::vector<uint64_t> off;
        Tokenizer::pack(texts, blob, off);
        const size_t d = dimensions(), n = texts.size();
        std::vector<float> flat(n * d + 1);
        check(cs_embedders_embed_texts(h_, tok_->handle(), blob.data(), off.data(), n, 0, flat.data(), shutdown));
        std::vector<std::vector<float>> out(n);
        for (size_t i = 0; i < n; ++i) out[i].assign(flat.begin() + i * d, flat.begin() + (i + 1) * d);
        return out;
    }
    // embed_chunks + insert_chunks_with_ids in one call on a SHARDED store -> the assigned ids (metadata: the caller's)
    std::vector<uint32_t> index_texts(cs_shards* store, const std::vector<std::string>& texts,
                                      const volatile int32_t* shutdown = nullptr) {
        if (!tok_) throw Error(CS_ERR_UNSUPPORTED, "Failed to generate embeddings: no tokenizer attached");
        std::string blob;
        std::vector<uint64_t> off;
        Tokenizer::pack(texts, blob, off);
        std::vector<uint32_t> ids(texts.size());
        check(cs_embedders_index_texts(h_, tok_->handle(), store, blob.data(), off.data(), texts.size(), 0, ids.data(), shutdown));
        return ids;
    }
    cs_embedders* handle() const { return h_; }

  private:
    cs_embedders* h_ = nullptr;
    const Tokenizer* tok_ = nullptr;
};

}  // namespace cs
