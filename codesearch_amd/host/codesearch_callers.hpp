// codesearch_callers.hpp — header-only C++ mirror of the callers either side of the hot path, for
// compiled hosts (the reference is compiled code): the text the embedder is fed and what happens to
// search results right after the scan.  No GPU work in here; the arithmetic is the C ABI's.
//
//   prepare_text / clean_docstring / BatchEmbedder / EmbeddingStats   /root/reference/src/embed/batch.rs:12-231
//   retrieval_limit / merge_variant_results / should_use_vector_only   src/search/mod.rs:494-611
//   rrf_fusion / vector_only / rrf_fusion_with_exact                   src/rerank/mod.rs:14-241
//
// Same names, argument meaning and known answers as the reference's own unit tests
// (tests/cpp/host_mirror_test.cpp re-expresses batch.rs:238-314 and rerank/mod.rs:273-337).
#pragma once

#include <algorithm>
#include <cstdint>
#include <map>
#include <optional>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "codesearch_gpu.hpp"

namespace cs {

namespace text_detail {

// Unicode White_Space (what Rust's char::is_whitespace / str::trim / split_whitespace use), on UTF-8.
// Returns the byte length of the whitespace scalar at s[i], 0 if there is none.
inline size_t ws_len(const std::string& s, size_t i) {
    const unsigned char c = (unsigned char)s[i];
    if (c == ' ' || (c >= 0x09 && c <= 0x0D)) return 1;
    if (c == 0xC2 && i + 1 < s.size()) {
        const unsigned char d = (unsigned char)s[i + 1];
        return (d == 0x85 || d == 0xA0) ? 2 : 0;
    }
    if (i + 2 < s.size()) {
        const unsigned char d = (unsigned char)s[i + 1], e = (unsigned char)s[i + 2];
        if (c == 0xE1 && d == 0x9A && e == 0x80) return 3;                                  // U+1680
        if (c == 0xE2 && d == 0x80 && ((e >= 0x80 && e <= 0x8A) || e == 0xA8 || e == 0xA9 || e == 0xAF)) return 3;
        if (c == 0xE2 && d == 0x81 && e == 0x9F) return 3;                                  // U+205F
        if (c == 0xE3 && d == 0x80 && e == 0x80) return 3;                                  // U+3000
    }
    return 0;
}

inline std::string trim(const std::string& s) {
    size_t lo = 0, hi = s.size();
    for (size_t n; lo < hi && (n = ws_len(s, lo)) != 0;) lo += n;
    for (;;) {  // the last scalar starts at most 3 bytes back
        bool cut = false;
        for (size_t back = 1; back <= 3 && back <= hi - lo; ++back)
            if (ws_len(s, hi - back) == back) { hi -= back; cut = true; break; }
        if (!cut) break;
    }
    return s.substr(lo, hi - lo);
}

inline std::vector<std::string> split_whitespace(const std::string& s) {
    std::vector<std::string> out;
    size_t i = 0, start = std::string::npos;
    while (i < s.size()) {
        const size_t n = ws_len(s, i);
        if (n) {
            if (start != std::string::npos) { out.push_back(s.substr(start, i - start)); start = std::string::npos; }
            i += n;
        } else {
            if (start == std::string::npos) start = i;
            ++i;
        }
    }
    if (start != std::string::npos) out.push_back(s.substr(start));
    return out;
}

// str::lines(): split on '\n', one trailing '\r' per line dropped, no final empty line
inline std::vector<std::string> lines(const std::string& s) {
    std::vector<std::string> out;
    size_t lo = 0;
    while (lo < s.size()) {
        size_t nl = s.find('\n', lo);
        if (nl == std::string::npos) nl = s.size();
        std::string line = s.substr(lo, nl - lo);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        out.push_back(std::move(line));
        lo = nl + 1;
    }
    return out;
}

inline bool strip_prefix(const std::string& s, const char* p, std::string& rest) {
    const size_t n = std::char_traits<char>::length(p);
    if (s.compare(0, n, p) != 0) return false;
    rest = s.substr(n);
    return true;
}

}  // namespace text_detail

// batch.rs:197-231
inline std::string clean_docstring(const std::string& doc) {
    using namespace text_detail;
    std::string result;
    for (const std::string& line : lines(doc)) {
        const std::string trimmed = trim(line);
        std::string cleaned;
        if (trimmed == "*/") {
            cleaned.clear();
        } else {
            cleaned = trimmed;
            for (const char* p : {"///", "//!", "//", "/**", "*", "\""}) {
                std::string rest;
                if (strip_prefix(trimmed, p, rest)) { cleaned = rest; break; }
            }
            cleaned = trim(cleaned);
        }
        if (!cleaned.empty()) {
            if (!result.empty()) result += ' ';
            result += cleaned;
        }
    }
    if (!result.empty() && result.back() == '"') result.pop_back();
    return trim(result);
}

// batch.rs:137-181: Context / Signature / Name / Documentation / Code
inline std::string prepare_text(const Chunk& chunk) {
    std::vector<std::string> parts;
    if (!chunk.context.empty()) {
        std::string j = "Context: ";
        for (size_t i = 0; i < chunk.context.size(); ++i) j += (i ? " > " : "") + chunk.context[i];
        parts.push_back(std::move(j));
    }
    if (chunk.signature) {
        parts.push_back("Signature: " + *chunk.signature);
        const auto words = text_detail::split_whitespace(*chunk.signature);
        if (words.size() > 1) {  // split_whitespace().nth(1), cut at the first of < ( {
            std::string name = words[1];
            for (char stop : {'<', '(', '{'}) name = name.substr(0, name.find(stop));
            parts.push_back("Name: " + name);
        }
    }
    if (chunk.docstring) {
        const std::string cleaned = clean_docstring(*chunk.docstring);
        if (!cleaned.empty()) parts.push_back("Documentation: " + cleaned);
    }
    parts.push_back("Code:\n" + chunk.content);
    std::string out;
    for (size_t i = 0; i < parts.size(); ++i) out += (i ? "\n" : "") + parts[i];
    return out;
}

// batch.rs:12-44
struct EmbeddingStats {
    size_t total_chunks = 0, embedded_chunks = 0, cached_chunks = 0, failed_chunks = 0;
    uint64_t total_time_ms = 0;
    double cache_hit_rate() const { return total_chunks ? (double)cached_chunks / (double)total_chunks : 0.0; }
    double success_rate() const { return total_chunks ? (double)embedded_chunks / (double)total_chunks : 0.0; }
    double chunks_per_second() const {
        return total_time_ms ? (double)embedded_chunks / (double)total_time_ms * 1000.0 : 0.0;
    }
};

// batch.rs:60-194.  batch_size defaults to the reference's 32; on the GPU hand over whole mini-batches
// (256) or everything at once: cs_embedder_embed_texts forms its own mini-batches.
template <class E, class = void> struct has_queue : std::false_type {};
template <class E>
struct has_queue<E, std::void_t<decltype(std::declval<E&>().submit(std::declval<const std::vector<std::string>&>()))>> : std::true_type {};

template <class Embedder>
class BatchEmbedder {
  public:
    explicit BatchEmbedder(Embedder& embedder, size_t batch_size = 32) : embedder_(embedder), batch_size_(batch_size) {}
    static BatchEmbedder with_batch_size(Embedder& embedder, size_t batch_size) {
        return BatchEmbedder(embedder, batch_size);
    }
    // batch.rs:84-115.  An embedder with a submission queue (cs::FastEmbedder: submit / wait) gets every slice SUBMITTED
    // first and collected afterwards — the first wait embeds all queued slices as full device batches — so the
    // reference's slice size of 32 keeps the large-batch rate; any other embedder is called slice by slice.
    std::vector<EmbeddedChunk> embed_chunks(const std::vector<Chunk>& chunks) {
        std::vector<EmbeddedChunk> out;
        out.reserve(chunks.size());
        if constexpr (has_queue<Embedder>::value) {
            if (chunks.size() > batch_size_) {
                std::vector<std::pair<uint64_t, size_t>> tickets;  // (ticket, first chunk)
                try {
                    for (size_t lo = 0; lo < chunks.size(); lo += batch_size_) {
                        const size_t hi = std::min(chunks.size(), lo + batch_size_);
                        std::vector<std::string> texts;
                        for (size_t i = lo; i < hi; ++i) texts.push_back(prepare_text(chunks[i]));
                        tickets.emplace_back(embedder_.submit(texts), lo);
                    }
                    while (!tickets.empty()) {
                        const size_t lo = tickets.front().second, hi = std::min(chunks.size(), lo + batch_size_);
                        auto embs = embedder_.wait(tickets.front().first, hi - lo);
                        tickets.erase(tickets.begin());
                        for (size_t i = lo; i < hi; ++i) out.push_back(EmbeddedChunk{chunks[i], std::move(embs[i - lo])});
                    }
                } catch (...) {
                    for (auto& t : tickets) { try { embedder_.discard(t.first); } catch (...) {} }
                    throw;
                }
                return out;
            }
        }
        for (size_t lo = 0; lo < chunks.size(); lo += batch_size_) {
            const size_t hi = std::min(chunks.size(), lo + batch_size_);
            std::vector<std::string> texts;
            for (size_t i = lo; i < hi; ++i) texts.push_back(prepare_text(chunks[i]));
            auto embs = embedder_.embed_batch(texts);
            for (size_t i = lo; i < hi; ++i) out.push_back(EmbeddedChunk{chunks[i], std::move(embs[i - lo])});
        }
        return out;
    }
    EmbeddedChunk embed_chunk(const Chunk& chunk) { return EmbeddedChunk{chunk, embedder_.embed_one(prepare_text(chunk))}; }
    size_t dimensions() const { return embedder_.dimensions(); }

  private:
    Embedder& embedder_;
    size_t batch_size_;
};

// ---- right after the scan: src/search/mod.rs:494-611 ---------------------------------------------------
constexpr float kHighConfidenceThreshold = 0.15f;  // mod.rs:598: distance < 0.15
constexpr size_t kEarlyTerminationTopN = 5;        // mod.rs:599

inline size_t retrieval_limit(size_t max_results, bool vector_only, bool is_identifier_query) {  // mod.rs:494-502
    if (vector_only) return max_results;
    if (is_identifier_query) return std::max<size_t>(max_results * 3, 100);
    return std::max<size_t>(max_results * 5, 200);
}

// mod.rs:513-590: per chunk id the best-scoring result across the query variants, the `limit` best, score descending
inline std::vector<SearchResult> merge_variant_results(const std::vector<std::vector<SearchResult>>& per_variant,
                                                       size_t limit) {
    std::map<uint32_t, SearchResult> best;
    std::vector<uint32_t> order;  // first-seen order: a stable tie-break
    for (const auto& results : per_variant)
        for (const auto& r : results) {
            auto it = best.find(r.id);
            if (it == best.end()) { best.emplace(r.id, r); order.push_back(r.id); }
            else if (r.score > it->second.score) it->second = r;
        }
    std::vector<SearchResult> merged;
    for (uint32_t id : order) merged.push_back(best[id]);
    std::stable_sort(merged.begin(), merged.end(), [](const SearchResult& a, const SearchResult& b) { return a.score > b.score; });
    if (merged.size() > limit) merged.resize(limit);
    return merged;
}

// mod.rs:601-611: skip FTS when the top five all have distance < 0.15
inline bool should_use_vector_only(const std::vector<SearchResult>& results, bool vector_only) {
    if (vector_only || results.empty()) return false;
    const size_t n = std::min(results.size(), kEarlyTerminationTopN);
    for (size_t i = 0; i < n; ++i)
        if (!(results[i].distance < kHighConfidenceThreshold)) return false;
    return true;
}

// ---- result fusion: src/rerank/mod.rs:14-241 -------------------------------------------------------------
constexpr float kDefaultRrfK = 20.0f;    // rerank/mod.rs:15
constexpr float kExactMatchRrfK = 5.0f;  // rerank/mod.rs:18

struct FtsResult { uint32_t chunk_id; float score; };  // (id, bm25 score) in rank order

struct FusedResult {  // rerank/mod.rs:21-36
    uint32_t chunk_id = 0;
    float rrf_score = 0.f;
    std::optional<float> vector_score, fts_score;
    std::optional<size_t> vector_rank, fts_rank;
};

namespace fusion_detail {
inline float term(float k, size_t rank0) { return 1.0f / (k + (float)rank0 + 1.0f); }  // rerank/mod.rs:59, f32
struct Acc {
    float rrf = 0.f;
    std::optional<float> vs, fs, es;
    std::optional<size_t> vr, fr, er;
    size_t seen = 0;
};
inline std::vector<FusedResult> finish(std::map<uint32_t, Acc>& acc, bool three_way) {
    std::vector<std::pair<size_t, FusedResult>> tmp;
    for (auto& kv : acc) {
        const Acc& a = kv.second;
        FusedResult f;
        f.chunk_id = kv.first;
        f.rrf_score = a.rrf;
        f.vector_score = a.vs;
        f.vector_rank = a.vr;
        if (three_way && a.fs && a.es) f.fts_score = (*a.fs + *a.es) / 2.0f;  // rerank/mod.rs:216-221
        else f.fts_score = a.fs ? a.fs : a.es;
        f.fts_rank = a.fr ? a.fr : a.er;                                       // rerank/mod.rs:229
        tmp.emplace_back(a.seen, std::move(f));
    }
    std::sort(tmp.begin(), tmp.end(), [](const auto& x, const auto& y) { return x.first < y.first; });  // first seen
    std::vector<FusedResult> out;
    for (auto& t : tmp) out.push_back(std::move(t.second));
    std::stable_sort(out.begin(), out.end(), [](const FusedResult& x, const FusedResult& y) { return x.rrf_score > y.rrf_score; });
    return out;
}
}  // namespace fusion_detail

// rerank/mod.rs:139-241 (exact identifier matches get the smaller k); two-way fusion = no exact list
inline std::vector<FusedResult> rrf_fusion_with_exact(const std::vector<SearchResult>& vector_results,
                                                      const std::vector<FtsResult>& fts_results,
                                                      const std::vector<FtsResult>& exact_results,
                                                      float vector_k = kDefaultRrfK, float fts_k = kDefaultRrfK,
                                                      float exact_k = kExactMatchRrfK) {
    using namespace fusion_detail;
    std::map<uint32_t, Acc> acc;
    size_t seen = 0;
    auto entry = [&](uint32_t id) -> Acc& {
        auto it = acc.find(id);
        if (it == acc.end()) { it = acc.emplace(id, Acc{}).first; it->second.seen = seen++; }
        return it->second;
    };
    for (size_t r = 0; r < vector_results.size(); ++r) {
        Acc& a = entry(vector_results[r].id);
        a.rrf += term(vector_k, r); a.vs = vector_results[r].score; a.vr = r + 1;
    }
    for (size_t r = 0; r < fts_results.size(); ++r) {
        Acc& a = entry(fts_results[r].chunk_id);
        a.rrf += term(fts_k, r); a.fs = fts_results[r].score; a.fr = r + 1;
    }
    for (size_t r = 0; r < exact_results.size(); ++r) {
        Acc& a = entry(exact_results[r].chunk_id);
        a.rrf += term(exact_k, r); a.es = exact_results[r].score; a.er = r + 1;
    }
    return finish(acc, true);
}

// rerank/mod.rs:48-108
inline std::vector<FusedResult> rrf_fusion(const std::vector<SearchResult>& vector_results,
                                           const std::vector<FtsResult>& fts_results, float k = kDefaultRrfK) {
    return rrf_fusion_with_exact(vector_results, fts_results, {}, k, k, k);
}

// rerank/mod.rs:111-124: pass-through, rrf_score = the vector score
inline std::vector<FusedResult> vector_only(const std::vector<SearchResult>& vector_results) {
    std::vector<FusedResult> out;
    for (size_t r = 0; r < vector_results.size(); ++r) {
        FusedResult f;
        f.chunk_id = vector_results[r].id;
        f.rrf_score = vector_results[r].score;
        f.vector_score = vector_results[r].score;
        f.vector_rank = r + 1;
        out.push_back(f);
    }
    return out;
}

}  // namespace cs
