// unigram.cpp — the `tokenizers` crate's pipeline for SentencePiece-unigram vocabularies, restated (see unigram.hpp).
// What the library does, component by component (tokenizers 0.22.2, the reference's pin; spm_precompiled 0.1):
//   added tokens   special tokens are cut out of the RAW text, leftmost-longest (lstrip / rstrip eat the neighbouring
//                  whitespace); every stretch between them runs through the rest on its own
//   Precompiled    SentencePiece's compiled character map (a darts-clone double array + replacement strings), applied
//                  grapheme by grapheme: a grapheme shorter than 6 bytes is looked up whole and replaced by the
//                  replacement of its SHORTEST matching prefix; otherwise (or on a miss) each of its characters is
//                  looked up alone (normalizers/precompiled.rs — "yes, this is weird" is the library's own comment)
//   Replace        Regex " {2,}" -> content (runs of two or more spaces); a literal String pattern
//   Strip          Unicode White_Space off either end
//   WhitespaceSplit / Metaspace   split at White_Space; ' ' -> U+2581, the prefix per prepend_scheme, split before every U+2581
//   Unigram        Viterbi over the piece trie, f64 scores, unknown characters at min_score - 10 and fused
//                  (models/unigram/model.rs encode_optimized)
//   TemplateProcessing "<s> $A </s>", truncation on the right to max_length - 2
// Host-only C++.
#include "unigram.hpp"

#include <algorithm>
#include <cstring>
#include <limits>
#include <unordered_map>

#include "common.hpp"
#include "grapheme_tables.hpp"

namespace cs {

namespace {

template <size_t N>
bool in_table(const CpRange (&t)[N], uint32_t cp) {
    size_t lo = 0, hi = N;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (cp < t[mid].lo) hi = mid;
        else if (cp > t[mid].hi) lo = mid + 1;
        else return true;
    }
    return false;
}
bool is_space(uint32_t cp) { return in_table(kWhiteSpace, cp); }

// one character of VALID UTF-8 at p (the input is sanitised first): its length and code point
inline size_t u8_len(unsigned char b) { return b < 0x80 ? 1 : (b < 0xE0 ? 2 : (b < 0xF0 ? 3 : 4)); }
inline uint32_t u8_cp(const char* p, size_t len) {
    const unsigned char* u = reinterpret_cast<const unsigned char*>(p);
    if (len == 1) return u[0];
    if (len == 2) return ((u[0] & 0x1Fu) << 6) | (u[1] & 0x3Fu);
    if (len == 3) return ((u[0] & 0x0Fu) << 12) | ((u[1] & 0x3Fu) << 6) | (u[2] & 0x3Fu);
    return ((u[0] & 0x07u) << 18) | ((u[1] & 0x3Fu) << 12) | ((u[2] & 0x3Fu) << 6) | (u[3] & 0x3Fu);
}

// bytes -> valid UTF-8, every ill-formed byte sequence replaced by U+FFFD (what a Rust caller's from_utf8_lossy hands over)
std::string sanitize(const char* s, size_t n) {
    std::string out;
    out.reserve(n);
    const unsigned char* u = reinterpret_cast<const unsigned char*>(s);
    size_t i = 0;
    while (i < n) {
        const unsigned char b = u[i];
        size_t len = 0;
        if (b < 0x80) len = 1;
        else if (b >= 0xC2 && b <= 0xDF) len = 2;
        else if (b >= 0xE0 && b <= 0xEF) len = 3;
        else if (b >= 0xF0 && b <= 0xF4) len = 4;
        bool ok = len != 0 && i + len <= n;
        for (size_t k = 1; ok && k < len; ++k) ok = (u[i + k] & 0xC0) == 0x80;
        if (ok && len == 3) {
            const uint32_t cp = u8_cp(s + i, 3);
            ok = cp >= 0x800 && !(cp >= 0xD800 && cp <= 0xDFFF);
        }
        if (ok && len == 4) {
            const uint32_t cp = u8_cp(s + i, 4);
            ok = cp >= 0x10000 && cp <= 0x10FFFF;
        }
        if (ok) { out.append(s + i, len); i += len; }
        else { out.append("\xEF\xBF\xBD"); ++i; }
    }
    return out;
}

// true when the bytes are well-formed UTF-8 (what sanitize would leave unchanged)
bool valid_utf8(const char* s, size_t n) {
    const unsigned char* u = reinterpret_cast<const unsigned char*>(s);
    size_t i = 0;
    while (i < n) {
        const unsigned char b = u[i];
        size_t len = 0;
        if (b < 0x80) len = 1;
        else if (b >= 0xC2 && b <= 0xDF) len = 2;
        else if (b >= 0xE0 && b <= 0xEF) len = 3;
        else if (b >= 0xF0 && b <= 0xF4) len = 4;
        if (len == 0 || i + len > n) return false;
        for (size_t k = 1; k < len; ++k)
            if ((u[i + k] & 0xC0) != 0x80) return false;
        if (len == 3) {
            const uint32_t cp = u8_cp(s + i, 3);
            if (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        }
        if (len == 4) {
            const uint32_t cp = u8_cp(s + i, 4);
            if (cp < 0x10000 || cp > 0x10FFFF) return false;
        }
        i += len;
    }
    return true;
}
bool valid_utf8(const std::string& s) { return valid_utf8(s.data(), s.size()); }
// length of the character at s[i], never past the end of the string (the strings below are valid UTF-8 by construction —
// sanitize() for the text, UnigramEngine::create for everything a tokenizer.json can splice into it — this is the belt)
inline size_t u8_len_at(const std::string& s, size_t i) { return std::min(u8_len((unsigned char)s[i]), s.size() - i); }

enum Gcb { G_OTHER, G_CONTROL, G_EXTEND, G_SPACING, G_PREPEND };
Gcb gcb(uint32_t cp) {
    if (cp < 0x300) {  // the fast path of ordinary text: only controls below U+0300 (plus U+00AD, in the table)
        if (cp < 0x20 || (cp >= 0x7F && cp <= 0x9F) || cp == 0xAD) return G_CONTROL;
        return G_OTHER;
    }
    if (in_table(kGcbExtend, cp)) return G_EXTEND;
    if (in_table(kGcbSpacingMark, cp)) return G_SPACING;
    if (in_table(kGcbControl, cp)) return G_CONTROL;
    if (in_table(kGcbPrepend, cp)) return G_PREPEND;
    return G_OTHER;
}

}  // namespace

int32_t UnigramEngine::create(UnigramSpec&& spec, std::shared_ptr<UnigramEngine>* out) {
    if (spec.vocab.empty()) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: empty unigram vocabulary");
    if (spec.unk_id < 0 || (size_t)spec.unk_id >= spec.vocab.size())
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: unigram unk_id %d outside the vocabulary", spec.unk_id);
    for (int32_t id : {spec.bos, spec.eos, spec.pad})
        if (id < 0 || (size_t)id >= spec.vocab.size())
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: <s> / </s> / <pad> missing from the unigram vocabulary");
    // Everything the file can splice INTO a text after sanitize() has run must itself be well-formed UTF-8: the pipeline
    // below walks strings character by character and trusts their lead bytes (a Replace content of one byte
    // 0xE2 made Strip run past the end of the string on a tokenizer worker thread).
    for (const auto& nz : spec.norms)
        if (!valid_utf8(nz.content) || !valid_utf8(nz.pattern))
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: a Replace normalizer's pattern or content is not valid UTF-8");
    for (const auto& pre : spec.pres)
        if (!valid_utf8(pre.replacement))
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: the Metaspace replacement is not valid UTF-8");
    for (size_t i = 0; i < spec.vocab.size(); ++i)
        if (!valid_utf8(spec.vocab[i].first))
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: unigram piece %zu is not valid UTF-8", i);
    for (const auto& a : spec.added)
        if (!valid_utf8(a.text))
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: added token %d is not valid UTF-8", a.id);
    auto eng = std::make_shared<UnigramEngine>();
    eng->spec_ = std::move(spec);
    double mn = std::numeric_limits<double>::infinity();
    for (const auto& v : eng->spec_.vocab) mn = std::min(mn, v.second);
    eng->min_score_ = mn;
    for (const auto& nz : eng->spec_.norms) {
        if (nz.kind != UnigramSpec::Norm::PRECOMPILED) continue;
        Charsmap m;
        const std::string& b = nz.blob;
        if (!b.empty()) {  // (an empty map normalises nothing)
            if (b.size() < 4) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: precompiled_charsmap is truncated");
            uint32_t tsize = 0;
            std::memcpy(&tsize, b.data(), 4);
            if (tsize % 4 || (size_t)tsize + 4 > b.size())
                return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: precompiled_charsmap trie size %u does not fit", tsize);
            m.trie.resize(tsize / 4);
            if (tsize) std::memcpy(m.trie.data(), b.data() + 4, tsize);
            m.normalized.assign(b.data() + 4 + tsize, b.size() - 4 - tsize);
            // every NUL-terminated replacement of the pool (transform() hands out [idx, next NUL) for an idx the trie
            // chooses: a replacement that starts inside a character is caught by checking every suffix start the trie
            // can name, i.e. the whole pool piecewise AND, at lookup time, the piece handed out)
            for (size_t lo = 0; lo < m.normalized.size();) {
                size_t hi = lo;
                while (hi < m.normalized.size() && m.normalized[hi] != '\0') ++hi;
                if (!valid_utf8(m.normalized.data() + lo, hi - lo))
                    return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: precompiled_charsmap holds a replacement that is not valid UTF-8");
                lo = hi + 1;
            }
        }
        eng->maps_.push_back(std::move(m));
    }
    // added tokens: longest first, so the first hit at a position is the longest
    std::stable_sort(eng->spec_.added.begin(), eng->spec_.added.end(),
                     [](const UnigramSpec::Added& a, const UnigramSpec::Added& b) { return a.text.size() > b.text.size(); });
    eng->build_trie();
    *out = std::move(eng);
    return CS_OK;
}

void UnigramEngine::build_trie() {
    // the library's token_to_ids is a map filled in id order: a repeated piece keeps its LAST id
    std::unordered_map<std::string, int32_t> last;
    last.reserve(spec_.vocab.size() * 2);
    for (size_t i = 0; i < spec_.vocab.size(); ++i) last[spec_.vocab[i].first] = (int32_t)i;
    struct Item { const std::string* s; int32_t id; };
    std::vector<Item> items;
    items.reserve(last.size());
    for (const auto& kv : last)
        if (!kv.first.empty()) items.push_back({&kv.first, kv.second});
    std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return *a.s < *b.s; });
    struct Range { uint32_t lo, hi, depth; };
    std::vector<Range> ranges;
    ranges.push_back({0, (uint32_t)items.size(), 0});
    id_.assign(1, -1);
    for (size_t n = 0; n < ranges.size(); ++n) {  // breadth first: a node's children are appended when the node is reached
        const Range r = ranges[n];
        begin_.push_back((uint32_t)labels_.size());
        uint32_t i = r.lo;
        if (i < r.hi && items[i].s->size() == r.depth) { id_[n] = items[i].id; ++i; }
        while (i < r.hi) {
            const unsigned char b = (unsigned char)(*items[i].s)[r.depth];
            uint32_t j = i;
            while (j < r.hi && (unsigned char)(*items[j].s)[r.depth] == b) ++j;
            labels_.push_back(b);
            child_.push_back((uint32_t)ranges.size());
            ranges.push_back({i, j, r.depth + 1});
            id_.push_back(-1);
            i = j;
        }
    }
    begin_.push_back((uint32_t)labels_.size());
}

int32_t UnigramEngine::token_to_id(const std::string& s) const {
    if (s.empty()) return -1;
    uint32_t node = 0;
    for (unsigned char b : s) {
        const uint8_t* lo = labels_.data() + begin_[node];
        const uint8_t* hi = labels_.data() + begin_[node + 1];
        const uint8_t* it = std::lower_bound(lo, hi, b);
        if (it == hi || *it != b) return -1;
        node = child_[(size_t)(it - labels_.data())];
    }
    return id_[node];
}

// spm_precompiled::Precompiled::transform: the replacement of the SHORTEST key that is a prefix of the chunk
bool UnigramEngine::transform(const Charsmap& m, const char* chunk, size_t n, const char** out, size_t* out_n) const {
    if (m.trie.empty()) return false;
    auto offset = [](uint32_t unit) { return (size_t)((unit >> 10) << ((unit & (1u << 9)) >> 6)); };
    size_t pos = 0;
    uint32_t unit = m.trie[0];
    pos ^= offset(unit);
    for (size_t i = 0; i < n; ++i) {
        const unsigned char c = (unsigned char)chunk[i];
        if (c == 0) break;
        pos ^= c;
        if (pos >= m.trie.size()) return false;
        unit = m.trie[pos];
        if ((unit & ((1u << 31) | 0xFFu)) != c) return false;
        pos ^= offset(unit);
        if ((unit >> 8) & 1u) {
            if (pos >= m.trie.size()) return false;
            const size_t idx = m.trie[pos] & 0x7FFFFFFFu;
            if (idx > m.normalized.size()) return false;
            size_t end = idx;
            while (end < m.normalized.size() && m.normalized[end] != '\0') ++end;
            if (!valid_utf8(m.normalized.data() + idx, end - idx)) return false;  // an index into the middle of a character
            *out = m.normalized.data() + idx;
            *out_n = end - idx;
            return true;
        }
    }
    return false;
}

void UnigramEngine::normalize(std::string& s) const {
    size_t map_i = 0;
    for (const auto& nz : spec_.norms) {
        switch (nz.kind) {
            case UnigramSpec::Norm::PRECOMPILED: {
                const Charsmap& m = maps_[map_i++];
                if (m.trie.empty()) break;
                std::string out;
                out.reserve(s.size() + 8);
                size_t i = 0;
                const size_t n = s.size();
                while (i < n) {
                    // one grapheme cluster [i, j).  Only clusters under six bytes are looked up whole, and the walk below is
                    // character by character otherwise, so the rules that only ever build longer clusters (Hangul
                    // sequences, regional-indicator pairs, emoji ZWJ sequences, Indic conjuncts) do not change the result
                    size_t len = u8_len_at(s, i);
                    uint32_t cp = u8_cp(s.data() + i, len);
                    size_t j = i + len;
                    Gcb g = gcb(cp);
                    if (cp == '\r' && j < n && s[j] == '\n') {
                        j += 1;
                    } else if (g != G_CONTROL) {
                        while (g == G_PREPEND && j < n) {  // GB9b: Prepend x (anything but a control)
                            const size_t l2 = u8_len_at(s, j);
                            const uint32_t c2 = u8_cp(s.data() + j, l2);
                            const Gcb g2 = gcb(c2);
                            if (g2 == G_CONTROL) break;
                            j += l2;
                            g = g2;
                        }
                        while (j < n) {  // GB9 / GB9a: x (Extend | ZWJ | SpacingMark)
                            const size_t l2 = u8_len_at(s, j);
                            const Gcb g2 = gcb(u8_cp(s.data() + j, l2));
                            if (g2 != G_EXTEND && g2 != G_SPACING) break;
                            j += l2;
                        }
                    }
                    const char* rep = nullptr;
                    size_t rep_n = 0;
                    if (j - i < 6 && transform(m, s.data() + i, j - i, &rep, &rep_n)) {
                        out.append(rep, rep_n);
                    } else {
                        for (size_t k = i; k < j;) {
                            const size_t l2 = std::min(u8_len((unsigned char)s[k]), j - k);
                            if (transform(m, s.data() + k, l2, &rep, &rep_n)) out.append(rep, rep_n);
                            else out.append(s.data() + k, l2);
                            k += l2;
                        }
                    }
                    i = j;
                }
                s.swap(out);
                break;
            }
            case UnigramSpec::Norm::REPLACE_SPACES: {
                std::string out;
                out.reserve(s.size());
                for (size_t i = 0; i < s.size();) {
                    if (s[i] == ' ') {
                        size_t j = i;
                        while (j < s.size() && s[j] == ' ') ++j;
                        if (j - i >= 2) out += nz.content;
                        else out.push_back(' ');
                        i = j;
                    } else {
                        out.push_back(s[i++]);
                    }
                }
                s.swap(out);
                break;
            }
            case UnigramSpec::Norm::REPLACE_STRING: {
                if (nz.pattern.empty()) break;
                std::string out;
                size_t i = 0;
                for (;;) {
                    const size_t hit = s.find(nz.pattern, i);
                    if (hit == std::string::npos) { out.append(s, i, std::string::npos); break; }
                    out.append(s, i, hit - i);
                    out += nz.content;
                    i = hit + nz.pattern.size();
                }
                s.swap(out);
                break;
            }
            case UnigramSpec::Norm::STRIP: {
                size_t lo = 0, hi = s.size();
                if (nz.left)
                    while (lo < hi) {
                        const size_t l = std::min(u8_len((unsigned char)s[lo]), hi - lo);
                        if (!is_space(u8_cp(s.data() + lo, l))) break;
                        lo += l;
                    }
                if (nz.right)
                    while (hi > lo) {
                        size_t k = hi - 1;
                        while (k > lo && ((unsigned char)s[k] & 0xC0) == 0x80) --k;
                        if (hi - k > 4 || !is_space(u8_cp(s.data() + k, hi - k))) break;
                        hi = k;
                    }
                s = s.substr(lo, hi - lo);  // lo <= hi <= size by the clamps above
                break;
            }
        }
    }
}

// models/unigram/model.rs encode_optimized + tokenize: the best segmentation's piece ids
void UnigramEngine::model_encode(const std::string& piece, std::vector<int32_t>& ids) const {
    const size_t size = piece.size();
    if (size == 0) return;
    struct Node { int32_t id; double score; int32_t starts_at; };
    std::vector<Node> best(size + 1, Node{0, 0.0, -1});
    const double unk_score = min_score_ - 10.0;
    size_t at = 0;
    while (at < size) {
        const double here = best[at].score;
        bool has_single = false;
        const size_t mblen = std::min(u8_len((unsigned char)piece[at]), size - at);
        uint32_t node = 0;
        for (size_t k = at; k < size; ++k) {
            const unsigned char b = (unsigned char)piece[k];
            const uint8_t* lo = labels_.data() + begin_[node];
            const uint8_t* hi = labels_.data() + begin_[node + 1];
            const uint8_t* it = std::lower_bound(lo, hi, b);
            if (it == hi || *it != b) break;
            node = child_[(size_t)(it - labels_.data())];
            const int32_t id = id_[node];
            if (id < 0) continue;
            const size_t end = k + 1, length = end - at;
            Node& t = best[end];
            const double cand = spec_.vocab[(size_t)id].second + here;
            if (t.starts_at < 0 || cand > t.score) { t.score = cand; t.starts_at = (int32_t)at; t.id = id; }
            if (!has_single && length == mblen) has_single = true;
        }
        if (!has_single) {
            Node& t = best[at + mblen];
            const double cand = unk_score + here;
            if (t.starts_at < 0 || cand > t.score) { t.score = cand; t.starts_at = (int32_t)at; t.id = spec_.unk_id; }
        }
        at += mblen;
    }
    // back to front; consecutive unknown pieces fuse into ONE unknown token (fuse_unk, the model's default)
    std::vector<int32_t> rev;
    size_t end = size;
    bool in_unk = false;
    while (end > 0) {
        const Node& nd = best[end];
        if (nd.id == spec_.unk_id) {
            if (!in_unk) { rev.push_back(spec_.unk_id); in_unk = true; }
        } else {
            rev.push_back(nd.id);
            in_unk = false;
        }
        end = (size_t)nd.starts_at;
    }
    ids.insert(ids.end(), rev.rbegin(), rev.rend());
}

void UnigramEngine::encode_segment(const char* p, size_t n, bool at_text_start, std::vector<int32_t>& ids) const {
    std::string s(p, n);
    normalize(s);
    if (s.empty()) return;
    struct Piece { std::string text; bool from_offset0; };
    std::vector<Piece> pieces;
    pieces.push_back({std::move(s), at_text_start});
    for (const auto& pre : spec_.pres) {
        std::vector<Piece> next;
        for (auto& pc : pieces) {
            const std::string& t = pc.text;
            if (pre.kind == UnigramSpec::Pre::WHITESPACE_SPLIT) {
                size_t i = 0, start = 0;
                bool first = true;
                auto flush = [&](size_t lo, size_t hi) {
                    if (hi > lo) next.push_back({t.substr(lo, hi - lo), pc.from_offset0 && first && lo == 0});
                    first = false;
                };
                while (i < t.size()) {
                    const size_t l = u8_len_at(t, i);
                    if (is_space(u8_cp(t.data() + i, l))) {
                        if (i > start) flush(start, i);
                        start = i + l;
                    }
                    i += l;
                }
                if (t.size() > start) flush(start, t.size());
            } else {
                std::string m;
                m.reserve(t.size() + 4);
                for (char c : t) {
                    if (c == ' ') m += pre.replacement;
                    else m.push_back(c);
                }
                const bool starts = m.compare(0, pre.replacement.size(), pre.replacement) == 0;
                if (!starts && (pre.prepend == 1 || (pre.prepend == 2 && pc.from_offset0))) m.insert(0, pre.replacement);
                if (!pre.split) { next.push_back({std::move(m), false}); continue; }
                // split at every replacement character, each one opening the piece that follows it (MergedWithNext)
                size_t start = 0, i = 0;
                while (i < m.size()) {
                    if (m.compare(i, pre.replacement.size(), pre.replacement) == 0) {
                        if (i > start) next.push_back({m.substr(start, i - start), false});
                        start = i;
                        i += pre.replacement.size();
                    } else {
                        i += u8_len_at(m, i);
                    }
                }
                if (m.size() > start) next.push_back({m.substr(start), false});
            }
        }
        pieces.swap(next);
    }
    for (const auto& pc : pieces) model_encode(pc.text, ids);
}

void UnigramEngine::encode(const char* utf8, size_t n, uint32_t body_max, std::vector<int32_t>& ids) const {
    ids.push_back(spec_.bos);
    const size_t base = ids.size();
    const std::string text = sanitize(utf8, n);
    size_t seg = 0, i = 0;
    bool first_seg = true;
    auto run_segment = [&](size_t lo, size_t hi) {
        if (hi > lo) encode_segment(text.data() + lo, hi - lo, first_seg && lo == 0, ids);
        first_seg = false;
    };
    while (i < text.size()) {
        const UnigramSpec::Added* hit = nullptr;
        if (!spec_.added.empty())
            for (const auto& a : spec_.added)
                if (!a.text.empty() && text.compare(i, a.text.size(), a.text) == 0) { hit = &a; break; }
        if (!hit) { i += u8_len_at(text, i); continue; }
        size_t lo = i, hi = i + hit->text.size();
        if (hit->lstrip)  // the token takes the whitespace in front of it
            while (lo > seg) {
                size_t k = lo - 1;
                while (k > seg && ((unsigned char)text[k] & 0xC0) == 0x80) --k;
                if (lo - k > 4 || !is_space(u8_cp(text.data() + k, lo - k))) break;
                lo = k;
            }
        if (hit->rstrip)
            while (hi < text.size()) {
                const size_t l = u8_len_at(text, hi);
                if (!is_space(u8_cp(text.data() + hi, l))) break;
                hi += l;
            }
        run_segment(seg, lo);
        ids.push_back(hit->id);
        seg = i = hi;
    }
    run_segment(seg, text.size());
    if (ids.size() - base > body_max) ids.resize(base + body_max);  // truncation: on the right, before the template
    ids.push_back(spec_.eos);
}

}  // namespace cs
