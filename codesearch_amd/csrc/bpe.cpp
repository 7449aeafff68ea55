// bpe.cpp — byte-level BPE as the `tokenizers` crate 0.22.2 runs it (bpe.hpp): added-token extraction, the ByteLevel
// pre-tokenizer (GPT-2 pattern), the byte -> character map, BPE merges by rank with the library's own queue discipline,
// <bos> $A <eos>.  Host-only.  The reference reaches this code through fastembed for the registry's
// JinaEmbeddingsV2BaseCode (/root/reference/src/embed/embedder.rs:40-41, :286-289); pinned id for id against the
// `tokenizers` 0.22.2 wheel — the same crate — by tests/test_bpe_tokenizer.py.
//
// The pattern  's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+  is matched by hand (ordered
// alternation, leftmost): the classes \p{L}, \p{N}, \s are the wheel's own, observed per code point
// (gen_bytelevel_tables.py -> bytelevel_tables.inc).
#include "bpe.hpp"

#include <algorithm>
#include <cstring>
#include <queue>

#include "common.hpp"

namespace cs {

namespace {

#include "bytelevel_tables.inc"
#include "nfc_tables.inc"

bool in_ranges(const uint32_t (*t)[2], uint32_t n, uint32_t cp) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (cp < t[mid][0]) hi = mid;
        else if (cp > t[mid][1]) lo = mid + 1;
        else return true;
    }
    return false;
}
inline bool is_letter(uint32_t cp) {
    if (cp < 0x80) return (cp >= 'a' && cp <= 'z') || (cp >= 'A' && cp <= 'Z');
    return in_ranges(kBlLetter, kBlLetter_N, cp);
}
inline bool is_number(uint32_t cp) {
    if (cp < 0x80) return cp >= '0' && cp <= '9';
    return in_ranges(kBlNumber, kBlNumber_N, cp);
}
inline bool is_space(uint32_t cp) {
    if (cp < 0x80) return cp == 0x20 || (cp >= 0x9 && cp <= 0xD);
    return in_ranges(kBlSpace, kBlSpace_N, cp);
}

inline size_t u8_len(unsigned char b) { return b < 0x80 ? 1 : (b >> 5) == 6 ? 2 : (b >> 4) == 14 ? 3 : (b >> 3) == 30 ? 4 : 1; }
inline uint32_t u8_cp(const char* p, size_t len) {
    const unsigned char* u = reinterpret_cast<const unsigned char*>(p);
    switch (len) {
        case 2: return ((u[0] & 0x1Fu) << 6) | (u[1] & 0x3Fu);
        case 3: return ((u[0] & 0x0Fu) << 12) | ((u[1] & 0x3Fu) << 6) | (u[2] & 0x3Fu);
        case 4: return ((u[0] & 0x07u) << 18) | ((u[1] & 0x3Fu) << 12) | ((u[2] & 0x3Fu) << 6) | (u[3] & 0x3Fu);
        default: return u[0];
    }
}
void append_cp(std::string& s, uint32_t cp) {
    if (cp < 0x80) s.push_back((char)cp);
    else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else { s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F))); }
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


// ---- NFC (the normaliser ModernBERT's tokenizer.json puts in front of its BPE model) ----------------------------------------
// Canonical decomposition (tables: the wheel's own NFD per code point; Hangul arithmetically), canonical reordering by
// combining class, canonical composition (the wheel's primary composites; Hangul arithmetically) — UAX #15.
uint32_t nfc_ccc(uint32_t cp) {
    if (cp < 0x300) return 0;
    uint32_t lo = 0, hi = kNfcCcc_N;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (kNfcCcc[mid].cp < cp) lo = mid + 1; else hi = mid;
    }
    return (lo < kNfcCcc_N && kNfcCcc[lo].cp == cp) ? kNfcCcc[lo].ccc : 0;
}
void nfc_decompose(uint32_t cp, std::vector<uint32_t>& out) {
    if (cp >= 0xAC00 && cp <= 0xD7A3) {  // Hangul syllable -> L V (T)
        const uint32_t s = cp - 0xAC00;
        out.push_back(0x1100 + s / 588);
        out.push_back(0x1161 + (s % 588) / 28);
        if (s % 28) out.push_back(0x11A7 + s % 28);
        return;
    }
    if (cp >= 0xC0) {
        uint32_t lo = 0, hi = kNfcDecomp_N;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) / 2;
            if (kNfcDecomp[mid].cp < cp) lo = mid + 1; else hi = mid;
        }
        if (lo < kNfcDecomp_N && kNfcDecomp[lo].cp == cp) {
            for (uint32_t i = 0; i < kNfcDecomp[lo].len; ++i) out.push_back(kNfcDecompPool[kNfcDecomp[lo].off + i]);
            return;
        }
    }
    out.push_back(cp);
}
uint32_t nfc_compose(uint32_t a, uint32_t b) {  // 0: the pair does not compose
    if (a >= 0x1100 && a <= 0x1112 && b >= 0x1161 && b <= 0x1175) return 0xAC00 + ((a - 0x1100) * 21 + (b - 0x1161)) * 28;
    if (a >= 0xAC00 && a <= 0xD7A3 && (a - 0xAC00) % 28 == 0 && b > 0x11A7 && b <= 0x11C2) return a + (b - 0x11A7);
    uint32_t lo = 0, hi = kNfcPairs_N;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        const NfcPair& p = kNfcPairs[mid];
        if (p.first < a || (p.first == a && p.second < b)) lo = mid + 1; else hi = mid;
    }
    return (lo < kNfcPairs_N && kNfcPairs[lo].first == a && kNfcPairs[lo].second == b) ? kNfcPairs[lo].composite : 0;
}
// s: valid UTF-8.  Text whose every code point lies below U+0300 is in NFC as it stands (no combining mark, nothing that
// decomposes to one... composes only with what follows): the common case of source code returns at the scan.
void nfc_normalize(std::string& s) {
    bool plain = true;
    for (size_t i = 0; i < s.size() && plain; ++i) {
        const unsigned char b = (unsigned char)s[i];
        if (b >= 0xCC) plain = false;  // lead bytes 0xCC.. start code points >= U+0300
    }
    if (plain) return;
    std::vector<uint32_t> d;
    d.reserve(s.size());
    for (size_t i = 0; i < s.size();) {
        const size_t l = std::min(u8_len((unsigned char)s[i]), s.size() - i);
        nfc_decompose(u8_cp(s.data() + i, l), d);
        i += l;
    }
    // canonical ordering: every run of non-starters sorted by class, stably
    for (size_t i = 0; i < d.size();) {
        if (nfc_ccc(d[i]) == 0) { ++i; continue; }
        size_t j = i;
        while (j < d.size() && nfc_ccc(d[j]) != 0) ++j;
        std::stable_sort(d.begin() + i, d.begin() + j, [](uint32_t x, uint32_t y) { return nfc_ccc(x) < nfc_ccc(y); });
        i = j;
    }
    // canonical composition (UAX #15 sample algorithm)
    std::vector<uint32_t> c;
    c.reserve(d.size());
    if (!d.empty()) {
        size_t starter_pos = 0;
        uint32_t starter = d[0];
        int last_class = (int)nfc_ccc(starter);
        if (last_class != 0) last_class = 256;  // a leading non-starter never composes with what follows
        c.push_back(starter);
        for (size_t i = 1; i < d.size(); ++i) {
            const uint32_t ch = d[i];
            const int cls = (int)nfc_ccc(ch);
            const uint32_t comp = nfc_compose(starter, ch);
            if (comp && (last_class < cls || last_class == 0)) {
                c[starter_pos] = comp;
                starter = comp;
            } else {
                if (cls == 0) { starter_pos = c.size(); starter = ch; }
                last_class = cls;
                c.push_back(ch);
            }
        }
    }
    s.clear();
    for (uint32_t cp : c) append_cp(s, cp);
}

// GPT-2's bytes_to_unicode: printable bytes stand for themselves, the others for U+0100 + n in order of appearance
struct ByteMap {
    uint32_t cp[256];
    ByteMap() {
        bool own[256] = {};
        for (int b = 33; b <= 126; ++b) own[b] = true;
        for (int b = 161; b <= 172; ++b) own[b] = true;
        for (int b = 174; b <= 255; ++b) own[b] = true;
        uint32_t n = 0;
        for (int b = 0; b < 256; ++b) cp[b] = own[b] ? (uint32_t)b : 256 + n++;
    }
};
const ByteMap kByteMap;

// One match of the GPT-2 pattern at byte offset i of s (valid UTF-8): its end offset (> i).
size_t gpt2_match(const std::string& s, size_t i) {
    const size_t n = s.size();
    auto cp_at = [&](size_t p, size_t& len) { len = std::min(u8_len((unsigned char)s[p]), n - p); return u8_cp(s.data() + p, len); };
    // 's|'t|'re|'ve|'m|'ll|'d
    if (s[i] == '\'' && i + 1 < n) {
        const char c1 = s[i + 1], c2 = i + 2 < n ? s[i + 2] : '\0';
        if (c1 == 's' || c1 == 't') return i + 2;
        if (c1 == 'r' && c2 == 'e') return i + 3;
        if (c1 == 'v' && c2 == 'e') return i + 3;
        if (c1 == 'm') return i + 2;
        if (c1 == 'l' && c2 == 'l') return i + 3;
        if (c1 == 'd') return i + 2;
    }
    //  ?\p{L}+ |  ?\p{N}+ |  ?[^\s\p{L}\p{N}]+   (the optional character is U+0020 only)
    {
        size_t p = i;
        if (s[p] == ' ' && p + 1 < n) ++p;
        size_t len;
        const uint32_t c = cp_at(p, len);
        const int cls = is_letter(c) ? 0 : is_number(c) ? 1 : is_space(c) ? 3 : 2;
        if (cls != 3) {
            size_t q = p + len;
            while (q < n) {
                size_t l2;
                const uint32_t c2 = cp_at(q, l2);
                const int k2 = is_letter(c2) ? 0 : is_number(c2) ? 1 : is_space(c2) ? 3 : 2;
                if (k2 != cls) break;
                q += l2;
            }
            return q;
        }
    }
    // \s+(?!\S) | \s+ : the whitespace run [i, j); followed by a non-space it gives up its last character when it has two
    size_t j = i, last = i;
    while (j < n) {
        size_t len;
        if (!is_space(cp_at(j, len))) break;
        last = j;
        j += len;
    }
    if (j == n || last == i) return j;  // at the end of the text, or a single whitespace character (\s+)
    return last;
}

struct Symbol { int32_t c; int prev, next; uint32_t len; };
struct QMerge {
    uint32_t rank; int pos; int32_t new_id;
    bool operator<(const QMerge& o) const { return rank != o.rank ? rank > o.rank : pos > o.pos; }  // lowest rank, then lowest pos, first
};

}  // namespace

int32_t BpeEngine::create(BpeSpec&& spec, std::shared_ptr<BpeEngine>* out) {
    if (spec.vocab.empty()) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: empty BPE vocabulary");
    if (spec.pres.empty() || spec.pres.back().kind != BpeSpec::Pre::BYTE_LEVEL)
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: a BPE tokenizer without a ByteLevel pre-tokenizer is not built");
    for (size_t k = 0; k + 1 < spec.pres.size(); ++k)
        if (spec.pres[k].kind == BpeSpec::Pre::BYTE_LEVEL)
            return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: ByteLevel must be the last pre-tokenizer step");
    std::shared_ptr<BpeEngine> e(new BpeEngine());
    int64_t max_id = -1;
    for (const auto& kv : spec.vocab) {
        if (kv.second < 0) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: negative id in the BPE vocabulary");
        e->ids_[kv.first] = kv.second;
        max_id = std::max<int64_t>(max_id, kv.second);
    }
    for (const auto& a : spec.added) {
        if (a.id < 0 || a.text.empty()) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: malformed added token");
        max_id = std::max<int64_t>(max_id, a.id);
    }
    e->vocab_size_ = (uint32_t)(max_id + 1);
    for (int b = 0; b < 256; ++b) {
        std::string ch;
        append_cp(ch, kByteMap.cp[b]);
        auto it = e->ids_.find(ch);
        e->byte_id_[b] = it == e->ids_.end() ? -1 : it->second;
    }
    if (!spec.unk_token.empty()) {
        auto it = e->ids_.find(spec.unk_token);
        if (it == e->ids_.end())
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: unk_token \"%s\" is not in the BPE vocabulary", spec.unk_token.c_str());
        e->unk_id_ = it->second;
    }
    uint32_t rank = 0;
    for (const auto& m : spec.merges) {
        auto a = e->ids_.find(m.first), b = e->ids_.find(m.second), ab = e->ids_.find(m.first + m.second);
        if (a == e->ids_.end() || b == e->ids_.end() || ab == e->ids_.end())
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: merge \"%s %s\" names a token outside the vocabulary",
                        m.first.c_str(), m.second.c_str());
        const uint64_t key = ((uint64_t)(uint32_t)a->second << 32) | (uint32_t)b->second;
        e->merge_.emplace(key, std::make_pair(rank, ab->second));  // (a repeated pair keeps its first rank, as a HashMap insert loop would not — the library's files hold none)
        ++rank;
    }
    e->spec_ = std::move(spec);
    // a normalized added token is matched in normalised text: its own content goes through the normaliser once, here
    for (auto& a : e->spec_.added)
        if (a.normalized) {
            if (e->spec_.nfc) nfc_normalize(a.text);
            if (a.text.empty()) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: malformed added token");
            e->has_normalized_added_ = true;
        }
    // added tokens: longest first, so the scan below is leftmost-longest
    std::stable_sort(e->spec_.added.begin(), e->spec_.added.end(),
                     [](const BpeSpec::Added& x, const BpeSpec::Added& y) { return x.text.size() > y.text.size(); });
    *out = e;
    return CS_OK;
}

int32_t BpeEngine::token_to_id(const std::string& s) const {
    for (const auto& a : spec_.added)
        if (a.text == s) return a.id;
    auto it = ids_.find(s);
    return it == ids_.end() ? -1 : it->second;
}

// One pre-token (raw bytes; every byte is one symbol of the byte-level alphabet) -> ids: Word::merge_all of the library
void BpeEngine::encode_word(const std::string& bytes, std::vector<int32_t>& ids) const {
    if (bytes.empty()) return;
    if (spec_.ignore_merges) {  // the whole word as one token when the vocabulary holds it
        std::string mapped;
        for (unsigned char b : bytes) append_cp(mapped, kByteMap.cp[b]);
        auto it = ids_.find(mapped);
        if (it != ids_.end()) { ids.push_back(it->second); return; }
    }
    std::vector<Symbol> sym;
    sym.reserve(bytes.size());
    bool last_unk = false;
    for (unsigned char b : bytes) {
        int32_t id = byte_id_[b];
        if (id < 0) {
            if (unk_id_ < 0) continue;  // no unk token: the library drops the symbol
            if (spec_.fuse_unk && last_unk) { sym.back().len += 1; continue; }
            id = unk_id_;
            last_unk = true;
        } else {
            last_unk = false;
        }
        sym.push_back({id, (int)sym.size() - 1, (int)sym.size() + 1, 1});
    }
    if (sym.empty()) return;
    sym.back().next = -1;
    auto find = [&](int32_t a, int32_t b) -> const std::pair<uint32_t, int32_t>* {
        auto it = merge_.find(((uint64_t)(uint32_t)a << 32) | (uint32_t)b);
        return it == merge_.end() ? nullptr : &it->second;
    };
    std::priority_queue<QMerge> q;
    for (size_t i = 0; i + 1 < sym.size(); ++i)
        if (const auto* m = find(sym[i].c, sym[i + 1].c)) q.push({m->first, (int)i, m->second});
    while (!q.empty()) {
        const QMerge top = q.top();
        q.pop();
        if (sym[top.pos].len == 0) continue;
        if (sym[top.pos].next == -1) continue;
        const int next_pos = sym[top.pos].next;
        const Symbol right = sym[next_pos];
        const auto* cur = find(sym[top.pos].c, right.c);  // an expired entry: the pair there now merges to something else (or not at all)
        if (!cur || cur->second != top.new_id) continue;
        sym[top.pos].c = top.new_id;
        sym[top.pos].len += right.len;
        sym[top.pos].next = right.next;
        sym[next_pos].len = 0;
        if (right.next > -1 && (size_t)right.next < sym.size()) sym[right.next].prev = top.pos;
        const Symbol& c = sym[top.pos];
        if (c.prev >= 0)
            if (const auto* m = find(sym[c.prev].c, c.c)) q.push({m->first, c.prev, m->second});
        if (c.next >= 0 && (size_t)c.next < sym.size())
            if (const auto* m = find(c.c, sym[c.next].c)) q.push({m->first, top.pos, m->second});
    }
    for (const Symbol& s : sym)
        if (s.len != 0) ids.push_back(s.c);
}

// A stretch of text between added tokens, normalised: the pre-tokenizer steps, then every pre-token through the model
void BpeEngine::encode_piece(std::string&& text, std::vector<int32_t>& ids) const {
    std::vector<std::string> pieces(1, std::move(text));
    for (const auto& pre : spec_.pres) {
        std::vector<std::string> next;
        for (std::string& t : pieces) {
            if (pre.kind == BpeSpec::Pre::DIGITS) {  // runs of numeric characters isolated (one piece per digit with individual_digits)
                size_t i = 0, start = 0;
                bool in_num = false;
                auto flush = [&](size_t lo, size_t hi) { if (hi > lo) next.push_back(t.substr(lo, hi - lo)); };
                while (i < t.size()) {
                    const size_t l = std::min(u8_len((unsigned char)t[i]), t.size() - i);
                    const bool num = is_number(u8_cp(t.data() + i, l));
                    if (num != in_num || (num && pre.individual_digits)) { flush(start, i); start = i; }
                    in_num = num;
                    i += l;
                }
                flush(start, t.size());
                continue;
            }
            if (pre.add_prefix_space && (t.empty() || t[0] != ' ')) t.insert(t.begin(), ' ');
            if (!pre.use_regex) { next.push_back(std::move(t)); continue; }
            size_t i = 0;
            while (i < t.size()) {
                const size_t j = gpt2_match(t, i);
                next.push_back(t.substr(i, j - i));
                i = j;
            }
        }
        pieces.swap(next);
    }
    for (const std::string& w : pieces) encode_word(w, ids);
}

template <class F>
void BpeEngine::split_on_added(const std::string& text, bool normalized, std::vector<int32_t>& ids, F&& piece) const {
    size_t seg = 0, i = 0;
    while (i < text.size()) {
        const BpeSpec::Added* hit = nullptr;
        for (const auto& a : spec_.added)  // (longest first)
            if (a.normalized == normalized && text.compare(i, a.text.size(), a.text) == 0) { hit = &a; break; }
        if (!hit) { i += std::min(u8_len((unsigned char)text[i]), text.size() - i); continue; }
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
                const size_t l = std::min(u8_len((unsigned char)text[hi]), text.size() - hi);
                if (!is_space(u8_cp(text.data() + hi, l))) break;
                hi += l;
            }
        if (lo > seg) piece(seg, lo);
        ids.push_back(hit->id);
        seg = i = hi;
    }
    if (text.size() > seg) piece(seg, text.size());
}

// A stretch of raw text between the verbatim (non-normalized) added tokens: normalised, cut at the normalized added tokens, the
// rest through the pre-tokenizer and the model
void BpeEngine::encode_segment(const char* p, size_t n, bool, std::vector<int32_t>& ids) const {
    std::string text(p, n);
    if (spec_.nfc) nfc_normalize(text);
    if (!has_normalized_added_) {
        encode_piece(std::move(text), ids);
        return;
    }
    split_on_added(text, true, ids, [&](size_t lo, size_t hi) { encode_piece(text.substr(lo, hi - lo), ids); });
}

void BpeEngine::encode(const char* utf8, size_t n, uint32_t body_max, std::vector<int32_t>& ids) const {
    if (spec_.bos >= 0) ids.push_back(spec_.bos);
    const size_t base = ids.size();
    const std::string text = sanitize(utf8, n);
    split_on_added(text, false, ids, [&](size_t lo, size_t hi) { encode_segment(text.data() + lo, hi - lo, lo == 0, ids); });
    if (ids.size() - base > body_max) ids.resize(base + body_max);  // truncation: on the right, before the template
    if (spec_.eos >= 0) ids.push_back(spec_.eos);
}

}  // namespace cs
