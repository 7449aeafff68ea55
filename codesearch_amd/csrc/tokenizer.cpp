// tokenizer.cpp — host-side BERT WordPiece tokenizer behind cs_tokenizer_* (SURVEY.md §8f-1).
//
// The reference tokenises inside fastembed with the `tokenizers` crate 0.22.2 (Cargo.lock;
// call site /root/reference/src/embed/embedder.rs:286-289), configured by the model's
// tokenizer.json the way every BERT-family checkpoint is:
//   added special tokens ([PAD] [UNK] [CLS] [SEP] [MASK], matched verbatim in the raw text)
//   -> BertNormalizer(clean_text, handle_chinese_chars, strip_accents = lowercase, lowercase)
//   -> BertPreTokenizer (split on whitespace, isolate punctuation)
//   -> WordPiece("##", [UNK], max_input_chars_per_word = 100), greedy longest match first
//   -> [CLS] A [SEP], truncation to max_length, padding to the batch's longest sequence.
// This file restates that published algorithm.  The per-code-point facts (which characters are
// dropped, are spaces, are punctuation, are CJK, and what lowercase(strip_Mn(NFD(c))) is) come from
// unicode_tables.inc, recorded from the 0.22.2 wheel by gen_unicode_tables.py.  Pinned by
// tests/golden/tokenizer_golden*.json (outputs of that same library).
//
// Known approximation: NFD's canonical reordering is not applied to the 23 spacing marks (Mc)
// that carry a non-zero combining class (Balinese/Javanese viramas, U+302E/F, musical symbols);
// every non-spacing mark is removed anyway, so the order of the survivors only differs when two of
// those 23 follow each other out of canonical order.

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"
#include "bpe.hpp"
#include "unigram.hpp"

namespace {

#include "unicode_tables.inc"

using cs::fail;

bool in_ranges(const uint32_t (*r)[2], uint32_t n, uint32_t c) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (c > r[mid][1]) lo = mid + 1;
        else hi = mid;
    }
    return lo < n && c >= r[lo][0];
}

const FoldEntry* find_fold(uint32_t c) {
    uint32_t lo = 0, hi = kFold_N;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (kFold[mid].cp < c) lo = mid + 1;
        else hi = mid;
    }
    return (lo < kFold_N && kFold[lo].cp == c) ? &kFold[lo] : nullptr;
}

// Character classes of one code point, packed for the ASCII/Latin fast path.
enum : uint8_t { kClsRemove = 1, kClsSpace = 2, kClsPunct = 4, kClsCjk = 8, kClsFold = 16 };

struct ClassTable {
    uint8_t low[0x3000];
    ClassTable() {
        for (uint32_t c = 0; c < 0x3000; ++c) low[c] = slow(c);
    }
    static uint8_t slow(uint32_t c) {
        uint8_t f = 0;
        if (in_ranges(kCleanRemove, kCleanRemove_N, c)) f |= kClsRemove;
        if (in_ranges(kSpace, kSpace_N, c)) f |= kClsSpace;
        if (in_ranges(kPunct, kPunct_N, c)) f |= kClsPunct;
        if (in_ranges(kCjk, kCjk_N, c)) f |= kClsCjk;
        if ((c >= 0xAC00 && c <= 0xD7A3) || find_fold(c)) f |= kClsFold;
        return f;
    }
    uint8_t get(uint32_t c) const { return c < 0x3000 ? low[c] : slow(c); }
};
const ClassTable& classes() {
    static const ClassTable t;
    return t;
}

// Decodes one UTF-8 scalar at s[i..n); ill-formed bytes decode as U+FFFD (what Rust's
// String::from_utf8_lossy hands the tokenizer), one per maximal ill-formed prefix byte.
uint32_t decode_utf8(const unsigned char* s, size_t n, size_t& i) {
    const unsigned char b0 = s[i];
    if (b0 < 0x80) { ++i; return b0; }
    int need = 0;
    uint32_t cp = 0, min = 0;
    if ((b0 & 0xE0) == 0xC0) { need = 1; cp = b0 & 0x1F; min = 0x80; }
    else if ((b0 & 0xF0) == 0xE0) { need = 2; cp = b0 & 0x0F; min = 0x800; }
    else if ((b0 & 0xF8) == 0xF0) { need = 3; cp = b0 & 0x07; min = 0x10000; }
    else { ++i; return 0xFFFD; }
    if (i + (size_t)need >= n) { ++i; return 0xFFFD; }  // truncated sequence
    for (int k = 1; k <= need; ++k) {
        const unsigned char b = s[i + k];
        if ((b & 0xC0) != 0x80) { ++i; return 0xFFFD; }
        cp = (cp << 6) | (b & 0x3F);
    }
    if (cp < min || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) { ++i; return 0xFFFD; }
    i += (size_t)need + 1;
    return cp;
}

void append_utf8(std::string& out, uint32_t c) {
    if (c < 0x80) out.push_back((char)c);
    else if (c < 0x800) { out.push_back((char)(0xC0 | (c >> 6))); out.push_back((char)(0x80 | (c & 0x3F))); }
    else if (c < 0x10000) {
        out.push_back((char)(0xE0 | (c >> 12))); out.push_back((char)(0x80 | ((c >> 6) & 0x3F)));
        out.push_back((char)(0x80 | (c & 0x3F)));
    } else {
        out.push_back((char)(0xF0 | (c >> 18))); out.push_back((char)(0x80 | ((c >> 12) & 0x3F)));
        out.push_back((char)(0x80 | ((c >> 6) & 0x3F))); out.push_back((char)(0x80 | (c & 0x3F)));
    }
}

inline uint64_t fnv_step(uint64_t h, unsigned char b) { return (h ^ b) * 1099511628211ull; }
constexpr uint64_t kFnvBasis = 1469598103934665603ull;

}  // namespace

struct cs_tokenizer {
    // open-addressing table over the vocabulary: key bytes live in `pool`
    struct Slot { uint64_t hash; uint32_t off; uint32_t len; int32_t id; };
    std::vector<Slot> slots;
    std::string pool;
    uint32_t size = 0;
    uint32_t slot_mask = 0;
    bool lowercase = true;
    uint32_t max_length = 512;
    uint32_t max_chars = 100;  // WordPiece max_input_chars_per_word
    int32_t unk = -1, cls = -1, sep = -1, pad = -1;
    uint64_t prefix_hash = kFnvBasis;  // FNV state after "##"
    uint32_t max_token_bytes = 0;
    struct Special { std::string text; int32_t id; };
    std::vector<Special> specials;  // matched verbatim in the raw text, longest first
    // a SentencePiece-unigram tokenizer.json (unigram.cpp) instead of the WordPiece tables above: cls / sep / pad are then
    // its <s> / </s> / <pad>, size its vocabulary
    std::shared_ptr<cs::UnigramEngine> unigram;
    // ... or a byte-level BPE tokenizer.json (bpe.cpp): cls / sep are its <bos> / <eos> (-1: the file's template has none)
    std::shared_ptr<cs::BpeEngine> bpe;

    int32_t find(uint64_t h, const char* a, uint32_t alen, const char* b, uint32_t blen) const {
        // key = a ++ b (a is the optional "##")
        for (uint32_t i = (uint32_t)h & slot_mask;; i = (i + 1) & slot_mask) {
            const Slot& s = slots[i];
            if (s.id < 0) return -1;
            // (a / b may be null with a zero length: memcmp's arguments must not be, whatever the length)
            if (s.hash == h && s.len == alen + blen && (alen == 0 || std::memcmp(pool.data() + s.off, a, alen) == 0) &&
                (blen == 0 || std::memcmp(pool.data() + s.off + alen, b, blen) == 0))
                return s.id;
        }
    }
    int32_t find(const std::string& k) const {
        uint64_t h = kFnvBasis;
        for (unsigned char c : k) h = fnv_step(h, c);
        return find(h, k.data(), (uint32_t)k.size(), nullptr, 0);
    }
};

namespace {

struct Encoder {
    const cs_tokenizer& t;
    std::vector<int32_t>& ids;
    std::string word;                 // current pre-token, UTF-8
    std::vector<uint32_t> char_off;   // byte offset of each char of `word` (+ end)
    std::vector<uint64_t> hashes;     // scratch: prefix hashes from one start position
    uint32_t limit;                   // stop once this many ids exist (truncation)

    Encoder(const cs_tokenizer& tok, std::vector<int32_t>& out, uint32_t lim) : t(tok), ids(out), limit(lim) {}

    bool full() const { return ids.size() >= limit; }

    // WordPiece over the finished pre-token in `word`.
    void flush_word() {
        if (word.empty()) return;
        const uint32_t nchar = (uint32_t)char_off.size();
        char_off.push_back((uint32_t)word.size());
        if (nchar > t.max_chars) {
            ids.push_back(t.unk);
        } else {
            const size_t mark = ids.size();
            uint32_t start = 0;
            bool bad = false;
            while (start < nchar) {
                // prefix hashes of word[start..end) for every end, seeded with "##" past the first piece
                uint64_t h = start ? t.prefix_hash : kFnvBasis;
                hashes.clear();
                const uint32_t b0 = char_off[start];
                uint32_t e = start;
                uint32_t bytes_cap = b0 + t.max_token_bytes;
                while (e < nchar && char_off[e + 1] <= bytes_cap) {
                    for (uint32_t b = char_off[e]; b < char_off[e + 1]; ++b) h = fnv_step(h, (unsigned char)word[b]);
                    hashes.push_back(h);
                    ++e;
                }
                int32_t found = -1;
                uint32_t end = e;
                for (; end > start; --end) {
                    found = t.find(hashes[end - start - 1], "##", start ? 2u : 0u, word.data() + b0,
                                   char_off[end] - b0);
                    if (found >= 0) break;
                }
                if (found < 0) { bad = true; break; }
                ids.push_back(found);
                start = end;
            }
            if (bad) {  // the whole word becomes [UNK]
                ids.resize(mark);
                ids.push_back(t.unk);
            }
        }
        word.clear();
        char_off.clear();
    }

    void push_char(uint32_t c) {
        char_off.push_back((uint32_t)word.size());
        append_utf8(word, c);
    }

    // One normalised code point into the pre-tokenizer.
    void feed(uint32_t c, uint8_t cls) {
        if (cls & kClsSpace) { flush_word(); return; }
        if (cls & kClsPunct) {
            flush_word();
            push_char(c);
            flush_word();
            return;
        }
        push_char(c);
    }

    // BertNormalizer + BertPreTokenizer + WordPiece over raw[lo, hi).
    void segment(const unsigned char* raw, size_t lo, size_t hi) {
        const ClassTable& ct = classes();
        size_t i = lo;
        while (i < hi && !full()) {
            const uint32_t c = decode_utf8(raw, hi, i);
            const uint8_t cls = ct.get(c);
            if (cls & kClsRemove) continue;            // clean_text: NUL, U+FFFD, control/format/private use
            if (cls & kClsSpace) { flush_word(); continue; }  // whitespace -> ' ' -> split
            if (cls & kClsCjk) {                       // handle_chinese_chars: ' c ' (folding below still applies)
                flush_word();
            }
            if (t.lowercase && (cls & kClsFold)) {     // strip_accents (NFD, drop Mn) then lowercase
                if (c >= 0xAC00 && c <= 0xD7A3) {      // Hangul syllable -> conjoining jamo
                    const uint32_t s = c - 0xAC00;
                    feed(0x1100 + s / 588, 0);
                    feed(0x1161 + (s % 588) / 28, 0);
                    if (s % 28) feed(0x11A7 + s % 28, 0);
                } else {
                    const FoldEntry* f = find_fold(c);
                    for (uint32_t k = 0; k < f->len; ++k) {
                        const uint32_t d = kFoldPool[f->off + k];
                        feed(d, ct.get(d) & (kClsSpace | kClsPunct));
                    }
                }
            } else {
                feed(c, cls);
            }
            if (cls & kClsCjk) flush_word();
        }
        flush_word();
    }

    void encode(const unsigned char* raw, size_t n) {
        // added special tokens are cut out of the raw text first (leftmost, longest)
        size_t seg = 0, i = 0;
        if (!t.specials.empty()) {
            while (i < n && !full()) {
                const cs_tokenizer::Special* hit = nullptr;
                for (const auto& sp : t.specials)
                    if (sp.text.size() <= n - i && std::memcmp(raw + i, sp.text.data(), sp.text.size()) == 0) {
                        hit = &sp;
                        break;
                    }
                if (!hit) { ++i; continue; }
                segment(raw, seg, i);
                if (!full()) ids.push_back(hit->id);
                i += hit->text.size();
                seg = i;
            }
        }
        if (!full()) segment(raw, seg, n);
        if (ids.size() > limit) ids.resize(limit);
    }
};

int32_t build_vocab(cs_tokenizer* t, const char* vocab, uint64_t bytes) {
    // vocab.txt: one token per line, id = line number ('\n' or "\r\n"; a final newline is optional)
    std::vector<std::pair<uint32_t, uint32_t>> lines;
    uint64_t lo = 0;
    for (uint64_t i = 0; i <= bytes; ++i) {
        if (i == bytes || vocab[i] == '\n') {
            if (i == bytes && lo == bytes) break;
            uint64_t hi = i;
            if (hi > lo && vocab[hi - 1] == '\r') --hi;
            lines.emplace_back((uint32_t)lo, (uint32_t)(hi - lo));
            lo = i + 1;
        }
    }
    if (lines.empty()) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: empty vocabulary");
    uint32_t cap = 16;
    while (cap < lines.size() * 2) cap <<= 1;
    t->slots.assign(cap, cs_tokenizer::Slot{0, 0, 0, -1});
    t->slot_mask = cap - 1;
    t->pool.assign(vocab, bytes);
    for (size_t id = 0; id < lines.size(); ++id) {
        uint64_t h = kFnvBasis;
        const char* p = t->pool.data() + lines[id].first;
        for (uint32_t k = 0; k < lines[id].second; ++k) h = fnv_step(h, (unsigned char)p[k]);
        uint32_t i = (uint32_t)h & t->slot_mask;
        bool dup = false;
        for (;; i = (i + 1) & t->slot_mask) {
            cs_tokenizer::Slot& s = t->slots[i];
            if (s.id < 0) break;
            if (s.hash == h && s.len == lines[id].second && std::memcmp(t->pool.data() + s.off, p, s.len) == 0) {
                s.id = (int32_t)id;  // a later line wins, as in a HashMap built by insertion
                dup = true;
                break;
            }
        }
        if (dup) continue;
        t->slots[i] = cs_tokenizer::Slot{h, lines[id].first, lines[id].second, (int32_t)id};
        t->max_token_bytes = std::max(t->max_token_bytes, lines[id].second);
        ++t->size;
    }
    t->prefix_hash = fnv_step(fnv_step(kFnvBasis, '#'), '#');
    return CS_OK;
}

uint32_t tokenizer_threads(uint32_t n_texts) {
    static const uint32_t conf = [] {
        if (const char* e = std::getenv("CS_TOKENIZER_THREADS")) {
            const long v = std::strtol(e, nullptr, 10);
            if (v > 0) return (uint32_t)v;
        }
        const uint32_t hw = std::thread::hardware_concurrency();
        return std::max(1u, std::min(hw ? hw : 1u, 16u));
    }();
    return std::max(1u, std::min(conf, n_texts / 8));  // a thread per >= 8 texts
}

}  // namespace

namespace cs {

// Tokenises texts [0, n) into per-text id lists ([CLS] ... [SEP], truncated to max_length).
void tokenize_texts(const cs_tokenizer* t, const char* utf8, const uint64_t* offsets, uint32_t n,
                    uint32_t max_length, std::vector<std::vector<int32_t>>& out) {
    out.assign(n, {});
    const uint32_t body = max_length >= 2 ? max_length - 2 : 0;
    auto work = [&](uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; ++i) {
            std::vector<int32_t>& ids = out[i];
            if (t->unigram) {
                t->unigram->encode(utf8 + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), body, ids);
                continue;
            }
            if (t->bpe) {
                const uint32_t sp = t->bpe->specials();
                t->bpe->encode(utf8 + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), max_length >= sp ? max_length - sp : 0, ids);
                continue;
            }
            ids.push_back(t->cls);
            Encoder enc(*t, ids, body + 1);
            enc.encode(reinterpret_cast<const unsigned char*>(utf8) + offsets[i], (size_t)(offsets[i + 1] - offsets[i]));
            ids.push_back(t->sep);
        }
    };
    const uint32_t nt = tokenizer_threads(n);
    if (nt <= 1) { work(0, n); return; }
    std::vector<std::thread> th;
    std::atomic<uint32_t> next{0};
    const uint32_t grain = 4;
    for (uint32_t k = 0; k < nt; ++k)
        th.emplace_back([&] {
            for (;;) {
                const uint32_t lo = next.fetch_add(grain);
                if (lo >= n) break;
                work(lo, std::min(n, lo + grain));
            }
        });
    for (auto& x : th) x.join();
}

int32_t tokenizer_from_bpe(BpeSpec&& spec, uint32_t max_length, cs_tokenizer** out) {
    if (!out) return fail(CS_ERR_BAD_ARG, "null out pointer");
    *out = nullptr;
    if (max_length < 2) return fail(CS_ERR_BAD_ARG, "max_length %u leaves no room for <s> and </s>", max_length);
    std::shared_ptr<BpeEngine> eng;
    CS_TRY(BpeEngine::create(std::move(spec), &eng));
    cs_tokenizer* t = new (std::nothrow) cs_tokenizer();
    if (!t) return fail(CS_ERR_OOM, "out of host memory");
    t->bpe = eng;
    t->lowercase = false;
    t->max_length = max_length;
    t->size = eng->vocab_size();
    t->cls = eng->bos();
    t->sep = eng->eos();
    t->pad = eng->pad() >= 0 ? eng->pad() : 0;
    t->unk = -1;
    *out = t;
    return CS_OK;
}

int32_t tokenizer_from_unigram(UnigramSpec&& spec, uint32_t max_length, cs_tokenizer** out) {
    if (!out) return fail(CS_ERR_BAD_ARG, "null out pointer");
    *out = nullptr;
    if (max_length < 2) return fail(CS_ERR_BAD_ARG, "max_length %u leaves no room for <s> and </s>", max_length);
    std::shared_ptr<UnigramEngine> eng;
    CS_TRY(UnigramEngine::create(std::move(spec), &eng));
    cs_tokenizer* t = new (std::nothrow) cs_tokenizer();
    if (!t) return fail(CS_ERR_OOM, "out of host memory");
    t->unigram = eng;
    t->lowercase = false;
    t->max_length = max_length;
    t->size = eng->vocab_size();
    t->cls = eng->bos();
    t->sep = eng->eos();
    t->pad = eng->pad();
    t->unk = -1;
    *out = t;
    return CS_OK;
}

}  // namespace cs

extern "C" {

int32_t cs_tokenizer_create(const char* vocab, uint64_t vocab_bytes, int32_t lowercase, uint32_t max_length,
                            cs_tokenizer** out) {
    if (!out) return fail(CS_ERR_BAD_ARG, "null out pointer");
    *out = nullptr;
    if (!vocab || vocab_bytes == 0 || vocab_bytes > 0xFFFFFFFFull)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: empty vocabulary");
    if (max_length < 2) return fail(CS_ERR_BAD_ARG, "max_length %u leaves no room for [CLS] and [SEP]", max_length);
    cs_tokenizer* t = new (std::nothrow) cs_tokenizer();
    if (!t) return fail(CS_ERR_OOM, "out of host memory");
    t->lowercase = lowercase != 0;
    t->max_length = max_length;
    const int32_t st = build_vocab(t, vocab, vocab_bytes);
    if (st != CS_OK) { delete t; return st; }
    t->unk = t->find("[UNK]");
    t->cls = t->find("[CLS]");
    t->sep = t->find("[SEP]");
    t->pad = t->find("[PAD]");
    if (t->unk < 0 || t->cls < 0 || t->sep < 0 || t->pad < 0) {
        delete t;
        return fail(CS_ERR_BAD_ARG,
                    "Failed to initialize embedding model: vocabulary lacks one of [PAD] [UNK] [CLS] [SEP]");
    }
    for (const char* s : {"[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"}) {
        const int32_t id = t->find(s);
        if (id >= 0) t->specials.push_back({s, id});
    }
    std::stable_sort(t->specials.begin(), t->specials.end(),
                     [](const cs_tokenizer::Special& a, const cs_tokenizer::Special& b) { return a.text.size() > b.text.size(); });
    *out = t;
    return CS_OK;
}

int32_t cs_tokenizer_create_from_file(const char* vocab_path, int32_t lowercase, uint32_t max_length,
                                      cs_tokenizer** out) {
    if (!out) return fail(CS_ERR_BAD_ARG, "null out pointer");
    *out = nullptr;
    if (!vocab_path) return fail(CS_ERR_BAD_ARG, "null vocabulary path");
    FILE* f = std::fopen(vocab_path, "rb");
    if (!f) return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: cannot open %s", vocab_path);
    std::string buf;
    char tmp[1 << 16];
    size_t got;
    while ((got = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.append(tmp, got);
    std::fclose(f);
    return cs_tokenizer_create(buf.data(), buf.size(), lowercase, max_length, out);
}

void cs_tokenizer_destroy(cs_tokenizer* t) { delete t; }

uint32_t cs_tokenizer_vocab_size(const cs_tokenizer* t) { return t ? t->size : 0; }
uint32_t cs_tokenizer_max_length(const cs_tokenizer* t) { return t ? t->max_length : 0; }
int32_t cs_tokenizer_pad_id(const cs_tokenizer* t) { return t ? t->pad : -1; }

int32_t cs_tokenizer_token_to_id(const cs_tokenizer* t, const char* token) {
    if (t && token && t->unigram) return t->unigram->token_to_id(std::string(token));
    if (t && token && t->bpe) return t->bpe->token_to_id(std::string(token));
    return (t && token) ? t->find(std::string(token)) : -1;
}

int32_t cs_tokenizer_encode_batch(const cs_tokenizer* t, const char* utf8, const uint64_t* offsets, uint32_t n,
                                  uint32_t max_length, int32_t* ids, int32_t* mask, uint32_t row_stride,
                                  uint32_t* out_len) {
    if (!t) return fail(CS_ERR_BAD_ARG, "null tokenizer handle");
    if (n && (!utf8 || !offsets)) return fail(CS_ERR_BAD_ARG, "null text buffer");
    if (max_length == 0) max_length = t->max_length;
    if (max_length < 2) return fail(CS_ERR_BAD_ARG, "max_length %u leaves no room for [CLS] and [SEP]", max_length);
    for (uint32_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(CS_ERR_BAD_ARG, "text offsets must be non-decreasing");
    std::vector<std::vector<int32_t>> enc;
    cs::tokenize_texts(t, utf8, offsets, n, max_length, enc);
    uint32_t L = 0;
    for (const auto& e : enc) L = std::max<uint32_t>(L, (uint32_t)e.size());
    if (out_len) *out_len = L;
    if (!ids && !mask) return CS_OK;  // length query
    if (row_stride < L)
        return fail(CS_ERR_BAD_ARG, "row_stride %u is shorter than the batch's longest sequence %u", row_stride, L);
    for (uint32_t i = 0; i < n; ++i) {
        const auto& e = enc[i];
        for (uint32_t j = 0; j < row_stride; ++j) {
            const bool live = j < e.size();
            if (ids) ids[(size_t)i * row_stride + j] = live ? e[j] : t->pad;
            if (mask) mask[(size_t)i * row_stride + j] = live ? 1 : 0;
        }
    }
    return CS_OK;
}

}  // extern "C"
