// bpe.hpp — the byte-level BPE text pipeline of the `tokenizers` crate (what fastembed loads for the registry's
// JinaEmbeddingsV2BaseCode: jinaai/jina-embeddings-v2-base-code ships a RoBERTa-style byte-level BPE tokenizer.json;
// /root/reference/src/embed/embedder.rs:40-41, :112) restated on the host: csrc/bpe.cpp.  A BpeSpec is what checkpoint.cpp
// reads out of tokenizer.json; the engine turns text into ids exactly as the library does for the components it knows
// (anything else in the file is refused at load).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

struct cs_tokenizer;

namespace cs {

struct BpeSpec {
    std::vector<std::pair<std::string, int32_t>> vocab;            // model.vocab {token: id}
    std::vector<std::pair<std::string, std::string>> merges;       // model.merges, in rank order
    std::string unk_token;                                         // "" = none (an unknown symbol is dropped, as the library does)
    bool fuse_unk = false, ignore_merges = false;
    bool nfc = false;                                              // normalizer {"type": "NFC"}: applied to the text between added tokens
    struct Pre {
        enum Kind { BYTE_LEVEL, DIGITS } kind = BYTE_LEVEL;
        bool add_prefix_space = false, use_regex = true;           // BYTE_LEVEL
        bool individual_digits = false;                            // DIGITS
    };
    std::vector<Pre> pres;                                         // must end in (or be) a BYTE_LEVEL step
    // added tokens (AddedVocabulary of the library).  normalized = false (special tokens as a rule): matched verbatim in the RAW
    // text, first.  normalized = true (ModernBERT's |||IP_ADDRESS||| ... and its runs of 2-24 spaces — plain tokens that indented
    // code is full of): matched in what is left, AFTER that text's normalisation (split_normalized_trie), leftmost-longest.
    struct Added { std::string text; int32_t id = -1; bool lstrip = false, rstrip = false, normalized = false; };
    std::vector<Added> added;
    int32_t bos = -1, eos = -1, pad = -1;                          // <bos> $A <eos> (RobertaProcessing / TemplateProcessing); -1 = none
};

class BpeEngine {
public:
    static int32_t create(BpeSpec&& spec, std::shared_ptr<BpeEngine>* out);
    // appends bos (if any), at most body_max ids of the text, eos (if any)
    void encode(const char* utf8, size_t n, uint32_t body_max, std::vector<int32_t>& ids) const;
    int32_t token_to_id(const std::string& s) const;
    uint32_t vocab_size() const { return vocab_size_; }
    int32_t pad() const { return spec_.pad; }
    int32_t bos() const { return spec_.bos; }
    int32_t eos() const { return spec_.eos; }
    uint32_t specials() const { return (spec_.bos >= 0) + (spec_.eos >= 0); }

private:
    BpeSpec spec_;
    uint32_t vocab_size_ = 0;
    std::unordered_map<std::string, int32_t> ids_;                 // token -> id
    std::unordered_map<uint64_t, std::pair<uint32_t, int32_t>> merge_;  // (left id, right id) -> (rank, merged id)
    int32_t byte_id_[256];                                         // id of the one-character token of byte b (-1: not in the vocabulary)
    int32_t unk_id_ = -1;

    void encode_word(const std::string& bytes, std::vector<int32_t>& ids) const;
    void encode_segment(const char* p, size_t n, bool at_text_start, std::vector<int32_t>& ids) const;
    void encode_piece(std::string&& text, std::vector<int32_t>& ids) const;
    // `text` cut at its leftmost-longest added tokens of one kind: piece(lo, hi) for every stretch between them, the token's id behind it
    template <class F> void split_on_added(const std::string& text, bool normalized, std::vector<int32_t>& ids, F&& piece) const;
    bool has_normalized_added_ = false;
};

// a cs_tokenizer handle (tokenizer.cpp) around an engine built from `spec`
int32_t tokenizer_from_bpe(BpeSpec&& spec, uint32_t max_length, struct ::cs_tokenizer** out);

}  // namespace cs
