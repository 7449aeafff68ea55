// unigram.hpp — the SentencePiece-unigram text pipeline of the `tokenizers` crate (what fastembed loads for the registry's
// XLM-R-vocabulary models: intfloat/multilingual-e5-small, paraphrase-multilingual-MiniLM-L12-v2;
// /root/reference/src/embed/embedder.rs:58,150) restated on the host: csrc/unigram.cpp.  A UnigramSpec is what
// checkpoint.cpp reads out of tokenizer.json; the engine turns text into ids exactly as the library does for the
// components it knows (anything else in the file is refused at load).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

struct cs_tokenizer;

namespace cs {

struct UnigramSpec {
    std::vector<std::pair<std::string, double>> vocab;  // model.vocab: [piece, log-probability], id = position
    int32_t unk_id = -1;
    struct Norm {
        enum Kind { PRECOMPILED, REPLACE_SPACES /* Regex " {2,}" */, REPLACE_STRING, STRIP } kind = PRECOMPILED;
        std::string blob;              // PRECOMPILED: the decoded precompiled_charsmap
        std::string pattern, content;  // REPLACE_*
        bool left = false, right = false;  // STRIP
    };
    std::vector<Norm> norms;
    struct Pre {
        enum Kind { WHITESPACE_SPLIT, METASPACE } kind = METASPACE;
        std::string replacement;  // METASPACE (one character, UTF-8)
        int prepend = 1;          // 0 never, 1 always, 2 first
        bool split = true;
    };
    std::vector<Pre> pres;
    struct Added { std::string text; int32_t id = -1; bool lstrip = false, rstrip = false; };
    std::vector<Added> added;  // matched verbatim in the raw text
    int32_t bos = -1, eos = -1, pad = -1;  // TemplateProcessing "<s> $A </s>", padding id
};

class UnigramEngine {
public:
    // CS_OK or an error (through fail()): the spec's pieces must be valid for the engine
    static int32_t create(UnigramSpec&& spec, std::shared_ptr<UnigramEngine>* out);
    // appends bos, at most body_max ids of the text, eos
    void encode(const char* utf8, size_t n, uint32_t body_max, std::vector<int32_t>& ids) const;
    int32_t token_to_id(const std::string& s) const;
    uint32_t vocab_size() const { return (uint32_t)spec_.vocab.size(); }
    int32_t pad() const { return spec_.pad; }
    int32_t bos() const { return spec_.bos; }
    int32_t eos() const { return spec_.eos; }

private:
    UnigramSpec spec_;
    double min_score_ = 0.0;
    // byte trie over the pieces: node n's children are labels_[begin_[n] .. begin_[n + 1]) (sorted) -> child_[...]; id_[n] >= 0 ends a piece
    std::vector<uint32_t> begin_;
    std::vector<uint8_t> labels_;
    std::vector<uint32_t> child_;
    std::vector<int32_t> id_;
    // Precompiled normaliser: darts-clone double array + the replacement strings
    struct Charsmap { std::vector<uint32_t> trie; std::string normalized; };
    std::vector<Charsmap> maps_;  // one per PRECOMPILED step, in order

    void build_trie();
    bool transform(const Charsmap& m, const char* chunk, size_t n, const char** out, size_t* out_n) const;
    void normalize(std::string& s) const;
    void model_encode(const std::string& piece, std::vector<int32_t>& ids) const;
    void encode_segment(const char* p, size_t n, bool at_text_start, std::vector<int32_t>& ids) const;
};

// a cs_tokenizer handle (tokenizer.cpp) around an engine built from `spec`
int32_t tokenizer_from_unigram(UnigramSpec&& spec, uint32_t max_length, struct ::cs_tokenizer** out);

}  // namespace cs
