// embedder.hip — cs_embedder_* (placeholder until the encoder kernels land).
#include "common.hpp"

using namespace cs;

struct cs_embedder { int device; };

extern "C" {

void cs_bert_config_bge_small(cs_bert_config* cfg) {
    if (!cfg) return;
    cfg->vocab_size = 30522; cfg->hidden = 384; cfg->layers = 12; cfg->heads = 12;
    cfg->intermediate = 1536; cfg->max_position = 512; cfg->type_vocab_size = 2;
    cfg->layer_norm_eps = 1e-12f; cfg->pooling = CS_POOL_CLS;
}
uint64_t cs_bert_param_count(const cs_bert_config*) { return 0; }
int32_t cs_embedder_create(const cs_bert_config*, const float*, uint64_t, int32_t, cs_embedder**) {
    return fail(CS_ERR_UNSUPPORTED, "encoder not built yet");
}
void cs_embedder_destroy(cs_embedder*) {}
uint32_t cs_embedder_dim(const cs_embedder*) { return 0; }
int32_t cs_embedder_embed_ids(cs_embedder*, const int32_t*, const int32_t*, uint64_t, uint32_t, uint32_t, float*, const volatile int32_t*) {
    return fail(CS_ERR_UNSUPPORTED, "encoder not built yet");
}
int32_t cs_embedder_embed_ids_device(cs_embedder*, const int32_t*, const int32_t*, uint64_t, uint32_t, uint32_t, float*, const volatile int32_t*) {
    return fail(CS_ERR_UNSUPPORTED, "encoder not built yet");
}
int32_t cs_embedder_last_hidden(cs_embedder*, float*, uint64_t) { return fail(CS_ERR_UNSUPPORTED, "encoder not built yet"); }
int32_t cs_embedder_profile_read(cs_embedder*, double*, uint64_t*, int32_t) { return fail(CS_ERR_UNSUPPORTED, "encoder not built yet"); }
}
