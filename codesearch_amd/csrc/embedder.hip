// embedder.hip — cs_embedder_*: the device half of the reference's FastEmbedder
// (/root/reference/src/embed/embedder.rs:201-322).  Starts from token ids (tokenisation is a
// host concern); runs the encoder kernels of encoder.hip; returns pooled, L2-normalised
// embeddings.  Mini-batching and the shutdown poll follow embed_batch_chunked
// (embedder.rs:266-295).
#include "embedder_state.hpp"
#ifdef CS_DIAGNOSTICS
#include "../../include/codesearch_gpu_diag.h"
#endif

using namespace cs;
using namespace cs::emb;

extern "C" {

void cs_bert_config_bge_small(cs_bert_config* cfg) {
    if (!cfg) return;
    cfg->vocab_size = 30522; cfg->hidden = 384; cfg->layers = 12; cfg->heads = 12;
    cfg->intermediate = 1536; cfg->max_position = 512; cfg->type_vocab_size = 2;
    cfg->layer_norm_eps = 1e-12f; cfg->pooling = CS_POOL_CLS;
    cfg->arch = CS_ARCH_BERT; cfg->rotary_base = 0.0f;
    cfg->rotary_base_local = 0.0f; cfg->local_window = 0; cfg->global_every = 0;
}

uint64_t cs_bert_param_count(const cs_bert_config* cfg) {
    if (!cfg) return 0;
    cs_bert_offsets off;
    cs_bert_layout(cfg, &off);
    return off.total;
}

static int32_t create_impl(const cs_bert_config* cfg, const float* params, uint64_t seed, const float* wscale,
                           int32_t device, cs_embedder** out) {
    if (!out) return fail(CS_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (!cfg) return fail(CS_ERR_BAD_ARG, "cfg is null");
    if (cfg->hidden == 0 || cfg->heads == 0 || cfg->hidden % cfg->heads || cfg->layers == 0 ||
        cfg->vocab_size == 0 || cfg->max_position == 0 || cfg->type_vocab_size == 0)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: inconsistent config");
    if (cfg->hidden != 384 && cfg->hidden != 768 && cfg->hidden != 1024)
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: hidden size %u not supported", cfg->hidden);
    // head_dim 64: BGE-base / BGE-large / mxbai-large
    if (cfg->hidden / cfg->heads != 32 && cfg->hidden / cfg->heads != 64)
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: head_dim %u not supported (32 or 64)",
                    cfg->hidden / cfg->heads);
    if (cfg->intermediate % 128 || cfg->hidden % 128)
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: sizes must be multiples of 128");
    if (cfg->pooling != CS_POOL_CLS && cfg->pooling != CS_POOL_MEAN)
        return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: unknown pooling %d", cfg->pooling);
    if (cfg->arch != CS_ARCH_BERT && !cs_arch_gated(cfg->arch))
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: unknown encoder family %u", cfg->arch);
    if (cs_arch_alibi(cfg->arch) && wscale)
        return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: the dynamic-quantisation mode is not built for "
                    "the JinaBert encoder (create it from the dequantised weights: cs_embedder_create)");
    if (cfg->arch == CS_ARCH_MODERN) {
        if (!(cfg->rotary_base > 1.0f) || !(cfg->rotary_base < 1.0e9f) || !(cfg->rotary_base_local > 1.0f) || !(cfg->rotary_base_local < 1.0e9f))
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: rotary bases %g / %g", (double)cfg->rotary_base,
                        (double)cfg->rotary_base_local);
        if (cfg->global_every == 0 || cfg->local_window == 0)
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: ModernBERT needs global_every and local_window");
        if (wscale) return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: the dynamic-quantisation mode is not built for "
                                "the ModernBERT encoder (create it from the dequantised weights: cs_embedder_create)");
    }
    if (cfg->arch == CS_ARCH_NOMIC) {
        if (!(cfg->rotary_base > 1.0f) || !(cfg->rotary_base < 1.0e9f))
            return fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: rotary base %g", (double)cfg->rotary_base);
        // the quantised export's graph quantises the rotated and gated tensors in places of its own: not restated
        if (wscale) return fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: the dynamic-quantisation mode is not built for "
                                "the Nomic encoder (create it from the dequantised weights: cs_embedder_create)");
    }
    int ndev = 0;
    CS_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(CS_ERR_HIP, "HIP device %d not available (%d visible); there is no CPU fallback", device, ndev);
    DeviceGuard g(device);
    cs_embedder* h = new cs_embedder();
    h->device = device;
    h->cfg = *cfg;
    cs_bert_layout(cfg, &h->off);
    const size_t H = cfg->hidden;
    auto cleanup = [&](int32_t s) { cs_embedder_destroy(h); return s; };
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess)
        return cleanup(fail(CS_ERR_HIP, "could not create stream/events"));
    if (hipMalloc(&h->d_params, h->off.total * sizeof(float)) != hipSuccess ||
        hipMalloc(&h->d_wqkv, (size_t)cfg->layers * 3 * H * H * sizeof(float)) != hipSuccess ||
        hipMalloc(&h->d_bqkv, (size_t)cfg->layers * 3 * H * sizeof(float)) != hipSuccess)
        return cleanup(fail(CS_ERR_OOM, "hipMalloc(parameters) failed"));
    int32_t s = CS_OK;
    if (params) {
        if (hipMemcpy(h->d_params, params, h->off.total * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return cleanup(fail(CS_ERR_HIP, "parameter upload failed"));
    } else {
        s = launch_synth_params(h->d_params, *cfg, seed, h->stream);
        if (s != CS_OK) return cleanup(s);
    }
    // pack query|key|value into one [3H, H] weight and [3H] bias per layer (E2 is one GEMM)
    for (uint32_t l = 0; l < cfg->layers && s == CS_OK; ++l) {
        cs_bert_layer_offsets lo;
        cs_bert_layer_layout(cfg, &h->off, l, &lo);
        const uint64_t w[3] = {lo.q_w, lo.k_w, lo.v_w}, b[3] = {lo.q_b, lo.k_b, lo.v_b};
        for (int i = 0; i < 3; ++i) {
            if (hipMemcpyAsync(h->d_wqkv + ((size_t)l * 3 + i) * H * H, h->d_params + w[i], H * H * sizeof(float),
                               hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
                hipMemcpyAsync(h->d_bqkv + ((size_t)l * 3 + i) * H, h->d_params + b[i], H * sizeof(float),
                               hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
                s = fail(CS_ERR_HIP, "QKV packing failed");
        }
    }
    if (s == CS_OK && cs_arch_alibi(cfg->arch)) {
        // JinaBert's `_get_alibi_head_slopes`: the geometric sequence from 2^(-8 / n) for the closest power of two n below the
        // head count, then every second slope of the doubled set — formed in double as the module forms them in Python floats
        const uint32_t nh = cfg->heads;
        std::vector<float> sl(2 * (size_t)nh);
        uint32_t closest = 1;
        while (closest * 2 <= nh) closest *= 2;
        auto slope = [](uint32_t n, uint32_t i) { const double start = std::exp2(-8.0 / (double)n); return start * std::pow(start, (double)i); };
        for (uint32_t i = 0; i < nh; ++i) sl[i] = (float)(i < closest ? slope(closest, i) : slope(2 * closest, 2 * (i - closest)));
        for (uint32_t i = 0; i < nh; ++i) sl[nh + i] = sl[i] * 1.4426950408889634f;
        if (hipMalloc(&h->d_alibi, sl.size() * sizeof(float)) != hipSuccess) return cleanup(fail(CS_ERR_OOM, "hipMalloc(parameters) failed"));
        if (hipMemcpy(h->d_alibi, sl.data(), sl.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            s = fail(CS_ERR_HIP, "ALiBi slope upload failed");
    }
    if (s == CS_OK && cs_arch_gated(cfg->arch)) {
        const size_t I = cfg->intermediate, half = H / cfg->heads / 2;
        const bool rotary = cfg->arch == CS_ARCH_NOMIC || cfg->arch == CS_ARCH_MODERN;
        if (hipMalloc(&h->d_bup, (size_t)cfg->layers * 2 * I * sizeof(float)) != hipSuccess ||
            (rotary && hipMalloc(&h->d_rope, (size_t)cfg->max_position * half * sizeof(float2)) != hipSuccess) ||
            (cfg->arch == CS_ARCH_MODERN && (hipMalloc(&h->d_rope_local, (size_t)cfg->max_position * half * sizeof(float2)) != hipSuccess ||
                                             hipMalloc(&h->d_zero_row, H * sizeof(float)) != hipSuccess ||
                                             hipMemsetAsync(h->d_zero_row, 0, H * sizeof(float), h->stream) != hipSuccess)))
            return cleanup(fail(CS_ERR_OOM, "hipMalloc(parameters) failed"));
        for (uint32_t l = 0; l < cfg->layers && s == CS_OK; ++l) {
            cs_bert_layer_offsets lo;
            cs_bert_layer_layout(cfg, &h->off, l, &lo);
            // value and gate interleaved in groups of 16 columns, the order of the packed weight (GW_OUT_SWIGLU, encoder.hpp)
            float* bl = h->d_bup + (size_t)l * 2 * I;
            if (hipMemcpy2DAsync(bl, 32 * sizeof(float), h->d_params + lo.up_b, 16 * sizeof(float), 16 * sizeof(float), I / 16,
                                 hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
                hipMemcpy2DAsync(bl + 16, 32 * sizeof(float), h->d_params + lo.gate_b, 16 * sizeof(float), 16 * sizeof(float), I / 16,
                                 hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
                s = fail(CS_ERR_HIP, "feed-forward bias packing failed");
        }
        // the module's cos / sin cache, formed as it forms it: inv_freq_i = 1 / base^(2i / d_h) and pos * inv_freq_i in f32
        std::vector<float2> rope(rotary ? (size_t)cfg->max_position * half : 0);
        const float dh = (float)(2 * half);
        for (int which = 0; which < (cfg->arch == CS_ARCH_MODERN ? 2 : 1) && !rope.empty() && s == CS_OK; ++which) {
            const float base = which ? cfg->rotary_base_local : cfg->rotary_base;
            for (size_t i = 0; i < half; ++i) {
                const float inv_freq = 1.0f / powf(base, (float)(2 * i) / dh);
                for (size_t p = 0; p < cfg->max_position; ++p) {
                    const float ang = (float)p * inv_freq;
                    rope[p * half + i] = make_float2(cosf(ang), sinf(ang));
                }
            }
            if (hipMemcpy(which ? h->d_rope_local : h->d_rope, rope.data(), rope.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
                s = fail(CS_ERR_HIP, "rotary table upload failed");
        }
    }
    // split-f16 copies of the four dense weights of every layer (split_f16.hpp)
    if (s == CS_OK) {
        const SplitLayer sl = split_layer(*cfg);
        const size_t I = cfg->intermediate;
        if (hipMalloc(&h->d_wsplit, (size_t)cfg->layers * sl.total * sizeof(_Float16)) != hipSuccess ||
            hipMalloc(&h->d_flag, sizeof(uint32_t)) != hipSuccess)
            return cleanup(fail(CS_ERR_OOM, "hipMalloc(split weights) failed"));
        if (hipMemsetAsync(h->d_flag, 0, sizeof(uint32_t), h->stream) != hipSuccess) s = fail(CS_ERR_HIP, "memset failed");
        // (a query's transfers from / to pageable memory are staged by the runtime, each a wait of its own: a few pinned KiB instead;
        // without them the pageable path below is taken)
        if (hipHostMalloc(reinterpret_cast<void**>(&h->h_pin), 128 << 10, hipHostMallocDefault) != hipSuccess) { h->h_pin = nullptr; (void)hipGetLastError(); }
        float* d_updown = nullptr;  // CS_ARCH_NOMIC: one layer's fc11 / fc12 rows interleaved in groups of 16, [2I][H]
        if (cs_arch_gated(cfg->arch) && hipMalloc(&d_updown, 2 * I * H * sizeof(float)) != hipSuccess)
            return cleanup(fail(CS_ERR_OOM, "hipMalloc(split weights) failed"));
        for (uint32_t l = 0; l < cfg->layers && s == CS_OK; ++l) {
            cs_bert_layer_offsets lo;
            cs_bert_layer_layout(cfg, &h->off, l, &lo);
            _Float16* ws = h->d_wsplit + (size_t)l * sl.total;
            s = launch_split_rows(h->d_wqkv + (size_t)l * 3 * H * H, ws + sl.qkv, 3 * H, (uint32_t)H, h->d_flag, h->stream);
            if (s == CS_OK) s = launch_split_rows(h->d_params + lo.ao_w, ws + sl.ao, H, (uint32_t)H, h->d_flag, h->stream);
            if (s == CS_OK && cs_arch_gated(cfg->arch)) {
                // one [2I, H] weight: raw output columns 32 u .. 32 u + 15 = fc11's rows 16 u .., the next sixteen fc12's, so that
                // a value and its gate meet in one lane of the product's epilogue (GW_OUT_SWIGLU) and in one line of its output
                const size_t grp = 16 * H * sizeof(float);
                if (hipMemcpy2DAsync(d_updown, 2 * grp, h->d_params + lo.up_w, grp, grp, I / 16, hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
                    hipMemcpy2DAsync(d_updown + 16 * H, 2 * grp, h->d_params + lo.gate_w, grp, grp, I / 16, hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
                    s = fail(CS_ERR_HIP, "feed-forward weight packing failed");
                if (s == CS_OK) s = launch_split_rows(d_updown, ws + sl.up, 2 * I, (uint32_t)H, h->d_flag, h->stream);
            } else if (s == CS_OK) {
                s = launch_split_rows(h->d_params + lo.up_w, ws + sl.up, I, (uint32_t)H, h->d_flag, h->stream);
            }
            if (s == CS_OK) s = launch_split_rows(h->d_params + lo.down_w, ws + sl.down, H, (uint32_t)I, h->d_flag, h->stream);
        }
        if (d_updown) {
            (void)hipStreamSynchronize(h->stream);
            (void)hipFree(d_updown);
        }
        uint32_t wflag = 0;
        if (s == CS_OK && (hipMemcpyAsync(&wflag, h->d_flag, sizeof(wflag), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                           hipStreamSynchronize(h->stream) != hipSuccess))
            s = fail(CS_ERR_HIP, "parameter setup failed");
        if (const char* env = std::getenv("CS_ENCODER_GEMM"))
            h->gemm_mode = (std::strcmp(env, "f32") == 0) ? CS_GEMM_F32 : CS_GEMM_SPLIT_F16;
        if (wflag) h->gemm_mode = CS_GEMM_F32;  // a weight outside the f16 range: exact path only
        if (s == CS_OK && !wflag) {  // may the one-accumulator kernels scale w_hi by 2^11 in f16?
            bool fit = false;
            s = sh_weights_fit_wide(h->d_wsplit, (uint64_t)cfg->layers * sl.total, h->d_flag, &fit, h->stream);
            const char* e = cs_lab_env("CS_GEMM_WIDE");  // "0": never
            h->wide_ok = fit && !(e && e[0] == '0');
        }
        bool denorm_ok = false;  // the split format relies on exact f16-subnormal MFMA inputs
        if (s == CS_OK) s = sh_denorm_selftest(&denorm_ok, h->stream);
        if (s == CS_OK && !denorm_ok) { h->gemm_mode = CS_GEMM_F32; h->split_unavailable = true; }
#ifdef CS_DIAGNOSTICS
        // the one-launch forward of short queries (small_forward.hip: bit-identical to the launch chain and slower, so it lives in the
        // diagnostic library only) reads the layers' pointers from a device table
        if (s == CS_OK && !cs_arch_gated(cfg->arch) && small_forward_supported((uint32_t)H, (uint32_t)I, cfg->heads, 1, 1)) {
            std::vector<SfLayer> tab(cfg->layers);
            for (uint32_t l = 0; l < cfg->layers; ++l) {
                cs_bert_layer_offsets lo;
                cs_bert_layer_layout(cfg, &h->off, l, &lo);
                const _Float16* wl = h->d_wsplit + (size_t)l * sl.total;
                const float* P = h->d_params;
                tab[l] = SfLayer{wl + sl.qkv, wl + sl.ao, wl + sl.up, wl + sl.down, h->d_bqkv + (size_t)l * 3 * H, P + lo.ao_b, P + lo.up_b,
                                 P + lo.down_b, P + lo.ao_ln_g, P + lo.ao_ln_b, P + lo.out_ln_g, P + lo.out_ln_b};
            }
            if (hipMalloc(&h->d_sf_layers, tab.size() * sizeof(SfLayer)) != hipSuccess || hipMalloc(&h->d_sf_sync, 16) != hipSuccess ||
                hipMemcpyAsync(h->d_sf_layers, tab.data(), tab.size() * sizeof(SfLayer), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess)
                s = fail(CS_ERR_OOM, "the one-launch forward's layer table could not be set up");
            if (s == CS_OK && cs_lab_env("CS_SMALL_FORWARD_DEBUG") && hipMalloc(&h->d_sf_dbg, (96 * 3 + 8) * sizeof(uint64_t)) != hipSuccess)
                h->d_sf_dbg = nullptr;
        }
#endif
        if (const char* env = std::getenv("CS_ENCODER_STREAMS")) {
            h->streams_forced = true;
            const int v = std::atoi(env);
            h->n_streams = v >= 4 ? 4 : (v >= 1 ? v : 1);
        }
        for (int i = 0; i + 2 < h->n_streams; ++i)
            if (hipStreamCreateWithFlags(&h->xstreams[i], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&h->xjoin[i], hipEventDisableTiming) != hipSuccess) {
                h->n_streams = 2;
                break;
            }
    }
    // dynamically quantised model: the s8 form of every Linear weight and its column metadata (gemm_q8.hip)
    if (s == CS_OK && wscale) {
        const size_t I = cfg->intermediate, cols = 5 * H + I;
        const Q8Layer ql = q8_layer((uint32_t)H, (uint32_t)I);
        float* d_ws = nullptr;
        uint32_t* d_bad = nullptr;
        if (I > 4 * H || I > 4096) s = fail(CS_ERR_UNSUPPORTED, "Failed to initialize embedding model: intermediate size above 4 x hidden or 4,096");
        if (s == CS_OK && (hipMalloc(&h->d_wq8, (size_t)cfg->layers * ql.total) != hipSuccess ||
                           hipMalloc(&h->d_cmeta, (size_t)cfg->layers * cols * sizeof(Q8ColMeta)) != hipSuccess ||
                           hipMalloc(&h->d_cmeta_tiles, (size_t)cfg->layers * cols * sizeof(Q8ColMeta)) != hipSuccess ||
                           (H % 64 == 0 && I % 64 == 0 && hipMalloc(&h->d_wq8_stages, (size_t)cfg->layers * (H * H + H * I)) != hipSuccess) ||
                           hipMalloc(&d_ws, (size_t)cfg->layers * cols * sizeof(float)) != hipSuccess ||
                           hipMalloc(&d_bad, sizeof(uint32_t)) != hipSuccess))
            s = fail(CS_ERR_OOM, "hipMalloc(quantised weights) failed");
        if (s == CS_OK && (hipMemcpyAsync(d_ws, wscale, (size_t)cfg->layers * cols * sizeof(float), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
                           hipMemsetAsync(d_bad, 0, sizeof(uint32_t), h->stream) != hipSuccess))
            s = fail(CS_ERR_HIP, "quantised weight setup failed");
        for (uint32_t l = 0; l < cfg->layers && s == CS_OK; ++l) {
            cs_bert_layer_offsets lo;
            cs_bert_layer_layout(cfg, &h->off, l, &lo);
            int8_t* wq = h->d_wq8 + (size_t)l * ql.total;
            Q8ColMeta* cm = h->d_cmeta + (size_t)l * cols;
            const float* sc = d_ws + (size_t)l * cols;
            s = launch_q8_pack_weight(h->d_wqkv + (size_t)l * 3 * H * H, sc, h->d_bqkv + (size_t)l * 3 * H, (uint32_t)(3 * H), (uint32_t)H, wq + ql.qkv, cm, d_bad, h->stream);
            if (s == CS_OK) s = launch_q8_pack_weight(h->d_params + lo.ao_w, sc + 3 * H, h->d_params + lo.ao_b, (uint32_t)H, (uint32_t)H, wq + ql.ao, cm + 3 * H, d_bad, h->stream);
            if (s == CS_OK) s = launch_q8_pack_weight(h->d_params + lo.up_w, sc + 4 * H, h->d_params + lo.up_b, (uint32_t)I, (uint32_t)H, wq + ql.up, cm + 4 * H, d_bad, h->stream);
            if (s == CS_OK) s = launch_q8_pack_weight(h->d_params + lo.down_w, sc + 4 * H + I, h->d_params + lo.down_b, (uint32_t)H, (uint32_t)I, wq + ql.down, cm + 4 * H + I, d_bad, h->stream);
            // QKV's and FFN-up's columns once more as structure-of-arrays tiles (the slab kernel's layout; column n of the layer at word 4 n)
            uint32_t* ct = h->d_cmeta_tiles + ((size_t)l * cols) * 4;
            if (s == CS_OK && (3 * H) % 128 == 0) s = launch_q8_cmeta_tiles(cm, (uint32_t)(3 * H), ct, h->stream);
            if (s == CS_OK && I % 128 == 0 && H % 128 == 0) s = launch_q8_cmeta_tiles(cm + 4 * H, (uint32_t)I, ct + 4 * 4 * H, h->stream);
            // out-proj and FFN-down once more, stage-major (the LayerNorm-fused products' weight stream)
            if (s == CS_OK && h->d_wq8_stages) {
                int8_t* ws = h->d_wq8_stages + (size_t)l * (H * H + H * I);
                s = launch_q8_stage_major(wq + ql.ao, (uint32_t)H, (uint32_t)H, ws, h->stream);
                if (s == CS_OK) s = launch_q8_stage_major(wq + ql.down, (uint32_t)H, (uint32_t)I, ws + H * H, h->stream);
            }
        }
        uint32_t bad = 0;
        if (s == CS_OK && (hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                           hipStreamSynchronize(h->stream) != hipSuccess))
            s = fail(CS_ERR_HIP, "quantised weight setup failed");
        if (d_ws) (void)hipFree(d_ws);
        if (d_bad) (void)hipFree(d_bad);
        if (s == CS_OK && bad)
            s = fail(CS_ERR_BAD_ARG, "Failed to initialize embedding model: %s", (bad & 2)
                         ? "a Linear weight is not an integer multiple of its column scale (not a quantised block)"
                         : "the integers of a weight column span more than 8 bits");
        if (s == CS_OK) {
            h->quantized = true;
            const char* env = std::getenv("CS_ENCODER_QUANT");  // "0": the f32 graph of the quantised weights
            // an explicit CS_ENCODER_GEMM is honoured for quantised models too: "f32" / "split" select the f32
            // graph of the dequantised weights on that arithmetic, exactly as CS_ENCODER_QUANT=0 does
            const bool explicit_gemm = std::getenv("CS_ENCODER_GEMM") != nullptr;
            if (!h->split_unavailable && !explicit_gemm && !(env && env[0] == '0')) h->gemm_mode = CS_GEMM_Q8_DYNAMIC;
        }
    }
    if (s == CS_OK && hipStreamSynchronize(h->stream) != hipSuccess) s = fail(CS_ERR_HIP, "parameter setup failed");
    if (s != CS_OK) return cleanup(s);
    *out = h;
    return CS_OK;
}

int32_t cs_embedder_create(const cs_bert_config* cfg, const float* params, uint64_t seed,
                           int32_t device, cs_embedder** out) {
    return create_impl(cfg, params, seed, nullptr, device, out);
}

uint64_t cs_bert_quant_columns(const cs_bert_config* cfg) {
    return cfg ? 5 * (uint64_t)cfg->hidden + cfg->intermediate : 0;
}

int32_t cs_embedder_create_quantized(const cs_bert_config* cfg, const float* params, const float* wscale,
                                     uint64_t n_wscale, int32_t device, cs_embedder** out) {
    if (out) *out = nullptr;
    if (!cfg || !params || !wscale) return fail(CS_ERR_BAD_ARG, "cs_embedder_create_quantized: null argument");
    if (n_wscale != (uint64_t)cfg->layers * cs_bert_quant_columns(cfg))
        return fail(CS_ERR_BAD_ARG, "cs_embedder_create_quantized: %llu column scales given, %llu expected (layers x (5 hidden + intermediate))",
                    (unsigned long long)n_wscale, (unsigned long long)((uint64_t)cfg->layers * cs_bert_quant_columns(cfg)));
    return create_impl(cfg, params, 0, wscale, device, out);
}

void cs_embedder_destroy(cs_embedder* h) {
    if (!h) return;
    DeviceGuard g(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    for (int i = 0; i < 2; ++i) {
        if (h->xstreams[i]) { (void)hipStreamSynchronize(h->xstreams[i]); (void)hipStreamDestroy(h->xstreams[i]); }
        if (h->xjoin[i]) (void)hipEventDestroy(h->xjoin[i]);
    }
    free_workspace(h);
    if (h->d_params) (void)hipFree(h->d_params);
    if (h->d_wqkv) (void)hipFree(h->d_wqkv);
    if (h->d_bqkv) (void)hipFree(h->d_bqkv);
    if (h->d_bup) (void)hipFree(h->d_bup);
    if (h->d_rope) (void)hipFree(h->d_rope);
    if (h->d_alibi) (void)hipFree(h->d_alibi);
    if (h->d_rope_local) (void)hipFree(h->d_rope_local);
    if (h->d_zero_row) (void)hipFree(h->d_zero_row);
    if (h->d_wsplit) (void)hipFree(h->d_wsplit);
    if (h->d_flag) (void)hipFree(h->d_flag);
    if (h->h_pin) (void)hipHostFree(h->h_pin);
    if (h->d_sf_layers) (void)hipFree(h->d_sf_layers);
    if (h->d_sf_sync) (void)hipFree(h->d_sf_sync);
    if (h->d_sf_dbg) (void)hipFree(h->d_sf_dbg);
    if (h->d_sp_ws) (void)hipFree(h->d_sp_ws);
    if (h->d_wq8) (void)hipFree(h->d_wq8);
    if (h->d_cmeta) (void)hipFree(h->d_cmeta);
    if (h->d_cmeta_tiles) (void)hipFree(h->d_cmeta_tiles);
    if (h->d_wq8_stages) (void)hipFree(h->d_wq8_stages);
    for (hipEvent_t e : h->stage_ev) (void)hipEventDestroy(e);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

uint32_t cs_embedder_dim(const cs_embedder* h) { return h ? h->cfg.hidden : 0; }

int32_t cs_embedder_embed_ids(cs_embedder* h, const int32_t* ids, const int32_t* mask, uint64_t n,
                              uint32_t seq_len, uint32_t batch, float* out,
                              const volatile int32_t* cancel) {
    return embed_ids_entry(h, ids, mask, n, seq_len, batch, out, false, cancel);
}

int32_t cs_embedder_embed_ids_device(cs_embedder* h, const int32_t* ids, const int32_t* mask,
                                     uint64_t n, uint32_t seq_len, uint32_t batch, float* d_out,
                                     const volatile int32_t* cancel) {
    return embed_ids_entry(h, ids, mask, n, seq_len, batch, d_out, true, cancel);
}

int32_t cs_embedder_embed_texts(cs_embedder* h, const cs_tokenizer* t, const char* utf8,
                                const uint64_t* offsets, uint64_t n, uint32_t batch, float* out,
                                const volatile int32_t* cancel) {
    return embed_texts_impl(h, t, utf8, offsets, n, batch, out, false, cancel);
}

int32_t cs_embedder_embed_texts_device(cs_embedder* h, const cs_tokenizer* t, const char* utf8,
                                       const uint64_t* offsets, uint64_t n, uint32_t batch, float* d_out,
                                       const volatile int32_t* cancel) {
    return embed_texts_impl(h, t, utf8, offsets, n, batch, d_out, true, cancel);
}

int32_t cs_embedder_last_hidden(cs_embedder* h, float* out, uint64_t n_tokens) {
    if (!h || !out) return fail(CS_ERR_BAD_ARG, "null argument");
    if (n_tokens > (uint64_t)h->last_B * h->last_L)
        return fail(CS_ERR_BAD_ARG, "only %u tokens in the last mini-batch", h->last_B * h->last_L);
    if (h->last_hidden_partial)
        return fail(CS_ERR_UNSUPPORTED, "the last forward computed its final layer for the CLS rows only (cls_tail.hip): "
                                        "set CS_ENCODER_CLS_TAIL=0 to read every token's last hidden state");
    DeviceGuard g(h->device);
    CS_HIP(hipStreamSynchronize(h->stream));
    CS_HIP(hipMemcpy(out, h->d_x, n_tokens * h->cfg.hidden * sizeof(float), hipMemcpyDeviceToHost));
    return CS_OK;
}

int32_t cs_embedder_profile_read(cs_embedder* h, double* forward_ms, uint64_t* forwards, int32_t reset) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (forward_ms) *forward_ms = h->forward_ms;
    if (forwards) *forwards = h->forwards;
    if (reset) { h->forward_ms = 0.0; h->forwards = 0; }
    return CS_OK;
}

int32_t cs_embedder_profile_stages(cs_embedder* h, int32_t enable) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    h->stage_profile = enable != 0;
    return CS_OK;
}

int32_t cs_embedder_profile_stages_read(cs_embedder* h, double* us_per_stage, uint64_t* forwards, int32_t reset) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (us_per_stage)
        for (int i = 0; i < CS_ENCODER_STAGES; ++i) us_per_stage[i] = h->stage_us[i];
    if (forwards) *forwards = h->stage_forwards;
    if (reset) {
        for (int i = 0; i < CS_ENCODER_STAGES; ++i) h->stage_us[i] = 0.0;
        h->stage_forwards = 0;
    }
    return CS_OK;
}

int32_t cs_embedder_set_gemm_mode(cs_embedder* h, int32_t mode) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (mode != CS_GEMM_F32 && mode != CS_GEMM_SPLIT_F16 && mode != CS_GEMM_Q8_DYNAMIC) return fail(CS_ERR_BAD_ARG, "unknown gemm mode %d", mode);
    if (mode == CS_GEMM_Q8_DYNAMIC && !h->quantized)
        return fail(CS_ERR_UNSUPPORTED, "dynamic-quantisation mode needs a quantised model (cs_embedder_create_quantized / a *Q model directory)");
    if (mode != CS_GEMM_F32 && h->split_unavailable)
        return fail(CS_ERR_UNSUPPORTED, "split-f16 mode needs exact f16-subnormal MFMA inputs, which this device/mode lacks");
    h->gemm_mode = mode;
    return CS_OK;
}

int32_t cs_embedder_gemm_mode(const cs_embedder* h) { return h ? h->gemm_mode : -1; }

int32_t cs_embedder_debug_counters(cs_embedder* h, uint64_t* split_forwards, uint64_t* f32_forwards,
                                   uint64_t* range_fallbacks) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (split_forwards) *split_forwards = h->split_forwards;
    if (f32_forwards) *f32_forwards = h->f32_forwards;
    if (range_fallbacks) *range_fallbacks = h->range_fallbacks;
    return CS_OK;
}

#ifdef CS_DIAGNOSTICS
int32_t cs_debug_small_forward_counters(cs_embedder* h, uint64_t* forwards, uint64_t* fallbacks) {
    if (!h) return fail(CS_ERR_BAD_ARG, "null embedder handle");
    if (forwards) *forwards = h->sf_forwards;
    if (fallbacks) *fallbacks = h->sf_fallbacks;
    return CS_OK;
}
#endif

}  // extern "C"
