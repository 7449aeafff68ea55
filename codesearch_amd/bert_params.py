"""Python mirror of include/cs_bert_params.h: flat parameter layout of the encoder, the
synthetic-weight rule, and the mapping to/from HF `BertModel` state-dict names (used to
load real checkpoints and to build golden vectors with transformers)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

from .synth import synth_below, synth_int

POOL_CLS, POOL_MEAN = 0, 1
ARCH_BERT, ARCH_NOMIC, ARCH_JINA, ARCH_JINA_QKNORM, ARCH_MODERN = 0, 1, 2, 3, 4  # cs_encoder_arch
GATED_ARCHS = (ARCH_NOMIC, ARCH_JINA, ARCH_JINA_QKNORM, ARCH_MODERN)      # no position table, a gate projection per layer

# kind -> (shift, base); cs_bert_synth_rule
_RULE = {
    "emb": (2, 0.0), "ln_g": (3, 1.0), "ln_b": (3, 0.0), "qk_w": (3, 0.0), "v_w": (4, 0.0),
    "ao_w": (4, 0.0), "up_w": (4, 0.0), "down_w": (5, 0.0), "bias": (4, 0.0),
}


@dataclass
class BertConfig:
    vocab_size: int = 30522
    hidden: int = 384
    layers: int = 12
    heads: int = 12
    intermediate: int = 1536
    max_position: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    pooling: int = POOL_CLS
    arch: int = ARCH_BERT          # ARCH_NOMIC: rotary positions on Q / K, fc2(fc11(x) * silu(fc12(x))), no position table
    #                                ARCH_JINA[_QKNORM]: ALiBi on the scores, down(value * gelu(gate)), no position table
    #                                (_QKNORM: LayerNorm on the whole query / key rows)
    rotary_base: float = 0.0
    #                                ARCH_MODERN: pre-norm layers, rotary positions (rotary_base on the global layers — every
    #                                global_every-th from 0 —, rotary_base_local on the others, which see |i - j| <= local_window),
    #                                Wo(gelu(Wi_a x) * Wi_b x), a final LayerNorm, no position / token-type table
    rotary_base_local: float = 0.0
    local_window: int = 0
    global_every: int = 0

    @staticmethod
    def bge_small() -> "BertConfig":
        """BAAI/bge-small-en-v1.5 architecture; CLS pooling (fastembed's choice for BGE)."""
        return BertConfig()

    def to_c(self):
        from ._lib import BertConfig as CBert

        return CBert(self.vocab_size, self.hidden, self.layers, self.heads, self.intermediate,
                     self.max_position, self.type_vocab_size, self.layer_norm_eps, self.pooling, self.arch,
                     self.rotary_base, self.rotary_base_local, self.local_window, self.global_every)


def tensor_table(cfg: BertConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """[(hf_name, shape, kind)] in flat order (= BertModel(add_pooling_layer=False).state_dict()).  ARCH_NOMIC: the same
    order without the position table and with the gate projection behind the up projection (cs_bert_params.h); the names
    are this table's own — `nomic_state_dict_names` maps a NomicBert checkpoint onto them."""
    H, I = cfg.hidden, cfg.intermediate
    nomic = cfg.arch in GATED_ARCHS
    qkn = cfg.arch == ARCH_JINA_QKNORM
    t = [
        ("embeddings.word_embeddings.weight", (cfg.vocab_size, H), "emb"),
    ] + ([] if nomic else [("embeddings.position_embeddings.weight", (cfg.max_position, H), "emb")]) + (
        [] if cfg.arch == ARCH_MODERN else [("embeddings.token_type_embeddings.weight", (cfg.type_vocab_size, H), "emb")]) + [
        ("embeddings.LayerNorm.weight", (H,), "ln_g"),
        ("embeddings.LayerNorm.bias", (H,), "ln_b"),
    ]
    for l in range(cfg.layers):
        p = f"encoder.layer.{l}."
        t += [
            (p + "attention.self.query.weight", (H, H), "qk_w"), (p + "attention.self.query.bias", (H,), "bias"),
            (p + "attention.self.key.weight", (H, H), "qk_w"), (p + "attention.self.key.bias", (H,), "bias"),
            (p + "attention.self.value.weight", (H, H), "v_w"), (p + "attention.self.value.bias", (H,), "bias"),
        ] + ([(p + "attention.self.layer_norm_q.weight", (H,), "ln_g"), (p + "attention.self.layer_norm_q.bias", (H,), "ln_b"),
              (p + "attention.self.layer_norm_k.weight", (H,), "ln_g"), (p + "attention.self.layer_norm_k.bias", (H,), "ln_b")]
             if qkn else []) + [
            (p + "attention.output.dense.weight", (H, H), "ao_w"), (p + "attention.output.dense.bias", (H,), "bias"),
            (p + "attention.output.LayerNorm.weight", (H,), "ln_g"), (p + "attention.output.LayerNorm.bias", (H,), "ln_b"),
            (p + "intermediate.dense.weight", (I, H), "up_w"), (p + "intermediate.dense.bias", (I,), "bias"),
        ] + ([(p + "intermediate.gate.weight", (I, H), "up_w"), (p + "intermediate.gate.bias", (I,), "bias")] if nomic else []) + [
            (p + "output.dense.weight", (H, I), "down_w"), (p + "output.dense.bias", (H,), "bias"),
            (p + "output.LayerNorm.weight", (H,), "ln_g"), (p + "output.LayerNorm.bias", (H,), "ln_b"),
        ]
    if cfg.arch == ARCH_MODERN:
        t += [("final_norm.weight", (H,), "ln_g"), ("final_norm.bias", (H,), "ln_b")]
    return t


def param_count(cfg: BertConfig) -> int:
    return sum(int(np.prod(s)) for _, s, _ in tensor_table(cfg))


def synth_params(cfg: BertConfig, seed: int) -> np.ndarray:
    """cs_bert_synth_param for the whole block -> float32 [param_count]."""
    out = np.empty(param_count(cfg), np.float32)
    off = 0
    for _, shape, kind in tensor_table(cfg):
        n = int(np.prod(shape))
        shift, base = _RULE[kind]
        idx = np.arange(off, off + n, dtype=np.uint64)
        v = synth_int(seed, idx).astype(np.float32) * np.float32(1.0 / 65536.0) * np.float32(1.0 / (1 << shift))
        out[off:off + n] = np.float32(base) + v
        off += n
    return out


def to_state_dict(cfg: BertConfig, flat: np.ndarray) -> Dict[str, np.ndarray]:
    sd, off = {}, 0
    for name, shape, _ in tensor_table(cfg):
        n = int(np.prod(shape))
        sd[name] = flat[off:off + n].reshape(shape)
        off += n
    return sd


def from_state_dict(cfg: BertConfig, sd) -> np.ndarray:
    """Flatten a HF state dict (numpy arrays or tensors; an optional 'bert.' prefix and
    pooler/position_ids entries are ignored) into the flat block."""
    out = np.empty(param_count(cfg), np.float32)
    off = 0
    for name, shape, _ in tensor_table(cfg):
        key = name if name in sd else ("bert." + name)
        a = sd[key]
        a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}: expected {shape}, got {a.shape}")
        n = int(np.prod(shape))
        out[off:off + n] = a.astype(np.float32).reshape(-1)
        off += n
    return out


def synth_token_batch(cfg: BertConfig, seed: int, B: int, L: int, ragged: bool):
    """Synthetic token ids + mask: ids uniform in [1000, vocab), [CLS]=101 first, [SEP]=102
    last valid, [PAD]=0 after; lengths U[L/8, L] when ragged else L.  """
    lo = min(1000, cfg.vocab_size // 2)
    idx = np.arange(B * L, dtype=np.uint64)
    ids = (synth_below(seed, idx, cfg.vocab_size - lo) + lo).astype(np.int32).reshape(B, L)
    if ragged:
        mn = max(2, L // 8)
        lens = (synth_below(seed + 1, np.arange(B, dtype=np.uint64), L - mn + 1) + mn).astype(np.int64)
        lens[0] = L  # batch-longest padding: at least one full row
    else:
        lens = np.full(B, L, np.int64)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int32)
    ids[:, 0] = 101 % cfg.vocab_size
    for b in range(B):
        ids[b, lens[b] - 1] = 102 % cfg.vocab_size
    ids = ids * mask
    return ids, mask


def token_batch_with_lens(cfg: BertConfig, seed: int, lens, L: int):
    """synth_token_batch with the rows' lengths given (tests of the local-attention family pick paddings that leave every
    padded position a valid key inside its window)."""
    B = len(lens)
    ids, _ = synth_token_batch(cfg, seed, B, L, False)
    lens = np.asarray(lens, np.int64)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int32)
    for b in range(B):
        ids[b, lens[b] - 1] = 102 % cfg.vocab_size
    return ids * mask, mask


# ---- real checkpoints ----------------------------------------------------------

def nomic_config_from_hf(cfg_json: dict, max_length: int = 512) -> BertConfig:
    """config.json of a NomicBert checkpoint (nomic-ai/nomic-embed-text-v1 / v1.5: model_type "nomic_bert", GPT-2 style
    key names) -> BertConfig(arch=ARCH_NOMIC).  Only the published configuration is taken: full rotary fraction,
    non-interleaved, no scale base, swiglu, post-norm — anything else is refused.  max_position is the sequence bound
    the tokenizer truncates to (there is no position table to size)."""
    if cfg_json.get("model_type") != "nomic_bert":
        raise ValueError(f"model_type {cfg_json.get('model_type')!r} is not nomic_bert")
    want = {"rotary_emb_fraction": 1.0, "rotary_emb_interleaved": False, "rotary_emb_scale_base": None,
            "activation_function": "swiglu", "prenorm": False}
    for key, val in want.items():
        if key in cfg_json and cfg_json[key] != val:
            raise ValueError(f"nomic_bert with {key} = {cfg_json[key]!r} is not built (only {val!r})")
    # dynamic-NTK scaling changes the rotary base only beyond max_trained_positions (2,048 by default): with the sequence
    # bound below that the table is the unscaled one whatever the factor (checkpoint.cpp nomic branch)
    factor = cfg_json.get("rotary_scaling_factor")
    trained = cfg_json.get("max_trained_positions") or 2048
    if factor not in (None, 1, 1.0) and min(max_length, cfg_json.get("n_positions", max_length)) > trained:
        raise ValueError("nomic_bert with a rotary scaling factor that applies at this length is not built")
    for key in ("vocab_size", "n_embd", "n_layer", "n_head"):
        if key not in cfg_json:
            raise ValueError(f"nomic_bert config.json lacks {key}")
    return BertConfig(vocab_size=cfg_json["vocab_size"], hidden=cfg_json["n_embd"], layers=cfg_json["n_layer"],
                      heads=cfg_json["n_head"], intermediate=cfg_json.get("n_inner") or 4 * cfg_json["n_embd"],
                      max_position=min(max_length, cfg_json.get("n_positions", max_length)),
                      type_vocab_size=cfg_json.get("type_vocab_size", 2),
                      layer_norm_eps=cfg_json.get("layer_norm_epsilon", 1e-12), pooling=POOL_MEAN, arch=ARCH_NOMIC,
                      rotary_base=float(cfg_json.get("rotary_emb_base", 10000.0)))


def jina_config_from_hf(cfg_json: dict, max_length: int = 512, qk_norm: bool = True) -> BertConfig:
    """config.json of a JinaBert checkpoint (jinaai/jina-embeddings-v2-base-code: model_type "bert" with
    position_embedding_type "alibi" and feed_forward_type "geglu") -> BertConfig(arch=ARCH_JINA[_QKNORM]).  Whether the
    attention carries layer_norm_q / layer_norm_k is a property of the modelling file the config's auto_map names
    ("...qk-post-norm..."), else of the checkpoint's tensors: `qk_norm` is the fallback.  max_position is the sequence
    bound the tokenizer truncates to (ALiBi has no table to size; the reference's chunks end at 512 tokens)."""
    if cfg_json.get("position_embedding_type") != "alibi":
        raise ValueError("not a JinaBert (alibi) configuration")
    if cfg_json.get("feed_forward_type", "original") != "geglu":
        raise ValueError(f"JinaBert with feed_forward_type {cfg_json.get('feed_forward_type')!r} is not built (only 'geglu')")
    if cfg_json.get("hidden_act", "gelu") != "gelu":
        raise ValueError("only erf-GELU encoders are supported")
    am = " ".join(str(v) for v in (cfg_json.get("auto_map") or {}).values())
    if am:
        qk_norm = "qk-post-norm" in am
    return BertConfig(vocab_size=cfg_json["vocab_size"], hidden=cfg_json["hidden_size"],
                      layers=cfg_json["num_hidden_layers"], heads=cfg_json["num_attention_heads"],
                      intermediate=cfg_json["intermediate_size"],
                      max_position=min(max_length, cfg_json.get("max_position_embeddings", max_length)),
                      type_vocab_size=cfg_json.get("type_vocab_size", 2),
                      layer_norm_eps=cfg_json.get("layer_norm_eps", 1e-12), pooling=POOL_MEAN,
                      arch=ARCH_JINA_QKNORM if qk_norm else ARCH_JINA)


def from_jina_state_dict(cfg: BertConfig, sd) -> np.ndarray:
    """A JinaBert state dict -> the flat block of an ARCH_JINA[_QKNORM] config.  BERT names throughout, except the
    feed-forward: `mlp.up_gated_layer` [2I, H] (qk-post-norm file: rows [0, I) are the value, rows [I, 2I) go through GELU)
    or `mlp.gated_layers` (the first modelling file: rows [0, I) go through GELU, rows [I, 2I) are the value), `mlp.down_layer`
    / `mlp.wo`, `mlp.layernorm`.  The up projection has no bias (zero slots)."""
    assert cfg.arch in (ARCH_JINA, ARCH_JINA_QKNORM)
    I = cfg.intermediate

    def get(name):
        for key in (name, "bert." + name, "model." + name):
            if key in sd:
                a = sd[key]
                return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
        return None

    mapped = {}
    for l in range(cfg.layers):
        p = f"encoder.layer.{l}."
        up = get(p + "mlp.up_gated_layer.weight")
        if up is not None:
            value, gate = up[:I], up[I:]
            down = "mlp.down_layer"
        else:
            up = get(p + "mlp.gated_layers.weight")
            if up is None:
                raise ValueError(f"the checkpoint holds no gated up projection for layer {l}")
            gate, value = up[:I], up[I:]
            down = "mlp.wo"
        if tuple(up.shape) != (2 * I, cfg.hidden):
            raise ValueError(f"{p}mlp up projection: expected {(2 * I, cfg.hidden)}, got {up.shape}")
        mapped[p + "intermediate.dense.weight"], mapped[p + "intermediate.gate.weight"] = value, gate
        mapped[p + "output.dense.weight"], mapped[p + "output.dense.bias"] = get(p + down + ".weight"), get(p + down + ".bias")
        mapped[p + "output.LayerNorm.weight"], mapped[p + "output.LayerNorm.bias"] = get(p + "mlp.layernorm.weight"), get(p + "mlp.layernorm.bias")
    out = np.zeros(param_count(cfg), np.float32)
    off = 0
    for name, shape, kind in tensor_table(cfg):
        n = int(np.prod(shape))
        a = mapped[name] if name in mapped else get(name)
        if a is None:
            if not (kind == "bias" and ".intermediate." in name):
                raise ValueError(f"the checkpoint holds nothing for {name}")
        else:
            if tuple(a.shape) != tuple(shape):
                raise ValueError(f"{name}: expected {shape}, got {a.shape}")
            out[off:off + n] = a.astype(np.float32).reshape(-1)
        off += n
    return out


def alibi_slopes(heads: int) -> np.ndarray:
    """JinaBert's `_get_alibi_head_slopes`: Python floats, then an f32 tensor."""
    import math

    def pow2(n):
        start = 2 ** (-(2 ** -(math.log2(n) - 3)))
        return [start * start ** i for i in range(n)]

    if math.log2(heads).is_integer():
        return np.asarray(pow2(heads), np.float32)
    closest = 2 ** math.floor(math.log2(heads))
    return np.asarray(pow2(closest) + pow2(2 * closest)[0::2][:heads - closest], np.float32)


def from_nomic_state_dict(cfg: BertConfig, sd) -> np.ndarray:
    """A NomicBert state dict (names of the model repository's modeling file: emb_ln, encoder.layers.N.attn.Wqkv /
    out_proj, norm1, mlp.fc11 / fc12 / fc2, norm2) -> the flat block of an ARCH_NOMIC config.  The fused Wqkv is cut into
    its query / key / value thirds; Linear biases the checkpoint does not hold (the published ones hold none) are zero."""
    assert cfg.arch == ARCH_NOMIC
    H = cfg.hidden

    def get(name):
        for key in (name, "bert." + name, "model." + name):
            if key in sd:
                a = sd[key]
                return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
        return None

    mapped = {
        "embeddings.word_embeddings.weight": get("embeddings.word_embeddings.weight"),
        "embeddings.token_type_embeddings.weight": get("embeddings.token_type_embeddings.weight"),
        "embeddings.LayerNorm.weight": get("emb_ln.weight"), "embeddings.LayerNorm.bias": get("emb_ln.bias"),
    }
    for l in range(cfg.layers):
        src, dst = f"encoder.layers.{l}.", f"encoder.layer.{l}."
        wqkv, bqkv = get(src + "attn.Wqkv.weight"), get(src + "attn.Wqkv.bias")
        if wqkv is None or tuple(wqkv.shape) != (3 * H, H):
            raise ValueError(f"{src}attn.Wqkv.weight: expected {(3 * H, H)}")
        for i, role in enumerate(("query", "key", "value")):
            mapped[dst + f"attention.self.{role}.weight"] = wqkv[i * H:(i + 1) * H]
            mapped[dst + f"attention.self.{role}.bias"] = None if bqkv is None else bqkv[i * H:(i + 1) * H]
        for ours, theirs in (("attention.output.dense", "attn.out_proj"), ("attention.output.LayerNorm", "norm1"),
                             ("intermediate.dense", "mlp.fc11"), ("intermediate.gate", "mlp.fc12"),
                             ("output.dense", "mlp.fc2"), ("output.LayerNorm", "norm2")):
            mapped[dst + ours + ".weight"] = get(src + theirs + ".weight")
            mapped[dst + ours + ".bias"] = get(src + theirs + ".bias")
    out = np.zeros(param_count(cfg), np.float32)
    off = 0
    for name, shape, kind in tensor_table(cfg):
        n = int(np.prod(shape))
        a = mapped.get(name)
        if a is None:
            if kind != "bias":
                raise ValueError(f"the checkpoint holds nothing for {name}")
        else:
            if tuple(a.shape) != tuple(shape):
                raise ValueError(f"{name}: expected {shape}, got {a.shape}")
            out[off:off + n] = a.astype(np.float32).reshape(-1)
        off += n
    return out


def config_from_hf(cfg_json: dict, pooling: int = POOL_CLS) -> BertConfig:
    """HF config.json -> BertConfig (BERT family; nomic_bert through nomic_config_from_hf)."""
    if cfg_json.get("model_type") == "nomic_bert":
        return nomic_config_from_hf(cfg_json)
    if cfg_json.get("position_embedding_type") == "alibi":
        return jina_config_from_hf(cfg_json)
    if cfg_json.get("model_type", "bert") != "bert":
        raise ValueError(f"model_type {cfg_json.get('model_type')!r} is not a BERT encoder")
    if cfg_json.get("hidden_act", "gelu") != "gelu":
        raise ValueError("only erf-GELU encoders are supported")
    return BertConfig(vocab_size=cfg_json["vocab_size"], hidden=cfg_json["hidden_size"],
                      layers=cfg_json["num_hidden_layers"], heads=cfg_json["num_attention_heads"],
                      intermediate=cfg_json["intermediate_size"],
                      max_position=cfg_json["max_position_embeddings"],
                      type_vocab_size=cfg_json.get("type_vocab_size", 2),
                      layer_norm_eps=cfg_json.get("layer_norm_eps", 1e-12), pooling=pooling)


def load_checkpoint(path: str, cfg: BertConfig) -> np.ndarray:
    """Read a `.safetensors` (or `.npz`) BertModel checkpoint into the flat block.  Tensor
    names may carry a `bert.` prefix; pooler / position_ids / extra heads are ignored."""
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file

        sd = load_file(path)
    elif path.endswith(".npz"):
        sd = dict(np.load(path))
    else:
        raise ValueError("expected a .safetensors or .npz checkpoint")
    if cfg.arch in (ARCH_JINA, ARCH_JINA_QKNORM):
        return from_jina_state_dict(cfg, sd)
    return from_nomic_state_dict(cfg, sd) if cfg.arch == ARCH_NOMIC else from_state_dict(cfg, sd)


# ---- dynamic-quantised Linear layers (the registry's *Q models) ------------------------------------

LINEAR_ROLES = ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense",
                "intermediate.dense", "output.dense")


def quant_columns(cfg: BertConfig) -> int:
    """Output columns of one layer's six Linear weights: query H | key H | value H | attention.output H |
    intermediate I | output H (cs_bert_quant_columns, include/cs_bert_params.h)."""
    return 5 * cfg.hidden + cfg.intermediate


def quantize_linear_weights(cfg: BertConfig, flat: np.ndarray, per_channel: bool = False, unsigned: bool = True,
                            reduce_range: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """What onnxruntime's quantize_dynamic does to the Linear weights of a BERT export, applied to a flat f32 block:
    every weight W [out, in] becomes (q - zero_point) * scale with q 8-bit (7-bit under reduce_range), one
    (scale, zero_point) pair per tensor or per output channel, asymmetric for UINT8 and symmetric for INT8.
    Returns (block with the dequantised weights in place, wscale [layers, 5H + I] float32) — the pair
    cs_embedder_create_quantized and the oracle take.  Biases, LayerNorm and embedding tables stay f32 (the *Q files
    may quantise their Gather tables too; the loader dequantises those, which is all the graph does with them)."""
    out = np.array(flat, np.float32, copy=True)
    H, I = cfg.hidden, cfg.intermediate
    wscale = np.zeros((cfg.layers, quant_columns(cfg)), np.float32)
    off = 0
    col = {r: c for r, c in zip(LINEAR_ROLES, (0, H, 2 * H, 3 * H, 4 * H, 4 * H + I))}
    if unsigned:
        qmin, qmax = (0, 127) if reduce_range else (0, 255)
    else:
        qmin, qmax = (-64, 64) if reduce_range else (-127, 127)
    for name, shape, _ in tensor_table(cfg):
        n = int(np.prod(shape))
        if name.endswith(".weight") and name.startswith("encoder.layer.") and len(shape) == 2:
            parts = name.split(".")
            layer, role = int(parts[2]), ".".join(parts[3:-1])
            w = out[off:off + n].reshape(shape).astype(np.float32)
            axis = 1 if per_channel else None
            lo = np.minimum(w.min(axis=axis, keepdims=True), np.float32(0))
            hi = np.maximum(w.max(axis=axis, keepdims=True), np.float32(0))
            if unsigned:
                scale = ((hi - lo) / np.float32(qmax - qmin)).astype(np.float32)
                scale = np.where(scale == 0, np.float32(1), scale)
                zp = np.clip(np.rint(np.float32(qmin) - lo / scale), qmin, qmax)
            else:
                scale = (np.maximum(np.abs(lo), np.abs(hi)) / np.float32(qmax)).astype(np.float32)
                scale = np.where(scale == 0, np.float32(1), scale)
                zp = np.zeros_like(scale)
            q = np.clip(np.rint(w / scale) + zp, qmin, qmax)
            out[off:off + n] = ((q - zp).astype(np.float32) * scale).astype(np.float32).reshape(-1)
            c0 = col[role]
            wscale[layer, c0:c0 + shape[0]] = np.broadcast_to(scale.reshape(-1), (shape[0],)) if per_channel \
                else np.float32(scale.reshape(-1)[0])
        off += n
    return out, wscale
