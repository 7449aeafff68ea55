"""ctypes binding of libcsgpu.so (include/codesearch_gpu.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded this module
raises, and every product entry point goes through it.
"""
from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# CS_LIBCSGPU: another build of the same library (A/B scripts under benchmarks/)
LIB_PATH = os.environ.get("CS_LIBCSGPU") or os.path.join(PKG_DIR, "libcsgpu.so")
# the diagnostic build of the same sources (csrc/Makefile): everything above plus cs_debug_*
DIAG_LIB_PATH = os.environ.get("CS_LIBCSGPU_DIAG") or os.path.join(PKG_DIR, "libcsgpu_diag.so")

CS_OK, CS_ERR_BAD_ARG, CS_ERR_DIM_MISMATCH, CS_ERR_NOT_BUILT = 0, 1, 2, 3
CS_ERR_CANCELLED, CS_ERR_OOM, CS_ERR_HIP, CS_ERR_UNSUPPORTED = 4, 5, 6, 7
CS_GEMM_F32, CS_GEMM_SPLIT_F16, CS_GEMM_Q8_DYNAMIC = 0, 1, 2
CS_MAX_K = 1024
CS_MAX_QUERIES = 4096

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)
vp = C.c_void_p


class BertConfig(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_uint32),
        ("hidden", C.c_uint32),
        ("layers", C.c_uint32),
        ("heads", C.c_uint32),
        ("intermediate", C.c_uint32),
        ("max_position", C.c_uint32),
        ("type_vocab_size", C.c_uint32),
        ("layer_norm_eps", C.c_float),
        ("pooling", C.c_int32),
        ("arch", C.c_uint32),
        ("rotary_base", C.c_float),
        ("rotary_base_local", C.c_float),
        ("local_window", C.c_uint32),
        ("global_every", C.c_uint32),
    ]


class CsError(RuntimeError):
    """A non-zero cs_status; .code is the status, str() the reference-worded message."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


# name -> (restype, argtypes); every symbol include/codesearch_gpu.h declares.
SIGNATURES = {
    "cs_last_error": (C.c_char_p, []),
    "cs_abi_version": (C.c_uint32, []),
    "cs_device_count": (C.c_int32, []),
    "cs_index_create": (C.c_int32, [C.c_uint32, C.c_uint64, C.c_int32, C.c_uint32, C.POINTER(vp)]),
    "cs_index_destroy": (None, [vp]),
    "cs_index_add": (C.c_int32, [vp, f32p, C.c_uint64, C.c_uint32, u32p]),
    "cs_index_add_device": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, u32p, vp]),
    "cs_index_reserve_rows": (C.c_int32, [vp, C.c_uint64, C.c_uint32, C.POINTER(vp)]),
    "cs_index_commit_rows": (C.c_int32, [vp, C.c_uint64, u32p]),
    "cs_index_add_synthetic": (C.c_int32, [vp, C.c_uint64, C.c_uint64, C.c_uint64, u32p]),
    "cs_index_remove": (C.c_int32, [vp, u32p, C.c_uint64, u64p]),
    "cs_index_build": (C.c_int32, [vp]),
    "cs_index_clear": (C.c_int32, [vp]),
    "cs_index_is_built": (C.c_int32, [vp]),
    "cs_index_len": (C.c_uint64, [vp]),
    "cs_index_stored_rows": (C.c_uint64, [vp]),
    "cs_index_next_id": (C.c_uint32, [vp]),
    "cs_index_dim": (C.c_uint32, [vp]),
    "cs_index_device": (C.c_int32, [vp]),
    "cs_index_search": (C.c_int32, [vp, f32p, C.c_uint32, C.c_uint32, C.c_uint32, f32p, u32p, u32p]),
    "cs_index_search_device": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp]),
    "cs_index_search_variants": (C.c_int32, [vp, f32p, C.c_uint32, C.c_uint32, C.c_uint32, f32p, u32p, u32p, i32p]),
    "cs_merge_variants_device": (C.c_int32, [C.c_int32, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp]),
    "cs_index_search_status": (C.c_int32, [vp, vp, u32p]),
    "cs_index_release_stream": (C.c_int32, [vp, vp]),
    "cs_merge_topk_device": (C.c_int32, [C.c_int32, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp]),
    "cs_shards_create": (C.c_int32, [C.c_uint32, C.c_uint32, i32p, C.c_uint64, C.c_uint64, C.POINTER(vp)]),
    "cs_shards_destroy": (None, [vp]),
    "cs_shards_add": (C.c_int32, [vp, f32p, C.c_uint64, C.c_uint32, u32p]),
    "cs_shards_add_device": (C.c_int32, [vp, vp, C.c_int32, C.c_uint64, C.c_uint32, u32p, vp]),
    "cs_shards_add_synthetic": (C.c_int32, [vp, C.c_uint64, C.c_uint64, C.c_uint64, u32p]),
    "cs_shards_remove": (C.c_int32, [vp, u32p, C.c_uint64, u64p]),
    "cs_shards_build": (C.c_int32, [vp]),
    "cs_shards_clear": (C.c_int32, [vp]),
    "cs_shards_is_built": (C.c_int32, [vp]),
    "cs_shards_len": (C.c_uint64, [vp]),
    "cs_shards_stored_rows": (C.c_uint64, [vp]),
    "cs_shards_next_id": (C.c_uint32, [vp]),
    "cs_shards_dim": (C.c_uint32, [vp]),
    "cs_shards_count": (C.c_uint32, [vp]),
    "cs_shards_shard_len": (C.c_uint64, [vp, C.c_uint32]),
    "cs_shards_direct_gather": (C.c_int32, [vp]),
    "cs_shards_search": (C.c_int32, [vp, f32p, C.c_uint32, C.c_uint32, C.c_uint32, f32p, u32p, u32p]),
    "cs_shards_search_device": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp]),
    "cs_shards_search_status": (C.c_int32, [vp, vp, u32p]),
    "cs_shards_root_device": (C.c_int32, [vp]),
    "cs_shards_shard_device": (C.c_int32, [vp, C.c_uint32]),
    "cs_shards_shard_index": (vp, [vp, C.c_uint32]),
    "cs_shards_search_variants": (C.c_int32, [vp, f32p, C.c_uint32, C.c_uint32, C.c_uint32, f32p, u32p, u32p, i32p]),
    "cs_shards_read_rows": (C.c_int32, [vp, C.c_uint64, C.c_uint64, f32p]),
    "cs_index_read_rows": (C.c_int32, [vp, C.c_uint64, C.c_uint64, f32p]),
    "cs_index_debug_counters": (C.c_int32, [vp, u64p, u64p]),
    "cs_index_filter_state": (C.c_int32, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_float), u64p]),
    "cs_index_filter_copies": (C.c_int32, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), u64p]),
    "cs_index_set_filter_min_queries": (C.c_int32, [vp, C.c_uint32]),
    "cs_index_set_single_query_route": (C.c_int32, [vp, C.c_int32]),
    "cs_index_profile": (C.c_int32, [vp, C.c_int32]),
    "cs_index_profile_read": (C.c_int32, [vp, f64p, u64p, f64p, C.c_int32]),
    "cs_bert_config_bge_small": (None, [C.POINTER(BertConfig)]),
    "cs_bert_param_count": (C.c_uint64, [C.POINTER(BertConfig)]),
    "cs_embedder_create": (C.c_int32, [C.POINTER(BertConfig), f32p, C.c_uint64, C.c_int32, C.POINTER(vp)]),
    "cs_embedder_create_quantized": (C.c_int32, [C.POINTER(BertConfig), f32p, f32p, C.c_uint64, C.c_int32, C.POINTER(vp)]),
    "cs_bert_quant_columns": (C.c_uint64, [C.POINTER(BertConfig)]),
    "cs_bert_params_from_onnx_q": (C.c_int32, [C.c_char_p, C.POINTER(BertConfig), f32p, C.c_uint64, f32p, C.c_uint64,
                                               C.POINTER(C.c_int32)]),
    "cs_bert_config_from_dir": (C.c_int32, [C.c_char_p, C.c_int32, C.POINTER(BertConfig)]),
    "cs_bert_params_from_safetensors": (C.c_int32, [C.c_char_p, C.POINTER(BertConfig), f32p, C.c_uint64]),
    "cs_bert_params_from_onnx": (C.c_int32, [C.c_char_p, C.POINTER(BertConfig), f32p, C.c_uint64]),
    "cs_embedder_create_from_dir": (C.c_int32, [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(vp)]),
    "cs_embedder_destroy": (None, [vp]),
    "cs_embedder_dim": (C.c_uint32, [vp]),
    "cs_embedder_embed_ids": (C.c_int32, [vp, i32p, i32p, C.c_uint64, C.c_uint32, C.c_uint32, f32p, i32p]),
    "cs_embedder_embed_ids_device": (C.c_int32, [vp, i32p, i32p, C.c_uint64, C.c_uint32, C.c_uint32, vp, i32p]),
    "cs_embedder_last_hidden": (C.c_int32, [vp, f32p, C.c_uint64]),
    "cs_embedder_profile_read": (C.c_int32, [vp, f64p, u64p, C.c_int32]),
    "cs_embedder_profile_stages": (C.c_int32, [vp, C.c_int32]),
    "cs_embedder_profile_stages_read": (C.c_int32, [vp, f64p, u64p, C.c_int32]),
    "cs_embedder_set_gemm_mode": (C.c_int32, [vp, C.c_int32]),
    "cs_embedder_gemm_mode": (C.c_int32, [vp]),
    "cs_embedder_debug_counters": (C.c_int32, [vp, u64p, u64p, u64p]),
    "cs_tokenizer_create": (C.c_int32, [C.c_char_p, C.c_uint64, C.c_int32, C.c_uint32, C.POINTER(vp)]),
    "cs_tokenizer_create_from_file": (C.c_int32, [C.c_char_p, C.c_int32, C.c_uint32, C.POINTER(vp)]),
    "cs_tokenizer_create_from_json": (C.c_int32, [C.c_char_p, C.c_uint32, C.POINTER(vp)]),
    "cs_tokenizer_create_from_dir": (C.c_int32, [C.c_char_p, C.c_uint32, C.POINTER(vp)]),
    "cs_tokenizer_destroy": (None, [vp]),
    "cs_tokenizer_vocab_size": (C.c_uint32, [vp]),
    "cs_tokenizer_max_length": (C.c_uint32, [vp]),
    "cs_tokenizer_pad_id": (C.c_int32, [vp]),
    "cs_tokenizer_token_to_id": (C.c_int32, [vp, C.c_char_p]),
    "cs_tokenizer_encode_batch": (C.c_int32, [vp, C.c_char_p, u64p, C.c_uint32, C.c_uint32, i32p, i32p,
                                              C.c_uint32, u32p]),
    "cs_embedder_embed_texts": (C.c_int32, [vp, vp, C.c_char_p, u64p, C.c_uint64, C.c_uint32, f32p, i32p]),
    "cs_embedder_embed_texts_device": (C.c_int32, [vp, vp, C.c_char_p, u64p, C.c_uint64, C.c_uint32, vp, i32p]),
    "cs_embedder_submit_texts": (C.c_int32, [vp, vp, C.c_char_p, u64p, C.c_uint64, u64p]),
    "cs_embedder_submit_ids": (C.c_int32, [vp, i32p, i32p, C.c_uint64, C.c_uint32, u64p]),
    "cs_embedder_wait": (C.c_int32, [vp, C.c_uint64, f32p, i32p]),
    "cs_embedder_wait_device": (C.c_int32, [vp, C.c_uint64, vp, i32p]),
    "cs_embedder_discard": (C.c_int32, [vp, C.c_uint64]),
    "cs_embedder_queued_rows": (C.c_uint64, [vp]),
    "cs_shards_plan_append": (C.c_int32, [vp, C.c_uint64, C.c_uint32, u32p, u64p, u64p, u32p]),
    "cs_shards_add_device_parts": (C.c_int32, [vp, C.c_uint32, C.POINTER(vp), i32p, u64p, C.c_uint32, u32p]),
    "cs_embedders_create": (C.c_int32, [C.POINTER(BertConfig), f32p, C.c_uint64, i32p, C.c_uint32, C.POINTER(vp)]),
    "cs_embedders_create_from_dir": (C.c_int32, [C.c_char_p, C.c_int32, i32p, C.c_uint32, C.POINTER(vp)]),
    "cs_embedders_destroy": (None, [vp]),
    "cs_embedders_count": (C.c_uint32, [vp]),
    "cs_embedders_dim": (C.c_uint32, [vp]),
    "cs_embedders_replica": (vp, [vp, C.c_uint32]),
    "cs_embedders_device": (C.c_int32, [vp, C.c_uint32]),
    "cs_embedders_embed_texts": (C.c_int32, [vp, vp, C.c_char_p, u64p, C.c_uint64, C.c_uint32, f32p, i32p]),
    "cs_embedders_embed_ids": (C.c_int32, [vp, i32p, i32p, C.c_uint64, C.c_uint32, C.c_uint32, f32p, i32p]),
    "cs_embedders_index_texts": (C.c_int32, [vp, vp, vp, C.c_char_p, u64p, C.c_uint64, C.c_uint32, u32p, i32p]),
    "cs_embedders_index_ids": (C.c_int32, [vp, vp, i32p, i32p, C.c_uint64, C.c_uint32, C.c_uint32, u32p, i32p]),
}

# include/codesearch_gpu_diag.h: exported by libcsgpu_diag.so only (operator-level parity tests, benchmarks/)
DIAG_SIGNATURES = {
    "cs_debug_small_forward_counters": (C.c_int32, [vp, u64p, u64p]),
    "cs_debug_gemm_time": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_int32, f64p]),
    "cs_debug_gemm_q8": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, f32p, f32p, f32p, f32p, f32p, f32p, C.c_uint32,
                                     C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), f32p, i32p]),
    "cs_debug_gemm_q8_units": (C.c_int32, [C.c_int32, C.c_int32, f32p, f32p, f32p, f32p, f32p, f32p, C.c_uint32, C.c_uint32,
                                           C.c_uint32, u32p, C.c_uint32, f32p, i32p]),
    "cs_debug_gemm": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, f32p, f32p, f32p, f32p, f32p,
                                  C.c_uint32, C.c_uint32, C.c_uint32, u32p]),
}

_LIB = None
_DIAG = None


def load() -> C.CDLL:
    """Load libcsgpu.so once; raise loudly when it is absent (run __graft_entry__.build())."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C codesearch_amd/csrc). There is no CPU fallback."
        )
    # torch bundles its own libamdhip64.so.7; importing it first makes both share ONE HIP
    # runtime in processes that also use torch.distributed (bench.py, multi-GPU tests).
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional plumbing
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError => ABI drift, fail loudly
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def load_diag() -> C.CDLL:
    """Load libcsgpu_diag.so once: its own copy of the library (linked -Bsymbolic, loaded RTLD_LOCAL, so it coexists
    with libcsgpu.so in one process) with the cs_debug_* entry points bound.  Errors of its calls are read with
    check_diag."""
    global _DIAG
    if _DIAG is not None:
        return _DIAG
    if not os.path.exists(DIAG_LIB_PATH):
        raise FileNotFoundError(f"{DIAG_LIB_PATH} is missing: build it with `make -C codesearch_amd/csrc`")
    load()  # (torch's HIP runtime first, as above)
    lib = C.CDLL(DIAG_LIB_PATH, mode=C.RTLD_LOCAL)
    for name, (res, args) in {**SIGNATURES, **DIAG_SIGNATURES}.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _DIAG = lib
    return lib


def check_diag(status: int) -> None:
    if status != CS_OK:
        raise CsError(status, load_diag().cs_last_error().decode("utf-8", "replace"))


def last_error() -> str:
    return load().cs_last_error().decode("utf-8", "replace")


def check(status: int) -> None:
    if status != CS_OK:
        raise CsError(status, last_error())
