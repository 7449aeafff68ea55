"""ctypes binding of oracle/liboracle.so — the CPU checker.  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_u32p = C.POINTER(C.c_uint32)
c_u64p = C.POINTER(C.c_uint64)
c_i32p = C.POINTER(C.c_int32)


def _ptr(a, ty):
    return a.ctypes.data_as(ty) if a is not None else None


def build_oracle() -> str:
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)
    return so


class Oracle:
    def __init__(self, path: str):
        self.lib = lib = C.CDLL(path)
        lib.cs_oracle_cosine.restype = C.c_float
        lib.cs_oracle_cosine.argtypes = [c_f32p, c_f32p, C.c_size_t]
        lib.cs_oracle_cosine_f64.restype = C.c_double
        lib.cs_oracle_cosine_f64.argtypes = [c_f32p, c_f32p, C.c_size_t]
        common = [c_f32p, C.c_uint64, C.c_uint32, c_f32p, C.c_uint32, c_u32p, C.c_uint32]
        lib.cs_oracle_scan_topk.restype = C.c_uint32
        lib.cs_oracle_scan_topk.argtypes = common + [c_f32p, c_u32p]
        lib.cs_oracle_scan_topk_f64.restype = C.c_uint32
        lib.cs_oracle_scan_topk_f64.argtypes = common + [c_f64p, c_u32p]
        lib.cs_oracle_scan_topk_omp.restype = C.c_uint32
        lib.cs_oracle_scan_topk_omp.argtypes = common + [C.c_int, c_f32p, c_u32p]
        lib.cs_oracle_merge_topk.restype = C.c_uint32
        lib.cs_oracle_merge_topk.argtypes = [c_f32p, c_u32p, c_u32p, C.c_uint32, C.c_uint32, c_f32p, c_u32p]
        lib.cs_oracle_num_threads.restype = C.c_int
        lib.cs_oracle_bert_forward.restype = None
        lib.cs_oracle_bert_forward.argtypes = [C.c_void_p, c_f32p, c_i32p, c_i32p, C.c_uint32, C.c_uint32, c_f32p, c_f32p, c_f32p]
        lib.cs_oracle_bert_synth_params.restype = None
        lib.cs_oracle_bert_forward_q8.restype = None
        lib.cs_oracle_bert_forward_q8.argtypes = [C.c_void_p, c_f32p, c_f32p, c_i32p, c_i32p, C.c_uint32, C.c_uint32, c_f32p, c_f32p, c_f32p]
        lib.cs_oracle_linear_q8.restype = None
        lib.cs_oracle_linear_q8.argtypes = [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_uint64, C.c_uint64, C.c_uint64]
        lib.cs_oracle_bert_synth_params.argtypes = [C.c_void_p, C.c_uint64, c_f32p]
        lib.cs_oracle_bert_param_count.restype = C.c_uint64
        lib.cs_oracle_bert_param_count.argtypes = [C.c_void_p]
        lib.cs_oracle_alibi_slopes.restype = None
        lib.cs_oracle_alibi_slopes.argtypes = [C.c_uint32, c_f32p]
        lib.cs_oracle_synth_rows.restype = None
        lib.cs_oracle_synth_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, c_f32p]
        lib.cs_oracle_synth_planted.restype = None
        lib.cs_oracle_synth_planted.argtypes = [C.c_uint64, C.c_uint64, c_u64p, C.c_uint64, C.c_uint32, c_f32p]

    # ---- scan -----------------------------------------------------------------
    def cosine(self, a, b) -> float:
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        return float(self.lib.cs_oracle_cosine(_ptr(a, c_f32p), _ptr(b, c_f32p), a.size))

    def cosine_f64(self, a, b) -> float:
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        return float(self.lib.cs_oracle_cosine_f64(_ptr(a, c_f32p), _ptr(b, c_f32p), a.size))

    def scan_topk(self, corpus, q, k, dead=None, id_base=0, mode="literal", threads=0):
        """-> (cos[count], ids[count]).  mode: literal | f64 | omp."""
        corpus = np.ascontiguousarray(corpus, np.float32)
        q = np.ascontiguousarray(q, np.float32)
        n, dim = corpus.shape
        assert q.shape == (dim,)
        ids = np.zeros(max(k, 1), np.uint32)
        dead_a = None if dead is None else np.ascontiguousarray(dead, np.uint32)
        args = [_ptr(corpus, c_f32p), n, dim, _ptr(q, c_f32p), k, _ptr(dead_a, c_u32p), id_base]
        if mode == "literal":
            cos = np.zeros(max(k, 1), np.float32)
            cnt = self.lib.cs_oracle_scan_topk(*args, _ptr(cos, c_f32p), _ptr(ids, c_u32p))
        elif mode == "f64":
            cos = np.zeros(max(k, 1), np.float64)
            cnt = self.lib.cs_oracle_scan_topk_f64(*args, _ptr(cos, c_f64p), _ptr(ids, c_u32p))
        elif mode == "omp":
            cos = np.zeros(max(k, 1), np.float32)
            cnt = self.lib.cs_oracle_scan_topk_omp(*args, threads, _ptr(cos, c_f32p), _ptr(ids, c_u32p))
        else:
            raise ValueError(mode)
        return cos[:cnt].copy(), ids[:cnt].copy()

    def merge_topk(self, cos, ids, counts, k):
        cos = np.ascontiguousarray(cos, np.float32)
        ids = np.ascontiguousarray(ids, np.uint32)
        counts = np.ascontiguousarray(counts, np.uint32)
        nl = counts.size
        oc = np.zeros(max(k, 1), np.float32)
        oi = np.zeros(max(k, 1), np.uint32)
        cnt = self.lib.cs_oracle_merge_topk(_ptr(cos, c_f32p), _ptr(ids, c_u32p), _ptr(counts, c_u32p), nl, k, _ptr(oc, c_f32p), _ptr(oi, c_u32p))
        return oc[:cnt].copy(), oi[:cnt].copy()

    def num_threads(self) -> int:
        return int(self.lib.cs_oracle_num_threads())

    # ---- encoder ------------------------------------------------------------------
    def bert_param_count(self, cfg) -> int:
        c = cfg.to_c()
        return int(self.lib.cs_oracle_bert_param_count(C.byref(c)))

    def alibi_slopes(self, heads: int):
        out = np.empty(heads, np.float32)
        self.lib.cs_oracle_alibi_slopes(heads, _ptr(out, c_f32p))
        return out

    def bert_synth_params(self, cfg, seed):
        c = cfg.to_c()
        out = np.empty(self.bert_param_count(cfg), np.float32)
        self.lib.cs_oracle_bert_synth_params(C.byref(c), seed, _ptr(out, c_f32p))
        return out

    def bert_forward(self, cfg, params, ids, mask, want_hidden=False, want_layers=False, wscale=None):
        """-> dict(pooled [B,H], hidden [B,L,H]?, layers [layers+1,B,L,H]?).  wscale [layers, 5H + I]: run every
        Linear as onnxruntime's dynamic quantiser rewrites it (cs_oracle_bert_forward_q8)."""
        c = cfg.to_c()
        params = np.ascontiguousarray(params, np.float32)
        ids = np.ascontiguousarray(ids, np.int32)
        mask = np.ascontiguousarray(mask, np.int32)
        B, L = ids.shape
        H = cfg.hidden
        pooled = np.empty((B, H), np.float32)
        hidden = np.empty((B, L, H), np.float32) if want_hidden else None
        layers = np.empty((cfg.layers + 1, B, L, H), np.float32) if want_layers else None
        if wscale is not None:
            wscale = np.ascontiguousarray(wscale, np.float32)
            assert wscale.shape == (cfg.layers, 5 * H + cfg.intermediate)
            self.lib.cs_oracle_bert_forward_q8(C.byref(c), _ptr(params, c_f32p), _ptr(wscale, c_f32p), _ptr(ids, c_i32p),
                                               _ptr(mask, c_i32p), B, L, _ptr(hidden, c_f32p), _ptr(pooled, c_f32p),
                                               _ptr(layers, c_f32p))
        else:
            self.lib.cs_oracle_bert_forward(C.byref(c), _ptr(params, c_f32p), _ptr(ids, c_i32p), _ptr(mask, c_i32p),
                                            B, L, _ptr(hidden, c_f32p), _ptr(pooled, c_f32p), _ptr(layers, c_f32p))
        return {"pooled": pooled, "hidden": hidden, "layers": layers}

    def linear_q8(self, x, w, wscale, b):
        """One dynamically quantised Linear (cs_oracle_linear_q8): x [T,K], w [N,K] dequantised, wscale [N], b [N] -> [T,N]."""
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        wscale = np.ascontiguousarray(wscale, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        y = np.empty((x.shape[0], w.shape[0]), np.float32)
        self.lib.cs_oracle_linear_q8(_ptr(x, c_f32p), _ptr(w, c_f32p), _ptr(wscale, c_f32p), _ptr(b, c_f32p), _ptr(y, c_f32p),
                                     x.shape[0], x.shape[1], w.shape[0])
        return y

    # ---- synthetic data --------------------------------------------------------
    def synth_rows(self, seed, first_row, n, dim):
        out = np.empty((n, dim), np.float32)
        self.lib.cs_oracle_synth_rows(seed, first_row, n, dim, _ptr(out, c_f32p))
        return out

    def synth_planted(self, seed_c, seed_q, rows, dim):
        rows = np.ascontiguousarray(rows, np.uint64)
        out = np.empty((rows.size, dim), np.float32)
        self.lib.cs_oracle_synth_planted(seed_c, seed_q, _ptr(rows, c_u64p), rows.size, dim, _ptr(out, c_f32p))
        return out


def load_oracle() -> Oracle:
    global _LIB
    if _LIB is None:
        _LIB = Oracle(build_oracle())
    return _LIB
