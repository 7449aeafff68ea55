"""The C-ABI library loads without a GPU and exports every symbol include/codesearch_gpu.h
declares; the Python binding table covers exactly that set.  CPU only (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="codesearch_gpu.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", text))
    inline = set(re.findall(r"static inline [a-z0-9_ ]+?\b(cs_[a-z0-9_]+)\s*\(", text))
    return names - inline


def test_library_exports_every_declared_symbol(gpu_lib):
    from codesearch_amd import _lib

    syms = declared_symbols()
    assert len(syms) >= 30
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in sorted(syms):
        assert hasattr(raw, s), f"{s} declared in include/codesearch_gpu.h but not exported"
    assert syms == set(_lib.SIGNATURES), syms ^ set(_lib.SIGNATURES)


def test_diagnostics_live_in_their_own_library(gpu_lib):
    """cs_debug_* (include/codesearch_gpu_diag.h) are exported by libcsgpu_diag.so — which also exports the whole product
    ABI — and by nothing a deployment ships: libcsgpu.so holds no debug entry point and no ablation state."""
    import subprocess

    from codesearch_amd import _lib

    diag = declared_symbols("codesearch_gpu_diag.h")
    assert diag == set(_lib.DIAG_SIGNATURES) and all(s.startswith("cs_debug_") for s in diag), diag ^ set(_lib.DIAG_SIGNATURES)
    raw = ctypes.CDLL(_lib.DIAG_LIB_PATH, mode=ctypes.RTLD_LOCAL)
    for s in sorted(diag | declared_symbols()):
        assert hasattr(raw, s), f"{s} not exported by libcsgpu_diag.so"
    exported = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    for needle in ("cs_debug_", "g_gw_stamps", "g_gemm_wide_ablation", "gemm_wide32", "small_forward"):
        assert needle not in exported, f"{needle} leaked into libcsgpu.so"
    # built with -fvisibility=hidden behind a version script: the dynamic symbol table IS the C ABI — no cs:: function, no
    # kernel stub, no template instantiation beside the entry points the header declares
    names = [l.split()[-1] for l in exported.splitlines() if l.strip()]
    assert sorted(names) == sorted(declared_symbols()), set(names) ^ declared_symbols()
    # the rejected variants are not merely unexported: their kernels are not in the product's code object at all
    blob = open(_lib.LIB_PATH, "rb").read()
    for needle in (b"gemm_wide32_kernel", b"small_forward_kernel", b"attention_sh2_kernel"):
        assert needle not in blob, needle
        assert needle in open(_lib.DIAG_LIB_PATH, "rb").read(), needle
    assert not (set(_lib.SIGNATURES) & set(_lib.DIAG_SIGNATURES))


def test_no_gpu_means_loud_failure_not_fallback(gpu_lib):
    """Without a device the product refuses to run (CS_ERR_HIP); it never computes on the CPU."""
    from codesearch_amd import CsError, VectorStore, _lib

    if gpu_lib.cs_device_count() > 0:
        return  # on the GPU box this path is covered by the -m gpu tests
    try:
        VectorStore(None, 384)
    except CsError as e:
        assert e.code in (_lib.CS_ERR_HIP, _lib.CS_ERR_OOM)
    else:
        raise AssertionError("VectorStore was created without a GPU")


def test_inline_key_helpers_match_python_mirror():
    import numpy as np

    from codesearch_amd.sharded import key_pack, key_unpack

    cos = np.array([1.0, 0.5, 0.0, -0.0, -0.25, 3e-8, -1.0], np.float32)
    ids = np.array([0, 1, 2, 3, 4, 0xFFFFFFFE, 7], np.uint32)
    keys = key_pack(cos, ids)
    c2, i2 = key_unpack(keys)
    assert np.array_equal(c2, np.abs(cos) * np.sign(cos) + np.float32(0)) and np.array_equal(i2, ids)
    order = np.argsort(keys)[::-1]
    assert order.tolist() == [0, 1, 5, 2, 3, 4, 6]  # cos desc, then id asc on the tie at 0.0
    assert (keys != 0).all()


def test_integration_md_binds_what_the_header_declares():
    """The Rust `extern "C"` block of INTEGRATION.md — what a maintainer of the reference would paste — names only functions
    the header declares, with the header's argument counts, and its CsBertConfig has the header's fields in the header's
    order (cs_bert_config_from_dir writes the whole struct: a stale binding is a buffer overrun)."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    hdr = open(os.path.join(root, "include", "codesearch_gpu.h")).read()
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)

    def count(args):
        args = re.sub(r"//[^\n]*", "", args).strip()
        return 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])

    bound = re.findall(r"pub fn (cs_[a-z0-9_]+)\(([^;]*?)\)\s*(?:->[^;]*)?;", md, re.S)
    assert len(bound) >= 40
    for name, args in bound:
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", hdr_nc, re.S)
        assert m, f"INTEGRATION.md binds {name}, the header does not declare it"
        assert count(args) == count(m.group(1)), f"{name}: {count(args)} arguments in INTEGRATION.md, {count(m.group(1))} in the header"
    c_fields = re.findall(r"^\s*(?:uint32_t|int32_t|float)\s+([a-z_]+);", re.search(r"typedef struct cs_bert_config \{(.*?)\} cs_bert_config;", hdr_nc, re.S).group(1), re.M)
    rs_body = re.sub(r"//[^\n]*", "", re.search(r"pub struct CsBertConfig \{(.*?)\n\}", md, re.S).group(1))
    rs_fields = re.findall(r"pub ([a-z_]+):", rs_body)
    assert rs_fields == c_fields, (rs_fields, c_fields)
    assert "cs_abi_version() -> u32" in md
