#!/usr/bin/env python3
"""Differential fuzz of the C++ unigram tokenizer (csrc/unigram.cpp) against the `tokenizers` library on the tokenizer.json
files tests/golden/make_unigram_golden.py builds: random strings over a pool of ordinary, accented, combining, CJK, Hangul
jamo, Indic, emoji, width-variant, control and special-token pieces.  usage: unigram_fuzz.py [seed] [n]"""
import ctypes as C
import os
import random
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

POOL = (list("abcdefghijklmnopqrstuvwxyzABCDEXYZ0123456789") + list(" \t\n\r.,;:!?()[]{}<>/\\\"'_-+=*&^%$#@~`|") + [" "] * 10 +
        list("éèêëàâäçñöüßøåÆœŁńşğİı"
             "ΩπλдяжёЖЩъы") +
        ["́", "̂", "̃", "̈", "̧", "⃣", "‍", "​", "‌", "­", "️", "؀", "۝",
         "ि", "्", "ำ", "་"] +
        list("快速的棕色狐狸日本語テキストひらがな한국어글자") +
        ["ᄀ", "ᅡ", "ᆨ", "가"] +
        list("العربيةעבריתहिन्दीไทย") +
        ["\U0001F600", "\U0001F44D", "\U0001F3FD", "\U0001F468", "\U0001F469", "\U0001F1EB", "\U0001F1F7", "☺", "©", "™"] +
        list("ｆｕｌｌ１２３") +
        ["　", " ", " ", " ", "", "ﬁ", "½", "²", "▁", "﻿", "�", "\x00", "\x01", "\x7f",
         "\x0b", "\x0c"] +
        ["<s>", "</s>", "<mask>", "<unk>", "<pad>", "<", ">", "mask"])


def rand_text(rng):
    n = rng.choice([0, 1, 2, 3, 5, 8, 13, 30, 80])
    return "".join(rng.choice(POOL) for _ in range(n))


def encode_all(lib, path, texts, max_length=0):
    from codesearch_amd import _lib
    import numpy as np

    h = C.c_void_p()
    _lib.check(lib.cs_tokenizer_create_from_json(path.encode(), max_length, C.byref(h)))
    enc = [t.encode("utf-8") for t in texts]
    blob = b"".join(enc)
    offs = [0]
    for e in enc:
        offs.append(offs[-1] + len(e))
    offs_c = (C.c_uint64 * len(offs))(*offs)
    L = C.c_uint32()
    _lib.check(lib.cs_tokenizer_encode_batch(h, blob, offs_c, len(texts), 0, None, None, 0, C.byref(L)))
    ids = np.zeros((len(texts), max(L.value, 1)), np.int32)
    mask = np.zeros_like(ids)
    _lib.check(lib.cs_tokenizer_encode_batch(h, blob, offs_c, len(texts), 0, ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                             mask.ctypes.data_as(C.POINTER(C.c_int32)), ids.shape[1], C.byref(L)))
    lib.cs_tokenizer_destroy(h)
    return [[int(x) for x in ids[i, :mask[i].sum()]] for i in range(len(texts))]


def mismatches(seed, n, styles=("published", "converter"), verbose=False):
    import make_unigram_golden as G
    from codesearch_amd import _lib
    from tokenizers import Tokenizer

    lib = _lib.load()
    rng = random.Random(seed)
    texts = [rand_text(rng) for _ in range(n)]
    mb = G.train()
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        for style in styles:
            path = os.path.join(d, f"unigram_{style}.json")
            G.build(path, style, mb)
            tok = Tokenizer.from_file(path)
            want = [e.ids for e in tok.encode_batch(texts)]
            got = encode_all(lib, path, texts)
            for t, g, w in zip(texts, got, want):
                if g != w:
                    bad += 1
                    if verbose and bad <= 12:
                        print(style, [hex(ord(c)) for c in t][:40])
                        print("  got ", g[:30])
                        print("  want", w[:30])
    return bad, len(styles) * n


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    bad, total = mismatches(seed, n, verbose=True)
    print("mismatches", bad, "of", total)
    sys.exit(1 if bad else 0)
