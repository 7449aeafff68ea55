#!/usr/bin/env python3
"""Max |GPU embedding - CPU oracle embedding| at BASELINE configs[2] (256 x 256 tokens, full BGE-small shape, ragged
mask) on a sample of rows; run with CS_GEMM_WIDE_LN_SPLIT_RESID=0 / 1 to compare the two residual-stream forms."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from codesearch_amd import BertConfig, FastEmbedder, ModelType  # noqa: E402
from codesearch_amd.bert_params import synth_token_batch  # noqa: E402
from tests.oracle_lib import load_oracle  # noqa: E402

oracle = load_oracle()
cfg = BertConfig.bge_small()
ids, mask = synth_token_batch(cfg, 999, 256, 256, True)
emb = FastEmbedder(ModelType.BGESmallENV15, config=cfg, seed=202)
got = emb.embed_ids(ids, mask)
rows = [0, 1, 33, 77, 128, 200, 254, 255]
ref = oracle.bert_forward(cfg, oracle.bert_synth_params(cfg, 202), ids[rows], mask[rows])["pooled"]
sub = emb.embed_ids(ids[rows], mask[rows])  # small batch: the unfused kernels, f32 residual stream
print("CS_GEMM_WIDE_LN_SPLIT_RESID=%s: max |gpu - oracle| %.3g (batch of 256), %.3g (the same rows alone), batch vs alone %.3g; "
      "min cosine with the oracle row %.9f" % (os.environ.get("CS_GEMM_WIDE_LN_SPLIT_RESID", "default"), np.abs(got[rows] - ref).max(),
                                                np.abs(sub - ref).max(), np.abs(got[rows] - sub).max(),
                                                float(np.min(np.sum(got[rows] * ref, axis=1)))))
