"""Diagnostic: quantised forward on the GPU against oracle/bert_oracle.c's, per weight-quantisation form."""
import sys; sys.path.insert(0, '.')
import numpy as np
from codesearch_amd import FastEmbedder, ModelType
from codesearch_amd.bert_params import *
from tests.oracle_lib import load_oracle
o = load_oracle()
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for pc, un in ((False, True), (False, False), (True, True), (True, False)):
    cfg = BertConfig(vocab_size=1500, hidden=384, layers=layers, heads=12, intermediate=1536, max_position=64, pooling=POOL_MEAN)
    params, ws = quantize_linear_weights(cfg, synth_params(cfg, 11), per_channel=pc, unsigned=un)
    ids, mask = synth_token_batch(cfg, 4, 24, 48, True)
    emb = FastEmbedder(ModelType.AllMiniLML6V2Q, config=cfg, params=params, wscale=ws)
    got = emb.embed_ids(ids, mask)
    hid = emb.last_hidden(24 * 48).reshape(24, 48, 384)
    ref = o.bert_forward(cfg, params, ids, mask, wscale=ws, want_hidden=True)
    f32g = o.bert_forward(cfg, params, ids, mask, want_hidden=True)
    v = mask.astype(bool)
    h1 = np.abs(hid[v] - ref["hidden"][v]); h2 = np.abs(f32g["hidden"][v] - ref["hidden"][v])
    rows_hit = (h1.max(axis=1) > 1e-5).mean()
    print("per_channel", pc, "unsigned", un, "hidden max/mean/median: gpu %.2e %.2e %.2e | f32graph %.2e %.2e %.2e" %
          (h1.max(), h1.mean(), np.median(h1), h2.max(), h2.mean(), np.median(h2)), "rows off by > 1e-5: %.3f" % rows_hit,
          "scale spread", float(ws.std() / ws.mean()))
    emb.close()
