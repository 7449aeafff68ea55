import sys; sys.path.insert(0, '.')
import numpy as np
from codesearch_amd import FastEmbedder, ModelType
from codesearch_amd.bert_params import *
from tests.oracle_lib import load_oracle
o = load_oracle()
for hidden, heads, inter in ((1024, 16, 4096), (768, 12, 3072)):
  for seed in (41, 42, 43, 44):
    cfg = BertConfig(vocab_size=600, hidden=hidden, layers=1, heads=heads, intermediate=inter, max_position=64, pooling=POOL_CLS)
    params, ws = quantize_linear_weights(cfg, synth_params(cfg, seed), per_channel=True, unsigned=True)
    ids, mask = synth_token_batch(cfg, 14 + seed, 20, 48, True)
    emb = FastEmbedder(ModelType.BGEBaseENV15, config=cfg, params=params, wscale=ws)
    got = emb.embed_ids(ids, mask)
    hid = emb.last_hidden(ids.size).reshape(ids.shape + (hidden,))
    want = o.bert_forward(cfg, params, ids, mask, wscale=ws, want_hidden=True)
    v = mask.astype(bool)
    err = np.abs(hid[v] - want["hidden"][v])
    print(hidden, seed, "median %.2e max %.2e rows>1e-5 %.2f" % (np.median(err), err.max(), (err.max(axis=1) > 1e-5).mean()))
    emb.close()
