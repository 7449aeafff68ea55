import os, sys
sys.path.insert(0, os.getcwd())
os.environ["CS_FILTER_SINGLE_MIN_K"] = "0"
import numpy as np
from codesearch_amd import VectorStore
def _shared_rows(rng, n, dim, c=1.2):
    mu = rng.normal(size=(1, dim)).astype(np.float32); mu /= np.linalg.norm(mu)
    x = rng.normal(size=(n, dim)).astype(np.float32) / np.sqrt(dim) + np.float32(c) * mu
    return x / np.linalg.norm(x, axis=1, keepdims=True)
dim, n, nq = 384, 300_000, 9
rng = np.random.default_rng(77)
x = _shared_rows(rng, n + 40_000, dim)
qs = _shared_rows(np.random.default_rng(78), nq, dim)
corpus = np.concatenate([x[:n], x[n:n + 20_000], -x[n + 20_000:]])
for upto in (n, n + 20_000, n + 40_000):
    st = VectorStore(None, dim)
    st.insert_embeddings(corpus[:n]); st.build_index()
    if upto > n:
        st.insert_embeddings(corpus[n:upto]); st.build_index()
    st.set_filter_min_queries(1)
    for i in (2, 3):
        r0 = st.filter_state()[2]
        st.search_raw(qs[i], 50)
        c = corpus[:upto] @ qs[i]
        srt = np.sort(c)[::-1]
        print(upto, 'query', i, 'overflow', st.filter_state()[2] - r0, 'tau', srt[49], 'rows within 0.02/0.05 below tau', int((c > srt[49]-0.02).sum()), int((c > srt[49]-0.05).sum()), 'max cos in tail part', c[n:upto].max() if upto > n else None, flush=True)
    st.close()
