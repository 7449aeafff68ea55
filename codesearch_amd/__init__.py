"""codesearch_amd — MI355X-native embedding + similarity hot path for flupkede/codesearch.

Only what the path needs: `csrc/` (HIP kernels + the C ABI of include/codesearch_gpu.h,
built into libcsgpu.so) and the host-side mirrors of the reference's two structs,
`VectorStore` (src/vectordb/store.rs) and `FastEmbedder` (src/embed/embedder.rs).
"""
from ._lib import CsError  # noqa: F401
from .vector_store import (  # noqa: F401
    Chunk,
    ChunkMetadata,
    EmbeddedChunk,
    SearchResult,
    StoreStats,
    VectorStore,
    cos_to_distance,
    cos_to_score,
)
from .bert_params import POOL_CLS, POOL_MEAN, BertConfig  # noqa: F401
from .embedder import EmbedderReplicas, FastEmbedder, ModelType  # noqa: F401
