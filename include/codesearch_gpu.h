/*
 * codesearch_gpu.h — C ABI of libcsgpu.so, the MI355X (gfx950) embedding + similarity
 * hot path for flupkede/codesearch.
 *
 * The reference has no FFI/plugin seam for this path: callers use
 * two concrete Rust structs directly.  Each entry point below therefore names the
 * reference METHOD it stands in for (paths relative to /root/reference):
 *
 *   VectorStore  (src/vectordb/store.rs:94-102)   -> cs_index_*
 *   FastEmbedder (src/embed/embedder.rs:201-322)  -> cs_embedder_*
 *
 * Conventions
 *   - Plain C: pointers, sizes, opaque handles.  No C++/torch types cross the boundary.
 *   - Every function returns a cs_status (0 = ok).  cs_last_error() returns a
 *     thread-local UTF-8 message that reproduces the reference's anyhow! wording.
 *   - The caller allocates every output buffer; inputs are borrowed for the call only.
 *   - "host" pointers are ordinary process memory; "_device" variants take HBM pointers
 *     on the handle's device and a hipStream_t passed as void* (NULL = the handle's
 *     own stream), and do not synchronise.
 *   - Threading mirrors the Rust receivers: functions standing in for `&mut self`
 *     methods (add/remove/build/clear, all of cs_embedder_*) need external exclusion
 *     per handle; cs_index_search* stands in for `&self` search and is re-entrant
 *     (src/search/mod.rs:508-511 calls it from rayon threads).
 *   - There is no CPU fallback: if no HIP device is usable, create() fails with
 *     CS_ERR_HIP.
 */
#ifndef CODESEARCH_GPU_H
#define CODESEARCH_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: CS_API marks the entry points — the only symbols it exports. */
#ifndef CS_API
#define CS_API __attribute__((visibility("default")))
#endif

typedef enum cs_status {
    CS_OK = 0,
    CS_ERR_BAD_ARG = 1,
    CS_ERR_DIM_MISMATCH = 2, /* store.rs:349-353, :433-437, :667-671 */
    CS_ERR_NOT_BUILT = 3,    /* store.rs:440-444 */
    CS_ERR_CANCELLED = 4,    /* embedder.rs:280-282 */
    CS_ERR_OOM = 5,
    CS_ERR_HIP = 6,
    CS_ERR_UNSUPPORTED = 7
} cs_status;

/* Largest k a single search call accepts.  The reference asks for
 * max(5*max_results, 200) per query variant (src/search/mod.rs:494-502): 200 at the default
 * max_results, 1024 covers max_results up to 204. */
#define CS_MAX_K 1024u
/* Largest number of queries per search call. */
#define CS_MAX_QUERIES 4096u
/* Largest number of query variants cs_index_search_variants merges (the reference expands a query into at
 * most 9, src/search/mod.rs:263-388). */
#define CS_MAX_VARIANTS 16u

CS_API const char* cs_last_error(void);
/* ABI version of this header; bumped on any signature change. */
CS_API uint32_t cs_abi_version(void);
/* Number of visible HIP devices (0 if none / no driver). */
CS_API int32_t cs_device_count(void);

/* ------------------------------------------------------------------------------------
 * Vector index  — the vector half of VectorStore.  Chunk metadata (store.rs:19-85) stays
 * host-side with the caller, keyed by the u32 ids this index hands out.
 * Rows live in one row-major [capacity, dim] f32 matrix in HBM; id == id_base + row.
 * ---------------------------------------------------------------------------------- */
typedef struct cs_index cs_index;

/* VectorStore::new(path, dimensions) — store.rs:110-176.  `capacity_rows` is a
 * reservation hint (the matrix grows by doubling); `device` is the HIP ordinal;
 * `id_base` is the first id this shard hands out (0 for a single-GPU store; the row
 * offset of the shard for a row-sharded store). */
CS_API int32_t cs_index_create(uint32_t dim, uint64_t capacity_rows, int32_t device,
                        uint32_t id_base, cs_index** out);
CS_API void cs_index_destroy(cs_index* h);

/* insert_chunks_with_ids — store.rs:618-686 (vector half: writer.add_item, :674).
 * Appends n rows of `dim` floats (host memory); ids are contiguous from next_id
 * (store.rs:659-685).  dim != index dim -> CS_ERR_DIM_MISMATCH with the reference text
 * "Embedding dimension mismatch: expected {}, got {}".  Marks the index not-built
 * (store.rs:682).  out_ids may be NULL. */
CS_API int32_t cs_index_add(cs_index* h, const float* rows, uint64_t n, uint32_t dim,
                     uint32_t* out_ids);
/* Same, rows already in HBM on the index's device (zero-copy hand-off from the
 * encoder's pooled+normalised output).  The copy is asynchronous on `stream`: d_rows must stay
 * alive and unmodified until that stream has passed this point (cs_index_build drains the
 * device, and so does a later append that has to grow the matrix). */
CS_API int32_t cs_index_add_device(cs_index* h, const float* d_rows, uint64_t n, uint32_t dim,
                            uint32_t* out_ids, void* stream);
/* Appends n rows produced in place by the counter-based generator of
 * include/cs_synth.h (row r, col c -> cs_synth_value(seed, (first_row + r)*dim + c)).
 * Exists so 10M..80M-row corpora never cross PCIe; not a reference method. */
/* The same without the copy (the encoder's pooling kernel writes straight into corpus rows): reserve room for
 * n rows, have them written at *d_rows (cs_embedder_embed_ids_device / _texts_device with that address as output), then
 * commit — they become rows with the next n ids (store.rs:659-685).  No other mutating call on the handle in between. */
CS_API int32_t cs_index_reserve_rows(cs_index* h, uint64_t n, uint32_t dim, float** d_rows);
CS_API int32_t cs_index_commit_rows(cs_index* h, uint64_t n, uint32_t* out_ids);
CS_API int32_t cs_index_add_synthetic(cs_index* h, uint64_t n, uint64_t seed, uint64_t first_row,
                               uint32_t* out_first_id);

/* delete_chunks — store.rs:548-610.  Tombstones the ids (a deleted row can never be
 * returned again); *removed counts ids that were live.  Unknown ids are ignored like
 * `del_item(..).is_ok()` failing (store.rs:594).  Marks not-built if any was removed
 * (store.rs:604-606). */
CS_API int32_t cs_index_remove(cs_index* h, const uint32_t* ids, uint64_t n, uint64_t* removed);

/* build_index — store.rs:386-430.  The exact scan needs no tree; this publishes the
 * appended rows to searchers and sets `indexed` (store.rs:428).  It is also where deleted rows are reclaimed, as arroy
 * drops deleted items at its next build (the incremental `index` deletes a changed file's chunks and re-inserts them,
 * src/index/mod.rs:525,544): when at least 10 % of the stored rows are tombstones (CS_INDEX_COMPACT_DEAD_PCT; 0 = never) the
 * live rows are moved together in HBM, in their order, and the filter copies are rebuilt over them.  Ids do not change
 * (a row -> id table in HBM, read for the few candidate rows of a search only), next_id does not go back (ids are never
 * reused, store.rs:101), and searches return the same ids and cosines bit for bit — they just no longer stream the dead
 * rows.  cs_index_stored_rows tells how many rows the matrix physically holds. */
CS_API int32_t cs_index_build(cs_index* h);
CS_API uint64_t cs_index_stored_rows(const cs_index* h);  /* rows held in HBM, tombstoned ones included (== len after a reclaiming build) */
/* clear — store.rs:690-707. */
CS_API int32_t cs_index_clear(cs_index* h);

/* is_indexed (store.rs:745), stats().total_chunks (store.rs:488-523), next_id, dims. */
CS_API int32_t cs_index_is_built(const cs_index* h);
CS_API uint64_t cs_index_len(const cs_index* h);      /* live rows (appended - removed) */
CS_API uint32_t cs_index_next_id(const cs_index* h);  /* store.rs:101 */
CS_API uint32_t cs_index_dim(const cs_index* h);
CS_API int32_t cs_index_device(const cs_index* h);

/* search — store.rs:431-486, for nq queries at once (the caller's par_iter over query
 * variants, src/search/mod.rs:508-511, becomes one call).
 *   queries : [nq, dim] f32 row-major
 *   out_cos : [nq, k] raw cosine, best first; order is (cosine desc, id asc)
 *   out_ids : [nq, k] u32
 *   out_counts : [nq] number of valid results per query (< k when fewer live rows)
 * Errors: dim mismatch -> "Query embedding dimension mismatch: expected {}, got {}"
 * (store.rs:432-438); not built -> "Index not built. Call build_index() after
 * inserting chunks." (store.rs:440-444).  The reference's distance/score pair is
 * derived from the cosine by cs_cos_to_distance()/cs_cos_to_score(). */
CS_API int32_t cs_index_search(cs_index* h, const float* queries, uint32_t nq, uint32_t dim,
                        uint32_t k, float* out_cos, uint32_t* out_ids,
                        uint32_t* out_counts);
/* Same with queries and outputs in HBM; asynchronous on `stream`: the call only enqueues work and
 * never waits for the device.  out_keys receives [nq, k] packed 64-bit sort keys (see cs_key_* below;
 * 0 = empty slot), best first — the form the shard merge consumes.  out_cos / out_ids / out_counts
 * may be NULL.  Scratch is kept per (stream, calling thread), so concurrent callers are safe on any
 * mix of streams.
 * Exactness: searches of two or more queries run a filter + refine pipeline whose per-query candidate
 * buffers (max(4096, 64 k) rows) can overflow on adversarial row orders.  Up to 16 queries per call the
 * exact list-based scan is enqueued behind the search, gated on the overflow word, so the outputs are
 * exact with no host involvement.  Calls with more than 16 queries over more than max(4096, 64 k) rows
 * report an overflow through cs_index_search_status() instead: check it once the results are needed and
 * rerun an overflowed search (in slices of <= 16 queries, or through cs_index_search, which reruns by
 * itself). */
CS_API int32_t cs_index_search_device(cs_index* h, const float* d_queries, uint32_t nq,
                               uint32_t dim, uint32_t k, uint64_t* d_out_keys,
                               float* d_out_cos, uint32_t* d_out_ids,
                               uint32_t* d_out_counts, void* stream);

/* search::search's vector leg in one call — src/search/mod.rs:508-611: every query variant is searched for
 * its k = retrieval_limit best rows (:508-511), the lists are unioned with a chunk id keeping its best score
 * (:547-566), the best k distinct ids are returned best-first (:570-590) and *out_high_confidence tells whether
 * the top five all have distance < 0.15 — the reference then skips its text-search leg (:595-611).  The merge
 * runs on the device behind the searches (scan.hip merge_variants_kernel): k results and two scalars cross
 * PCIe instead of nq lists.  out_cos / out_ids: [k]; *out_count <= k valid entries.  nq <= CS_MAX_VARIANTS.
 * Equal scores are ordered (cosine desc, id asc), where the reference's HashMap order is unspecified. */
CS_API int32_t cs_index_search_variants(cs_index* h, const float* queries, uint32_t nq, uint32_t dim, uint32_t k,
                                 float* out_cos, uint32_t* out_ids, uint32_t* out_count,
                                 int32_t* out_high_confidence);
/* The merge alone, on device buffers: d_keys [nv, k] packed keys as cs_index_search_device (or the shard
 * merge) leaves them -> the best `limit` distinct ids.  Asynchronous on `stream`; outputs optional. */
CS_API int32_t cs_merge_variants_device(int32_t device, const uint64_t* d_keys, uint32_t nv, uint32_t k, uint32_t limit,
                                 uint64_t* d_out_keys, float* d_out_cos, uint32_t* d_out_ids,
                                 uint32_t* d_out_count, uint32_t* d_out_high_confidence, void* stream);

/* Synchronises `stream` and reports in *overflowed whether any cs_index_search_device call of more
 * than 16 queries issued by this thread on it since the previous status call overflowed a candidate buffer
 * (its results are then incomplete); clears the condition. */
CS_API int32_t cs_index_search_status(cs_index* h, void* stream, uint32_t* overflowed);
/* Frees the device-API scratch (partial lists, candidate buffers, split queries, merge ping-pong) the handle keeps
 * per (stream, calling thread) for `stream`, for every thread that used it; synchronises the stream first.  Call it
 * before destroying a stream that searched, or from a request thread that is about to exit: scratch is otherwise
 * kept for the life of the index (cs_index_destroy frees all of it). */
CS_API int32_t cs_index_release_stream(cs_index* h, void* stream);

/* Shard merge: given `nlists` per-shard key lists [nlists, nq, k]
 * (as all-gathered over RCCL), write the merged best-k per query.  Device pointers. */
CS_API int32_t cs_merge_topk_device(int32_t device, const uint64_t* d_keys, uint32_t nlists,
                             uint32_t nq, uint32_t k, uint64_t* d_out_keys,
                             float* d_out_cos, uint32_t* d_out_ids,
                             uint32_t* d_out_counts, void* stream);
/* (asynchronous like cs_index_search_device: scratch for multi-level merges is pooled per stream) */

/* ------------------------------------------------------------------------------------
 * Row-sharded index over several GPUs of one node, owned by ONE process — what a Rust VectorStore
 * can call from `codesearch search` (store.rs:431-486 is called from a single process,
 * src/search/mod.rs:508-511).  SURVEY.md §8e; codesearch_amd/csrc/shards.hip.
 * Ids stay contiguous from next_id (store.rs:659-685); rows are dealt to the shards in stripes of
 * `rows_per_stripe` consecutive ids, round-robin (stripe = rows per GPU gives the contiguous ranges of
 * BASELINE.json config 5: shard g holds ids [g*stripe, (g+1)*stripe)).  A search broadcasts the queries,
 * runs every shard's cs_index search on its own stream, gathers nq*k*8 bytes per shard on the first
 * device (one peer copy per shard over xGMI; CS_SHARDS_DIRECT=1: written there by the search's last kernel)
 * and merges them: the result is the single-index result bit for bit.  Same error texts, threading rules and entry-point meaning as
 * cs_index_* (add/remove/build/clear need external exclusion; search is re-entrant).
 * ---------------------------------------------------------------------------------- */
typedef struct cs_shards cs_shards;
CS_API int32_t cs_shards_create(uint32_t dim, uint32_t nshards, const int32_t* devices, uint64_t rows_per_stripe,
                         uint64_t capacity_rows /* whole store, reservation hint */, cs_shards** out);
CS_API void cs_shards_destroy(cs_shards* h);
CS_API int32_t cs_shards_add(cs_shards* h, const float* rows, uint64_t n, uint32_t dim, uint32_t* out_ids);
/* Same with the rows in HBM of device `src_device` (an encoder replica's output): each run of rows goes to its
 * shard by one asynchronous copy on `stream` (a stream of src_device; in place when the shard lives there, over
 * xGMI otherwise) — the multi-GPU form of cs_index_add_device, src/index/mod.rs:692-723.  Appends are
 * all-or-nothing: capacity is reserved on every touched shard before any row moves.  `stream` must stay alive until
 * the store's next build, search or read has drained it (the store tracks one event per source stream). */
CS_API int32_t cs_shards_add_device(cs_shards* h, const float* d_rows, int32_t src_device, uint64_t n, uint32_t dim,
                             uint32_t* out_ids, void* stream);
/* Where the next n appended rows will live (ids are contiguous from next_id, so this is known before the rows exist):
 * run i = rows [first[i], first[i] + count[i]) of the append, all on shard[i]; runs ascend and cover [0, n).
 * *n_runs = the number of runs; the arrays are filled only when max_runs >= *n_runs (call with 0 to size them). */
CS_API int32_t cs_shards_plan_append(const cs_shards* h, uint64_t n, uint32_t max_runs, uint32_t* shard, uint64_t* first,
                              uint64_t* count, uint32_t* n_runs);
/* cs_shards_add_device with the rows in several buffers: part i = the next counts[i] rows, at d_rows[i] in HBM of
 * src_devices[i]; the parts must be the runs of cs_shards_plan_append for their total, in order.  One asynchronous
 * copy per part on the null stream of its source device. */
CS_API int32_t cs_shards_add_device_parts(cs_shards* h, uint32_t nparts, const float* const* d_rows,
                                   const int32_t* src_devices, const uint64_t* counts, uint32_t dim, uint32_t* out_ids);
CS_API int32_t cs_shards_add_synthetic(cs_shards* h, uint64_t n, uint64_t seed, uint64_t first_row,
                                uint32_t* out_first_id);
CS_API int32_t cs_shards_remove(cs_shards* h, const uint32_t* ids, uint64_t n, uint64_t* removed);
CS_API int32_t cs_shards_build(cs_shards* h);
CS_API int32_t cs_shards_clear(cs_shards* h);
CS_API int32_t cs_shards_is_built(const cs_shards* h);
CS_API uint64_t cs_shards_len(const cs_shards* h);
CS_API uint64_t cs_shards_stored_rows(const cs_shards* h);   /* over all shards (cs_index_stored_rows; every shard reclaims its own deleted rows at cs_shards_build) */
CS_API uint32_t cs_shards_next_id(const cs_shards* h);
CS_API uint32_t cs_shards_dim(const cs_shards* h);
CS_API uint32_t cs_shards_count(const cs_shards* h);                        /* number of shards */
CS_API uint64_t cs_shards_shard_len(const cs_shards* h, uint32_t shard);    /* live rows on one shard */
CS_API int32_t cs_shards_direct_gather(const cs_shards* h);                 /* 1 = shards write into the root's buffer (CS_SHARDS_DIRECT=1) */
CS_API int32_t cs_shards_search(cs_shards* h, const float* queries, uint32_t nq, uint32_t dim, uint32_t k,
                         float* out_cos, uint32_t* out_ids, uint32_t* out_counts);
/* Device-pointer form (cs_index_search_device's counterpart): d_queries [nq, dim] and the outputs live in HBM of the
 * store's FIRST device (cs_shards_root_device); asynchronous on `stream`, a stream of that device — the call only
 * enqueues: the shard streams wait for an event of `stream`, fetch the queries over xGMI, search, send their keys
 * back, and `stream` waits for one event per shard before the merge.  Nothing waits for the host, so consecutive
 * searches overlap their launch cost with the previous scan.  Exactness above 16 queries per call as for
 * cs_index_search_device: ask cs_shards_search_status once the results are needed. */
CS_API int32_t cs_shards_search_device(cs_shards* h, const float* d_queries, uint32_t nq, uint32_t dim, uint32_t k,
                                uint64_t* d_out_keys, float* d_out_cos, uint32_t* d_out_ids, uint32_t* d_out_counts,
                                void* stream);
CS_API int32_t cs_shards_search_status(cs_shards* h, void* stream, uint32_t* overflowed);
CS_API int32_t cs_shards_root_device(const cs_shards* h);
CS_API int32_t cs_shards_shard_device(const cs_shards* h, uint32_t shard);
/* Borrowed handle of one shard's cs_index, for diagnostics only (cs_index_profile*, cs_index_debug_counters,
 * cs_index_search* of that shard alone: local row numbers, not ids).  Never mutate or destroy it. */
CS_API cs_index* cs_shards_shard_index(cs_shards* h, uint32_t shard);
/* cs_index_search_variants over the sharded store: per-variant searches on every shard, the shard merge, then the
 * variant merge (src/search/mod.rs:513-611) on the first device. */
CS_API int32_t cs_shards_search_variants(cs_shards* h, const float* queries, uint32_t nq, uint32_t dim, uint32_t k,
                                  float* out_cos, uint32_t* out_ids, uint32_t* out_count, int32_t* out_high_confidence);
CS_API int32_t cs_shards_read_rows(cs_shards* h, uint64_t first_id, uint64_t n, float* out_rows);

/* Copy the rows of ids [id_base + first_row, id_base + first_row + n) back to host memory (test and
 * persistence aid; VectorStore has no direct counterpart).  Rows are named by their ids' offsets, wherever a reclaiming
 * build has moved them; an id whose row was deleted and reclaimed is CS_ERR_BAD_ARG. */
CS_API int32_t cs_index_read_rows(cs_index* h, uint64_t first_row, uint64_t n, float* out_rows);

/* Kernel timing for bench.py: when enabled, every scan launch is bracketed by HIP
 * events on the launching stream.  read() synchronises and returns accumulated
 * milliseconds and launch count since the last reset. */
CS_API int32_t cs_index_profile(cs_index* h, int32_t enable);
CS_API int32_t cs_index_profile_read(cs_index* h, double* scan_ms, uint64_t* scan_launches,
                              double* merge_ms, int32_t reset);

/* Diagnostics: how many searches took the batched-query (MFMA) path, and how many of those
 * overflowed their candidate buffers and were rerun on the exact list-based scan. */
/* Searches of at least `min_queries` queries take the filter-and-refine path (an f16 MFMA filter
 * over the half-size unit-vector copy built at cs_index_build, then an exact f32 re-score of the
 * candidates: results bit-identical to the streaming f32 scan).  Default 2 (CS_FILTER_MIN_Q); 1
 * routes single queries through it too (1.45 ms instead of 2.26 ms over 10M x 384). */
CS_API int32_t cs_index_set_filter_min_queries(cs_index* h, uint32_t min_queries);
/* How ONE query is answered over a large index.  The reference's commonest searches are exactly that shape: MCP
 * (src/mcp/mod.rs:252: one query, k = limit * 3) and the HTTP handler (src/server/mod.rs:547: k = 25).  Every route
 * returns the same bits (exact f32 re-score of the filter's candidates; tests/test_gpu_scan.py).
 *   CS_ROUTE_COST   (default) over >= 32,768 rows below k = 48 and >= 300,000 rows from there on (CS_FILTER_SINGLE_MIN_ROWS,
 *                   CS_FILTER_SINGLE_MIN_ROWS_LONG: the measured crossovers) the filter + refine path whenever the int8
 *                   copy serves (it streams 1 byte per element instead of 4: 0.065 vs 0.083 ms over 100,000 x 384, 0.13
 *                   vs 0.26 ms over 1M, 0.64 vs 2.16 ms over 10M at k = 10), and from k = 100 on over >= 2M rows
 *                   (CS_FILTER_SINGLE_MIN_K) when only the f16 copy does; else the streaming scan;
 *   CS_ROUTE_STREAM always the f32 streaming scan (scan.hip — the kernel BASELINE.json's roofline target is quoted
 *                   on; bench.py selects it for `value`);  CS_FILTER_SINGLE_MIN_K=0 makes it a handle's default;
 *   CS_ROUTE_FILTER the filter path whenever a filter copy can serve, whatever the row count. */
enum { CS_ROUTE_COST = 0, CS_ROUTE_STREAM = 1, CS_ROUTE_FILTER = 2 };
CS_API int32_t cs_index_set_single_query_route(cs_index* h, int32_t route);
CS_API int32_t cs_index_debug_counters(cs_index* h, uint64_t* batched_searches,
                                uint64_t* batched_fallbacks);
/* Diagnostics: which copy of the corpus feeds the filter of batched searches right now — 0 none (no filter copy:
 * CS_INDEX_SPLIT=0 or unsupported width), 1 the f16 unit rows, 2 the int8 unit rows — the spread
 * statistic of the last build (median over tiles of max |u - mu| sqrt(dim): ~4.4 for evenly spread coordinates; above
 * CS_FILTER_INT8_MAX_SPREAD = 7 the int8 copy is not used) and how many host-API searches through the int8 copy
 * overflowed and were answered by the f16 copy instead (the int8 copy is retired until cs_index_clear once two searches
 * through it have overflowed and they are more than one in sixteen of its searches).
 * Results are exact and bit-identical whichever copy filters. */
CS_API int32_t cs_index_filter_state(cs_index* h, int32_t* copy, float* spread, uint64_t* int8_reruns);
/* Which filter copies exist in HBM right now, and the bytes they occupy.  The int8 copy (1 byte per element) is kept by
 * every index of a supported width; the f16 copy (2 bytes per element) only where the int8 copy does not serve — no int8
 * copy (CS_FILTER_INT8=0, no room), retired at a build by its spread or later by two overflowed searches, or at most
 * 1,024 rows — and is then built by cs_index_build, or by the first search after a retirement (that one search waits
 * ~5 ms per 10M x 384 for the conversion; if there is no room for it the search takes the exact paths, no error).
 * CS_FILTER_F16_EAGER=1 keeps both at every build (round 3's behaviour: 7 bytes per element instead of 5). */
CS_API int32_t cs_index_filter_copies(cs_index* h, int32_t* has_int8, int32_t* has_f16, uint64_t* filter_bytes);

/* Score mapping.  store.rs:477-478: score = 1 - distance, distance = arroy 0.5.0
 * Cosine = (1 - cos) / 2  (third-party). */
static inline float cs_cos_to_distance(float c) { return (1.0f - c) * 0.5f; }
static inline float cs_cos_to_score(float c) { return 1.0f - (1.0f - c) * 0.5f; }

/* Packed sort key: high 32 bits = order-preserving image of the f32 cosine, low 32 =
 * ~id, so that a larger key is a better result under (cosine desc, id asc). */
static inline uint64_t cs_key_pack(float c, uint32_t id) {
    union { float f; uint32_t u; } v; v.f = c + 0.0f; /* -0 -> +0 */
    uint32_t o = (v.u & 0x80000000u) ? ~v.u : (v.u | 0x80000000u);
    return ((uint64_t)o << 32) | (uint64_t)(~id);
}
static inline float cs_key_cos(uint64_t key) {
    uint32_t o = (uint32_t)(key >> 32);
    union { float f; uint32_t u; } v;
    v.u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return v.f;
}
static inline uint32_t cs_key_id(uint64_t key) { return ~(uint32_t)key; }

/* ------------------------------------------------------------------------------------
 * Embedder — FastEmbedder.  The device side starts from token ids (the tokenizer is a
 * host concern).  Architecture = HF BertModel without pooler.
 * ---------------------------------------------------------------------------------- */
typedef struct cs_embedder cs_embedder;

typedef enum cs_pooling { CS_POOL_CLS = 0, CS_POOL_MEAN = 1 } cs_pooling;

/* Encoder family.  CS_ARCH_BERT: absolute position embeddings, GELU feed-forward (every BERT / MiniLM / BGE / E5 entry of
 * the reference's registry, embedder.rs:7-48).  CS_ARCH_NOMIC: the NomicBert encoder of the registry's three Nomic entries
 * (embedder.rs:30-35: nomic-embed-text-v1 / v1.5 / v1.5-Q) — no position table; rotary angles on Q and K (non-interleaved
 * halves, pos * base^(-2i/d)); a gated feed-forward  fc2( fc11(x) * silu(fc12(x)) );  post-LayerNorm and everything else as
 * BERT.  Flat parameter order: cs_bert_params.h. */
/* CS_ARCH_JINA / CS_ARCH_JINA_QKNORM: JinaBert (embedder.rs:40-41,69: JinaEmbeddingsV2BaseCode) — no position table, the
 * symmetric ALiBi bias -slope_h |i - j| on the attention scores, a GELU-gated feed-forward; _QKNORM adds the LayerNorm on the
 * whole query and key rows that the "qk-post-norm" modelling file (the one jina-embeddings-v2-base-code names) applies.
 * CS_ARCH_MODERN: ModernBERT (embedder.rs:47, :72: ModernBertEmbedLarge) — PRE-norm layers (x += attn(LN(x)); x += mlp(LN(x)),
 * layer 0 without its first LayerNorm, a final LayerNorm), no position or token-type table, rotary positions on Q / K with one
 * base for the global-attention layers (every `global_every`-th, from layer 0) and another for the local ones, whose scores
 * are masked outside |i - j| <= local_window, and the feed-forward  Wo( gelu(Wi_a x) * Wi_b x ). */
typedef enum cs_encoder_arch { CS_ARCH_BERT = 0, CS_ARCH_NOMIC = 1, CS_ARCH_JINA = 2, CS_ARCH_JINA_QKNORM = 3, CS_ARCH_MODERN = 4 } cs_encoder_arch;

typedef struct cs_bert_config {
    uint32_t vocab_size;        /* 30522 for bge-small-en-v1.5 */
    uint32_t hidden;            /* 384  (== ModelType::dimensions, embedder.rs:76-96) */
    uint32_t layers;            /* 12 */
    uint32_t heads;             /* 12 */
    uint32_t intermediate;      /* 1536 */
    uint32_t max_position;      /* 512 */
    uint32_t type_vocab_size;   /* 2 */
    float layer_norm_eps;       /* 1e-12 */
    int32_t pooling;            /* cs_pooling */
    uint32_t arch;              /* cs_encoder_arch; 0 = BERT (every field above means what it did before this field existed) */
    float rotary_base;          /* CS_ARCH_NOMIC: base of the rotary angles (1000 for nomic-embed-text-v1 / v1.5); CS_ARCH_MODERN: of
                                 * the global-attention layers (160000); else ignored */
    /* CS_ARCH_MODERN only (zero for every other family; ABI 6): */
    float rotary_base_local;    /* base of the rotary angles of the local-attention layers (10000) */
    uint32_t local_window;      /* local layers attend to keys with |i - j| <= local_window (64 = local_attention / 2) */
    uint32_t global_every;      /* layer l attends globally when l % global_every == 0 (3) */
} cs_bert_config;

/* Fills *cfg with the BAAI/bge-small-en-v1.5 architecture (CLS pooling). */
CS_API void cs_bert_config_bge_small(cs_bert_config* cfg);
/* Number of f32 parameters a config needs, in the flat order documented in
 * codesearch_amd/csrc/bert_params.h (HF BertModel tensor order). */
CS_API uint64_t cs_bert_param_count(const cs_bert_config* cfg);

/* FastEmbedder::with_cache_dir — embedder.rs:218-245.  `params` = flat f32 parameter
 * block (host memory, cs_bert_param_count floats); NULL => weights are generated on the
 * device from `seed` by the counter-based generator of include/cs_synth.h (synthetic-weight mode). */
CS_API int32_t cs_embedder_create(const cs_bert_config* cfg, const float* params, uint64_t seed,
                           int32_t device, cs_embedder** out);
/* Real checkpoints (the model-loading half of with_cache_dir).  `model_dir` is a model directory as hf-hub
 * caches it: config.json (BERT family, erf-GELU, absolute positions) and the weights as EITHER
 *   - model.safetensors (the PyTorch snapshot: HF BertModel tensor names, optional "bert." prefix), or
 *   - the ONNX export fastembed itself downloads and runs — onnx/model.onnx (Xenova/bge-small-en-v1.5),
 *     model.onnx or model_optimized.onnx: the graph's initialisers are read by hand (no protobuf library):
 *     named parameters directly, Linear weights through the MatMul/Gemm feeding each bias's Add — or, in files that went
 *     through onnxruntime's transformer optimiser, the fused node that swallowed that Add (SkipLayerNormalization,
 *     BiasGelu / FastGelu) — the packed QKV of such files through their fused Attention / QAttention nodes.
 * F32, F16 or BF16; pooler / position_ids / other extras ignored.  Dynamically quantised exports (the *Q models of
 * the registry, among them the reference's default AllMiniLML6V2Q, embedder.rs:12-13: onnx/model_quantized.onnx — INT8 /
 * UINT8 weights with scale and zero point behind MatMulInteger) are read as (q - zero_point) * scale into the f32 block and,
 * when every Linear of every layer is quantised, the embedder runs them the way the file's graph does (CS_GEMM_Q8_DYNAMIC
 * below: activations re-quantised to 8 bits per call, integer products); CS_ENCODER_QUANT=0 in the environment keeps the
 * f32 graph of the quantised weights instead.  All loaders are host-only.
 * A NomicBert directory (config.json with model_type "nomic_bert": nomic-ai/nomic-embed-text-v1 / v1.5, the registry's
 * Nomic entries) is read too — its GPT-2 style keys (n_embd, n_head, n_layer, n_inner, rotary_emb_base, ...; only the
 * published arrangement: full non-interleaved rotary positions without scaling, swiglu, post-norm) into a CS_ARCH_NOMIC
 * config, and its model.safetensors by the model repository's tensor names (emb_ln, encoder.layers.N.attn.Wqkv cut into
 * query | key | value, attn.out_proj, norm1, mlp.fc11 / fc12 / fc2, norm2; absent Linear biases are zero) — or its ONNX export,
 * which is what fastembed caches for the registry's Nomic entries (onnx/model.onnx; onnx/model_quantized.onnx for the *Q
 * entry, read as (q - zero_point) * scale and run as the f32 graph of those weights): the bias-free Linear weights are
 * anonymous there and are taken by graph structure — the MatMul / MatMulInteger nodes with a 2-D initialiser, in order
 * Wqkv | out_proj | fc11, fc12 | fc2 per layer, the gate being the [H, I] product whose result reaches a Sigmoid.
 * A JinaBert directory (config.json with position_embedding_type "alibi" and feed_forward_type "geglu":
 * jinaai/jina-embeddings-v2-base-code, the registry's JinaEmbeddingsV2BaseCode) is read from model.safetensors by either
 * modelling file's tensor names (mlp.up_gated_layer / down_layer / layernorm with the value rows first, or mlp.gated_layers /
 * wo with the activated rows first; attention.self.layer_norm_q / _k present or not decides CS_ARCH_JINA_QKNORM against
 * CS_ARCH_JINA, the config's auto_map where there is no checkpoint to ask); max_position is capped at the 512 tokens
 * fastembed truncates to.  Its ONNX export (onnx/model.onnx: what fastembed caches) is read as well: BERT's names for the
 * embeddings and the attention block (weights behind the Add of their named bias), the bias-free gated up projection as the
 * one [H, 2I] weight product of each layer, mlp.wo / mlp.down_layer telling the two modelling files apart.
 * A ModernBERT directory (config.json with model_type "modernbert": lightonai/modernbert-embed-large, the registry's
 * ModernBertEmbedLarge) is read from model.safetensors by HF ModernBertModel's names (embeddings.tok_embeddings / norm,
 * layers.N.attn_norm / attn.Wqkv / attn.Wo / mlp_norm / mlp.Wi / mlp.Wo, final_norm; optional "model." prefix; biases optional):
 * config keys global_attn_every_n_layers, local_attention, global_rope_theta / local_rope_theta (or rope_parameters), norm_eps;
 * intermediate_size is rounded up to a multiple of 128 in the cs_bert_config (2,624 -> 2,688) and the loader fills the
 * difference with zero rows / columns.  Its ONNX export (onnx/model.onnx) is read as well: the LayerNorm weights by name, the
 * four bias-free Linear weights of a layer (Wqkv, attn.Wo, mlp.Wi, mlp.Wo) as the graph's weight products in order. */
/* pooling: CS_POOL_CLS, CS_POOL_MEAN, or -1 = what <model_dir>/1_Pooling/config.json says (the
 * sentence-transformers module: mean for MiniLM / E5, CLS for BGE), CLS when that file is absent (mean for a
 * nomic_bert directory: fastembed's pooling for the family). */
CS_API int32_t cs_bert_config_from_dir(const char* model_dir, int32_t pooling, cs_bert_config* cfg);
CS_API int32_t cs_bert_params_from_safetensors(const char* path, const cs_bert_config* cfg,
                                        float* params, uint64_t n_params);
CS_API int32_t cs_bert_params_from_onnx(const char* path, const cs_bert_config* cfg, float* params,
                                 uint64_t n_params);
/* The same, also reporting the file's quantisation: wscale (optional, cs_bert_quant_columns(cfg) * layers floats) receives
 * the scale of every output column of the six Linear weights of every layer (a per-tensor scale repeated over its columns),
 * in the order query | key | value | attention.output | intermediate | output; *quantized = 1 when ALL of them are INT8 /
 * UINT8 initialisers behind MatMulInteger (then `params` holds exact multiples of these scales), else 0. */
CS_API int32_t cs_bert_params_from_onnx_q(const char* path, const cs_bert_config* cfg, float* params, uint64_t n_params,
                                   float* wscale, uint64_t n_wscale, int32_t* quantized);
/* Output columns of one layer's Linear weights: 5 * hidden + intermediate. */
CS_API uint64_t cs_bert_quant_columns(const cs_bert_config* cfg);
/* A dynamically quantised model (what onnxruntime's quantize_dynamic writes; see CS_GEMM_Q8_DYNAMIC).  `params` as for
 * cs_embedder_create, with every Linear weight W[n][k] an integer multiple of wscale[layer][column n] whose integers span at
 * most 8 bits per column (anything else -> CS_ERR_BAD_ARG); wscale: layers * cs_bert_quant_columns(cfg) floats. */
CS_API int32_t cs_embedder_create_quantized(const cs_bert_config* cfg, const float* params, const float* wscale,
                                     uint64_t n_wscale, int32_t device, cs_embedder** out);
/* config.json + weights -> embedder on `device` (the tokenizer of the same directory comes from
 * cs_tokenizer_create_from_dir). */
CS_API int32_t cs_embedder_create_from_dir(const char* model_dir, int32_t pooling, int32_t device,
                                    cs_embedder** out);
CS_API void cs_embedder_destroy(cs_embedder* h);
CS_API uint32_t cs_embedder_dim(const cs_embedder* h);   /* dimensions(), embedder.rs:307 */

/* embed_batch_chunked — embedder.rs:266-295, from token ids.
 *   ids, mask : [n, seq_len] i32 row-major (mask 1 = token, 0 = pad)
 *   batch     : mini-batch size (embed_batch's 256/128/64 policy, embedder.rs:251-261;
 *               0 = that policy, honouring CODESEARCH_BATCH_SIZE)
 *   out       : [n, dim] f32 — L2-normalised pooled embeddings
 *   cancel    : optional flag polled between mini-batches (embedder.rs:280); non-zero
 *               -> CS_ERR_CANCELLED "Embedding interrupted by shutdown request". */
CS_API int32_t cs_embedder_embed_ids(cs_embedder* h, const int32_t* ids, const int32_t* mask,
                              uint64_t n, uint32_t seq_len, uint32_t batch, float* out,
                              const volatile int32_t* cancel);
/* Same, but leaves the [n, dim] result in HBM at d_out (e.g. to feed
 * cs_index_add_device without a PCIe round trip). */
CS_API int32_t cs_embedder_embed_ids_device(cs_embedder* h, const int32_t* ids,
                                     const int32_t* mask, uint64_t n, uint32_t seq_len,
                                     uint32_t batch, float* d_out,
                                     const volatile int32_t* cancel);
/* Debug/parity aid: last_hidden_state [n*seq_len, hidden] of the most recent
 * mini-batch (n <= batch), copied to host.  A CLS-pooled model whose mini-batch holds at least 4,096 tokens computes
 * only the CLS rows of its LAST layer (the only rows the embedding reads; csrc/cls_tail.hip): this call then returns
 * CS_ERR_UNSUPPORTED (the buffer holds the layer before outside the CLS rows).  CS_ENCODER_CLS_TAIL=0 in the environment
 * restores the full last layer; rows longer than 512 tokens always run it. */
CS_API int32_t cs_embedder_last_hidden(cs_embedder* h, float* out, uint64_t n_tokens);
CS_API int32_t cs_embedder_profile_read(cs_embedder* h, double* forward_ms, uint64_t* forwards,
                                 int32_t reset);
/* Per-kernel-class timing for bench.py's encoder roofline: while enabled, a forward runs on ONE
 * stream with a HIP event after every kernel.  read() returns the summed
 * microseconds per class since the last reset and the number of forwards they cover. */
enum {
    CS_STAGE_EMBED_LN = 0,  /* E1 */
    CS_STAGE_QKV = 1,       /* E2 */
    CS_STAGE_ATTENTION = 2, /* E3 */
    CS_STAGE_OUT_PROJ = 3,  /* E4 GEMM */
    CS_STAGE_LN_ATTN = 4,   /* E4 LayerNorm */
    CS_STAGE_FFN_UP = 5,    /* E5 */
    CS_STAGE_FFN_DOWN = 6,  /* E6 GEMM */
    CS_STAGE_LN_FFN = 7,    /* E6 LayerNorm */
    CS_STAGE_POOL = 8,      /* E7 + E8 */
    CS_ENCODER_STAGES = 9
};
CS_API int32_t cs_embedder_profile_stages(cs_embedder* h, int32_t enable);
CS_API int32_t cs_embedder_profile_stages_read(cs_embedder* h, double* us_per_stage /*[CS_ENCODER_STAGES]*/,
                                        uint64_t* forwards, int32_t reset);

/* ------------------------------------------------------------------------------------
 * Text entry points — what FastEmbedder::embed_batch(Vec<String>) takes (embedder.rs:249-295).
 * fastembed tokenises with the `tokenizers` crate 0.22.2 configured from the model's
 * tokenizer.json; cs_tokenizer restates that BERT pipeline on the host in C++ (special tokens
 * cut out verbatim -> BertNormalizer -> BertPreTokenizer -> WordPiece("##", [UNK], 100) ->
 * [CLS] A [SEP] -> truncation -> batch-longest padding).  Pure host code: usable without a GPU.
 * ---------------------------------------------------------------------------------- */
typedef struct cs_tokenizer cs_tokenizer;

/* `vocab` = the bytes of a vocab.txt (one token per line, id = line number).  `lowercase`
 * is BertNormalizer's flag (strip_accents follows it, as in tokenizer.json's null);
 * `max_length` the truncation length (512 for bge-small).  The vocabulary must hold
 * [PAD] [UNK] [CLS] [SEP]. */
CS_API int32_t cs_tokenizer_create(const char* vocab, uint64_t vocab_bytes, int32_t lowercase,
                            uint32_t max_length, cs_tokenizer** out);
CS_API int32_t cs_tokenizer_create_from_file(const char* vocab_path, int32_t lowercase,
                                      uint32_t max_length, cs_tokenizer** out);
/* tokenizer.json of the `tokenizers` crate — the file fastembed builds its tokenizer from: WordPiece
 * model.vocab, BertNormalizer.lowercase, truncation.max_length.  max_length 0 = the file's truncation
 * length, 512 (fastembed's default) when it has none.  Anything that is not the BERT pipeline
 * cs_tokenizer implements (another model type, prefix, unk token or normalizer) is refused.
 * A file whose model.type is "Unigram" (SentencePiece vocabularies: the registry's multilingual entries,
 * /root/reference/src/embed/embedder.rs:58,70) is read by csrc/unigram.cpp instead: model.vocab [[piece, score]],
 * unk_id; normalizer Precompiled / Replace(Regex " {2,}" or a String) / Strip (or a Sequence of them);
 * pre_tokenizer WhitespaceSplit / Metaspace; post_processor TemplateProcessing <bos> $A <eos>; special added
 * tokens.  The handle then encodes <s> ... </s> and pads with <pad>; any other component is refused.
 * A file whose model.type is "BPE" (byte-level BPE: the registry's JinaEmbeddingsV2BaseCode,
 * /root/reference/src/embed/embedder.rs:40-41) is read by csrc/bpe.cpp: model.vocab {token: id}, model.merges ("a b" or
 * [a, b]), unk_token, fuse_unk, ignore_merges; normalizer none or NFC (ModernBERT's file); pre_tokenizer ByteLevel (add_prefix_space, use_regex: the
 * GPT-2 pattern) optionally behind Digits; post_processor RobertaProcessing / TemplateProcessing [<bos>] $A [<eos>] /
 * ByteLevel; special added tokens (lstrip / rstrip honoured).  Dropout, word prefixes / suffixes and byte_fallback are
 * refused.  Merges run in the crate's own queue order (lowest rank, then leftmost), so ids equal the crate's. */
CS_API int32_t cs_tokenizer_create_from_json(const char* tokenizer_json_path, uint32_t max_length,
                                      cs_tokenizer** out);
/* A model directory: tokenizer.json when present, else vocab.txt with tokenizer_config.json's
 * do_lower_case; truncation at min(max_length or 512, tokenizer_config.json's model_max_length). */
CS_API int32_t cs_tokenizer_create_from_dir(const char* model_dir, uint32_t max_length, cs_tokenizer** out);
CS_API void cs_tokenizer_destroy(cs_tokenizer* t);
CS_API uint32_t cs_tokenizer_vocab_size(const cs_tokenizer* t);
CS_API uint32_t cs_tokenizer_max_length(const cs_tokenizer* t);  /* the handle's truncation length */
CS_API int32_t cs_tokenizer_pad_id(const cs_tokenizer* t);       /* [PAD] (WordPiece) or <pad> (unigram): what encode_batch pads with */
CS_API int32_t cs_tokenizer_token_to_id(const cs_tokenizer* t, const char* token); /* -1 = absent */
/* Tokenizer::encode_batch.  Text i is utf8[offsets[i] .. offsets[i+1]) (n+1 offsets).
 * max_length 0 = the handle's.  *out_len = the batch's longest sequence L (<= max_length).
 * ids/mask: [n, row_stride] i32 with row_stride >= L, padded with [PAD] / 0; pass both NULL
 * to query L only.  Re-entrant. */
CS_API int32_t cs_tokenizer_encode_batch(const cs_tokenizer* t, const char* utf8,
                                  const uint64_t* offsets, uint32_t n, uint32_t max_length,
                                  int32_t* ids, int32_t* mask, uint32_t row_stride,
                                  uint32_t* out_len);

/* embed_batch / embed_batch_chunked from strings — embedder.rs:249-295.  Mini-batches of
 * `batch` texts (0 = the 256/128/64 policy, CODESEARCH_BATCH_SIZE honoured), each padded to
 * ITS longest sequence as fastembed does (truncation at the model's max_position).  Texts are
 * taken in windows of 16 mini-batches: the next window is tokenised on host threads while the
 * device runs the current one, and inside a window texts are grouped into mini-batches by token
 * count (padding is masked out of attention and pooling, so an embedding does not depend on its
 * batch-mates beyond f32 rounding; CS_EMBED_LENGTH_SORT=0 keeps consecutive texts together).
 * Row i of `out` is always text i.  `cancel` is polled between mini-batches.
 * out: [n, dim] f32 host memory. */
CS_API int32_t cs_embedder_embed_texts(cs_embedder* h, const cs_tokenizer* t, const char* utf8,
                                const uint64_t* offsets, uint64_t n, uint32_t batch,
                                float* out, const volatile int32_t* cancel);
/* Same, result left in HBM at d_out. */
CS_API int32_t cs_embedder_embed_texts_device(cs_embedder* h, const cs_tokenizer* t, const char* utf8,
                                       const uint64_t* offsets, uint64_t n, uint32_t batch,
                                       float* d_out, const volatile int32_t* cancel);

/* Queueing entry points for callers that keep the reference's call shape.  BatchEmbedder::embed_chunks hands the
 * embedder slices of 32 chunks, one file at a time, under a mutex (src/embed/batch.rs:70,84-115; src/embed/mod.rs:41;
 * src/index/mod.rs:692): a 32-row device batch leaves seven eighths of the chip idle.  submit_* copies the rows into the
 * library (texts are tokenised on the calling thread; inputs are borrowed for the call only) and returns a ticket;
 * wait* returns that ticket's [n, dim] embeddings.  The first wait that needs an unfinished ticket embeds EVERYTHING
 * queued so far, packed into length-grouped mini-batches of the embed_batch size (256/128/64, embedder.rs:251-261), so
 * eight queued slices of 32 run as one 256-row forward; every ticket gets its own rows back in its own order, and a
 * row's value does not depend on what it was batched with beyond f32 rounding (as cs_embedder_embed_texts).
 * submit / wait / discard are safe from several threads on one handle (the reference's callers are rayon workers
 * behind an Arc<Mutex<..>>); they must not overlap the handle's other entry points.  A wait interrupted through
 * `cancel` returns CS_ERR_CANCELLED and leaves the queue intact; a ticket is consumed by the wait that returns its
 * rows (or its error) and by discard. */
CS_API int32_t cs_embedder_submit_texts(cs_embedder* h, const cs_tokenizer* t, const char* utf8, const uint64_t* offsets,
                                 uint64_t n, uint64_t* ticket);
CS_API int32_t cs_embedder_submit_ids(cs_embedder* h, const int32_t* ids, const int32_t* mask, uint64_t n, uint32_t seq_len,
                               uint64_t* ticket);
CS_API int32_t cs_embedder_wait(cs_embedder* h, uint64_t ticket, float* out, const volatile int32_t* cancel);
CS_API int32_t cs_embedder_wait_device(cs_embedder* h, uint64_t ticket, float* d_out, const volatile int32_t* cancel);
CS_API int32_t cs_embedder_discard(cs_embedder* h, uint64_t ticket);
CS_API uint64_t cs_embedder_queued_rows(cs_embedder* h);   /* rows submitted and not yet embedded */

/* ------------------------------------------------------------------------------------
 * Encoder replicas: one cs_embedder per GPU inside ONE process, and the reference's index loop
 * (src/index/mod.rs:626-762: embed_chunks :692 -> insert_chunks_with_ids :723) over a row-sharded store.
 * replicas only — weights replicated, every GPU embeds the chunks destined for its own shard and
 * the rows are written in place; no collective.  codesearch_amd/csrc/embedders.hip.
 * One caller at a time per handle (`&mut self`), as cs_embedder_*.
 * ---------------------------------------------------------------------------------- */
typedef struct cs_embedders cs_embedders;
/* One replica per entry of `devices` (a device may appear twice: two replicas share it). */
CS_API int32_t cs_embedders_create(const cs_bert_config* cfg, const float* params, uint64_t seed, const int32_t* devices,
                            uint32_t n, cs_embedders** out);
/* (a dynamically quantised model directory brings every replica up in CS_GEMM_Q8_DYNAMIC: a call tensor is then what ONE
 * replica receives — cs_embedders_embed_* hand each replica whole mini-batches of the caller's order; in the index loop a
 * replica's mini-batch is the chunks of ITS shards, not the reference's contiguous slice.) */
CS_API int32_t cs_embedders_create_from_dir(const char* model_dir, int32_t pooling, const int32_t* devices, uint32_t n,
                                     cs_embedders** out);
CS_API void cs_embedders_destroy(cs_embedders* e);
CS_API uint32_t cs_embedders_count(const cs_embedders* e);
CS_API uint32_t cs_embedders_dim(const cs_embedders* e);
CS_API cs_embedder* cs_embedders_replica(cs_embedders* e, uint32_t i);   /* borrowed: never destroy it */
CS_API int32_t cs_embedders_device(const cs_embedders* e, uint32_t i);
/* embed_batch over all replicas: the inputs are cut into one contiguous range of whole mini-batches per replica, each
 * embedded on its own device by its own host thread; row i of `out` (host memory) is input i. */
CS_API int32_t cs_embedders_embed_texts(cs_embedders* e, const cs_tokenizer* t, const char* utf8, const uint64_t* offsets,
                                 uint64_t n, uint32_t batch, float* out, const volatile int32_t* cancel);
CS_API int32_t cs_embedders_embed_ids(cs_embedders* e, const int32_t* ids, const int32_t* mask, uint64_t n, uint32_t seq_len,
                               uint32_t batch, float* out, const volatile int32_t* cancel);
/* Embed and append in one call: every replica embeds the inputs whose ids fall on the shards it serves (the shards on
 * its own device; a shard whose device has no replica is served by replica shard %% count, its rows crossing xGMI
 * once), leaving them in its own HBM; when all replicas are done the runs are appended in id order
 * (cs_shards_add_device_parts).  out_ids (optional): the n assigned ids, contiguous from next_id (store.rs:659-685).
 * An error or a shutdown request leaves the store as it was.  Use a stripe of one mini-batch (256 rows) or a few
 * for an even spread of one call's inputs over the GPUs. */
CS_API int32_t cs_embedders_index_texts(cs_embedders* e, const cs_tokenizer* t, cs_shards* store, const char* utf8,
                                 const uint64_t* offsets, uint64_t n, uint32_t batch, uint32_t* out_ids,
                                 const volatile int32_t* cancel);
CS_API int32_t cs_embedders_index_ids(cs_embedders* e, cs_shards* store, const int32_t* ids, const int32_t* mask, uint64_t n,
                               uint32_t seq_len, uint32_t batch, uint32_t* out_ids, const volatile int32_t* cancel);

/* Arithmetic of the dense layers.  CS_GEMM_SPLIT_F16 (default): every f32 operand as two f16
 * values on the f16 MFMA, three MFMAs per product block, f32 accumulation — error per product
 * <= ~3 * 2^-22, the order of f32 rounding itself (codesearch_amd/csrc/split_f16.hpp).
 * CS_GEMM_F32: the exact-f32 MFMA (bit-for-bit an fmaf chain), 16/3 x the matrix-pipe time.
 * A mini-batch whose activations leave the f16 range (|x| > 65504) is recomputed with
 * CS_GEMM_F32 automatically; cs_embedder_debug_counters reports how often.
 * The environment variable CS_ENCODER_GEMM=f32|split sets the default at create time.
 * CS_GEMM_Q8_DYNAMIC (embedders of quantised models only, and their default): every Linear as onnxruntime's dynamic
 * quantiser rewrites it — DynamicQuantizeLinear of the layer's input (uint8, one range per call tensor, padding rows
 * included) -> MatMulInteger -> * (x_scale * W_scale) -> + bias — with the integer product on the int8 MFMA (exact).  What
 * "one call tensor" is: the sequences of ONE embed call's mini-batch (`batch` rows in the caller's order, padded to the
 * longest of them, as fastembed hands them to ORT); length grouping, token-budget batches, stream slices and the CLS tail
 * are off in this mode because each would change that tensor.  The submission queue (cs_embedder_submit_*) still embeds
 * several submissions in one device batch: each stays its own quantisation unit there (its own range per tensor, its rows
 * beyond its own padded length kept out of it), so a ticket's rows are what the call alone would have produced.
 * Attention, LayerNorm, GELU and pooling stay f32-class.  A mini-batch whose Q / K / V, attention output or GELU output
 * leaves the f16 range of that hand-over is run again as the f32 graph of the dequantised weights (CS_GEMM_F32) and
 * counted in range_fallbacks: onnxruntime has no such limit, so the call never fails for it. */
typedef enum cs_gemm_mode { CS_GEMM_F32 = 0, CS_GEMM_SPLIT_F16 = 1, CS_GEMM_Q8_DYNAMIC = 2 } cs_gemm_mode;
CS_API int32_t cs_embedder_set_gemm_mode(cs_embedder* h, int32_t mode);
/* The mode in force (a cs_gemm_mode; -1 for a null handle): CS_GEMM_Q8_DYNAMIC for a quantised model unless switched off. */
CS_API int32_t cs_embedder_gemm_mode(const cs_embedder* h);
CS_API int32_t cs_embedder_debug_counters(cs_embedder* h, uint64_t* split_forwards,
                                   uint64_t* f32_forwards, uint64_t* range_fallbacks);

/* Operator-level diagnostics (cs_debug_*: single dense layers on host buffers for the kernels' unit parity tests, timed
 * launches and ablations for the A/B scripts under benchmarks/) are NOT part of this library: they are declared in
 * include/codesearch_gpu_diag.h and exported by codesearch_amd/libcsgpu_diag.so, a second build of the same sources with
 * -DCS_DIAGNOSTICS. */

#ifdef __cplusplus
}
#endif
#endif /* CODESEARCH_GPU_H */
