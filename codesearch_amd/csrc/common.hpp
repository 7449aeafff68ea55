// common.hpp — shared host/device helpers of libcsgpu (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/codesearch_gpu.h"

namespace cs {

// ---- error plumbing: status code + thread-local message (cs_last_error) ---------------
std::string& last_error_ref();
int32_t fail(int32_t code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// onnx_reader.cpp: 1 when an initialiser's name contains `needle`, 0 when none does, -1 when the file cannot be read
int onnx_initializer_mentions(const char* path, const char* needle);

// Environment knobs.  A deployment's knobs (CODESEARCH_BATCH_SIZE, CS_ENCODER_*, CS_INDEX_SPLIT, the routing
// thresholds ...) are read with std::getenv in every build.  LABORATORY knobs — tile shapes, rejected kernel variants, fault
// injection, A/B switches of experiments that are decided — exist only in the diagnostic build (libcsgpu_diag.so,
// -DCS_DIAGNOSTICS): in the product library cs_lab_env is a constant null and the branches behind it compile away.
inline const char* cs_lab_env(const char* name) {
#ifdef CS_DIAGNOSTICS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

#define CS_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return ::cs::fail(_e == hipErrorOutOfMemory ? CS_ERR_OOM : CS_ERR_HIP,        \
                              "HIP error %d (%s) at %s:%d: %s", (int)_e,                  \
                              hipGetErrorString(_e), __FILE__, __LINE__, #expr);          \
    } while (0)

#define CS_TRY(expr)                      \
    do {                                  \
        int32_t _s = (expr);              \
        if (_s != CS_OK) return _s;       \
    } while (0)

// Runs `f` once per device (the current one), under a lock: kernel function attributes such as the dynamic-LDS limit
// belong to the device they were set on, and one process may drive several (cs_shards_*, one embedder per GPU).
struct PerDeviceOnce {
    std::mutex mu;
    uint64_t done = 0;
    template <class F>
    int32_t run(F&& f) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) d = 0;
        const uint64_t bit = 1ull << (d & 63);
        std::lock_guard<std::mutex> lk(mu);
        if (done & bit) return CS_OK;
        const int32_t s = f();
        if (s == CS_OK) done |= bit;
        return s;
    }
};

// RAII: make `device` current for the calling thread, restore on scope exit.
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) {
            changed = (hipSetDevice(device) == hipSuccess);
        }
    }
    ~DeviceGuard() {
        if (changed) (void)hipSetDevice(prev);
    }
};

// ---- packed sort key (mirrors cs_key_* in the public header) ------------------------------
// The id of a corpus row.  A never-compacted index numbers its rows in storage order (id = base + row, store.rs:659-685); once
// cs_index_build has squeezed deleted rows out, the surviving rows keep their ids through a row -> id table in HBM (ascending, so
// "(cosine desc, id asc)" and the strict-> insert rule are untouched).  The table is read on the candidate path only — a few
// rows per search, never per row scanned.
struct RowIds {
    uint32_t base;
    const uint32_t* ids;  // [rows] or null (identity)
    __host__ __device__ RowIds(uint32_t b = 0, const uint32_t* t = nullptr) : base(b), ids(t) {}
    __device__ __forceinline__ uint32_t of(uint64_t row) const { return ids ? ids[row] : base + (uint32_t)row; }
};

__host__ __device__ __forceinline__ uint64_t key_pack(float c, uint32_t id) {
    c = c + 0.0f;  // -0 -> +0
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t u = __float_as_uint(c);
#else
    union { float f; uint32_t u; } v; v.f = c; uint32_t u = v.u;
#endif
    uint32_t o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((uint64_t)o << 32) | (uint64_t)(~id);
}
__host__ __device__ __forceinline__ float key_cos(uint64_t key) {
    uint32_t o = (uint32_t)(key >> 32);
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    union { float f; uint32_t u; } v; v.u = u; return v.f;
#endif
}
__host__ __device__ __forceinline__ uint32_t key_id(uint64_t key) { return ~(uint32_t)key; }

// tokenizer.cpp: per-text id lists ([CLS] ... [SEP], truncated to max_length), texts in parallel
void tokenize_texts(const cs_tokenizer* t, const char* utf8, const uint64_t* offsets, uint32_t n,
                    uint32_t max_length, std::vector<std::vector<int32_t>>& out);

inline uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace cs
