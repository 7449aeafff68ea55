// diagnostics.hip — cs_debug_*: operator-level entry points for the parity tests of single kernels and for the A/B scripts
// under benchmarks/.  NOT part of libcsgpu.so: built into libcsgpu_diag.so only (-DCS_DIAGNOSTICS, include/codesearch_gpu_diag.h).
#include "embedder_state.hpp"
#include "scan.hpp"  // launch_synth_fill (cs_debug_gemm_time)
#include "../../include/codesearch_gpu_diag.h"

using namespace cs;
using namespace cs::emb;

extern "C" {

int32_t cs_debug_gemm(int32_t device, int32_t mode, int32_t epilogue, const float* A, const float* W,
                      const float* bias, const float* resid, float* C, uint32_t M, uint32_t N, uint32_t K,
                      uint32_t* range_flag) {
    if (!A || !W || !bias || !C || ((epilogue == 2 || epilogue == 3) && !resid)) return fail(CS_ERR_BAD_ARG, "null buffer");
    const bool wide = mode == 2;  // diagnostics only: the 128 x 384 one-accumulator kernel whatever M is
    if (wide) mode = CS_GEMM_SPLIT_F16;
    // epilogue 3 (wide only, N = 384): + resid, LayerNorm with gamma = bias + 1, beta = -bias, eps 1e-12; C receives
    // the f32 output re-assembled from the SPLIT output (hi + lo / 2048), so both stores are exercised
    if (epilogue == 3 && !(wide && N == 384)) return fail(CS_ERR_UNSUPPORTED, "epilogue 3 needs mode 2 and N = 384");
    if (epilogue < 0 || epilogue > 4 || (mode != CS_GEMM_F32 && mode != CS_GEMM_SPLIT_F16))
        return fail(CS_ERR_BAD_ARG, "unknown epilogue/mode");
    if (wide && !gemm_wide_supported(N, K)) return fail(CS_ERR_UNSUPPORTED, "wide kernel needs N %% 384 == 0");
    if (M == 0 || N % 128 || K % 32 || K == 0) return fail(CS_ERR_UNSUPPORTED, "cs_debug_gemm needs M > 0, N %% 128 == 0, K %% 32 == 0");
    int ndev = 0;
    CS_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(CS_ERR_HIP, "HIP device %d not available (%d visible)", device, ndev);
    DeviceGuard g(device);
    const size_t a_n = (size_t)M * K, w_n = (size_t)N * K, c_n = (size_t)M * N;
    float *dA = nullptr, *dW = nullptr, *dB = nullptr, *dR = nullptr, *dC = nullptr;
    _Float16 *sA = nullptr, *sW = nullptr, *sC = nullptr;
    uint32_t* dF = nullptr;
    int32_t st = CS_OK;
    auto run = [&]() -> int32_t {
        CS_HIP(hipMalloc(&dA, a_n * 4)); CS_HIP(hipMalloc(&dW, w_n * 4)); CS_HIP(hipMalloc(&dB, (size_t)N * 4));
        CS_HIP(hipMalloc(&dC, c_n * 4)); CS_HIP(hipMalloc(&dF, 4));
        CS_HIP(hipMemcpy(dA, A, a_n * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dW, W, w_n * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dB, bias, (size_t)N * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemset(dF, 0, 4));
        if (epilogue >= 2) {
            CS_HIP(hipMalloc(&dR, c_n * 4));
            CS_HIP(hipMemcpy(dR, resid, c_n * 4, hipMemcpyHostToDevice));
        }
        if (mode == CS_GEMM_F32) {
            CS_TRY(launch_gemm(epilogue, dA, dW, dB, dR, dC, M, N, K, nullptr));
        } else {
            CS_HIP(hipMalloc(&sA, a_n * 4)); CS_HIP(hipMalloc(&sW, w_n * 4));
            CS_TRY(launch_split_rows(dA, sA, M, K, dF, nullptr));
            CS_TRY(launch_split_rows(dW, sW, N, K, dF, nullptr));
            auto run_gemm = [&](int e, const _Float16* a_, const _Float16* w_, const float* b_, const float* r_, float* c_, _Float16* cs_,
                                uint32_t m_, uint32_t n_, uint32_t k_, uint32_t* f_, hipStream_t st_) {
                return wide ? launch_gemm_wide(e, a_, w_, b_, r_, c_, cs_, m_, n_, k_, f_, st_, 0) : launch_gemm_split(e, a_, w_, b_, r_, c_, cs_, m_, n_, k_, f_, st_);
            };
            if (epilogue == 4) {  // LayerNorm epilogue, residual given (and overwritten) in split form, no f32 output
                std::vector<float> gam(N), bet(N);
                for (uint32_t n = 0; n < N; ++n) { gam[n] = bias[n] + 1.0f; bet[n] = -bias[n]; }
                float *dG = nullptr, *dBe = nullptr;
                CS_HIP(hipMalloc(&dG, (size_t)N * 4)); CS_HIP(hipMalloc(&dBe, (size_t)N * 4));
                CS_HIP(hipMemcpy(dG, gam.data(), (size_t)N * 4, hipMemcpyHostToDevice));
                CS_HIP(hipMemcpy(dBe, bet.data(), (size_t)N * 4, hipMemcpyHostToDevice));
                CS_HIP(hipMalloc(&sC, c_n * 4));
                CS_TRY(launch_split_rows(dR, sC, M, N, dF, nullptr));
                const int32_t st4 = launch_gemm_wide_ln(sA, sW, dB, nullptr, dG, dBe, 1e-12f, nullptr, sC, M, K, dF, nullptr, sC);
                CS_HIP(hipDeviceSynchronize());
                (void)hipFree(dG); (void)hipFree(dBe);
                CS_TRY(st4);
            } else if (epilogue == 3) {
                std::vector<float> gam(N), bet(N);
                for (uint32_t n = 0; n < N; ++n) { gam[n] = bias[n] + 1.0f; bet[n] = -bias[n]; }
                float *dG = nullptr, *dBe = nullptr;
                CS_HIP(hipMalloc(&dG, (size_t)N * 4)); CS_HIP(hipMalloc(&dBe, (size_t)N * 4));
                CS_HIP(hipMemcpy(dG, gam.data(), (size_t)N * 4, hipMemcpyHostToDevice));
                CS_HIP(hipMemcpy(dBe, bet.data(), (size_t)N * 4, hipMemcpyHostToDevice));
                CS_HIP(hipMalloc(&sC, c_n * 4));
                const int32_t st3 = launch_gemm_wide_ln(sA, sW, dB, dR, dG, dBe, 1e-12f, dR, sC, M, K, dF, nullptr);  // in place over resid
                CS_HIP(hipDeviceSynchronize());
                std::vector<float> f32out(c_n);
                CS_HIP(hipMemcpy(f32out.data(), dR, c_n * 4, hipMemcpyDeviceToHost));
                (void)hipFree(dG); (void)hipFree(dBe);
                CS_TRY(st3);
                // the two outputs must describe the same values: checked here, the split one is what C receives below
                std::vector<_Float16> hs(c_n * 2);
                CS_HIP(hipMemcpy(hs.data(), sC, c_n * 4, hipMemcpyDeviceToHost));
                for (size_t m = 0; m < M; ++m)
                    for (size_t n = 0; n < N; ++n) {
                        const _Float16* line = hs.data() + (m * (N / 32) + n / 32) * 64;
                        const float v = (float)line[n % 32] + (float)line[32 + n % 32] * (1.0f / 2048.0f);
                        if (!(fabsf(v - f32out[m * N + n]) <= 1e-6f * fmaxf(1.0f, fabsf(v))))
                            return fail(CS_ERR_HIP, "LayerNorm epilogue: f32 and split outputs disagree at (%zu, %zu): %g vs %g", m, n,
                                        (double)f32out[m * N + n], (double)v);
                    }
            } else if (epilogue == 1) {  // the GELU epilogue writes split form: read it back through hi + lo / 2048
                CS_HIP(hipMalloc(&sC, c_n * 4));
                CS_TRY(run_gemm(SH_OUT_SPLIT_GELU, sA, sW, dB, nullptr, nullptr, sC, M, N, K, dF, nullptr));
            } else {
                CS_TRY(run_gemm(epilogue == 2 ? SH_OUT_F32_RESID : SH_OUT_F32, sA, sW, dB, dR, dC, nullptr, M, N, K, dF, nullptr));
            }
        }
        CS_HIP(hipDeviceSynchronize());
        if (sC) {
            std::vector<_Float16> hs(c_n * 2);
            CS_HIP(hipMemcpy(hs.data(), sC, c_n * 4, hipMemcpyDeviceToHost));
            const size_t nch = N / 32;
            for (size_t m = 0; m < M; ++m)
                for (size_t n = 0; n < N; ++n) {
                    const _Float16* line = hs.data() + (m * nch + n / 32) * 64;
                    C[m * N + n] = (float)line[n % 32] + (float)line[32 + n % 32] * (1.0f / 2048.0f);
                }
        } else {
            CS_HIP(hipMemcpy(C, dC, c_n * 4, hipMemcpyDeviceToHost));
        }
        if (range_flag) CS_HIP(hipMemcpy(range_flag, dF, 4, hipMemcpyDeviceToHost));
        return CS_OK;
    };
    st = run();
    for (void* p : {(void*)dA, (void*)dW, (void*)dB, (void*)dR, (void*)dC, (void*)sA, (void*)sW, (void*)sC, (void*)dF})
        if (p) (void)hipFree(p);
    return st;
}

int32_t cs_debug_gemm_q8(int32_t device, int32_t epilogue, int32_t a_split, const float* A, const float* W,
                         const float* wscale, const float* bias, const float* resid, float* C, uint32_t M, uint32_t N,
                         uint32_t K, uint8_t* xq_out, float* xparams, int32_t* acc_out) {
    if (!A || !W || !wscale || !bias || !C || (epilogue == 2 && !resid)) return fail(CS_ERR_BAD_ARG, "null buffer");
    if (epilogue != 0 && epilogue != 1 && epilogue != 2 && epilogue != 4 && epilogue != 5) return fail(CS_ERR_BAD_ARG, "unknown epilogue %d", epilogue);
    if (M == 0 || N % 128 || K % 128 || K == 0) return fail(CS_ERR_UNSUPPORTED, "cs_debug_gemm_q8 needs M > 0, N %% 128 == 0, K %% 128 == 0");
    int ndev = 0;
    CS_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(CS_ERR_HIP, "HIP device %d not available (%d visible)", device, ndev);
    DeviceGuard g(device);
    const size_t a_n = (size_t)M * K, w_n = (size_t)N * K, c_n = (size_t)M * N;
    float *dA = nullptr, *dW = nullptr, *dS = nullptr, *dB = nullptr, *dR = nullptr, *dC = nullptr;
    _Float16 *sA = nullptr, *sC = nullptr;
    int8_t *dXq = nullptr, *dWq = nullptr;
    Q8RowMeta* dRm = nullptr;
    Q8ColMeta* dCm = nullptr;
    uint32_t *dF = nullptr, *dRange = nullptr, *dCmT = nullptr;
    int32_t* dAcc = nullptr;
    auto run = [&]() -> int32_t {
        CS_HIP(hipMalloc(&dA, a_n * 4)); CS_HIP(hipMalloc(&dW, w_n * 4)); CS_HIP(hipMalloc(&dS, (size_t)N * 4));
        CS_HIP(hipMalloc(&dB, (size_t)N * 4)); CS_HIP(hipMalloc(&dC, c_n * 4)); CS_HIP(hipMalloc(&dF, 16));
        CS_HIP(hipMalloc(&dXq, a_n)); CS_HIP(hipMalloc(&dWq, w_n)); CS_HIP(hipMalloc(&dRm, (size_t)M * sizeof(Q8RowMeta)));
        CS_HIP(hipMalloc(&dCm, (size_t)N * sizeof(Q8ColMeta))); CS_HIP(hipMalloc(&dRange, Q8_RANGE_WORDS * 4));
        CS_HIP(hipMemcpy(dA, A, a_n * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dW, W, w_n * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dS, wscale, (size_t)N * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dB, bias, (size_t)N * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemset(dF, 0, 16));
        CS_HIP(hipMemset(dRange, 0, Q8_RANGE_WORDS * 4));
        if (epilogue == 2) {
            CS_HIP(hipMalloc(&dR, c_n * 4));
            CS_HIP(hipMemcpy(dR, resid, c_n * 4, hipMemcpyHostToDevice));
        }
        if (acc_out && epilogue != 5) CS_HIP(hipMalloc(&dAcc, c_n * 4));
        CS_TRY(launch_q8_pack_weight(dW, dS, dB, N, K, dWq, dCm, dF + 1, nullptr));
        if (a_split & 16) {
            if ((a_split & 1) || K != 384 || (epilogue != 4 && epilogue != 5)) return fail(CS_ERR_UNSUPPORTED, "cs_debug_gemm_q8: the slab kernel takes f32 rows, K = 384, epilogue 4 | 5");
            CS_HIP(hipMalloc(&dCmT, (size_t)N * sizeof(Q8ColMeta)));
            CS_TRY(launch_q8_cmeta_tiles(dCm, N, dCmT, nullptr));
        }
        if (a_split & 1) {
            CS_HIP(hipMalloc(&sA, a_n * 4));
            CS_TRY(launch_split_rows(dA, sA, M, K, dF, nullptr));
            CS_TRY(launch_q8_quantize(Q8_SRC_SPLIT, sA, M, K, dRange, nullptr, dXq, dRm, nullptr));
        } else {
            CS_TRY(launch_q8_quantize(Q8_SRC_F32, dA, M, K, dRange, nullptr, dXq, dRm, nullptr));
        }
        if (epilogue == 5) {  // GELU -> re-quantised (the two-pass FFN-up): C = the uint8 output, xparams[2..3] = its scale / zero point
            int8_t* dOut = nullptr;
            Q8RowMeta* dRm2 = nullptr;
            uint32_t* dRange2 = nullptr;
            CS_HIP(hipMalloc(&dOut, c_n)); CS_HIP(hipMalloc(&dRm2, (size_t)M * sizeof(Q8RowMeta))); CS_HIP(hipMalloc(&dRange2, Q8_RANGE_WORDS * 4));
            CS_HIP(hipMemset(dRange2, 0, Q8_RANGE_WORDS * 4));
            // (a_split & 16: the slab kernel, whatever M — gemm_q8_slab.hip)
            int32_t st5 = (a_split & 16) ? launch_gemm_q8_slab_gelu_requant(dA, dRange, dWq, dCmT, M, N, K, dRange2, dOut, dRm2, q8_gelu_table_on(), nullptr, (a_split & 32) ? dXq : nullptr)
                        : (a_split & 8) ? launch_gemm_q8_gelu_requant_from_source(dA, dRange, dWq, dCm, dB, M, N, K, dRange2, dOut, dRm2, nullptr)
                                        : launch_gemm_q8_gelu_requant(dXq, dRm, dWq, dCm, dB, M, N, K, dRange2, dOut, dRm2, nullptr);
            if (st5 == CS_OK && hipDeviceSynchronize() != hipSuccess) st5 = fail(CS_ERR_HIP, "requant GEMM failed");
            std::vector<int8_t> ho(c_n);
            std::vector<Q8RowMeta> hr(M);
            if (st5 == CS_OK && (hipMemcpy(ho.data(), dOut, c_n, hipMemcpyDeviceToHost) != hipSuccess ||
                                 hipMemcpy(hr.data(), dRm2, (size_t)M * sizeof(Q8RowMeta), hipMemcpyDeviceToHost) != hipSuccess))
                st5 = fail(CS_ERR_HIP, "requant GEMM read-back failed");
            (void)hipFree(dOut); (void)hipFree(dRm2); (void)hipFree(dRange2);
            CS_TRY(st5);
            for (size_t i = 0; i < c_n; ++i) C[i] = (float)((int)ho[i] + 128);
            if (acc_out) for (size_t m = 0; m < M; ++m) acc_out[m] = hr[m].rowsum + 128 * (int32_t)N;  // row sums of the uint8 output
            if (xparams) { xparams[2] = hr[0].xs; xparams[3] = (float)(hr[0].za + 128); }
            uint32_t flags5[2] = {0, 0};
            CS_HIP(hipMemcpy(flags5, dF, 8, hipMemcpyDeviceToHost));
            if (flags5[1]) return fail(CS_ERR_BAD_ARG, "cs_debug_gemm_q8: W is not a quantised matrix for these column scales (flag %u)", flags5[1]);
            if (xq_out || xparams) {
                std::vector<int8_t> hq(a_n);
                Q8RowMeta rm0;
                CS_HIP(hipMemcpy(hq.data(), dXq, a_n, hipMemcpyDeviceToHost));
                CS_HIP(hipMemcpy(&rm0, dRm, sizeof(rm0), hipMemcpyDeviceToHost));
                if (xq_out) for (size_t i = 0; i < a_n; ++i) xq_out[i] = (uint8_t)((int)hq[i] + 128);
                if (xparams) { xparams[0] = rm0.xs; xparams[1] = (float)(rm0.za + 128); }
            }
            return CS_OK;
        }
        const int epi = epilogue == 0 ? SH_OUT_F32 : epilogue == 1 ? SH_OUT_SPLIT_GELU : epilogue == 2 ? SH_OUT_F32_RESID : SH_OUT_SPLIT;
        if (epi == SH_OUT_SPLIT_GELU || epi == SH_OUT_SPLIT) CS_HIP(hipMalloc(&sC, c_n * 4));
        if (a_split & 4) {  // the few-rows kernel: its "pairs" are the one (lo, hi) in the slot (the words are the floats' bits)
            if (dAcc) CS_HIP(hipMemset(dAcc, 0, c_n * 4));
            CS_TRY(launch_gemm_q8_skinny(epi, (a_split & 1) ? Q8_SRC_SPLIT : Q8_SRC_F32, (a_split & 1) ? (const void*)sA : (const void*)dA,
                                         reinterpret_cast<const float*>(dRange), 1, dWq, dCm, dR, dC, sC, M, N, K, dF, nullptr, nullptr, nullptr));
        } else if (a_split & 16) {  // the slab kernel (acc is not reported)
            if (dAcc) CS_HIP(hipMemset(dAcc, 0, c_n * 4));
            CS_TRY(launch_gemm_q8_slab_split(dA, dRange, dWq, dCmT, sC, M, N, K, dF, nullptr));
        } else if (a_split & 8) {  // the products that quantise their own rows on the way in (row-block kernel; acc is not reported)
            if (dAcc) CS_HIP(hipMemset(dAcc, 0, c_n * 4));
            CS_TRY(launch_gemm_q8_from_source(epi, (a_split & 1) ? Q8_SRC_SPLIT : Q8_SRC_F32, (a_split & 1) ? (const void*)sA : (const void*)dA, dRange,
                                              dWq, dCm, dB, dR, dC, sC, M, N, K, dF, nullptr));
        } else
        CS_TRY(launch_gemm_q8(epi, dXq, dRm, dWq, dCm, dB, dR, dC, sC, M, N, K, dF, nullptr, dAcc));
        CS_HIP(hipDeviceSynchronize());
        uint32_t flags[2] = {0, 0};
        CS_HIP(hipMemcpy(flags, dF, 8, hipMemcpyDeviceToHost));
        if (flags[1]) return fail(CS_ERR_BAD_ARG, "cs_debug_gemm_q8: W is not a quantised matrix for these column scales (flag %u)", flags[1]);
        if (sC) {
            std::vector<_Float16> hs(c_n * 2);
            CS_HIP(hipMemcpy(hs.data(), sC, c_n * 4, hipMemcpyDeviceToHost));
            const size_t nch = N / 32;
            for (size_t m = 0; m < M; ++m)
                for (size_t n = 0; n < N; ++n) {
                    const _Float16* line = hs.data() + (m * nch + n / 32) * 64;
                    C[m * N + n] = (float)line[n % 32] + (float)line[32 + n % 32] * (1.0f / 2048.0f);
                }
        } else {
            CS_HIP(hipMemcpy(C, dC, c_n * 4, hipMemcpyDeviceToHost));
        }
        if (acc_out) CS_HIP(hipMemcpy(acc_out, dAcc, c_n * 4, hipMemcpyDeviceToHost));
        if (xq_out || xparams) {
            std::vector<int8_t> hq(a_n);
            Q8RowMeta rm0;
            CS_HIP(hipMemcpy(hq.data(), dXq, a_n, hipMemcpyDeviceToHost));
            CS_HIP(hipMemcpy(&rm0, dRm, sizeof(rm0), hipMemcpyDeviceToHost));
            if (xq_out) for (size_t i = 0; i < a_n; ++i) xq_out[i] = (uint8_t)((int)hq[i] + 128);
            if (xparams) { xparams[0] = rm0.xs; xparams[1] = (float)(rm0.za + 128); }
        }
        return CS_OK;
    };
    const int32_t st = run();
    for (void* p : {(void*)dA, (void*)dW, (void*)dS, (void*)dB, (void*)dR, (void*)dC, (void*)sA, (void*)sC, (void*)dXq, (void*)dWq,
                    (void*)dRm, (void*)dCm, (void*)dF, (void*)dRange, (void*)dAcc, (void*)dCmT})
        if (p) (void)hipFree(p);
    return st;
}

// Diagnostics: the row-block products over a tensor of SEVERAL quantisation units (queued calls in one device batch):
// row_slot [M] as launch_q8_quantize takes it.  epilogue 4 (f32 source -> split store), 2 (split source, + residual) or
// 5 (FFN-up: GELU, quantised again per unit).  row_params [M][4] = per row (x_scale, x_zero_point, out_scale,
// out_zero_point) — the last two only for epilogue 5, where rowsums [M] receives each output row's sum of uint8 values.
int32_t cs_debug_gemm_q8_units(int32_t device, int32_t epilogue, const float* A, const float* W, const float* wscale,
                               const float* bias, const float* resid, float* C, uint32_t M, uint32_t N, uint32_t K,
                               const uint32_t* row_slot, uint32_t units, float* row_params, int32_t* rowsums) {
    if (!A || !W || !wscale || !bias || !C || !row_slot || (epilogue == 2 && !resid)) return fail(CS_ERR_BAD_ARG, "null buffer");
    if (epilogue != 2 && epilogue != 4 && epilogue != 5) return fail(CS_ERR_BAD_ARG, "unknown epilogue %d", epilogue);
    if (M == 0 || units == 0 || N % 128 || !q8_rows_from_source(M, K))
        return fail(CS_ERR_UNSUPPORTED, "cs_debug_gemm_q8_units: M=%u N=%u K=%u is not a row-block product", M, N, K);
    for (uint32_t m = 0; m < M; ++m)
        if ((row_slot[m] & 0x7fffffffu) >= units || (m && (row_slot[m] & 0x7fffffffu) < (row_slot[m - 1] & 0x7fffffffu)))
            return fail(CS_ERR_BAD_ARG, "row_slot[%u]: units must be consecutive runs of rows, in order", m);
    int ndev = 0;
    CS_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(CS_ERR_HIP, "HIP device %d not available (%d visible)", device, ndev);
    DeviceGuard g(device);
    const size_t a_n = (size_t)M * K, w_n = (size_t)N * K, c_n = (size_t)M * N;
    float *dA = nullptr, *dW = nullptr, *dS = nullptr, *dB = nullptr, *dR = nullptr, *dC = nullptr;
    _Float16 *sA = nullptr, *sC = nullptr;
    int8_t *dXq = nullptr, *dWq = nullptr, *dOut = nullptr;
    Q8RowMeta *dRm = nullptr, *dRm2 = nullptr;
    Q8ColMeta* dCm = nullptr;
    uint32_t *dF = nullptr, *dRange = nullptr, *dRange2 = nullptr, *dSlot = nullptr;
    auto run = [&]() -> int32_t {
        const size_t rbytes = (size_t)units * Q8_RANGE_WORDS * 4;
        CS_HIP(hipMalloc(&dA, a_n * 4)); CS_HIP(hipMalloc(&dW, w_n * 4)); CS_HIP(hipMalloc(&dS, (size_t)N * 4));
        CS_HIP(hipMalloc(&dB, (size_t)N * 4)); CS_HIP(hipMalloc(&dC, c_n * 4)); CS_HIP(hipMalloc(&dF, 16));
        CS_HIP(hipMalloc(&dXq, a_n)); CS_HIP(hipMalloc(&dWq, w_n)); CS_HIP(hipMalloc(&dRm, (size_t)M * sizeof(Q8RowMeta)));
        CS_HIP(hipMalloc(&dCm, (size_t)N * sizeof(Q8ColMeta))); CS_HIP(hipMalloc(&dRange, rbytes)); CS_HIP(hipMalloc(&dRange2, rbytes));
        CS_HIP(hipMalloc(&dSlot, (size_t)M * 4));
        CS_HIP(hipMemcpy(dA, A, a_n * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dW, W, w_n * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dS, wscale, (size_t)N * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dB, bias, (size_t)N * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemcpy(dSlot, row_slot, (size_t)M * 4, hipMemcpyHostToDevice));
        CS_HIP(hipMemset(dF, 0, 16));
        CS_HIP(hipMemset(dRange, 0, rbytes));
        CS_HIP(hipMemset(dRange2, 0, rbytes));
        CS_TRY(launch_q8_pack_weight(dW, dS, dB, N, K, dWq, dCm, dF + 1, nullptr));
        // the units' ranges by a pass over the tensor (the quantised rows this also writes only serve row_params)
        if (epilogue == 2) {
            CS_HIP(hipMalloc(&sA, a_n * 4));
            CS_HIP(hipMalloc(&dR, c_n * 4));
            CS_HIP(hipMemcpy(dR, resid, c_n * 4, hipMemcpyHostToDevice));
            CS_TRY(launch_split_rows(dA, sA, M, K, dF, nullptr));
            CS_TRY(launch_q8_quantize(Q8_SRC_SPLIT, sA, M, K, dRange, dSlot, dXq, dRm, nullptr));
            CS_TRY(launch_gemm_q8_from_source(SH_OUT_F32_RESID, Q8_SRC_SPLIT, sA, dRange, dWq, dCm, dB, dR, dC, nullptr, M, N, K, dF, nullptr, dSlot));
        } else {
            CS_TRY(launch_q8_quantize(Q8_SRC_F32, dA, M, K, dRange, dSlot, dXq, dRm, nullptr));
            if (epilogue == 4) {
                CS_HIP(hipMalloc(&sC, c_n * 4));
                CS_TRY(launch_gemm_q8_from_source(SH_OUT_SPLIT, Q8_SRC_F32, dA, dRange, dWq, dCm, dB, nullptr, nullptr, sC, M, N, K, dF, nullptr, dSlot));
            } else {
                CS_HIP(hipMalloc(&dOut, c_n));
                CS_HIP(hipMalloc(&dRm2, (size_t)M * sizeof(Q8RowMeta)));
                CS_TRY(launch_gemm_q8_gelu_requant_from_source(dA, dRange, dWq, dCm, dB, M, N, K, dRange2, dOut, dRm2, nullptr, dSlot));
            }
        }
        CS_HIP(hipDeviceSynchronize());
        uint32_t flags[2] = {0, 0};
        CS_HIP(hipMemcpy(flags, dF, 8, hipMemcpyDeviceToHost));
        if (flags[1]) return fail(CS_ERR_BAD_ARG, "cs_debug_gemm_q8_units: W is not a quantised matrix for these column scales (flag %u)", flags[1]);
        std::vector<Q8RowMeta> hr(M), hr2;
        CS_HIP(hipMemcpy(hr.data(), dRm, (size_t)M * sizeof(Q8RowMeta), hipMemcpyDeviceToHost));
        if (epilogue == 5) {
            std::vector<int8_t> ho(c_n);
            hr2.resize(M);
            CS_HIP(hipMemcpy(ho.data(), dOut, c_n, hipMemcpyDeviceToHost));
            CS_HIP(hipMemcpy(hr2.data(), dRm2, (size_t)M * sizeof(Q8RowMeta), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < c_n; ++i) C[i] = (float)((int)ho[i] + 128);
            if (rowsums) for (size_t m = 0; m < M; ++m) rowsums[m] = hr2[m].rowsum + 128 * (int32_t)N;
        } else if (sC) {
            std::vector<_Float16> hs(c_n * 2);
            CS_HIP(hipMemcpy(hs.data(), sC, c_n * 4, hipMemcpyDeviceToHost));
            const size_t nch = N / 32;
            for (size_t m = 0; m < M; ++m)
                for (size_t n = 0; n < N; ++n) {
                    const _Float16* line = hs.data() + (m * nch + n / 32) * 64;
                    C[m * N + n] = (float)line[n % 32] + (float)line[32 + n % 32] * (1.0f / 2048.0f);
                }
        } else {
            CS_HIP(hipMemcpy(C, dC, c_n * 4, hipMemcpyDeviceToHost));
        }
        if (row_params)
            for (size_t m = 0; m < M; ++m) {
                row_params[4 * m] = hr[m].xs;
                row_params[4 * m + 1] = (float)(hr[m].za + 128);
                row_params[4 * m + 2] = epilogue == 5 ? hr2[m].xs : 0.0f;
                row_params[4 * m + 3] = epilogue == 5 ? (float)(hr2[m].za + 128) : 0.0f;
            }
        return CS_OK;
    };
    const int32_t st = run();
    for (void* p : {(void*)dA, (void*)dW, (void*)dS, (void*)dB, (void*)dR, (void*)dC, (void*)sA, (void*)sC, (void*)dXq, (void*)dWq,
                    (void*)dOut, (void*)dRm, (void*)dRm2, (void*)dCm, (void*)dF, (void*)dRange, (void*)dRange2, (void*)dSlot})
        if (p) (void)hipFree(p);
    return st;
}

// Diagnostics: device time of one dense layer on synthetic operands already in HBM (no PCIe, no allocation inside the
// timed region).  mode as cs_debug_gemm (0 f32 MFMA, 1 split-f16 128 x 128 / skinny kernels, 2 split-f16 wide kernel);
// epilogue 0 f32 store, 1 GELU -> split store, 2 + residual, 3 LayerNorm-fused (mode 2, N = 384), 4 bias -> split store
// (the QKV projection).  `ablation` (mode 2, epilogue 4 only): 1 no LDS-DMA, 2 no MFMA, 3 DMAs issued at the step start.
int32_t cs_debug_gemm_time(int32_t device, int32_t mode, int32_t epilogue, uint32_t M, uint32_t N, uint32_t K,
                           uint32_t iters, int32_t ablation, double* ms_per_launch) {
    if (!ms_per_launch || iters == 0 || M == 0 || N % 128 || K % 32 || K == 0) return fail(CS_ERR_BAD_ARG, "bad arguments");
    int ndev = 0;
    CS_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(CS_ERR_HIP, "HIP device %d not available (%d visible)", device, ndev);
    DeviceGuard g(device);
    const size_t a_n = (size_t)M * K, w_n = (size_t)N * K, c_n = (size_t)M * N;
    float *dA = nullptr, *dW = nullptr, *dB = nullptr, *dR = nullptr, *dC = nullptr;
    _Float16 *sA = nullptr, *sW = nullptr, *sC = nullptr;
    uint32_t* dF = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto run = [&]() -> int32_t {
        CS_HIP(hipMalloc(&dA, a_n * 4)); CS_HIP(hipMalloc(&dW, w_n * 4)); CS_HIP(hipMalloc(&dB, (size_t)N * 4));
        CS_HIP(hipMalloc(&dC, c_n * 4)); CS_HIP(hipMalloc(&dR, c_n * 4)); CS_HIP(hipMalloc(&dF, 4));
        CS_HIP(hipMalloc(&sA, a_n * 4)); CS_HIP(hipMalloc(&sW, w_n * 4)); CS_HIP(hipMalloc(&sC, c_n * 4));
        // operands from the counter-based generator: unit-scale activations, weights / 20
        CS_TRY(launch_synth_fill(dA, M, K, 11, 0, nullptr));
        CS_TRY(launch_synth_fill(dW, N, K, 12, 0, nullptr));
        CS_TRY(launch_synth_fill(dR, M, N, 13, 0, nullptr));
        if (cs_lab_env("CS_DEBUG_GEMM_ZERO")) {  // all-zero operands: what the clock (DVFS), not the schedule, is worth
            CS_HIP(hipMemset(dA, 0, a_n * 4)); CS_HIP(hipMemset(dW, 0, w_n * 4)); CS_HIP(hipMemset(dR, 0, c_n * 4));
        }
        CS_HIP(hipMemset(dB, 0, (size_t)N * 4));
        CS_HIP(hipMemset(dF, 0, 4));
        CS_TRY(launch_split_rows(dA, sA, M, K, dF, nullptr));
        CS_TRY(launch_split_rows(dW, sW, N, K, dF, nullptr));
        CS_HIP(hipEventCreate(&e0)); CS_HIP(hipEventCreate(&e1));
        auto once = [&]() -> int32_t {
            if (mode == CS_GEMM_F32)
                return launch_gemm(epilogue == 1 ? GEMM_GELU : epilogue == 2 ? GEMM_RESID : GEMM_BIAS, dA, dW, dB, dR, dC, M, N, K, nullptr);
            if (epilogue == 3) return launch_gemm_wide_ln(sA, sW, dB, dR, dB, dB, 1e-12f, dR, sC, M, K, dF, nullptr);
            const int epi = epilogue == 0 ? SH_OUT_F32 : epilogue == 1 ? SH_OUT_SPLIT_GELU : epilogue == 2 ? SH_OUT_F32_RESID : SH_OUT_SPLIT;
            if (mode == 2) return launch_gemm_wide(epi, sA, sW, dB, dR, dC, sC, M, N, K, dF, nullptr, 0);
            return launch_gemm_split(epi, sA, sW, dB, dR, dC, sC, M, N, K, dF, nullptr);
        };
        // ablation >= 100: DMA schedule ablation - 100 of the product kernel (gemm_wide.hip gw_dma_slot), any epilogue
        // ablation 192 / 384: that block shape of the product kernel, any epilogue
        // ablation 3192 / 3384: that block shape with the main loop on the 32 x 32 x 16 MFMA (gemm_wide32.hip); 1192 / 1384:
        // the 16 x 16 x 32 form whatever CS_GEMM_WIDE_MFMA says
        if (mode == 2 && (ablation == 3192 || ablation == 3384 || ablation == 1192 || ablation == 1384)) {
            cs::g_gemm_wide_mfma = ablation >= 3000 ? 32 : 16;
            ablation %= 1000;
        }
        cs::g_gemm_wide_shape = (mode == 2 && (ablation == 192 || ablation == 384 || ablation == 256)) ? ablation : 0;  // (256: the 256 x 192 block)
        cs::g_gemm_wide_ablation = (mode == 2 && epilogue == 4 && ablation < 100) ? ablation : 0;
        for (int i = 0; i < 3; ++i) CS_TRY(once());
        CS_HIP(hipEventRecord(e0, nullptr));
        for (uint32_t i = 0; i < iters; ++i) CS_TRY(once());
        CS_HIP(hipEventRecord(e1, nullptr));
        CS_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        CS_HIP(hipEventElapsedTime(&ms, e0, e1));
        *ms_per_launch = (double)ms / iters;
        if (cs::g_gemm_wide_ablation == 7)  // stamped build: the clock the blocks of the LAST launch ran at
        {
            double mc = 0.0, ec = 0.0;
            const double ghz = cs::gemm_wide_read_clock_ghz(&mc, &ec);
            fprintf(stderr, "gemm_wide in-kernel clock: %.3f GHz (median over blocks, last of %u launches, %.1f us each); per tile: "
                            "k loop %.0f cycles, epilogue %.0f cycles\n", ghz, iters, (double)ms / iters * 1e3, mc, ec);
        }
        return CS_OK;
    };
    const int32_t st = run();
    cs::g_gemm_wide_ablation = 0;
    cs::g_gemm_wide_shape = 0;
    cs::g_gemm_wide_mfma = 0;
    (void)hipDeviceSynchronize();
    for (void* p : {(void*)dA, (void*)dW, (void*)dB, (void*)dR, (void*)dC, (void*)sA, (void*)sW, (void*)sC, (void*)dF})
        if (p) (void)hipFree(p);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return st;
}

}  // extern "C"
