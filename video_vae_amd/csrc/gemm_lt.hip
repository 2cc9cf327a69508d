// Linear layer whose residual add rides in the library GEMM:  y = x W + bias + res  (hipBLASLt: D = alpha A B + beta C with C != D,
// bias epilogue).  Reference train/layers.py:212-221 (x = x + Attention(...), x = x + MLP(...)): the Linear that closes a branch is
// followed by the residual add and the next block's LayerNorm.  Round 1 folded that add into the LayerNorm kernel (4 stream passes
// there: skip and branch in, sum and normalised rows out, at HBM rate); here the GEMM -- which is not HBM-bound -- reads the
// residual as its C operand and writes the sum, and the LayerNorm kernel goes back to 2 passes: +2.7 us per product, -7.9 us per
// LayerNorm (tools/beta_gemm_probe.py).  This is the one place the library is called from this side of the C ABI: a plain library
// GEMM with operands the framework's addmm cannot express without a copy (its out-of-place form copies C into D first).
#include "common.hpp"
#include <hipblaslt/hipblaslt.h>
#include <map>
#include <mutex>
#include <tuple>

namespace {

struct LtPlan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr, ld = nullptr;
    hipblasLtMatmulAlgo_t algo;
};
typedef std::tuple<int, int, int, int, int, int, int, int, int, int, size_t> LtKey;

hipblasLtHandle_t g_lt = nullptr;
std::mutex g_lt_mu;
std::map<LtKey, LtPlan> g_lt_plans;
int g_lt_algo = 0;              // which of the heuristic's ranked solutions new plans take (vvae_linear_residual_algo)

#define LT_OK(call) do { if ((call) != HIPBLAS_STATUS_SUCCESS) return VVAE_ERR_LIBRARY; } while (0)

int lt_plan(const LtKey& key, int M, int N, int K, int ldx, int ldw, int ldr, int ldy, int bias_kind, bool has_res, bool wt, size_t ws_bytes,
            LtPlan** out)
{
    auto it = g_lt_plans.find(key);
    if (it != g_lt_plans.end()) { *out = &it->second; return 0; }
    if (!g_lt) LT_OK(hipblasLtCreate(&g_lt));
    LtPlan p;
    LT_OK(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    const hipblasOperation_t opn = HIPBLAS_OP_N, opa = wt ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    LT_OK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa)));
    LT_OK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opn, sizeof(opn)));
    if (bias_kind) {
        const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
        LT_OK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
        const hipDataType bt = bias_kind == 2 ? HIP_R_32F : HIP_R_16BF;
        LT_OK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
    }
    // row-major y (M, N) = x (M, K) w (K, N)  <=>  column-major y^T (N, M) = w^T (N, K) x^T (K, M): A = w, B = x, no transposes.
    // wt: the weight is the (N, K) row-major shadow = a column-major (K, N) matrix, taken transposed (the library's "TN" kernels).
    if (wt) LT_OK(hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16BF, K, N, ldw));
    else LT_OK(hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16BF, N, K, ldw));
    LT_OK(hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_16BF, K, M, ldx));
    LT_OK(hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_16BF, N, M, has_res ? ldr : ldy));
    LT_OK(hipblasLtMatrixLayoutCreate(&p.ld, HIP_R_16BF, N, M, ldy));
    hipblasLtMatmulPreference_t pref = nullptr;
    LT_OK(hipblasLtMatmulPreferenceCreate(&pref));
    LT_OK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
    hipblasLtMatmulHeuristicResult_t res[16];
    int found = 0;
    const int want = g_lt_algo < 0 ? 1 : (g_lt_algo > 15 ? 16 : g_lt_algo + 1);
    const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(g_lt, p.desc, p.la, p.lb, p.lc, p.ld, pref, want, res, &found);
    hipblasLtMatmulPreferenceDestroy(pref);
    if (st != HIPBLAS_STATUS_SUCCESS || found < 1) return VVAE_ERR_LIBRARY;
    p.algo = res[found >= want ? want - 1 : found - 1].algo;          // the library's first choice unless the tuning hook asks further down
    *out = &g_lt_plans.emplace(key, p).first->second;
    return 0;
}

}  // namespace

// y (M, N) = x (M, K) w (K, N) + bias (N) + res (M, N); bf16 operands, fp32 accumulation, one rounding.  bias: NULL, bf16
// (bias_dtype = VVAE_DT_BF16) or fp32 (VVAE_DT_F32).  res: NULL = no residual.  Row pitches in elements; ws: >= ws_bytes of
// 16-byte-aligned device scratch for the library (0 / NULL allowed: the library then picks a solution that needs none).
// The (shape, pitches) -> solution choice is made once per process by the library's heuristic and cached.
static int linear_residual(const void* x, int ldx, const void* w, int ldw, bool wt, const void* bias, int bias_dtype, const void* res,
                           int ldr, void* y, int ldy, int M, int N, int K, void* ws, size_t ws_bytes, void* stream)
{
    if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldw < (wt ? K : N) || ldy < N || (res && ldr < N)) return VVAE_ERR_BAD_ARG;
    if (bias && bias_dtype != VVAE_DT_BF16 && bias_dtype != VVAE_DT_F32) return VVAE_ERR_BAD_ARG;
    if (((uintptr_t)x % 16) || ((uintptr_t)w % 16) || ((uintptr_t)y % 16) || ((uintptr_t)res % 16) || ((uintptr_t)ws % 16)) return VVAE_ERR_BAD_ARG;
    const int bias_kind = !bias ? 0 : (bias_dtype == VVAE_DT_F32 ? 2 : 1);
    std::lock_guard<std::mutex> lock(g_lt_mu);
    LtPlan* p = nullptr;
    const LtKey key{M, N, K, ldx, ldw, res ? ldr : 0, ldy, bias_kind, res ? 1 : 0, wt ? 1 : 0, ws_bytes};
    const int rc = lt_plan(key, M, N, K, ldx, ldw, ldr, ldy, bias_kind, res != nullptr, wt, ws_bytes, &p);
    if (rc) return rc;
    if (bias_kind) LT_OK(hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
    const float alpha = 1.f, beta = res ? 1.f : 0.f;
    LT_OK(hipblasLtMatmul(g_lt, p->desc, &alpha, w, p->la, x, p->lb, &beta, res ? res : y, p->lc, y, p->ld, &p->algo, ws, ws_bytes,
                          (hipStream_t)stream));
    return 0;
}

extern "C" int vvae_linear_residual_bf16(const void* x, int ldx, const void* w, int ldw, const void* bias, int bias_dtype, const void* res,
                                         int ldr, void* y, int ldy, int M, int N, int K, void* ws, size_t ws_bytes, void* stream)
{
    return linear_residual(x, ldx, w, ldw, false, bias, bias_dtype, res, ldr, y, ldy, M, N, K, ws, ws_bytes, stream);
}

// The same product with the weight handed over as its (N, K) row-major transpose (the optimizer's second bf16 shadow, pitch ldwt >= K):
// the library's K-contiguous-both-sides kernels are 8-13 % faster on the 768 <-> 1536 products of the trunk.
extern "C" int vvae_linear_residual_wt_bf16(const void* x, int ldx, const void* wt, int ldwt, const void* bias, int bias_dtype,
                                            const void* res, int ldr, void* y, int ldy, int M, int N, int K, void* ws, size_t ws_bytes,
                                            void* stream)
{
    return linear_residual(x, ldx, wt, ldwt, true, bias, bias_dtype, res, ldr, y, ldy, M, N, K, ws, ws_bytes, stream);
}

// Test / tuning hook: plans made from now on take the idx-th solution of the library's ranked list (0 = its first choice, the default); cached
// plans are dropped.
extern "C" int vvae_linear_residual_algo(int idx)
{
    if (idx < 0 || idx > 15) return VVAE_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(g_lt_mu);
    g_lt_algo = idx;
    g_lt_plans.clear();                                             // descriptors of dropped plans are a bounded leak of a test hook
    return 0;
}
