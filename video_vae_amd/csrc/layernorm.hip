// LayerNorm(eps=1e-6, fast variance, fp32 statistics) over the last axis, forward and backward.
//
// Replaces nnx.LayerNorm at /root/reference/train/layers.py:17,152,178 (PatchEmbedding.norm, Attention.input_norm,
// MLP.norm) and :155-156 (q_norm / k_norm, use_bias=False) -- 127 LayerNorms per training step of the production model.
// Pure HBM streams: a row lives in registers (16-byte vector loads, LPR lanes per row), statistics by xor-shuffles inside
// the LPR-lane group, one read of x (+dy) and one write per pass.  Backward keeps per-lane column sums of dy*xhat and dy
// across the rows a wave walks and emits ONE partial row per workgroup (the caller sums them): no atomics, deterministic.
//
// Rows may be strided in two levels (row r at base + (r / inner) * outer_pitch + (r % inner) * inner_pitch) so a per-head
// slice of a fused QKV buffer is normalised in place without a gather copy.
#include "common.hpp"

namespace {

struct LnDims { long rows; int C; int inner; long outer_pitch; long inner_pitch; float eps; };

__device__ __forceinline__ long row_off(const LnDims& d, long r) { return (r / d.inner) * d.outer_pitch + (r % d.inner) * d.inner_pitch; }

template <int LPR> __device__ __forceinline__ float group_sum(float v) { return butterfly_sum<LPR / 2, 1>(v); }   // VALU-only (common.hpp)

// raw 16-byte vector -> fp32 lanes (the backward kernel keeps x and dy RAW between its two passes: 8 instead of 16 live
// registers per bf16 vector, which is what lets four waves share a SIMD)
__device__ __forceinline__ void unpack(const uint4& r, float (&v)[4]) {
    v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y); v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
}
__device__ __forceinline__ void unpack(const uint4& r, float (&v)[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ void opaque(uint4& r) { asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w)); }

// y rows are written contiguously (pitch C).  mean / rstd: fp32 [rows] (saved for backward).
// addend != NULL: the row normalised is round(x + addend) (addend, xsum contiguous (rows, C)) and that sum is written to xsum: the
// residual add that closes one pre-norm block, fused with the LayerNorm that opens the next (one pass over the stream, not two).
template <typename T_, int LPR, int VPL, bool LATE_STAGE>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T_* __restrict__ x, T_* __restrict__ y, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, const T_* __restrict__ addend, T_* __restrict__ xsum,
                                                            LnDims d)
{
    constexpr int V = VecWidth<T_>::value, RPW = 64 / LPR;          // rows per wave-iteration
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, ll = lane % LPR;
    // gamma / beta reach the lanes through LDS (one coalesced pass per workgroup): per-lane strided dword loads of the affine
    // cost ~30x the L1 line accesses of the row itself and were what bounded the first version of this kernel.
    __shared__ __attribute__((aligned(16))) float sg[LPR * VPL * V], sb[LPR * VPL * V];
    // The affine is requested first, the first rows right behind it, and only then is it parked in LDS and the workgroup
    // synchronised (LDS-only wait + raw barrier): the row loads are already in flight while the staging completes.
    constexpr int NG = (LPR * VPL * V + 255) / 256;
    float gpre[NG], bpre[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int i = threadIdx.x + j * 256;
        gpre[j] = (i < LPR * VPL * V && i < d.C) ? gamma[i] : 0.f;
        bpre[j] = (beta && i < LPR * VPL * V && i < d.C) ? beta[i] : 0.f;
    }
    bool staged = false;
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int i = threadIdx.x + j * 256;
            if (i < LPR * VPL * V) { sg[i] = gpre[j]; sb[i] = bpre[j]; }
        }
        if (LATE_STAGE) {
            __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0) only: the row loads stay in flight across the barrier
            __builtin_amdgcn_s_barrier();
        } else {
            __syncthreads();
        }
        staged = true;
    };
    if (!LATE_STAGE) stage();                           // affine parked and the workgroup synchronised BEFORE any row is requested
    // gamma / beta are re-read from LDS at the point of use: ~50 VGPRs instead of ~118, i.e. 8 waves per SIMD, twice the bytes
    // in flight per CU
    const long rstride = (long)gridDim.x * 4 * RPW;
    for (long r0 = ((long)blockIdx.x * 4 + wave) * RPW; r0 < d.rows; r0 += rstride) {
        const long r = r0 + sub;
        const bool rv = r < d.rows;
        const long rs = rv ? r : 0;
        const T_* xr = x + row_off(d, rs);
        const bool has_add = addend != nullptr;
        // all of the row's loads are requested before any is consumed, unconditionally (out-of-range lanes read row / column 0
        // and contribute zeros): predicated loads cost one branch + one full memory wait per vector
        float v[VPL][V], a[VPL][V];
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int c = (k * LPR + ll) * V;
            VecIO<T_, V>::load(xr + (c < d.C ? c : 0), v[k]);
        }
        if (has_add) {
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                const int c = (k * LPR + ll) * V;
                VecIO<T_, V>::load(addend + rs * d.C + (c < d.C ? c : 0), a[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int c = (k * LPR + ll) * V;
            const bool ok = rv && c < d.C;
            if (has_add) {
#pragma unroll
                for (int e = 0; e < V; ++e) v[k][e] = round_to<T_>(v[k][e] + a[k][e]);
                if (ok) VecIO<T_, V>::store(xsum + r * d.C + c, v[k]);
            }
#pragma unroll
            for (int e = 0; e < V; ++e) {
                if (!ok) v[k][e] = 0.f;
                s += v[k][e]; ss += v[k][e] * v[k][e];
            }
        }
        if (!staged) stage();
        s = group_sum<LPR>(s); ss = group_sum<LPR>(ss);
        const float mean = s / d.C;
        float var = ss / d.C - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float rstd = rsqrtf(var + d.eps);
        if (rv && ll == 0) { mean_out[r] = mean; rstd_out[r] = rstd; }
        T_* yr = y + r * d.C;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int c = (k * LPR + ll) * V;
            if (rv && c < d.C) {
                float o[V];
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = (v[k][e] - mean) * (rstd * sg[c + e]) + sb[c + e];
                VecIO<T_, V>::store(yr + c, o);
            }
        }
    }
    if (!staged) stage();                                // a wave without rows still owes the workgroup its barrier
}

// dy rows contiguous (pitch C); dx rows contiguous.  part: fp32 [gridDim.x][2][C] = per-workgroup (sum dy*xhat | sum dy).
constexpr int LN_BW = 8;                      // waves per workgroup of the backward kernel (one partial row per workgroup)

template <typename T_, int LPR, int VPL>
__global__ __launch_bounds__(64 * LN_BW) void layernorm_bwd_kernel(const T_* __restrict__ x, const T_* __restrict__ dy, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                            const T_* __restrict__ dres, T_* __restrict__ dx, float* __restrict__ part,
                                                            LnDims d)
{
    constexpr int V = VecWidth<T_>::value, RPW = 64 / LPR;
    __shared__ __attribute__((aligned(16))) float red[LN_BW][LPR][VPL * V];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, ll = lane % LPR;
    // gamma stays in LDS (see the forward kernel for why it gets there in one coalesced pass) and is re-read per use:
    // registers go to the two column accumulators instead
    __shared__ __attribute__((aligned(16))) float sg[LPR * VPL * V];
    for (int i = threadIdx.x; i < LPR * VPL * V; i += 64 * LN_BW) sg[i] = i < d.C ? gamma[i] : 0.f;
    __syncthreads();
    float ag[VPL][V], ab[VPL][V];
#pragma unroll
    for (int k = 0; k < VPL; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) { ag[k][e] = 0.f; ab[k][e] = 0.f; }
    const long rstride = (long)gridDim.x * LN_BW * RPW;
    for (long r0 = ((long)blockIdx.x * LN_BW + wave) * RPW; r0 < d.rows; r0 += rstride) {
        const long r = r0 + sub;
        const bool rv = r < d.rows;
        const long rs = rv ? r : 0;
        const T_* xr = x + row_off(d, rs);
        const T_* gr = dy + rs * d.C;
        const float mean = mean_in[rs], rstd = rstd_in[rs];
        // x and dy are requested unconditionally (out-of-range lanes read row / column 0 and are zeroed afterwards): predicated
        // loads compiled to one branch + one full memory wait per vector, i.e. several serialised latencies per iteration
        uint4 rx[VPL], rg[VPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int c = (k * LPR + ll) * V;
            const int cs = c < d.C ? c : 0;
            rx[k] = *reinterpret_cast<const uint4*>(xr + cs);
            rg[k] = *reinterpret_cast<const uint4*>(gr + cs);
        }
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int c = (k * LPR + ll) * V;
            if (!(rv && c < d.C)) { rx[k] = make_uint4(0, 0, 0, 0); rg[k] = make_uint4(0, 0, 0, 0); }
        }
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int c = (k * LPR + ll) * V;
            if (rv && c < d.C) {
                float xh[V], g[V];
                unpack(rx[k], xh); unpack(rg[k], g);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    xh[e] = (xh[e] - mean) * rstd;
                    ag[k][e] += g[e] * xh[e];
                    ab[k][e] += g[e];
                    const float gg = g[e] * sg[c + e];
                    s1 += gg; s2 += gg * xh[e];
                }
            }
            opaque(rx[k]); opaque(rg[k]);               // second pass re-decodes instead of keeping 2*V floats alive
        }
        s1 = group_sum<LPR>(s1) / d.C; s2 = group_sum<LPR>(s2) / d.C;
        T_* dr = dx + r * d.C;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
            const int c = (k * LPR + ll) * V;
            if (rv && c < d.C) {
                float xh[V], g[V], o[V];
                unpack(rx[k], xh); unpack(rg[k], g);
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = rstd * (g[e] * sg[c + e] - s1 - (xh[e] - mean) * rstd * s2);
                if (dres) {                              // pre-norm residual block: the skip path's gradient joins here
                    float rr[V];
                    VecIO<T_, V>::load(dres + r * d.C + c, rr);
#pragma unroll
                    for (int e = 0; e < V; ++e) o[e] = round_to<T_>(o[e]) + rr[e];
                }
                VecIO<T_, V>::store(dr + c, o);
            }
        }
    }
    // column partials: fold the RPW row-slots of a wave (lanes with equal ll), then the 4 waves, fixed order
    float* pg = part + (long)blockIdx.x * 2 * d.C;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int k = 0; k < VPL; ++k)
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float t = pass ? ab[k][e] : ag[k][e];
                if constexpr (LPR < 64) t = butterfly_sum<32, LPR>(t);       // the row slots of a wave (lanes with equal ll)
                if (lane < LPR) red[wave][lane][k * V + e] = t;
            }
        __syncthreads();
        for (int i = threadIdx.x; i < LPR * VPL * V; i += 64 * LN_BW) {
            const int l2 = i / (VPL * V), kv = i % (VPL * V);
            const int c = ((kv / V) * LPR + l2) * V + kv % V;
            if (c < d.C) {
                float t = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < LN_BW; w2 += 2) t += red[w2][l2][kv] + red[w2 + 1][l2][kv];      // fixed order
                pg[pass * d.C + c] = t;
            }
        }
        __syncthreads();
    }
}

int g_ln_bwd_cap = 384;      // workgroups (= partial dgamma / dbeta rows) of the backward kernel.  Every row is 6 KB written here and read again by the grouped fold:
                             // 84 launches x 512 rows were 264 MB each way per step.  Whole step (tools/ab_hook.py, 5 alternating rounds): 256 -> 34.05 ms, 384 -> 33.96,
                             // 512 -> 34.03, 1024 -> 34.31
int g_ln_fwd_cap = 1024;     // workgroups of the bf16 forward kernel (tuning hook vvae_layernorm_fwd_config)
int g_ln_fwd_late = 0;        // forward kernel: 1 = stage the affine behind the first rows' loads (LDS-only wait + raw barrier)

inline int ln_blocks(long rows, int lpr, int cap = 0)
{
    const int waves = cap == 0 ? LN_BW : 4;
    if (cap == 0) cap = g_ln_bwd_cap;
    // backward: <= 1024 workgroups (each emits one partial row of dgamma/dbeta).  forward has no epilogue, so it takes one
    // row-slot per wave (cap 16384): every row's loads are in flight at once instead of 4 rows queued behind each other.
    const long per = (long)waves * (64 / lpr);
    long b = (rows + per - 1) / per;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

// pick lanes-per-row: the SMALLEST power of two (>= 8) that covers the row with <= 4 vectors per lane -- a wave then walks
// 64/lpr rows at once with 3-4 sixteen-byte loads in flight per lane (C = 768 bf16: 32 lanes x 3 vectors, two rows per wave)
inline bool ln_pick(int C, int V, int& lpr, int& vpl)
{
    if (C % V) return false;
    const int nv = C / V;
    lpr = 8;
    while (lpr < 64 && (nv + lpr - 1) / lpr > 4) lpr <<= 1;
    vpl = (nv + lpr - 1) / lpr;
    return vpl <= 4;
}

template <typename T_>
bool ln_ok(const LnDims& d, const void* x, int& lpr, int& vpl)
{
    constexpr int V = VecWidth<T_>::value;
    if (d.rows <= 0 || d.C <= 0 || d.inner <= 0) return false;
    if (!ln_pick(d.C, V, lpr, vpl)) return false;
    return ((uintptr_t)x % 16) == 0 && d.outer_pitch % V == 0 && d.inner_pitch % V == 0;
}

}  // namespace

#define LN_CASE(KERNEL, T_, L, P, ...) case L * 8 + P: hipLaunchKernelGGL((KERNEL<T_, L, P LN_EXTRA>), grid, dim3(ln_threads), 0, s, __VA_ARGS__); break;
#define LN_EXTRA
#define LN_SWITCH(KERNEL, T_, ...)                                                                                                   \
    do {                                                                                                                             \
        switch (lpr * 8 + vpl) {                                                                                                     \
            LN_CASE(KERNEL, T_, 8, 1, __VA_ARGS__) LN_CASE(KERNEL, T_, 8, 2, __VA_ARGS__) LN_CASE(KERNEL, T_, 8, 3, __VA_ARGS__)     \
            LN_CASE(KERNEL, T_, 8, 4, __VA_ARGS__) LN_CASE(KERNEL, T_, 16, 3, __VA_ARGS__) LN_CASE(KERNEL, T_, 16, 4, __VA_ARGS__)   \
            LN_CASE(KERNEL, T_, 32, 3, __VA_ARGS__) LN_CASE(KERNEL, T_, 32, 4, __VA_ARGS__) LN_CASE(KERNEL, T_, 64, 3, __VA_ARGS__)  \
            LN_CASE(KERNEL, T_, 64, 4, __VA_ARGS__)                                                                                  \
            default: return VVAE_ERR_BAD_ARG;                                                                                        \
        }                                                                                                                            \
    } while (0)

#define LN_FCASE(LATE, T_, L, P, ...) case L * 8 + P: hipLaunchKernelGGL((layernorm_fwd_kernel<T_, L, P, LATE>), grid, dim3(ln_threads), 0, s, __VA_ARGS__); break;
#define LN_FWD_SWITCH(LATE, T_, ...)                                                                                                  \
    do {                                                                                                                             \
        switch (lpr * 8 + vpl) {                                                                                                     \
            LN_FCASE(LATE, T_, 8, 1, __VA_ARGS__) LN_FCASE(LATE, T_, 8, 2, __VA_ARGS__) LN_FCASE(LATE, T_, 8, 3, __VA_ARGS__)     \
            LN_FCASE(LATE, T_, 8, 4, __VA_ARGS__) LN_FCASE(LATE, T_, 16, 3, __VA_ARGS__) LN_FCASE(LATE, T_, 16, 4, __VA_ARGS__)   \
            LN_FCASE(LATE, T_, 32, 3, __VA_ARGS__) LN_FCASE(LATE, T_, 32, 4, __VA_ARGS__) LN_FCASE(LATE, T_, 64, 3, __VA_ARGS__)  \
            LN_FCASE(LATE, T_, 64, 4, __VA_ARGS__)                                                                                  \
            default: return VVAE_ERR_BAD_ARG;                                                                                        \
        }                                                                                                                            \
    } while (0)

// 1 if vvae_layernorm_* take this shape (C a multiple of the 16-byte vector, C <= 2048 bf16 / 1024 fp32).
extern "C" int vvae_layernorm_supported(int C, int dtype)
{
    int lpr, vpl;
    return ln_pick(C, dtype == VVAE_DT_F32 ? 4 : 8, lpr, vpl) ? 1 : 0;
}

// Workgroups the backward kernel uses = rows of its partial buffer (each 2*C floats).
extern "C" int vvae_layernorm_bwd_blocks(long rows, int C, int dtype)
{
    int lpr, vpl;
    if (!ln_pick(C, dtype == VVAE_DT_F32 ? 4 : 8, lpr, vpl)) return 0;
    return ln_blocks(rows, lpr);
}

// x: rows of C elements, row r at x + (r / inner) * outer_pitch + (r % inner) * inner_pitch (elements); y contiguous (rows, C).
// gamma fp32 [C]; beta fp32 [C] or NULL; mean, rstd fp32 [rows] written.
// addend, xsum: both NULL, or contiguous (rows, C): normalise round(x + addend) and write that sum to xsum (fused residual add).
extern "C" int vvae_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta, float* mean, float* rstd,
                                  const void* addend, void* xsum, long rows, int C, int inner, long outer_pitch, long inner_pitch, float eps,
                                  int dtype, void* stream)
{
    if (!x || !y || !gamma || !mean || !rstd || (addend == nullptr) != (xsum == nullptr) || ((uintptr_t)addend % 16) || ((uintptr_t)xsum % 16))
        return VVAE_ERR_BAD_ARG;
    LnDims d{rows, C, inner, outer_pitch, inner_pitch, eps};
    hipStream_t s = (hipStream_t)stream;
    int lpr, vpl;
    if (dtype == VVAE_DT_F32) {
        if (!ln_ok<float>(d, x, lpr, vpl) || ((uintptr_t)y % 16)) return VVAE_ERR_BAD_ARG;
        dim3 grid(ln_blocks(rows, lpr, 2048));
        const int ln_threads = 256;
        if (g_ln_fwd_late) { LN_FWD_SWITCH(true, float, (const float*)x, (float*)y, gamma, beta, mean, rstd, (const float*)addend, (float*)xsum, d); }
        else { LN_FWD_SWITCH(false, float, (const float*)x, (float*)y, gamma, beta, mean, rstd, (const float*)addend, (float*)xsum, d); }
    } else if (dtype == VVAE_DT_BF16) {
        if (!ln_ok<bf16_t>(d, x, lpr, vpl) || ((uintptr_t)y % 16)) return VVAE_ERR_BAD_ARG;
        dim3 grid(ln_blocks(rows, lpr, g_ln_fwd_cap));  // 1024 workgroups = 4096 waves: all resident at 6 waves/SIMD (2048 left a third-full second round)
        const int ln_threads = 256;
        if (g_ln_fwd_late) { LN_FWD_SWITCH(true, bf16_t, (const bf16_t*)x, (bf16_t*)y, gamma, beta, mean, rstd, (const bf16_t*)addend, (bf16_t*)xsum, d); }
        else { LN_FWD_SWITCH(false, bf16_t, (const bf16_t*)x, (bf16_t*)y, gamma, beta, mean, rstd, (const bf16_t*)addend, (bf16_t*)xsum, d); }
    } else return VVAE_ERR_BAD_ARG;
    VVAE_LAUNCH_CHECK();
    return 0;
}

// dy, dx contiguous (rows, C).  part: fp32 (vvae_layernorm_bwd_blocks(...), 2, C): [sum dy*xhat | sum dy] per workgroup.
// dres: NULL, or a contiguous (rows, C) gradient added to dx (the skip path of a pre-norm residual block: x + f(LN(x))).
extern "C" int vvae_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, const void* dres,
                                  void* dx, float* part, long rows, int C, int inner, long outer_pitch, long inner_pitch, int dtype,
                                  void* stream)
{
    if (!x || !dy || !gamma || !mean || !rstd || !dx || !part || ((uintptr_t)dres % 16)) return VVAE_ERR_BAD_ARG;
    LnDims d{rows, C, inner, outer_pitch, inner_pitch, 0.f};
    hipStream_t s = (hipStream_t)stream;
    int lpr, vpl;
    if (dtype == VVAE_DT_F32) {
        if (!ln_ok<float>(d, x, lpr, vpl) || ((uintptr_t)dy % 16) || ((uintptr_t)dx % 16)) return VVAE_ERR_BAD_ARG;
        dim3 grid(ln_blocks(rows, lpr));
        const int ln_threads = 64 * LN_BW;
        LN_SWITCH(layernorm_bwd_kernel, float, (const float*)x, (const float*)dy, gamma, mean, rstd, (const float*)dres, (float*)dx, part, d);
    } else if (dtype == VVAE_DT_BF16) {
        if (!ln_ok<bf16_t>(d, x, lpr, vpl) || ((uintptr_t)dy % 16) || ((uintptr_t)dx % 16)) return VVAE_ERR_BAD_ARG;
        dim3 grid(ln_blocks(rows, lpr));
        const int ln_threads = 64 * LN_BW;
        LN_SWITCH(layernorm_bwd_kernel, bf16_t, (const bf16_t*)x, (const bf16_t*)dy, gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, part, d);
    } else return VVAE_ERR_BAD_ARG;
    VVAE_LAUNCH_CHECK();
    return 0;
}

// Tuning hook: workgroups (= partial rows) of the backward kernel.
extern "C" int vvae_layernorm_config(int bwd_cap)
{
    g_ln_bwd_cap = bwd_cap > 0 ? bwd_cap : 384;
    return 0;
}

// Test hook: forward-kernel variant.  0 (default): gamma / beta are parked in LDS and the workgroup synchronised (__syncthreads)
// before any row is requested.  1: the round-1 form -- parked behind the first rows' loads, LDS-only wait + raw s_barrier.
extern "C" int vvae_layernorm_fwd_mode(int late_stage)
{
    g_ln_fwd_late = late_stage ? 1 : 0;
    return 0;
}

// Tuning hook: workgroups of the bf16 forward kernel (default 1024).
extern "C" int vvae_layernorm_fwd_config(int fwd_cap)
{
    g_ln_fwd_cap = fwd_cap > 0 ? fwd_cap : 1024;
    return 0;
}
