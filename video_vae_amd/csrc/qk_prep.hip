// q/k-norm + RoPE "prep" pass around a library attention core (the spatial half of FactoredAttention, sequence = h*w
// patches, reference train/layers.py:153-170,217-221).
//
// forward : qkv (tokens, 3*H*D) -> qk (tokens, 2*H*D) = [rope(q_norm(q)) | rope(k_norm(k))] per head; v is consumed in place
//           (a strided view of qkv), so the pass reads 2/3 of qkv once and writes the rotated q, k once.
// backward: (dq', dk', dv) in any (token stride, head stride) layout -> dqkv (tokens, 3*H*D) in ONE launch: q and k rows go
//           back through RoPE and the bias-free LayerNorm (xhat recomputed from the raw row), v rows are copied; the q/k-norm
//           scale gradients leave as per-workgroup partials (fixed-order sum by the caller: deterministic).
// It replaces ~25 framework launches per attention layer (chunk / LayerNorm x2 / rotate-half cat / mul / add / transposes
// and their autograd mirrors) with two.  Rows are split over lanes as in attn_rows.hpp.
#include "attn_rows.hpp"

namespace {

struct PrepDims { long tokens; int S, H; float eps; };

// grid (blocks, 2): y = 0 q rows, 1 k rows
template <typename T_, int D, int LPR>
__global__ __launch_bounds__(256) void qk_prep_fwd_kernel(const T_* __restrict__ qkv, int ld, T_* __restrict__ out, int ldo,
                                                          const float* __restrict__ q_scale, const float* __restrict__ k_scale,
                                                          const float* __restrict__ cosT, const float* __restrict__ sinT, PrepDims d)
{
    constexpr int DL = D / LPR;
    const int type = blockIdx.y;
    const int p = threadIdx.x % LPR;
    const float* scale = type ? k_scale : q_scale;
    const long nrows = d.tokens * d.H;
    const long step = (long)gridDim.x * (256 / LPR);
    for (long row = ((long)blockIdx.x * 256 + threadIdx.x) / LPR; row < nrows; row += step) {
        const long token = row / d.H;
        const int h = (int)(row - token * d.H);
        const int pos = (int)(token % d.S);
        const int col = (type * d.H + h) * D;
        float x[DL];
        load_row<T_, D, LPR>(qkv + token * ld + col, p, x);
        ln_rope_row<T_, D, LPR>(x, p, scale, d.eps, cosT + (long)pos * D, sinT + (long)pos * D);
        store_row<T_, D, LPR>(out + token * ldo + col, p, x);
    }
}

struct GradSrc { const void* p; long ts, hs; };     // element strides per token / per head

// grid (blocks, 3): y = 0 q rows, 1 k rows, 2 v rows (copy).  part: (gridDim.x, 2, D) fp32.
template <typename T_, int D, int LPR>
__global__ __launch_bounds__(256) void qk_prep_bwd_kernel(const T_* __restrict__ qkv, int ld, GradSrc gq, GradSrc gk, GradSrc gv,
                                                          T_* __restrict__ dqkv, int lddq, const float* __restrict__ q_scale,
                                                          const float* __restrict__ k_scale, const float* __restrict__ cosT,
                                                          const float* __restrict__ sinT, float* __restrict__ part, PrepDims d)
{
    using S = Slice<T_, D, LPR>;
    constexpr int DL = D / LPR;
    __shared__ float red[4][D];
    const int type = blockIdx.y;
    const int p = threadIdx.x % LPR, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nrows = d.tokens * d.H;
    const long step = (long)gridDim.x * (256 / LPR);
    const long row0 = ((long)blockIdx.x * 256 + threadIdx.x) / LPR;
    if (type == 2) {                                    // workgroup-uniform
        const T_* src = (const T_*)gv.p;
        for (long row = row0; row < nrows; row += step) {
            const long token = row / d.H;
            const int h = (int)(row - token * d.H);
            float x[DL];
            load_row<T_, D, LPR>(src + token * gv.ts + h * gv.hs, p, x);
            store_row<T_, D, LPR>(dqkv + token * lddq + (2 * d.H + h) * D, p, x);
        }
        return;
    }
    const GradSrc gs = type ? gk : gq;
    const T_* src = (const T_*)gs.p;
    const float* scale = type ? k_scale : q_scale;
    float acc[DL];
#pragma unroll
    for (int i = 0; i < DL; ++i) acc[i] = 0.f;
    for (long row = row0; row < nrows; row += step) {
        const long token = row / d.H;
        const int h = (int)(row - token * d.H);
        const int pos = (int)(token % d.S);
        const int col = (type * d.H + h) * D;
        float g[DL], x[DL];
        load_row<T_, D, LPR>(src + token * gs.ts + h * gs.hs, p, g);
        load_row<T_, D, LPR>(qkv + token * ld + col, p, x);
        const float rs = xhat_row<DL, LPR, D>(x, d.eps);
        rope_ln_bwd_row<T_, D, LPR>(g, x, p, rs, scale, cosT + (long)pos * D, sinT + (long)pos * D);
        store_row<T_, D, LPR>(dqkv + token * lddq + col, p, g);
#pragma unroll
        for (int i = 0; i < DL; ++i) acc[i] += x[i];
    }
#pragma unroll
    for (int i = 0; i < DL; ++i) {
        const float tot = frames_sum<LPR>(acc[i]);
        if (lane < LPR) red[wave][S::ch(i, p)] = tot;
    }
    __syncthreads();
    if (threadIdx.x < D) {
        const int t = threadIdx.x;
        part[((long)blockIdx.x * 2 + type) * D + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    }
}

constexpr int prep_lpr(int D) { return D >= 64 ? 4 : D >= 32 ? 2 : 1; }

int prep_blocks(long tokens, int H, int D)
{
    const int lpr = prep_lpr(D);
    const long need = (tokens * H * lpr + 255) / 256;
    return (int)(need < 1024 ? need : 1024);             // 4 workgroups per CU; rows are strided over them
}

bool prep_ok(int D, int dtype)
{
    return (D == 8 || D == 16 || D == 32 || D == 64) && (dtype == VVAE_DT_F32 || dtype == VVAE_DT_BF16);
}

template <typename T_, int D>
int launch_prep_fwd(const void* qkv, int ld, void* out, int ldo, const float* qs, const float* ks, const float* cosT, const float* sinT,
                    PrepDims d, hipStream_t s)
{
    hipLaunchKernelGGL((qk_prep_fwd_kernel<T_, D, prep_lpr(D)>), dim3(prep_blocks(d.tokens, d.H, D), 2), dim3(256), 0, s, (const T_*)qkv, ld,
                       (T_*)out, ldo, qs, ks, cosT, sinT, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}
template <typename T_, int D>
int launch_prep_bwd(const void* qkv, int ld, GradSrc gq, GradSrc gk, GradSrc gv, void* dqkv, int lddq, const float* qs, const float* ks,
                    const float* cosT, const float* sinT, float* part, PrepDims d, hipStream_t s)
{
    hipLaunchKernelGGL((qk_prep_bwd_kernel<T_, D, prep_lpr(D)>), dim3(prep_blocks(d.tokens, d.H, D), 3), dim3(256), 0, s, (const T_*)qkv, ld,
                       gq, gk, gv, (T_*)dqkv, lddq, qs, ks, cosT, sinT, part, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

}  // namespace

#define PREP_DISPATCH(FN, ...)                                                                                  \
    do {                                                                                                         \
        if (dtype == VVAE_DT_F32) {                                                                              \
            switch (D) { case 8: return FN<float, 8>(__VA_ARGS__); case 16: return FN<float, 16>(__VA_ARGS__);   \
                         case 32: return FN<float, 32>(__VA_ARGS__); default: return FN<float, 64>(__VA_ARGS__); } \
        } else {                                                                                                 \
            switch (D) { case 8: return FN<bf16_t, 8>(__VA_ARGS__); case 16: return FN<bf16_t, 16>(__VA_ARGS__); \
                         case 32: return FN<bf16_t, 32>(__VA_ARGS__); default: return FN<bf16_t, 64>(__VA_ARGS__); } \
        }                                                                                                        \
    } while (0)

// 1 if the prep kernels take this head_dim / dtype (row pitches and strides must be multiples of 16 bytes).
extern "C" int vvae_qk_prep_supported(int D, int dtype) { return prep_ok(D, dtype) ? 1 : 0; }

// Rows of the (rows, 2, D) fp32 scale-gradient partial buffer vvae_qk_prep_bwd writes.
extern "C" int vvae_qk_prep_blocks(long tokens, int heads, int D) { return prep_ok(D, VVAE_DT_F32) ? prep_blocks(tokens, heads, D) : 0; }

// qkv: (tokens, >= 3*heads*D) row pitch ld; out: (tokens, >= 2*heads*D) row pitch ldo, [q heads | k heads] rotated.
// RoPE position of a token = token % S.  q_scale/k_scale fp32 (D); cos/sin fp32 (>= S, D).
extern "C" int vvae_qk_prep_fwd(const void* qkv, int ld, void* out, int ldo, const float* q_scale, const float* k_scale,
                                const float* cos_table, const float* sin_table, long tokens, int S, int heads, int D, float eps,
                                int dtype, void* stream)
{
    const int v = dtype == VVAE_DT_F32 ? 4 : 8;
    if (!qkv || !out || !q_scale || !k_scale || !cos_table || !sin_table || tokens <= 0 || S <= 0 || heads <= 0 || !prep_ok(D, dtype) ||
        ld < 3 * heads * D || ldo < 2 * heads * D || ld % v || ldo % v || ((uintptr_t)qkv % 16) || ((uintptr_t)out % 16))
        return VVAE_ERR_BAD_ARG;
    PrepDims d{tokens, S, heads, eps};
    hipStream_t s = (hipStream_t)stream;
    PREP_DISPATCH(launch_prep_fwd, qkv, ld, out, ldo, q_scale, k_scale, cos_table, sin_table, d, s);
}

// dq/dk/dv: gradients w.r.t. the rotated q, k and v, element (token, head, c) at p + token*ts + head*hs + c (strides in
// elements).  dqkv: (tokens, 3*heads*D) row pitch lddq, fully written.  part: (vvae_qk_prep_blocks(...), 2, D) fp32 written.
extern "C" int vvae_qk_prep_bwd(const void* qkv, int ld, const void* dq, long dq_ts, long dq_hs, const void* dk, long dk_ts, long dk_hs,
                                const void* dv, long dv_ts, long dv_hs, void* dqkv, int lddq, const float* q_scale, const float* k_scale,
                                const float* cos_table, const float* sin_table, float* part, long tokens, int S, int heads, int D,
                                float eps, int dtype, void* stream)
{
    const int v = dtype == VVAE_DT_F32 ? 4 : 8;
    if (!qkv || !dq || !dk || !dv || !dqkv || !part || !q_scale || !k_scale || !cos_table || !sin_table || tokens <= 0 || S <= 0 ||
        heads <= 0 || !prep_ok(D, dtype) || ld < 3 * heads * D || lddq < 3 * heads * D || ld % v || lddq % v || dq_ts % v || dq_hs % v ||
        dk_ts % v || dk_hs % v || dv_ts % v || dv_hs % v || ((uintptr_t)qkv % 16) || ((uintptr_t)dq % 16) || ((uintptr_t)dk % 16) ||
        ((uintptr_t)dv % 16) || ((uintptr_t)dqkv % 16))
        return VVAE_ERR_BAD_ARG;
    PrepDims d{tokens, S, heads, eps};
    GradSrc gq{dq, dq_ts, dq_hs}, gk{dk, dk_ts, dk_hs}, gv{dv, dv_ts, dv_hs};
    hipStream_t s = (hipStream_t)stream;
    PREP_DISPATCH(launch_prep_bwd, qkv, ld, gq, gk, gv, dqkv, lddq, q_scale, k_scale, cos_table, sin_table, part, d, s);
}
