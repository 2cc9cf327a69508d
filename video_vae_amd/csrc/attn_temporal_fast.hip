// Temporal-attention core, "LPR lanes = one frame" form: the production path for head_dim D in {8,16,32,64}.
//
// Same math and reference lines as attn_temporal.hip (train/layers.py:159-170), restructured for CDNA4:
//   * a wavefront carries 64/(T*LPR) whole (sequence, head) items; the LPR adjacent lanes of frame t keep the q / k / v ROW
//     of that frame in registers, DL = D/LPR channels each (LPR = D/16 where the wave has room: 16 channels per lane keeps
//     the backward pass near 100 VGPRs, 4-5 waves per SIMD, instead of 256 VGPRs and one wave at D = 64).  Lane p owns
//     channels [p*HL, (p+1)*HL) and [D/2 + p*HL, D/2 + (p+1)*HL), HL = DL/2, so the RoPE rotate-half partner of every
//     channel sits in the SAME lane; LayerNorm sums and q.k dots finish with one or two quad-DPP adds;
//   * keys and values go to LDS once (storage dtype); every lane then walks the T keys with broadcast ds_read_b128 and
//     an online softmax -- q, k, v are read from HBM once and o written once;
//   * forward also emits the row log-sum-exp; backward uses it plus delta = dO.O (no second softmax pass), computes
//     dS/P row-wise (lanes = query), parks them in LDS and accumulates dK/dV column-wise (lanes = key);
//   * q/k-norm scale gradients are written as per-workgroup partials (summed by the caller): deterministic.
#include "attn_rows.hpp"

// matrix-core form for bf16, head_dim 64, T = 16 (attn_temporal_mfma.hip)
int tmfma_supported(int T, int D, int ld, int ldo, int dtype);
int tmfma_bwd_rows(long items);
int tmfma_fwd(const void* qkv, int ld, void* out, int ldo, float* lse, const float* qs, const float* ks, const float* cosT, const float* sinT,
              const uint8_t* mask, int mask_div, int inner, int A, int heads, float eps, hipStream_t s);
int tmfma_bwd(const void* qkv, int ld, const void* out, int ldo, const void* dout, int lddo, const float* lse, void* dqkv, int lddq,
              const float* qs, const float* ks, const float* cosT, const float* sinT, const uint8_t* mask, int mask_div, int inner,
              float* part, int A, int heads, float eps, hipStream_t s);

// matrix-core form for bf16, head_dim 64, T = 32 / 64 (attn_temporal_mfma32.hip); one partial row per (sequence, head)
int tm32_supported(int T, int D, int ld, int ldo, int dtype);
int tm32_fwd(const void* qkv, int ld, void* out, int ldo, float* lse, const float* qs, const float* ks, const float* cosT, const float* sinT,
             const uint8_t* mask, int mask_div, int inner, int A, int T, int heads, float eps, hipStream_t s);
int tm32_bwd(const void* qkv, int ld, const void* out, int ldo, const void* dout, int lddo, const float* lse, void* dqkv, int lddq,
             const float* qs, const float* ks, const float* cosT, const float* sinT, const uint8_t* mask, int mask_div, int inner,
             float* part, int A, int T, int heads, float eps, hipStream_t s);

namespace {

struct FAttnDims { int A, T, heads, mask_div; float eps; long items; int inner; };

// Token (row of the (tokens, channels) matrix) of frame `row` of sequence `a`.  inner = 1: sequences are contiguous (A, T, C);
// inner = hw: the tensor is (b, t, hw, C) and sequence a = b*hw + i walks frames with stride hw -- the FactoredAttention
// layout, so the temporal half needs no "b t hw c -> (b hw) t c" transpose copies (reference train/layers.py:211,215).
__device__ __forceinline__ long token_of(const FAttnDims& d, int a, int row) {
    return (long)(a / d.inner) * d.T * d.inner + (long)row * d.inner + (a % d.inner);
}

// partial dot(reg slice, LDS row slice) and axpy(reg slice += a * LDS row slice); LDS rows hold all D channels in the
// storage dtype, read with 16-byte broadcast loads.
template <typename T_, int D, int LPR>
__device__ __forceinline__ float dot_lds(const float (&r)[D / LPR], const T_* row, int p) {
    using S = Slice<T_, D, LPR>;
    float s = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const T_* src = row + (half ? S::hi(p) : S::lo(p));
#pragma unroll
        for (int c = 0; c < S::HL / S::V; ++c) {
            float t[S::V];
            VecIO<T_, S::V>::load(src + c * S::V, t);
#pragma unroll
            for (int e = 0; e < S::V; ++e) s += r[half * S::HL + c * S::V + e] * t[e];
        }
    }
    return s;
}
template <typename T_, int D, int LPR>
__device__ __forceinline__ void axpy_lds(float (&r)[D / LPR], float a, const T_* row, int p) {
    using S = Slice<T_, D, LPR>;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const T_* src = row + (half ? S::hi(p) : S::lo(p));
#pragma unroll
        for (int c = 0; c < S::HL / S::V; ++c) {
            float t[S::V];
            VecIO<T_, S::V>::load(src + c * S::V, t);
#pragma unroll
            for (int e = 0; e < S::V; ++e) r[half * S::HL + c * S::V + e] += a * t[e];
        }
    }
}

constexpr int kItemPad = 16;     // bytes between items in an LDS array: two items in one ds_read_b128 lane group hit different slots

template <typename T_, int D>
__host__ __device__ inline int item_stride_bytes(int T) { return T * D * (int)sizeof(T_) + kItemPad; }

template <typename T_, int D, int LPR>
__global__ __launch_bounds__(64) void tattn_fwd_fast(const T_* __restrict__ qkv, int ld, T_* __restrict__ out, int ldo, float* __restrict__ lse,
                                                     const float* __restrict__ q_scale, const float* __restrict__ k_scale,
                                                     const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                     const uint8_t* __restrict__ mask, FAttnDims d)
{
    constexpr int DL = D / LPR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = d.T, lane = threadIdx.x;
    const int p = lane % LPR, fr = lane / LPR;
    const int ipw = 64 / (T * LPR);
    const int item = fr / T, row = fr - item * T;
    const long gi = (long)blockIdx.x * ipw + item;
    const bool valid = item < ipw && gi < d.items;
    const long gic = valid ? gi : 0;
    const int a = (int)(gic / d.heads), h = (int)(gic % d.heads);
    const int HD = d.heads * D;
    const long tok = token_of(d, a, row);
    const int istride = item_stride_bytes<T_, D>(T);
    T_* Ks = reinterpret_cast<T_*>(smem + (valid ? item : 0) * istride);
    T_* Vs = reinterpret_cast<T_*>(smem + ipw * istride + (valid ? item : 0) * istride);

    float q[DL], kv[DL];
    const T_* g = qkv + tok * ld + h * D;
    if (valid) {
        load_row<T_, D, LPR>(g + HD, p, kv);
        ln_rope_row<T_, D, LPR>(kv, p, k_scale, d.eps, cosT + row * D, sinT + row * D);
        store_row<T_, D, LPR>(Ks + row * D, p, kv);
        load_row<T_, D, LPR>(g + 2 * HD, p, kv);
        store_row<T_, D, LPR>(Vs + row * D, p, kv);
        load_row<T_, D, LPR>(g, p, q);
        ln_rope_row<T_, D, LPR>(q, p, q_scale, d.eps, cosT + row * D, sinT + row * D);
    } else {
#pragma unroll
        for (int i = 0; i < DL; ++i) q[i] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();

    const uint8_t* mrow = mask ? mask + (long)(a / d.mask_div) * T : nullptr;
    const float scale = rsqrtf((float)D);
    float o[DL];
#pragma unroll
    for (int i = 0; i < DL; ++i) o[i] = 0.f;
    float m = -3.0e38f, l = 0.f;
    for (int j = 0; j < T; ++j) {
        if (mrow && !mrow[j]) continue;           // per item, hence uniform over the lanes of a frame
        const float s = lpr_sum<LPR>(dot_lds<T_, D, LPR>(q, Ks + j * D, p)) * scale;
        const float mn = fmaxf(m, s);
        const float alpha = __expf(m - mn), pr = __expf(s - mn);
        l = l * alpha + pr;
#pragma unroll
        for (int i = 0; i < DL; ++i) o[i] *= alpha;
        axpy_lds<T_, D, LPR>(o, pr, Vs + j * D, p);
        m = mn;
    }
    if (valid) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
        for (int i = 0; i < DL; ++i) o[i] *= inv;
        store_row<T_, D, LPR>(out + tok * ldo + h * D, p, o);
        if (p == 0) lse[gi * T + row] = l > 0.f ? m + __logf(l) : 0.f;
    }
}

template <typename T_, int D, int LPR>
__global__ __launch_bounds__(64) void tattn_bwd_fast(const T_* __restrict__ qkv, int ld, const T_* __restrict__ out, int ldo,
                                                     const T_* __restrict__ dout, int lddo, const float* __restrict__ lse,
                                                     T_* __restrict__ dqkv, int lddq, const float* __restrict__ q_scale,
                                                     const float* __restrict__ k_scale, const float* __restrict__ cosT,
                                                     const float* __restrict__ sinT, const uint8_t* __restrict__ mask,
                                                     float* __restrict__ dscale_part, FAttnDims d)
{
    using S = Slice<T_, D, LPR>;
    constexpr int DL = D / LPR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = d.T, lane = threadIdx.x;
    const int p = lane % LPR, fr = lane / LPR;
    const int ipw = 64 / (T * LPR);
    const int item = fr / T, row = fr - item * T;
    const long gi = (long)blockIdx.x * ipw + item;
    const bool valid = item < ipw && gi < d.items;
    const long gic = valid ? gi : 0;
    const int a = (int)(gic / d.heads), h = (int)(gic % d.heads);
    const int HD = d.heads * D;
    const long tok = token_of(d, a, row);
    const int istride = item_stride_bytes<T_, D>(T);
    const int it = valid ? item : 0;
    T_* Ks = reinterpret_cast<T_*>(smem + it * istride);
    T_* Vs = reinterpret_cast<T_*>(smem + (ipw + it) * istride);
    T_* Qs = reinterpret_cast<T_*>(smem + (2 * ipw + it) * istride);
    T_* Gs = reinterpret_cast<T_*>(smem + (3 * ipw + it) * istride);          // dO rows
    float* Ps = reinterpret_cast<float*>(smem + 4 * ipw * istride) + it * T * (T + 1);
    float* Ss = reinterpret_cast<float*>(smem + 4 * ipw * istride) + (ipw + it) * T * (T + 1);

    const T_* g = qkv + tok * ld + h * D;
    const float* cosr = cosT + row * D;
    const float* sinr = sinT + row * D;
    float q[DL], go[DL], t0[DL];
    float rs;
    float delta = 0.f;
    if (valid) {
        load_row<T_, D, LPR>(g + HD, p, t0);
        ln_rope_row<T_, D, LPR>(t0, p, k_scale, d.eps, cosr, sinr);
        store_row<T_, D, LPR>(Ks + row * D, p, t0);
        load_row<T_, D, LPR>(g + 2 * HD, p, t0);
        store_row<T_, D, LPR>(Vs + row * D, p, t0);
        load_row<T_, D, LPR>(g, p, q);
        ln_rope_row<T_, D, LPR>(q, p, q_scale, d.eps, cosr, sinr);
        store_row<T_, D, LPR>(Qs + row * D, p, q);
        load_row<T_, D, LPR>(dout + tok * lddo + h * D, p, go);
        store_row<T_, D, LPR>(Gs + row * D, p, go);
        load_row<T_, D, LPR>(out + tok * ldo + h * D, p, t0);
#pragma unroll
        for (int i = 0; i < DL; ++i) delta += go[i] * t0[i];
    } else {
#pragma unroll
        for (int i = 0; i < DL; ++i) { q[i] = 0.f; go[i] = 0.f; }
    }
    delta = lpr_sum<LPR>(delta);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();

    // ---- phase A: lanes = query row.  dq_rot, and P / dS rows into LDS ----
    const uint8_t* mrow = mask ? mask + (long)(a / d.mask_div) * T : nullptr;
    const float scale = rsqrtf((float)D);
    const float lse_i = valid ? lse[gi * T + row] : 0.f;
    float dq[DL];
#pragma unroll
    for (int i = 0; i < DL; ++i) dq[i] = 0.f;
    for (int j = 0; j < T; ++j) {
        float pr = 0.f, ds = 0.f;
        if (valid && !(mrow && !mrow[j])) {          // per item: uniform over the lanes of a frame
            const float s = lpr_sum<LPR>(dot_lds<T_, D, LPR>(q, Ks + j * D, p)) * scale;
            pr = __expf(s - lse_i);
            const float dp = lpr_sum<LPR>(dot_lds<T_, D, LPR>(go, Vs + j * D, p));
            ds = pr * (dp - delta) * scale;
            axpy_lds<T_, D, LPR>(dq, ds, Ks + j * D, p);
        }
        if (valid && p == 0) {
            Ps[row * (T + 1) + j] = round_to<T_>(pr);  // the reference multiplies V by probabilities cast to the value dtype
            Ss[row * (T + 1) + j] = ds;
        }
    }
    if (valid) {                                        // dq through RoPE and q_norm (recompute xhat from the raw row)
        load_row<T_, D, LPR>(g, p, t0);
        rs = xhat_row<DL, LPR, D>(t0, d.eps);
        rope_ln_bwd_row<T_, D, LPR>(dq, t0, p, rs, q_scale, cosr, sinr);
        store_row<T_, D, LPR>(dqkv + tok * lddq + h * D, p, dq);
    } else {
#pragma unroll
        for (int i = 0; i < DL; ++i) t0[i] = 0.f;
    }
    float* part = dscale_part + (long)blockIdx.x * 2 * D;
#pragma unroll
    for (int i = 0; i < DL; ++i) {
        const float tot = frames_sum<LPR>(t0[i]);
        if (lane < LPR) part[S::ch(i, p)] = tot;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();

    // ---- phase B: lanes = key row.  dk_rot = sum_i dS[i][j] q_i ; dv = sum_i P[i][j] dO_i ----
    float dk[DL], dv[DL];
#pragma unroll
    for (int i = 0; i < DL; ++i) { dk[i] = 0.f; dv[i] = 0.f; }
    if (valid) {
        for (int i = 0; i < T; ++i) {
            const float pij = Ps[i * (T + 1) + row], dsij = Ss[i * (T + 1) + row];
            axpy_lds<T_, D, LPR>(dk, dsij, Qs + i * D, p);
            axpy_lds<T_, D, LPR>(dv, pij, Gs + i * D, p);
        }
        store_row<T_, D, LPR>(dqkv + tok * lddq + 2 * HD + h * D, p, dv);
        load_row<T_, D, LPR>(g + HD, p, t0);
        rs = xhat_row<DL, LPR, D>(t0, d.eps);
        rope_ln_bwd_row<T_, D, LPR>(dk, t0, p, rs, k_scale, cosr, sinr);
        store_row<T_, D, LPR>(dqkv + tok * lddq + HD + h * D, p, dk);
    } else {
#pragma unroll
        for (int i = 0; i < DL; ++i) t0[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < DL; ++i) {
        const float tot = frames_sum<LPR>(t0[i]);
        if (lane < LPR) part[D + S::ch(i, p)] = tot;
    }
}

// lanes per frame: 16 channels per lane where the wave has room for a whole sequence.
int pick_lpr(int T, int D)
{
    int lpr = D / 16;
    if (lpr < 1) lpr = 1;
    if (lpr > 4) lpr = 4;
    while (lpr > 1 && T * lpr > 64) lpr /= 2;
    return lpr;
}

template <typename T_, int D, int LPR>
int launch_fwd(const void* qkv, int ld, void* out, int ldo, float* lse, const float* qs, const float* ks, const float* cosT,
               const float* sinT, const uint8_t* mask, FAttnDims d, hipStream_t s)
{
    const int ipw = 64 / (d.T * LPR);
    const size_t lds = (size_t)2 * ipw * item_stride_bytes<T_, D>(d.T);
    auto k = tattn_fwd_fast<T_, D, LPR>;
    static size_t attr_lds = 65536;                // grow-only: the attribute is set once per (kernel, larger size)
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_lds = lds;
    }
    hipLaunchKernelGGL(k, dim3(ceil_div(d.items, ipw)), dim3(64), lds, s, (const T_*)qkv, ld, (T_*)out, ldo, lse, qs, ks, cosT, sinT, mask, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

template <typename T_, int D, int LPR>
int launch_bwd(const void* qkv, int ld, const void* out, int ldo, const void* dout, int lddo, const float* lse, void* dqkv, int lddq,
               const float* qs, const float* ks, const float* cosT, const float* sinT, const uint8_t* mask, float* part, FAttnDims d,
               hipStream_t s)
{
    const int ipw = 64 / (d.T * LPR);
    const size_t lds = (size_t)4 * ipw * item_stride_bytes<T_, D>(d.T) + (size_t)2 * ipw * d.T * (d.T + 1) * sizeof(float);
    if (lds > 160 * 1024) return VVAE_ERR_BAD_ARG;
    auto k = tattn_bwd_fast<T_, D, LPR>;
    static size_t attr_lds = 65536;                // grow-only: the attribute is set once per (kernel, larger size)
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_lds = lds;
    }
    hipLaunchKernelGGL(k, dim3(ceil_div(d.items, ipw)), dim3(64), lds, s, (const T_*)qkv, ld, (const T_*)out, ldo, (const T_*)dout, lddo, lse,
                       (T_*)dqkv, lddq, qs, ks, cosT, sinT, mask, part, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

bool fast_ok(int T, int D, int ld, int ldo, int dtype)
{
    if (!(D == 8 || D == 16 || D == 32 || D == 64) || T < 1 || T > 64) return false;
    const int v = dtype == VVAE_DT_F32 ? 4 : 8;
    return ld % v == 0 && ldo % v == 0;
}

}  // namespace

#define FATTN_DISPATCH_T(FN, TT, ...)                                                  \
    switch (D * 8 + lpr) {                                                              \
        case 8 * 8 + 1: return FN<TT, 8, 1>(__VA_ARGS__);                               \
        case 16 * 8 + 1: return FN<TT, 16, 1>(__VA_ARGS__);                             \
        case 32 * 8 + 1: return FN<TT, 32, 1>(__VA_ARGS__);                             \
        case 32 * 8 + 2: return FN<TT, 32, 2>(__VA_ARGS__);                             \
        case 64 * 8 + 1: return FN<TT, 64, 1>(__VA_ARGS__);                             \
        case 64 * 8 + 2: return FN<TT, 64, 2>(__VA_ARGS__);                             \
        case 64 * 8 + 4: return FN<TT, 64, 4>(__VA_ARGS__);                             \
        default: return VVAE_ERR_BAD_ARG;                                               \
    }
#define FATTN_DISPATCH(FN, ...)                                                         \
    do {                                                                                \
        const int lpr = pick_lpr(T, D);                                                 \
        if (dtype == VVAE_DT_F32) { FATTN_DISPATCH_T(FN, float, __VA_ARGS__) }          \
        else { FATTN_DISPATCH_T(FN, bf16_t, __VA_ARGS__) }                              \
    } while (0)

// 1 if the lane-per-frame kernels take this shape (else callers use the generic vvae_temporal_attn_fwd/_bwd).
extern "C" int vvae_temporal_attn_fast_supported(int T, int D, int ld, int ldo, int dtype) { return fast_ok(T, D, ld, ldo, dtype) ? 1 : 0; }

// Rows of the dscale partial buffer (each 2*D floats: [dq_scale | dk_scale]) vvae_temporal_attn_bwd_fast writes for this shape:
// one per workgroup of the VALU kernels, one per persistent wave of the matrix-core kernels (bf16, D = 64, T = 16).
extern "C" int vvae_temporal_attn_fast_blocks(int A, int T, int heads, int D, int dtype)
{
    if (T < 1 || T > 64 || D < 1) return 0;
    if (tmfma_supported(T, D, 8, 8, dtype)) return tmfma_bwd_rows((long)A * heads);
    if (tm32_supported(T, D, 8, 8, dtype)) return A * heads;
    return ceil_div((long)A * heads, 64 / (T * pick_lpr(T, D)));
}

// lse: fp32 (A*heads, T) written.  Other arguments as vvae_temporal_attn_fwd.
extern "C" int vvae_temporal_attn_fwd_fast(const void* qkv, int ld, void* out, int ldo, float* lse, const float* q_scale,
                                           const float* k_scale, const float* cos_table, const float* sin_table, const uint8_t* mask,
                                           int mask_div, int inner, int A, int T, int heads, int D, float eps, int dtype, void* stream)
{
    if (inner <= 0 || A % inner) return VVAE_ERR_BAD_ARG;
    if (!qkv || !out || !lse || !q_scale || !k_scale || !cos_table || !sin_table || A <= 0 || heads <= 0 || mask_div <= 0 ||
        ld < 3 * heads * D || ldo < heads * D || !fast_ok(T, D, ld, ldo, dtype) || ((uintptr_t)qkv % 16) || ((uintptr_t)out % 16))
        return VVAE_ERR_BAD_ARG;
    FAttnDims d{A, T, heads, mask_div, eps, (long)A * heads, inner};
    hipStream_t s = (hipStream_t)stream;
    if (tmfma_supported(T, D, ld, ldo, dtype) && (!mask || ((uintptr_t)mask % 4) == 0))
        return tmfma_fwd(qkv, ld, out, ldo, lse, q_scale, k_scale, cos_table, sin_table, mask, mask_div, inner, A, heads, eps, s);
    if (tm32_supported(T, D, ld, ldo, dtype) && (!mask || ((uintptr_t)mask % 4) == 0))
        return tm32_fwd(qkv, ld, out, ldo, lse, q_scale, k_scale, cos_table, sin_table, mask, mask_div, inner, A, T, heads, eps, s);
    FATTN_DISPATCH(launch_fwd, qkv, ld, out, ldo, lse, q_scale, k_scale, cos_table, sin_table, mask, d, s);
}

// out, lse: forward results.  dscale_part: fp32 (vvae_temporal_attn_fast_blocks(...), 2*D) written; the caller sums rows.
extern "C" int vvae_temporal_attn_bwd_fast(const void* qkv, int ld, const void* out, int ldo, const void* dout, int lddo, const float* lse,
                                           void* dqkv, int lddq, const float* q_scale, const float* k_scale, const float* cos_table,
                                           const float* sin_table, const uint8_t* mask, int mask_div, int inner, float* dscale_part,
                                           int A, int T, int heads, int D, float eps, int dtype, void* stream)
{
    if (inner <= 0 || A % inner) return VVAE_ERR_BAD_ARG;
    if (!qkv || !out || !dout || !lse || !dqkv || !dscale_part || !q_scale || !k_scale || !cos_table || !sin_table || A <= 0 ||
        heads <= 0 || mask_div <= 0 || ld < 3 * heads * D || lddq < 3 * heads * D || ldo < heads * D || lddo < heads * D ||
        !fast_ok(T, D, ld, ldo, dtype) || !fast_ok(T, D, lddq, lddo, dtype) || ((uintptr_t)qkv % 16) || ((uintptr_t)out % 16) ||
        ((uintptr_t)dout % 16) || ((uintptr_t)dqkv % 16)) return VVAE_ERR_BAD_ARG;
    FAttnDims d{A, T, heads, mask_div, eps, (long)A * heads, inner};
    hipStream_t s = (hipStream_t)stream;
    if (tmfma_supported(T, D, ld, ldo, dtype) && tmfma_supported(T, D, lddq, lddo, dtype) && (!mask || ((uintptr_t)mask % 4) == 0))
        return tmfma_bwd(qkv, ld, out, ldo, dout, lddo, lse, dqkv, lddq, q_scale, k_scale, cos_table, sin_table, mask, mask_div, inner,
                         dscale_part, A, heads, eps, s);
    if (tm32_supported(T, D, ld, ldo, dtype) && tm32_supported(T, D, lddq, lddo, dtype) && (!mask || ((uintptr_t)mask % 4) == 0))
        return tm32_bwd(qkv, ld, out, ldo, dout, lddo, lse, dqkv, lddq, q_scale, k_scale, cos_table, sin_table, mask, mask_div, inner,
                        dscale_part, A, T, heads, eps, s);
    FATTN_DISPATCH(launch_bwd, qkv, ld, out, ldo, dout, lddo, lse, dqkv, lddq, q_scale, k_scale, cos_table, sin_table, mask, dscale_part, d, s);
}
