// Temporal-attention core, "one lane = one frame" form: the production path for head_dim D in {8,16,32,64}.
//
// Same math and reference lines as attn_temporal.hip (train/layers.py:159-170), restructured for CDNA4:
//   * a wavefront carries 64/T whole (sequence, head) items; lane (item, t) keeps the q / k / v ROW of frame t in
//     registers (16-byte vector loads of the contiguous head slice), so LayerNorm and RoPE (whose rotate-half partner is
//     in the same row) need no cross-lane traffic at all;
//   * keys and values go to LDS once (storage dtype); every lane then walks the T keys with broadcast ds_read_b128 and
//     an online softmax -- q, k, v are read from HBM once and o written once;
//   * forward also emits the row log-sum-exp; backward uses it plus delta = dO.O (no second softmax pass), computes
//     dS/P row-wise (lane = query), parks them in LDS and accumulates dK/dV column-wise (lane = key);
//   * q/k-norm scale gradients are written as per-workgroup partials (summed by the caller): deterministic.
#include "common.hpp"

namespace {

struct FAttnDims { int A, T, heads, mask_div; float eps; long items; int inner; };

// Token (row of the (tokens, channels) matrix) of frame `row` of sequence `a`.  inner = 1: sequences are contiguous (A, T, C);
// inner = hw: the tensor is (b, t, hw, C) and sequence a = b*hw + i walks frames with stride hw -- the FactoredAttention
// layout, so the temporal half needs no "b t hw c -> (b hw) t c" transpose copies (reference train/layers.py:211,215).
__device__ __forceinline__ long token_of(const FAttnDims& d, int a, int row) {
    return (long)(a / d.inner) * d.T * d.inner + (long)row * d.inner + (a % d.inner);
}

template <typename T_> struct Vw;                                // elements per 16-byte vector
template <> struct Vw<float> { static constexpr int n = 4; };
template <> struct Vw<bf16_t> { static constexpr int n = 8; };

template <typename T_, int D>
__device__ __forceinline__ void load_row(const T_* __restrict__ p, float (&r)[D]) {
    constexpr int V = Vw<T_>::n;
#pragma unroll
    for (int c = 0; c < D / V; ++c) {
        float t[V];
        VecIO<T_, V>::load(p + c * V, t);
#pragma unroll
        for (int e = 0; e < V; ++e) r[c * V + e] = t[e];
    }
}
template <typename T_, int D>
__device__ __forceinline__ void store_row(T_* __restrict__ p, const float (&r)[D]) {
    constexpr int V = Vw<T_>::n;
#pragma unroll
    for (int c = 0; c < D / V; ++c) {
        float t[V];
#pragma unroll
        for (int e = 0; e < V; ++e) t[e] = r[c * V + e];
        VecIO<T_, V>::store(p + c * V, t);
    }
}

// y = round(xhat * scale); optionally returns xhat and rstd (for backward).
template <typename T_, int D, bool KEEP>
__device__ __forceinline__ void ln_row(float (&x)[D], const float* __restrict__ scale, float eps, float (&xhat)[KEEP ? D : 1], float& rstd) {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) { s += x[i]; ss += x[i] * x[i]; }
    const float mean = s / D;
    float var = ss / D - mean * mean;
    var = var < 0.f ? 0.f : var;
    rstd = rsqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float xh = (x[i] - mean) * rstd;
        if (KEEP) xhat[i] = xh;
        x[i] = round_to<T_>(xh * scale[i]);
    }
}

template <typename T_, int D>
__device__ __forceinline__ void rope_row(float (&x)[D], const float* __restrict__ cosr, const float* __restrict__ sinr) {
    constexpr int H = D / 2;
    float y[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float rx = i < H ? -x[i + H] : x[i - H];
        const float c = round_to<T_>(cosr[i]), s = round_to<T_>(sinr[i]);
        y[i] = round_to<T_>(round_to<T_>(x[i] * c) + round_to<T_>(rx * s));
    }
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = y[i];
}

// dy (w.r.t. the RoPE output) -> dx (w.r.t. the raw q/k row), through RoPE and the bias-free LayerNorm.
// dsc[i] receives dy_ln[i] * xhat[i] (this row's contribution to the scale gradient).
// In place, register-lean: g: dy_rot -> dx;  xh: xhat -> this row's scale-gradient contribution dy_ln * xhat.
template <typename T_, int D>
__device__ __forceinline__ void rope_ln_bwd_row(float (&g)[D], float (&xh)[D], float rstd, const float* __restrict__ scale,
                                                const float* __restrict__ cosr, const float* __restrict__ sinr) {
    constexpr int H = D / 2;
#pragma unroll
    for (int i = 0; i < H; ++i) {                       // RoPE transpose on the (i, i+H) pair
        const float lo = g[i], hi = g[i + H];
        g[i] = lo * round_to<T_>(cosr[i]) + hi * round_to<T_>(sinr[i + H]);
        g[i + H] = hi * round_to<T_>(cosr[i + H]) - lo * round_to<T_>(sinr[i]);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float dxh = g[i] * scale[i];
        s1 += dxh; s2 += dxh * xh[i];
    }
    s1 /= D; s2 /= D;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float dy = g[i], x = xh[i];
        g[i] = rstd * (dy * scale[i] - s1 - x * s2);
        xh[i] = dy * x;
    }
}

// x -> xhat in place; returns rstd.
template <int D>
__device__ __forceinline__ float xhat_row(float (&x)[D], float eps) {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) { s += x[i]; ss += x[i] * x[i]; }
    const float mean = s / D;
    float var = ss / D - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = (x[i] - mean) * rstd;
    return rstd;
}

// dot(reg row, LDS row) and axpy(reg row += a * LDS row), LDS row in the storage dtype, 16-byte broadcast reads.
template <typename T_, int D>
__device__ __forceinline__ float dot_lds(const float (&r)[D], const T_* row) {
    constexpr int V = Vw<T_>::n;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D / V; ++c) {
        float t[V];
        VecIO<T_, V>::load(row + c * V, t);
#pragma unroll
        for (int e = 0; e < V; ++e) s += r[c * V + e] * t[e];
    }
    return s;
}
template <typename T_, int D>
__device__ __forceinline__ void axpy_lds(float (&r)[D], float a, const T_* row) {
    constexpr int V = Vw<T_>::n;
#pragma unroll
    for (int c = 0; c < D / V; ++c) {
        float t[V];
        VecIO<T_, V>::load(row + c * V, t);
#pragma unroll
        for (int e = 0; e < V; ++e) r[c * V + e] += a * t[e];
    }
}

constexpr int kItemPad = 16;     // bytes between items in an LDS array: two items in one ds_read_b128 lane group hit different slots

template <typename T_, int D>
__host__ __device__ inline int item_stride_bytes(int T) { return T * D * (int)sizeof(T_) + kItemPad; }

template <typename T_, int D>
__global__ __launch_bounds__(64) void tattn_fwd_fast(const T_* __restrict__ qkv, int ld, T_* __restrict__ out, int ldo, float* __restrict__ lse,
                                                     const float* __restrict__ q_scale, const float* __restrict__ k_scale,
                                                     const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                     const uint8_t* __restrict__ mask, FAttnDims d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = d.T, lane = threadIdx.x;
    const int ipw = 64 / T;
    const int item = lane / T, row = lane - item * T;
    const long gi = (long)blockIdx.x * ipw + item;
    const bool valid = item < ipw && gi < d.items;
    const long gic = valid ? gi : 0;
    const int a = (int)(gic / d.heads), h = (int)(gic % d.heads);
    const int HD = d.heads * D;
    const long tok = token_of(d, a, row);
    const int istride = item_stride_bytes<T_, D>(T);
    T_* Ks = reinterpret_cast<T_*>(smem + (valid ? item : 0) * istride);
    T_* Vs = reinterpret_cast<T_*>(smem + ipw * istride + (valid ? item : 0) * istride);

    float q[D], kv[D];
    const T_* g = qkv + tok * ld + h * D;
    float dummy[1], rs;
    if (valid) {
        load_row<T_, D>(g + HD, kv);
        ln_row<T_, D, false>(kv, k_scale, d.eps, dummy, rs);
        rope_row<T_, D>(kv, cosT + row * D, sinT + row * D);
        store_row<T_, D>(Ks + row * D, kv);
        load_row<T_, D>(g + 2 * HD, kv);
        store_row<T_, D>(Vs + row * D, kv);
        load_row<T_, D>(g, q);
        ln_row<T_, D, false>(q, q_scale, d.eps, dummy, rs);
        rope_row<T_, D>(q, cosT + row * D, sinT + row * D);
    } else {
#pragma unroll
        for (int i = 0; i < D; ++i) q[i] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();

    const uint8_t* mrow = mask ? mask + (long)(a / d.mask_div) * T : nullptr;
    const float scale = rsqrtf((float)D);
    float o[D];
#pragma unroll
    for (int i = 0; i < D; ++i) o[i] = 0.f;
    float m = -3.0e38f, l = 0.f;
    for (int j = 0; j < T; ++j) {
        if (mrow && !mrow[j]) continue;
        const float s = dot_lds<T_, D>(q, Ks + j * D) * scale;
        const float mn = fmaxf(m, s);
        const float alpha = __expf(m - mn), p = __expf(s - mn);
        l = l * alpha + p;
#pragma unroll
        for (int i = 0; i < D; ++i) o[i] *= alpha;
        axpy_lds<T_, D>(o, p, Vs + j * D);
        m = mn;
    }
    if (valid) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
        for (int i = 0; i < D; ++i) o[i] *= inv;
        store_row<T_, D>(out + tok * ldo + h * D, o);
        lse[gi * T + row] = l > 0.f ? m + __logf(l) : 0.f;
    }
}

template <typename T_, int D>
__global__ __launch_bounds__(64) void tattn_bwd_fast(const T_* __restrict__ qkv, int ld, const T_* __restrict__ out, int ldo,
                                                     const T_* __restrict__ dout, int lddo, const float* __restrict__ lse,
                                                     T_* __restrict__ dqkv, int lddq, const float* __restrict__ q_scale,
                                                     const float* __restrict__ k_scale, const float* __restrict__ cosT,
                                                     const float* __restrict__ sinT, const uint8_t* __restrict__ mask,
                                                     float* __restrict__ dscale_part, FAttnDims d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T = d.T, lane = threadIdx.x;
    const int ipw = 64 / T;
    const int item = lane / T, row = lane - item * T;
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
    float q[D], go[D], t0[D];
    float dummy[1], rs;
    float delta = 0.f;
    if (valid) {
        load_row<T_, D>(g + HD, t0);
        ln_row<T_, D, false>(t0, k_scale, d.eps, dummy, rs);
        rope_row<T_, D>(t0, cosr, sinr);
        store_row<T_, D>(Ks + row * D, t0);
        load_row<T_, D>(g + 2 * HD, t0);
        store_row<T_, D>(Vs + row * D, t0);
        load_row<T_, D>(g, q);
        ln_row<T_, D, false>(q, q_scale, d.eps, dummy, rs);
        rope_row<T_, D>(q, cosr, sinr);
        store_row<T_, D>(Qs + row * D, q);
        load_row<T_, D>(dout + tok * lddo + h * D, go);
        store_row<T_, D>(Gs + row * D, go);
        load_row<T_, D>(out + tok * ldo + h * D, t0);
#pragma unroll
        for (int i = 0; i < D; ++i) delta += go[i] * t0[i];
    } else {
#pragma unroll
        for (int i = 0; i < D; ++i) { q[i] = 0.f; go[i] = 0.f; }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();

    // ---- phase A: lane = query row.  dq_rot, and P / dS rows into LDS ----
    const uint8_t* mrow = mask ? mask + (long)(a / d.mask_div) * T : nullptr;
    const float scale = rsqrtf((float)D);
    const float lse_i = valid ? lse[gi * T + row] : 0.f;
    float dq[D];
#pragma unroll
    for (int i = 0; i < D; ++i) dq[i] = 0.f;
    for (int j = 0; j < T; ++j) {
        float p = 0.f, ds = 0.f;
        if (valid && !(mrow && !mrow[j])) {
            const float s = dot_lds<T_, D>(q, Ks + j * D) * scale;
            p = __expf(s - lse_i);
            const float dp = dot_lds<T_, D>(go, Vs + j * D);
            ds = p * (dp - delta) * scale;
            axpy_lds<T_, D>(dq, ds, Ks + j * D);
        }
        if (valid) {
            Ps[row * (T + 1) + j] = round_to<T_>(p);   // the reference multiplies V by probabilities cast to the value dtype
            Ss[row * (T + 1) + j] = ds;
        }
    }
    if (valid) {                                        // dq through RoPE and q_norm (recompute xhat from the raw row)
        load_row<T_, D>(g, t0);
        rs = xhat_row<D>(t0, d.eps);
        rope_ln_bwd_row<T_, D>(dq, t0, rs, q_scale, cosr, sinr);
        store_row<T_, D>(dqkv + tok * lddq + h * D, dq);
    } else {
#pragma unroll
        for (int i = 0; i < D; ++i) t0[i] = 0.f;
    }
    float* part = dscale_part + (long)blockIdx.x * 2 * D;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float tot = wave_sum(t0[i]);
        if (lane == (i & 63)) part[i] = tot;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();

    // ---- phase B: lane = key row.  dk_rot = sum_i dS[i][j] q_i ; dv = sum_i P[i][j] dO_i ----
    float dk[D], dv[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { dk[i] = 0.f; dv[i] = 0.f; }
    if (valid) {
        for (int i = 0; i < T; ++i) {
            const float pij = Ps[i * (T + 1) + row], dsij = Ss[i * (T + 1) + row];
            axpy_lds<T_, D>(dk, dsij, Qs + i * D);
            axpy_lds<T_, D>(dv, pij, Gs + i * D);
        }
        store_row<T_, D>(dqkv + tok * lddq + 2 * HD + h * D, dv);
        load_row<T_, D>(g + HD, t0);
        rs = xhat_row<D>(t0, d.eps);
        rope_ln_bwd_row<T_, D>(dk, t0, rs, k_scale, cosr, sinr);
        store_row<T_, D>(dqkv + tok * lddq + HD + h * D, dk);
    } else {
#pragma unroll
        for (int i = 0; i < D; ++i) t0[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float tot = wave_sum(t0[i]);
        if (lane == (i & 63)) part[D + i] = tot;
    }
}

template <typename T_, int D>
int launch_fwd(const void* qkv, int ld, void* out, int ldo, float* lse, const float* qs, const float* ks, const float* cosT,
               const float* sinT, const uint8_t* mask, FAttnDims d, hipStream_t s)
{
    const int ipw = 64 / d.T;
    const size_t lds = (size_t)2 * ipw * item_stride_bytes<T_, D>(d.T);
    auto k = tattn_fwd_fast<T_, D>;
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

template <typename T_, int D>
int launch_bwd(const void* qkv, int ld, const void* out, int ldo, const void* dout, int lddo, const float* lse, void* dqkv, int lddq,
               const float* qs, const float* ks, const float* cosT, const float* sinT, const uint8_t* mask, float* part, FAttnDims d,
               hipStream_t s)
{
    const int ipw = 64 / d.T;
    const size_t lds = (size_t)4 * ipw * item_stride_bytes<T_, D>(d.T) + (size_t)2 * ipw * d.T * (d.T + 1) * sizeof(float);
    if (lds > 160 * 1024) return VVAE_ERR_BAD_ARG;
    auto k = tattn_bwd_fast<T_, D>;
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

#define FATTN_DISPATCH(FN, ...)                                                                                   \
    do {                                                                                                           \
        if (dtype == VVAE_DT_F32) {                                                                                \
            switch (D) { case 8: return FN<float, 8>(__VA_ARGS__); case 16: return FN<float, 16>(__VA_ARGS__);     \
                         case 32: return FN<float, 32>(__VA_ARGS__); default: return FN<float, 64>(__VA_ARGS__); } \
        } else {                                                                                                   \
            switch (D) { case 8: return FN<bf16_t, 8>(__VA_ARGS__); case 16: return FN<bf16_t, 16>(__VA_ARGS__);   \
                         case 32: return FN<bf16_t, 32>(__VA_ARGS__); default: return FN<bf16_t, 64>(__VA_ARGS__); } \
        }                                                                                                          \
    } while (0)

// 1 if the lane-per-frame kernels take this shape (else callers use the generic vvae_temporal_attn_fwd/_bwd).
extern "C" int vvae_temporal_attn_fast_supported(int T, int D, int ld, int ldo, int dtype) { return fast_ok(T, D, ld, ldo, dtype) ? 1 : 0; }

// Workgroups (= rows of the dscale partial buffer, each 2*D floats: [dq_scale | dk_scale]) for this shape.
extern "C" int vvae_temporal_attn_fast_blocks(int A, int T, int heads) { return ceil_div((long)A * heads, 64 / T); }

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
    FATTN_DISPATCH(launch_bwd, qkv, ld, out, ldo, dout, lddo, lse, dqkv, lddq, q_scale, k_scale, cos_table, sin_table, mask, dscale_part, d, s);
}
