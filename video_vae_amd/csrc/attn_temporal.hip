// Fused temporal-attention core: per-head LayerNorm(q), LayerNorm(k) (no bias, eps 1e-6) -> RoPE (rotate-half)
// -> masked softmax(q k^T / sqrt(D)) v, forward and backward, one wavefront per (sequence, head).
//
// Replaces lines 159-170 of Attention.__call__ (/root/reference/train/layers.py) as called for the temporal half
// of FactoredAttention (layers.py:212-213): q_norm/k_norm (layers.py:164-165), RotaryEmbedding (layers.py:105-129)
// and jax.nn.dot_product_attention with a boolean key-padding mask (layers.py:168).  The QKV / output projections
// and input LayerNorm stay on hipBLASLt.  Sequences are short (T <= 64 frames), so q, k, v of one (sequence, head)
// live in LDS and q/k/v/o each cross HBM exactly once (forward) -- the unfused chain round-trips them 6+ times.
//
// qkv layout: (A, T, 3*heads*D) row pitch ld; q | k | v thirds, head h at h*D (the reference's jnp.split + rearrange).
#include "common.hpp"

namespace {

struct AttnDims { int A, T, heads, D, mask_div; float eps; };

// LDS carve per wave (floats): buffers of T*(D+1) plus one T*(T+1) score tile.
__host__ __device__ inline int buf_floats(int T, int D) { return T * (D + 1); }

template <typename T_>
__device__ __forceinline__ void load_rows(const T_* __restrict__ g, long pitch, int T, int D, float* lds, int lane) {
    for (int idx = lane; idx < T * D; idx += 64) {
        const int t = idx / D, dd = idx - t * D;
        lds[t * (D + 1) + dd] = ldf(g + (long)t * pitch + dd);
    }
}

// In-place LayerNorm (no bias) over D for each of T rows, output rounded to the storage dtype.
// If xhat != nullptr also stores the normalised value before the scale, and rstd per row.
template <typename T_>
__device__ __forceinline__ void ln_rows(float* buf, const float* __restrict__ scale, int T, int D, float eps, int lane,
                                        float* xhat, float* rstd_out) {
    for (int t = 0; t < T; ++t) {
        float s = 0.f, ss = 0.f;
        for (int dd = lane; dd < D; dd += 64) { const float x = buf[t * (D + 1) + dd]; s += x; ss += x * x; }
        s = wave_sum(s); ss = wave_sum(ss);
        const float mean = s / D;
        float var = ss / D - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float rstd = rsqrtf(var + eps);
        for (int dd = lane; dd < D; dd += 64) {
            const float xh = (buf[t * (D + 1) + dd] - mean) * rstd;
            if (xhat) xhat[t * (D + 1) + dd] = xh;
            buf[t * (D + 1) + dd] = round_to<T_>(xh * scale[dd]);
        }
        if (rstd_out && lane == 0) rstd_out[t] = rstd;
    }
}

// RoPE in place: y = x*cos + rotate_half(x)*sin, every product and the sum rounded to the storage dtype
// (the reference casts the tables to q.dtype and computes in that dtype, layers.py:123-127).
template <typename T_>
__device__ __forceinline__ void rope_rows(float* buf, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                          int T, int D, int lane) {
    const int h = D / 2;
    for (int t = 0; t < T; ++t) {
        float y[2]; int n = 0;
        for (int dd = lane; dd < D; dd += 64, ++n) {
            const float x = buf[t * (D + 1) + dd];
            const float rx = dd < h ? -buf[t * (D + 1) + dd + h] : buf[t * (D + 1) + dd - h];
            const float c = round_to<T_>(cosT[t * D + dd]), s = round_to<T_>(sinT[t * D + dd]);
            y[n] = round_to<T_>(round_to<T_>(x * c) + round_to<T_>(rx * s));
        }
        __builtin_amdgcn_wave_barrier();
        n = 0;
        for (int dd = lane; dd < D; dd += 64, ++n) buf[t * (D + 1) + dd] = y[n];
        __builtin_amdgcn_wave_barrier();
    }
}

// S = softmax(mask(QR KR^T / sqrt(D))) rounded to the storage dtype; P in lds [T][T+1]
template <typename T_>
__device__ __forceinline__ void scores_softmax(const float* QR, const float* KR, float* P, const uint8_t* __restrict__ mrow,
                                               int T, int D, int lane) {
    const float scale = rsqrtf((float)D);
    for (int idx = lane; idx < T * T; idx += 64) {
        const int i = idx / T, j = idx - i * T;
        float s = 0.f;
        for (int dd = 0; dd < D; ++dd) s += QR[i * (D + 1) + dd] * KR[j * (D + 1) + dd];
        s *= scale;
        if (mrow && !mrow[j]) s = -0.7f * 3.4028234663852886e38f;
        P[i * (T + 1) + j] = s;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < T; i += 64) {
        float m = -3.4028234663852886e38f;
        for (int j = 0; j < T; ++j) m = fmaxf(m, P[i * (T + 1) + j]);
        float sum = 0.f;
        for (int j = 0; j < T; ++j) { const float e = __expf(P[i * (T + 1) + j] - m); P[i * (T + 1) + j] = e; sum += e; }
        const float inv = 1.f / sum;
        for (int j = 0; j < T; ++j) P[i * (T + 1) + j] = round_to<T_>(P[i * (T + 1) + j] * inv);
    }
    __builtin_amdgcn_wave_barrier();
}

template <typename T_>
__global__ __launch_bounds__(256) void temporal_attn_fwd_kernel(const T_* __restrict__ qkv, int ld, T_* __restrict__ out, int ldo,
                                                                const float* __restrict__ q_scale, const float* __restrict__ k_scale,
                                                                const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                                const uint8_t* __restrict__ mask, AttnDims d, int waves_per_block)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= waves_per_block) return;
    const long item = (long)blockIdx.x * waves_per_block + wave;
    if (item >= (long)d.A * d.heads) return;
    const int a = (int)(item / d.heads), h = (int)(item % d.heads);
    const int T = d.T, D = d.D, HD = d.heads * d.D;
    float* base = smem + (long)wave * (3 * buf_floats(T, D) + T * (T + 1));
    float* QR = base; float* KR = QR + buf_floats(T, D); float* VV = KR + buf_floats(T, D); float* P = VV + buf_floats(T, D);
    const T_* g = qkv + (long)a * T * ld + h * D;
    load_rows(g, ld, T, D, QR, lane);
    load_rows(g + HD, ld, T, D, KR, lane);
    load_rows(g + 2 * HD, ld, T, D, VV, lane);
    __builtin_amdgcn_wave_barrier();
    ln_rows<T_>(QR, q_scale, T, D, d.eps, lane, nullptr, nullptr);
    ln_rows<T_>(KR, k_scale, T, D, d.eps, lane, nullptr, nullptr);
    __builtin_amdgcn_wave_barrier();
    rope_rows<T_>(QR, cosT, sinT, T, D, lane);
    rope_rows<T_>(KR, cosT, sinT, T, D, lane);
    const uint8_t* mrow = mask ? mask + (long)(a / d.mask_div) * T : nullptr;
    scores_softmax<T_>(QR, KR, P, mrow, T, D, lane);
    T_* o = out + (long)a * T * ldo + h * D;
    for (int idx = lane; idx < T * D; idx += 64) {
        const int i = idx / D, dd = idx - i * D;
        float s = 0.f;
        for (int j = 0; j < T; ++j) s += P[i * (T + 1) + j] * VV[j * (D + 1) + dd];
        stf(o + (long)i * ldo + dd, s);
    }
}

// RoPE backward in place on a gradient buffer, then LayerNorm backward (no bias) -> writes dx rows to global
// and accumulates dscale[dd] += sum_rows dy*xhat (one atomic per lane per (sequence, head)).
template <typename T_>
__device__ __forceinline__ void rope_ln_bwd(float* G, const float* XH, const float* rstd, const float* __restrict__ scale,
                                            const float* __restrict__ cosT, const float* __restrict__ sinT, T_* __restrict__ dst,
                                            long pitch, float* __restrict__ dscale, int T, int D, int lane) {
    const int h = D / 2;
    float dsc[2] = {0.f, 0.f};
    for (int t = 0; t < T; ++t) {
        float dy[2]; int n = 0;
        for (int dd = lane; dd < D; dd += 64, ++n) {
            const float c = round_to<T_>(cosT[t * D + dd]);
            const float g0 = G[t * (D + 1) + dd];
            float o;
            if (dd < h) o = G[t * (D + 1) + dd + h] * round_to<T_>(sinT[t * D + dd + h]);
            else o = -G[t * (D + 1) + dd - h] * round_to<T_>(sinT[t * D + dd - h]);
            dy[n] = g0 * c + o;
        }
        // LN backward: y = xhat * gamma
        float s1 = 0.f, s2 = 0.f; n = 0;
        for (int dd = lane; dd < D; dd += 64, ++n) {
            const float xh = XH[t * (D + 1) + dd];
            dsc[n] += dy[n] * xh;
            const float dxh = dy[n] * scale[dd];
            s1 += dxh; s2 += dxh * xh;
        }
        s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
        n = 0;
        for (int dd = lane; dd < D; dd += 64, ++n) {
            const float xh = XH[t * (D + 1) + dd];
            stf(dst + (long)t * pitch + dd, rstd[t] * (dy[n] * scale[dd] - s1 - xh * s2));
        }
    }
    int n = 0;
    for (int dd = lane; dd < D; dd += 64, ++n) dscale[dd] = dsc[n];        // this (sequence, head)'s partial row: folded in index order afterwards
}

template <typename T_>
__global__ __launch_bounds__(256) void temporal_attn_bwd_kernel(const T_* __restrict__ qkv, int ld, const T_* __restrict__ dout, int lddo,
                                                                T_* __restrict__ dqkv, int lddq,
                                                                const float* __restrict__ q_scale, const float* __restrict__ k_scale,
                                                                const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                                const uint8_t* __restrict__ mask, float* __restrict__ dq_scale,
                                                                float* __restrict__ dk_scale, AttnDims d, int waves_per_block)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= waves_per_block) return;
    const long item = (long)blockIdx.x * waves_per_block + wave;
    if (item >= (long)d.A * d.heads) return;
    const int a = (int)(item / d.heads), h = (int)(item % d.heads);
    const int T = d.T, D = d.D, HD = d.heads * d.D, BF = buf_floats(T, D);
    float* base = smem + (long)wave * (6 * BF + T * (T + 1) + 2 * T);
    float* QR = base; float* KR = QR + BF; float* VV = KR + BF; float* DO = VV + BF; float* XQ = DO + BF; float* XK = XQ + BF;
    float* P = XK + BF; float* rq = P + T * (T + 1); float* rk = rq + T;
    const T_* g = qkv + (long)a * T * ld + h * D;
    load_rows(g, ld, T, D, QR, lane);
    load_rows(g + HD, ld, T, D, KR, lane);
    load_rows(g + 2 * HD, ld, T, D, VV, lane);
    load_rows(dout + (long)a * T * lddo + h * D, lddo, T, D, DO, lane);
    __builtin_amdgcn_wave_barrier();
    ln_rows<T_>(QR, q_scale, T, D, d.eps, lane, XQ, rq);
    ln_rows<T_>(KR, k_scale, T, D, d.eps, lane, XK, rk);
    __builtin_amdgcn_wave_barrier();
    rope_rows<T_>(QR, cosT, sinT, T, D, lane);
    rope_rows<T_>(KR, cosT, sinT, T, D, lane);
    const uint8_t* mrow = mask ? mask + (long)(a / d.mask_div) * T : nullptr;
    scores_softmax<T_>(QR, KR, P, mrow, T, D, lane);
    T_* dg = dqkv + (long)a * T * lddq + h * D;
    // dV[j][d] = sum_i P[i][j] dO[i][d]
    for (int idx = lane; idx < T * D; idx += 64) {
        const int j = idx / D, dd = idx - j * D;
        float s = 0.f;
        for (int i = 0; i < T; ++i) s += P[i * (T + 1) + j] * DO[i * (D + 1) + dd];
        stf(dg + 2 * HD + (long)j * lddq + dd, s);
    }
    __builtin_amdgcn_wave_barrier();
    // dP -> dS in place: dS = P * (dP - sum_j P*dP) / sqrt(D)
    const float scale = rsqrtf((float)D);
    for (int i = lane; i < T; i += 64) {
        float dot = 0.f;
        // first pass: dP into registers is too large for T=64; recompute twice instead
        for (int j = 0; j < T; ++j) {
            float dp = 0.f;
            for (int dd = 0; dd < D; ++dd) dp += DO[i * (D + 1) + dd] * VV[j * (D + 1) + dd];
            dot += P[i * (T + 1) + j] * dp;
        }
        for (int j = 0; j < T; ++j) {
            float dp = 0.f;
            for (int dd = 0; dd < D; ++dd) dp += DO[i * (D + 1) + dd] * VV[j * (D + 1) + dd];
            P[i * (T + 1) + j] = P[i * (T + 1) + j] * (dp - dot) * scale;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // dQR -> DO buffer, dKR -> VV buffer (both dead now)
    for (int idx = lane; idx < T * D; idx += 64) {
        const int i = idx / D, dd = idx - i * D;
        float s = 0.f, u = 0.f;
        for (int j = 0; j < T; ++j) {
            s += P[i * (T + 1) + j] * KR[j * (D + 1) + dd];     // dq_rot[i][dd]
            u += P[j * (T + 1) + i] * QR[j * (D + 1) + dd];     // dk_rot[i][dd] = sum_j dS[j][i] q_rot[j][dd]
        }
        DO[i * (D + 1) + dd] = s;
        VV[i * (D + 1) + dd] = u;
    }
    __builtin_amdgcn_wave_barrier();
    // dq_scale here is the partial buffer [item][2 D]: q-norm scale gradients in columns [0, D), k-norm in [D, 2 D) (no atomics)
    rope_ln_bwd<T_>(DO, XQ, rq, q_scale, cosT, sinT, dg, lddq, dq_scale + item * 2 * D, T, D, lane);
    rope_ln_bwd<T_>(VV, XK, rk, k_scale, cosT, sinT, dg + HD, lddq, dq_scale + item * 2 * D + D, T, D, lane);
}

bool attn_ok(const AttnDims& d) {
    return d.A > 0 && d.T > 0 && d.T <= 64 && d.heads > 0 && d.D >= 2 && d.D <= 128 && (d.D % 2) == 0 && d.mask_div > 0;
}

constexpr size_t kMaxLds = 160 * 1024;

}  // namespace

// mask: uint8 (ceil(A/mask_div), T), 1 = attend, or NULL.  cos/sin: fp32 tables (>= T rows, D columns).
extern "C" int vvae_temporal_attn_fwd(const void* qkv, int ld, void* out, int ldo, const float* q_scale, const float* k_scale,
                                      const float* cos_table, const float* sin_table, const uint8_t* mask, int mask_div,
                                      int A, int T, int heads, int D, float eps, int dtype, void* stream)
{
    AttnDims d{A, T, heads, D, mask_div, eps};
    if (!qkv || !out || !q_scale || !k_scale || !cos_table || !sin_table || !attn_ok(d) || ld < 3 * heads * D || ldo < heads * D)
        return VVAE_ERR_BAD_ARG;
    const size_t per_wave = sizeof(float) * (3 * (size_t)buf_floats(T, D) + (size_t)T * (T + 1));
    int wpb = (int)(65536 / per_wave); if (wpb > 4) wpb = 4;
    size_t lds = per_wave * (wpb > 0 ? wpb : 1);
    if (wpb < 1) { wpb = 1; if (lds > kMaxLds) return VVAE_ERR_BAD_ARG; }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(ceil_div((long)A * heads, wpb));
    hipError_t e;
    if (dtype == VVAE_DT_F32) {
        auto k = temporal_attn_fwd_kernel<float>;
        static size_t attr_lds = 65536;
        if (lds > attr_lds) {
            if ((e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return (int)e;
            attr_lds = lds;
        }
        hipLaunchKernelGGL(k, grid, dim3(64 * wpb), lds, s, (const float*)qkv, ld, (float*)out, ldo, q_scale, k_scale, cos_table, sin_table, mask, d, wpb);
    } else if (dtype == VVAE_DT_BF16) {
        auto k = temporal_attn_fwd_kernel<bf16_t>;
        static size_t attr_lds = 65536;
        if (lds > attr_lds) {
            if ((e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return (int)e;
            attr_lds = lds;
        }
        hipLaunchKernelGGL(k, grid, dim3(64 * wpb), lds, s, (const bf16_t*)qkv, ld, (bf16_t*)out, ldo, q_scale, k_scale, cos_table, sin_table, mask, d, wpb);
    } else return VVAE_ERR_BAD_ARG;
    VVAE_LAUNCH_CHECK();
    return 0;
}

// dq_scale / dk_scale: fp32 [D], overwritten.
// Scratch of vvae_temporal_attn_bwd: one partial row of 2 D floats per (sequence, head).
extern "C" size_t vvae_temporal_attn_bwd_ws_bytes(int A, int heads, int D)
{
    return (A > 0 && heads > 0 && D > 0) ? (size_t)A * heads * 2 * D * sizeof(float) : 0;
}

// dq_scale / dk_scale (D floats each) are overwritten: per-(sequence, head) partial rows in ws, folded in index order (no atomics).
extern "C" int vvae_temporal_attn_bwd(const void* qkv, int ld, const void* dout, int lddo, void* dqkv, int lddq,
                                      const float* q_scale, const float* k_scale, const float* cos_table, const float* sin_table,
                                      const uint8_t* mask, int mask_div, float* dq_scale, float* dk_scale,
                                      int A, int T, int heads, int D, float eps, int dtype, void* ws, size_t ws_bytes, void* stream)
{
    AttnDims d{A, T, heads, D, mask_div, eps};
    if (!qkv || !dout || !dqkv || !q_scale || !k_scale || !cos_table || !sin_table || !dq_scale || !dk_scale || !attn_ok(d) ||
        ld < 3 * heads * D || lddq < 3 * heads * D || lddo < heads * D) return VVAE_ERR_BAD_ARG;
    if (!ws || ws_bytes < vvae_temporal_attn_bwd_ws_bytes(A, heads, D) || ((uintptr_t)ws % 16)) return VVAE_ERR_WORKSPACE;
    float* part = (float*)ws;
    const size_t per_wave = sizeof(float) * (6 * (size_t)buf_floats(T, D) + (size_t)T * (T + 1) + 2 * (size_t)T);
    int wpb = (int)(65536 / per_wave); if (wpb > 4) wpb = 4;
    size_t lds = per_wave * (wpb > 0 ? wpb : 1);
    if (wpb < 1) { wpb = 1; if (lds > kMaxLds) return VVAE_ERR_BAD_ARG; }
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    dim3 grid(ceil_div((long)A * heads, wpb));
    if (dtype == VVAE_DT_F32) {
        auto k = temporal_attn_bwd_kernel<float>;
        static size_t attr_lds = 65536;
        if (lds > attr_lds) {
            if ((e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return (int)e;
            attr_lds = lds;
        }
        hipLaunchKernelGGL(k, grid, dim3(64 * wpb), lds, s, (const float*)qkv, ld, (const float*)dout, lddo, (float*)dqkv, lddq,
                           q_scale, k_scale, cos_table, sin_table, mask, part, part, d, wpb);
    } else if (dtype == VVAE_DT_BF16) {
        auto k = temporal_attn_bwd_kernel<bf16_t>;
        static size_t attr_lds = 65536;
        if (lds > attr_lds) {
            if ((e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return (int)e;
            attr_lds = lds;
        }
        hipLaunchKernelGGL(k, grid, dim3(64 * wpb), lds, s, (const bf16_t*)qkv, ld, (const bf16_t*)dout, lddo, (bf16_t*)dqkv, lddq,
                           q_scale, k_scale, cos_table, sin_table, mask, part, part, d, wpb);
    } else return VVAE_ERR_BAD_ARG;
    VVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(vvae_reduce_rows_kernel, dim3(ceil_div(2 * D, 32)), dim3(256), 0, s, (const float*)part, A * heads, (long)2 * D, 2 * D, dq_scale, D,
                       dk_scale);
    VVAE_LAUNCH_CHECK();
    return 0;
}
