// Row-slice helpers shared by the fused attention kernels (attn_temporal_fast.hip, qk_prep.hip).
//
// A head row of D channels is split over LPR adjacent lanes, DL = D/LPR channels each.  Lane p owns channels
// [p*HL, (p+1)*HL) and [D/2 + p*HL, D/2 + (p+1)*HL), HL = DL/2, so the RoPE rotate-half partner of every channel sits in
// the SAME lane; LayerNorm sums finish with one or two quad-DPP adds (reference train/layers.py:100-128,159-163).
//
// RoPE tables: the kernels built on these helpers take cos / sin tables whose values are ALREADY rounded to the activation dtype
// (the reference multiplies by tables cast to q's dtype, layers.py:113-114): the caller rounds them once, the kernels do not
// spend four conversions per rotated pair on it.
#pragma once
#include "common.hpp"

namespace {

template <typename T_> struct Vw;                                // elements per 16-byte vector
template <> struct Vw<float> { static constexpr int n = 4; };
template <> struct Vw<bf16_t> { static constexpr int n = 8; };

// Geometry of one lane's slice of a D-channel row.
template <typename T_, int D, int LPR> struct Slice {
    static constexpr int DL = D / LPR;                            // channels per lane
    static constexpr int HL = DL / 2;                             // ... per half (lo / hi)
    static constexpr int H = D / 2;
    static constexpr int V = Vw<T_>::n < HL ? Vw<T_>::n : HL;     // vector width of the global / LDS accesses
    static_assert(HL % V == 0 && HL >= 4, "slice halves must be whole vectors");
    // channel of register r for lane p
    static __device__ __forceinline__ int ch(int r, int p) { return r < HL ? p * HL + r : H + p * HL + (r - HL); }
    static __device__ __forceinline__ int lo(int p) { return p * HL; }
    static __device__ __forceinline__ int hi(int p) { return H + p * HL; }
};

// sum over the LPR lanes of a frame (adjacent lanes: quad_perm DPP, no LDS traffic)
__device__ __forceinline__ float dpp_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
}
template <int LPR> __device__ __forceinline__ float lpr_sum(float v) {
    static_assert(LPR == 1 || LPR == 2 || LPR == 4, "frames are split over at most a lane quad");
    if (LPR >= 2) v += dpp_xor1(v);
    if (LPR >= 4) v += dpp_xor2(v);
    return v;
}
// sum over the lanes that hold the SAME channels (every LPR-th lane of the wave)
template <int LPR> __device__ __forceinline__ float frames_sum(float v) { return butterfly_sum<32, LPR>(v); }   // VALU only (common.hpp)

template <typename T_, int D, int LPR>
__device__ __forceinline__ void load_row(const T_* __restrict__ row, int p, float (&r)[D / LPR]) {
    using S = Slice<T_, D, LPR>;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const T_* src = row + (half ? S::hi(p) : S::lo(p));
#pragma unroll
        for (int c = 0; c < S::HL / S::V; ++c) {
            float t[S::V];
            VecIO<T_, S::V>::load(src + c * S::V, t);
#pragma unroll
            for (int e = 0; e < S::V; ++e) r[half * S::HL + c * S::V + e] = t[e];
        }
    }
}
template <typename T_, int D, int LPR>
__device__ __forceinline__ void store_row(T_* __restrict__ row, int p, const float (&r)[D / LPR]) {
    using S = Slice<T_, D, LPR>;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        T_* dst = row + (half ? S::hi(p) : S::lo(p));
#pragma unroll
        for (int c = 0; c < S::HL / S::V; ++c) {
            float t[S::V];
#pragma unroll
            for (int e = 0; e < S::V; ++e) t[e] = r[half * S::HL + c * S::V + e];
            VecIO<T_, S::V>::store(dst + c * S::V, t);
        }
    }
}
// fp32 table slice (scale / cos / sin rows), same channel mapping
template <typename T_, int D, int LPR>
__device__ __forceinline__ void load_tab(const float* __restrict__ row, int p, float (&r)[D / LPR]) {
    using S = Slice<T_, D, LPR>;
#pragma unroll
    for (int i = 0; i < S::DL; ++i) r[i] = row[S::ch(i, p)];
}

// x -> xhat in place; returns rstd.
template <int DL, int LPR, int D>
__device__ __forceinline__ float xhat_row(float (&x)[DL], float eps) {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < DL; ++i) { s += x[i]; ss += x[i] * x[i]; }
    s = lpr_sum<LPR>(s); ss = lpr_sum<LPR>(ss);
    const float mean = s / D;
    float var = ss / D - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < DL; ++i) x[i] = (x[i] - mean) * rstd;
    return rstd;
}

// q/k-norm (bias-free LayerNorm, y = round(xhat * scale)) followed by RoPE, in place on the lane's slice.
template <typename T_, int D, int LPR>
__device__ __forceinline__ void ln_rope_row(float (&x)[D / LPR], int p, const float* __restrict__ scale, float eps,
                                            const float* __restrict__ cosr, const float* __restrict__ sinr) {
    using S = Slice<T_, D, LPR>;
    xhat_row<S::DL, LPR, D>(x, eps);
#pragma unroll
    for (int i = 0; i < S::HL; ++i) {                    // rotate-half pair (i, i + HL) = channels (c, c + D/2); roundings in pairs
        const int cl = S::ch(i, p), chh = S::ch(i + S::HL, p);
        float lo = x[i] * scale[cl], hi = x[i + S::HL] * scale[chh];
        round2<T_>(lo, hi);
        float a = lo * cosr[cl], b = -hi * sinr[cl], c = hi * cosr[chh], e = lo * sinr[chh];
        round2<T_>(a, b);
        round2<T_>(c, e);
        float y0 = a + b, y1 = c + e;
        round2<T_>(y0, y1);
        x[i] = y0; x[i + S::HL] = y1;
    }
}

// dy (w.r.t. the RoPE output) -> dx (w.r.t. the raw q/k row), through RoPE and the bias-free LayerNorm.
// In place, register-lean: g: dy_rot -> dx;  xh: xhat -> this row's scale-gradient contribution dy_ln * xhat.
template <typename T_, int D, int LPR>
__device__ __forceinline__ void rope_ln_bwd_row(float (&g)[D / LPR], float (&xh)[D / LPR], int p, float rstd,
                                                const float* __restrict__ scale, const float* __restrict__ cosr,
                                                const float* __restrict__ sinr) {
    using S = Slice<T_, D, LPR>;
#pragma unroll
    for (int i = 0; i < S::HL; ++i) {                    // RoPE transpose on the (i, i+HL) pair
        const float lo = g[i], hi = g[i + S::HL];
        const int cl = S::ch(i, p), chh = S::ch(i + S::HL, p);
        g[i] = lo * cosr[cl] + hi * sinr[chh];
        g[i + S::HL] = hi * cosr[chh] - lo * sinr[cl];
    }
    float sc[S::DL];
    load_tab<T_, D, LPR>(scale, p, sc);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < S::DL; ++i) {
        const float dxh = g[i] * sc[i];
        s1 += dxh; s2 += dxh * xh[i];
    }
    s1 = lpr_sum<LPR>(s1) / D; s2 = lpr_sum<LPR>(s2) / D;
#pragma unroll
    for (int i = 0; i < S::DL; ++i) {
        const float dy = g[i], x = xh[i];
        g[i] = rstd * (dy * sc[i] - s1 - x * s2);
        xh[i] = dy * x;
    }
}

// The two row transforms with scale / cos / sin already in registers (load_tab order: element i = channel ch(i, p)): a kernel
// that issues every global load of a tile up front, before the first dependent instruction, uses these.
template <typename T_, int D, int LPR>
__device__ __forceinline__ void ln_rope_row_reg(float (&x)[D / LPR], float eps, const float (&sc)[D / LPR], const float (&cs)[D / LPR],
                                                const float (&sn)[D / LPR]) {
    using S = Slice<T_, D, LPR>;
    xhat_row<S::DL, LPR, D>(x, eps);
#pragma unroll
    for (int i = 0; i < S::HL; ++i) {
        float lo = x[i] * sc[i], hi = x[i + S::HL] * sc[i + S::HL];
        round2<T_>(lo, hi);
        float a = lo * cs[i], b = -hi * sn[i], c = hi * cs[i + S::HL], e = lo * sn[i + S::HL];
        round2<T_>(a, b);
        round2<T_>(c, e);
        float y0 = a + b, y1 = c + e;
        round2<T_>(y0, y1);
        x[i] = y0; x[i + S::HL] = y1;
    }
}
template <typename T_, int D, int LPR>
__device__ __forceinline__ void rope_ln_bwd_row_reg(float (&g)[D / LPR], float (&xh)[D / LPR], float rstd, const float (&sc)[D / LPR],
                                                    const float (&cs)[D / LPR], const float (&sn)[D / LPR]) {
    using S = Slice<T_, D, LPR>;
#pragma unroll
    for (int i = 0; i < S::HL; ++i) {
        const float lo = g[i], hi = g[i + S::HL];
        g[i] = lo * cs[i] + hi * sn[i + S::HL];
        g[i + S::HL] = hi * cs[i + S::HL] - lo * sn[i];
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < S::DL; ++i) {
        const float dxh = g[i] * sc[i];
        s1 += dxh; s2 += dxh * xh[i];
    }
    s1 = lpr_sum<LPR>(s1) / D; s2 = lpr_sum<LPR>(s2) / D;
#pragma unroll
    for (int i = 0; i < S::DL; ++i) {
        const float dy = g[i], x = xh[i];
        g[i] = rstd * (dy * sc[i] - s1 - x * s2);
        xh[i] = dy * x;
    }
}

}  // namespace
