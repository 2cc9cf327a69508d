// Temporal-attention core on the matrix cores for T = 32 and T = 64 frames (bf16, head_dim 64): the later stages of the reference's
// batch <-> frames curriculum (train/rl_nonadversarial.py:287-295) and config C5 (B = 2 x 3 x 32 x 256 x 256).
//
// Same math and reference lines as attn_temporal_fast.hip / attn_temporal_mfma.hip (train/layers.py:159-170: q/k LayerNorm, RoPE,
// masked softmax(Q K^T / sqrt(D)) V over the frames of one (patch, head) sequence).  The tile algebra is the spatial kernels'
// (attn_spatial.hip, shared pieces in attn_tile32.hpp) with ONE WAVE per (sequence, head) and four of them per workgroup:
//   * K' = rope(k_norm(K)) and V (backward: also Q' and dO) sit in per-wave swizzled LDS images of T rows x 128 bytes; scores are
//     computed transposed on v_mfma_f32_32x32x16_bf16, so a lane holds 16 keys of ITS query per 32-key tile, softmax is lane-local +
//     one permlane step, and the exponentiated tile is the column operand of the next product as it stands;
//   * rows are frames of one patch: token(a, t) = (a / inner) * T * inner + t * inner + a % inner (the (b, t, hw, c) layout of
//     FactoredAttention, no transposes); global memory is touched in row layout only (4 lanes per row, attn_rows.hpp), accumulator
//     tiles cross over through a 2 KB scratch image;
//   * key mask (uint8 per (sequence / mask_div, frame)): masked scores never enter the maximum, their probabilities are exact zeros;
//     a fully masked sequence gives zero output and lse = 0, like the other temporal kernels;
//   * lse is the natural-log-sum-exp of the scaled scores, the convention of the other temporal kernels (either backward pairs
//     with either forward).
// No workgroup barrier anywhere: every image is private to its wave (wave_lds_fence orders a wave's own LDS traffic).
#include "attn_tile32.hpp"

namespace tm32 {

struct Dims { int A, heads, mask_div, inner, T; float eps; long items; };

__device__ __forceinline__ long token_of(const Dims& d, int a, int row)
{
    return (long)(a / d.inner) * d.T * d.inner + (long)row * d.inner + (a % d.inner);
}

// 16 keys of a score tile: register r <-> key 32 kb + 8 (r / 4) + 4 kh + r % 4.  -> bit r set = key attended.
__device__ __forceinline__ uint32_t key_bits(const uint8_t* __restrict__ mrow, int kb, int kh)
{
    if (!mrow) return 0xffffu;
    uint32_t bits = 0;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(mrow + 32 * kb + 8 * rg + 4 * kh);
#pragma unroll
        for (int e = 0; e < 4; ++e) bits |= ((w >> (8 * e)) & 0xffu) ? (1u << (4 * rg + e)) : 0u;
    }
    return bits;
}

// Stage T rows of one (sequence, head) into a wave-private image: 16 rows x 4 lanes per pass.  NORM: q/k-norm + RoPE on the way.
template <int S_, bool NORM>
__device__ __forceinline__ void stage_image(const bf16_t* __restrict__ src, int ld, const Dims& d, int a, unsigned char* img, int lane,
                                            const float* __restrict__ scale, const float* __restrict__ cosT, const float* __restrict__ sinT)
{
    const int p = lane & 3;
    float sc[16];
    if (NORM) load_tab<bf16_t, SD, 4>(scale, p, sc);
#pragma unroll
    for (int ps = 0; ps < S_ / 16; ++ps) {
        const int row = 16 * ps + (lane >> 2);
        const bf16_t* r = src + token_of(d, a, row) * ld;
        if (NORM) {
            float x[16], cs[16], sn[16];
            load_row<bf16_t, SD, 4>(r, p, x);
            load_tab<bf16_t, SD, 4>(cosT + (long)row * SD, p, cs);
            load_tab<bf16_t, SD, 4>(sinT + (long)row * SD, p, sn);
            ln_rope_row_reg<bf16_t, SD, 4>(x, d.eps, sc, cs, sn);
            float lo[8], hi[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { lo[e] = x[e]; hi[e] = x[8 + e]; }
            VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(img + img_off(row, p)), lo);
            VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(img + img_off(row, 4 + p)), hi);
        } else {
            *reinterpret_cast<uint4*>(img + img_off(row, p)) = *reinterpret_cast<const uint4*>(r + 8 * p);
            *reinterpret_cast<uint4*>(img + img_off(row, 4 + p)) = *reinterpret_cast<const uint4*>(r + 32 + 8 * p);
        }
    }
}

template <int NKB>          // T = 32 * NKB frames
__global__ __launch_bounds__(256) void tattn32_fwd_kernel(const bf16_t* __restrict__ qkv, int ld, bf16_t* __restrict__ out, int ldo,
                                                          float* __restrict__ lse, const float* __restrict__ q_scale,
                                                          const float* __restrict__ k_scale, const float* __restrict__ cosT,
                                                          const float* __restrict__ sinT, const uint8_t* __restrict__ mask, Dims d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int S = 32 * NKB, IMG = S * SROW, PER_WAVE = 2 * IMG + SCR_BYTES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* Ks = smem + wave * PER_WAVE;
    unsigned char* Vs = Ks + IMG;
    unsigned char* scr = Vs + IMG;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= d.items) return;                                   // wave-uniform; no workgroup barrier in this kernel
    const int a = (int)(item / d.heads), h = (int)(item % d.heads);
    const int HD = d.heads * SD;
    const bf16_t* base = qkv + h * SD;
    const uint8_t* mrow = mask ? mask + (long)(a / d.mask_div) * S : nullptr;

    stage_image<S, true>(base + HD, ld, d, a, Ks, lane, k_scale, cosT, sinT);
    stage_image<S, false>(base + 2 * HD, ld, d, a, Vs, lane, nullptr, nullptr, nullptr);
    wave_lds_fence();

    const int j = lane & 31, kh = lane >> 5;
    const FragAddr fa(lane);
    const float sm_scale = rsqrtf((float)SD);
    const float c2 = sm_scale * 1.44269504088896341f;

#pragma unroll 1
    for (int qb = 0; qb < NKB; ++qb) {
        const int qrow = qb * 32 + j;
        bf16x8 qf[4];
        {
            float x[2][16], cs[2][16], sn[2][16], sc[16];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int rr = qb * 32 + 16 * hf + (lane >> 2);
                load_row<bf16_t, SD, 4>(base + token_of(d, a, rr) * ld, lane & 3, x[hf]);
                load_tab<bf16_t, SD, 4>(cosT + (long)rr * SD, lane & 3, cs[hf]);
                load_tab<bf16_t, SD, 4>(sinT + (long)rr * SD, lane & 3, sn[hf]);
            }
            load_tab<bf16_t, SD, 4>(q_scale, lane & 3, sc);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                ln_rope_row_reg<bf16_t, SD, 4>(x[hf], d.eps, sc, cs[hf], sn[hf]);
                wave_lds_fence();
                rows_put(scr, lane, x[hf]);
                wave_lds_fence();
                if ((j >> 4) == hf) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        qf[ks] = *reinterpret_cast<const bf16x8*>(scr + (j & 15) * SROW + (((2 * ks + kh) ^ gsw(j & 15)) << 4));
                }
            }
        }
        // S^T tiles: s[kb][r] = score(key 32 kb + 8 (r/4) + 4 kh + r%4, query j)
        f32x16 s[NKB];
        uint32_t bits[NKB];
        float m = -3.0e38f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.rowfrag(Ks, kb, ks), qf[ks], s[kb], 0, 0, 0);
            bits[kb] = key_bits(mrow, kb, kh);
#pragma unroll
            for (int e = 0; e < 16; ++e) m = ((bits[kb] >> e) & 1u) ? fmaxf(m, s[kb][e]) : m;
        }
        m = fmaxf(m, xor32(m));
        float l = 0.f;
        const float mc2 = m * c2;                                       // one fma per score instead of a subtraction and a product
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe = ((bits[kb] >> e) & 1u) ? exp2_fast(__builtin_fmaf(s[kb][e], c2, -mc2)) : 0.f;
                s[kb][e] = pe;
                l += pe;
            }
        l += xor32(l);
        const float inv = l > 0.f ? 1.f / l : 0.f;
        // O^T = V^T P^T with the probabilities normalised in fp32 and rounded to bf16 first (where the reference casts them)
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float pv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) pv[e] = s[kb][8 * u + e] * inv;
                const bf16x8 pf = pack8(pv);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.trfrag(Vs, kb, u, dt), pf, o[dt], 0, 0, 0);
            }
        float og[2][16];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) og[dt][e] = o[dt][e];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            wave_lds_fence();
            acc_half_put(scr, j, kh, hf, og);
            wave_lds_fence();
            float x[16];
            rows_get(scr, lane, x);
            store_row<bf16_t, SD, 4>(out + token_of(d, a, qb * 32 + 16 * hf + (lane >> 2)) * ldo + h * SD, lane & 3, x);
        }
        if (kh == 0) lse[item * S + qrow] = l > 0.f ? m * sm_scale + __logf(l) : 0.f;
    }
}

// part: fp32 (items, 2 * D): one row per (sequence, head), [dq_scale | dk_scale] contributions (the caller sums rows).
template <int NKB>
__global__ __launch_bounds__(256, 2) void tattn32_bwd_kernel(const bf16_t* __restrict__ qkv, int ld, const bf16_t* __restrict__ out, int ldo,
                                                          const bf16_t* __restrict__ dout, int lddo, const float* __restrict__ lse,
                                                          bf16_t* __restrict__ dqkv, int lddq, const float* __restrict__ q_scale,
                                                          const float* __restrict__ k_scale, const float* __restrict__ cosT,
                                                          const float* __restrict__ sinT, const uint8_t* __restrict__ mask,
                                                          float* __restrict__ part, Dims d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int S = 32 * NKB, IMG = S * SROW, PER_WAVE = 4 * IMG + 2 * S * 4 + SCR_BYTES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* Qs = smem + wave * PER_WAVE;
    unsigned char* Ks = Qs + IMG;
    unsigned char* Vs = Ks + IMG;
    unsigned char* Gs = Vs + IMG;                      // dO
    float* lseS = reinterpret_cast<float*>(Gs + IMG);  // base-2 log-sum-exp per query
    float* delS = lseS + S;
    unsigned char* scr = reinterpret_cast<unsigned char*>(delS + S);
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= d.items) return;                                   // wave-uniform; no workgroup barrier in this kernel
    const int a = (int)(item / d.heads), h = (int)(item % d.heads);
    const int HD = d.heads * SD;
    const bf16_t* base = qkv + h * SD;
    const bf16_t* gbase = dout + h * SD;
    const uint8_t* mrow = mask ? mask + (long)(a / d.mask_div) * S : nullptr;
    const int j = lane & 31, kh = lane >> 5, p = lane & 3;
    const FragAddr fa(lane);
    const float sm_scale = rsqrtf((float)SD);
    const float c2 = sm_scale * 1.44269504088896341f;

    // ---- staging: Q' and K' (norm + RoPE), V, dO, delta = rowsum(dO * O), lse (to base 2)
    stage_image<S, true>(base, ld, d, a, Qs, lane, q_scale, cosT, sinT);
    stage_image<S, true>(base + HD, ld, d, a, Ks, lane, k_scale, cosT, sinT);
    stage_image<S, false>(base + 2 * HD, ld, d, a, Vs, lane, nullptr, nullptr, nullptr);
#pragma unroll
    for (int ps = 0; ps < S / 16; ++ps) {
        const int row = 16 * ps + (lane >> 2);
        const long tok = token_of(d, a, row);
        float go[16], oo[16];
        load_row<bf16_t, SD, 4>(gbase + tok * lddo, p, go);
        load_row<bf16_t, SD, 4>(out + tok * ldo + h * SD, p, oo);
        float dl = 0.f, lo[8], hi[8];
#pragma unroll
        for (int i = 0; i < 16; ++i) dl += go[i] * oo[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) { lo[e] = go[e]; hi[e] = go[8 + e]; }
        VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(Gs + img_off(row, p)), lo);
        VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(Gs + img_off(row, 4 + p)), hi);
        dl = lpr_sum<4>(dl);
        if (p == 0) { delS[row] = dl; lseS[row] = lse[item * S + row] * 1.44269504088896341f; }
    }
    wave_lds_fence();

    // a finished 32-row gradient tile leaves through the scratch image in row layout (see attn_spatial.hip)
    auto put_v_tile = [&](const float (&g)[2][16], int row0) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            wave_lds_fence();
            acc_half_put(scr, j, kh, hf, g);
            wave_lds_fence();
            float x[16];
            rows_get(scr, lane, x);
            store_row<bf16_t, SD, 4>(dqkv + token_of(d, a, row0 + 16 * hf + (lane >> 2)) * lddq + 2 * HD + h * SD, p, x);
        }
    };
    struct RowCtx { float xr[2][16], cs[2][16], sn[2][16], sc[16]; };
    auto qk_loads = [&](RowCtx& c, int row0, int which, const float* __restrict__ scale) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int rr = row0 + 16 * hf + (lane >> 2);
            load_row<bf16_t, SD, 4>(base + which * HD + token_of(d, a, rr) * ld, p, c.xr[hf]);
            load_tab<bf16_t, SD, 4>(cosT + (long)rr * SD, p, c.cs[hf]);
            load_tab<bf16_t, SD, 4>(sinT + (long)rr * SD, p, c.sn[hf]);
        }
        load_tab<bf16_t, SD, 4>(scale, p, c.sc);
    };
    // -> adds this tile's scale-gradient contributions (summed over its 32 rows, valid in lanes 0-3) to acc[16]
    auto qk_finish = [&](RowCtx& c, const float (&g)[2][16], int row0, int which, float (&acc)[16]) {
        float ds[16];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int rr = row0 + 16 * hf + (lane >> 2);
            wave_lds_fence();
            acc_half_put(scr, j, kh, hf, g);
            wave_lds_fence();
            float gx[16];
            rows_get(scr, lane, gx);
            const float rstd = xhat_row<16, 4, SD>(c.xr[hf], d.eps);
            rope_ln_bwd_row_reg<bf16_t, SD, 4>(gx, c.xr[hf], rstd, c.sc, c.cs[hf], c.sn[hf]);
            store_row<bf16_t, SD, 4>(dqkv + token_of(d, a, rr) * lddq + which * HD + h * SD, p, gx);
#pragma unroll
            for (int i = 0; i < 16; ++i) ds[i] = hf ? ds[i] + c.xr[hf][i] : c.xr[hf][i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += frames_sum<4>(ds[i]);
    };
    float accq[16], acck[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { accq[i] = 0.f; acck[i] = 0.f; }

    // ---- phase A: dV, dK per key tile (scores as [query][key]: the lane IS the key)
#pragma unroll 1
    for (int kt = 0; kt < NKB; ++kt) {
        bf16x8 kc[4], vc[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kc[ks] = fa.rowfrag(Ks, kt, ks); vc[ks] = fa.rowfrag(Vs, kt, ks); }
        const bool key_on = mrow ? mrow[kt * 32 + j] != 0 : true;
        f32x16 dv[2], dk[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dv[dt][e] = 0.f; dk[dt][e] = 0.f; }
#pragma unroll 1
        for (int qb = 0; qb < NKB; ++qb) {
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.rowfrag(Qs, qb, ks), kc[ks], s, 0, 0, 0);        // [query][key]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.rowfrag(Gs, qb, ks), vc[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float p8[8], s8[8];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const float4 lv = *reinterpret_cast<const float4*>(lseS + qb * 32 + 8 * (2 * u + hh) + 4 * kh);
                    const float4 dv4 = *reinterpret_cast<const float4*>(delS + qb * 32 + 8 * (2 * u + hh) + 4 * kh);
                    const float lq[4] = {lv.x, lv.y, lv.z, lv.w}, dq4[4] = {dv4.x, dv4.y, dv4.z, dv4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = key_on ? exp2_fast(s[8 * u + 4 * hh + e] * c2 - lq[e]) : 0.f;
                        p8[4 * hh + e] = pe;
                        s8[4 * hh + e] = pe * (dp[8 * u + 4 * hh + e] - dq4[e]);
                    }
                }
                const bf16x8 pf = pack8(p8), sf = pack8(s8);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.trfrag(Gs, qb, u, dt), pf, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.trfrag(Qs, qb, u, dt), sf, dk[dt], 0, 0, 0);
                }
            }
        }
        RowCtx c;
        qk_loads(c, kt * 32, 1, k_scale);
        float g[2][16];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[dt][r] = dv[dt][r];
        put_v_tile(g, kt * 32);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[dt][r] = dk[dt][r] * sm_scale;
        qk_finish(c, g, kt * 32, 1, acck);
    }

    // ---- phase B: dQ per query tile (scores as [key][query]: the lane IS the query)
#pragma unroll 1
    for (int qt = 0; qt < NKB; ++qt) {
        const int qrow = qt * 32 + j;
        bf16x8 qc[4], gc[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qc[ks] = fa.rowfrag(Qs, qt, ks); gc[ks] = fa.rowfrag(Gs, qt, ks); }
        const float lq = lseS[qrow], dl = delS[qrow];
        f32x16 dq[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;
#pragma unroll 1
        for (int kb = 0; kb < NKB; ++kb) {
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.rowfrag(Ks, kb, ks), qc[ks], s, 0, 0, 0);        // [key][query]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.rowfrag(Vs, kb, ks), gc[ks], dp, 0, 0, 0);
            }
            const uint32_t bits = key_bits(mrow, kb, kh);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float s8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    s8[e] = ((bits >> (8 * u + e)) & 1u) ? exp2_fast(s[8 * u + e] * c2 - lq) * (dp[8 * u + e] - dl) : 0.f;
                const bf16x8 sf = pack8(s8);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.trfrag(Ks, kb, u, dt), sf, dq[dt], 0, 0, 0);
            }
        }
        float g[2][16];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[dt][r] = dq[dt][r] * sm_scale;
        RowCtx c;
        qk_loads(c, qt * 32, 0, q_scale);
        qk_finish(c, g, qt * 32, 0, accq);
    }
    if (lane < 4) {
        float* pr = part + item * 2 * SD;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ch = Slice<bf16_t, SD, 4>::ch(i, p);
            pr[ch] = accq[i];
            pr[SD + ch] = acck[i];
        }
    }
}

int g_enable = 1;

bool shape_ok(int T_, int D_, int ld, int ldo, int dtype)
{
    return g_enable && dtype == VVAE_DT_BF16 && (T_ == 32 || T_ == 64) && D_ == SD && ld % 8 == 0 && ldo % 8 == 0;
}

template <int NKB>
int launch_fwd(const void* qkv, int ld, void* out, int ldo, float* lse, const float* qs, const float* ks, const float* cosT, const float* sinT,
               const uint8_t* mask, Dims d, hipStream_t s)
{
    constexpr int lds = 4 * (2 * 32 * NKB * SROW + SCR_BYTES);
    auto k = tattn32_fwd_kernel<NKB>;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)((d.items + 3) / 4)), dim3(256), lds, s, (const bf16_t*)qkv, ld, (bf16_t*)out, ldo, lse, qs, ks, cosT,
                       sinT, mask, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

template <int NKB>
int launch_bwd(const void* qkv, int ld, const void* out, int ldo, const void* dout, int lddo, const float* lse, void* dqkv, int lddq,
               const float* qs, const float* ks, const float* cosT, const float* sinT, const uint8_t* mask, float* part, Dims d, hipStream_t s)
{
    constexpr int lds = 4 * (4 * 32 * NKB * SROW + 2 * 32 * NKB * 4 + SCR_BYTES);
    auto k = tattn32_bwd_kernel<NKB>;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)((d.items + 3) / 4)), dim3(256), lds, s, (const bf16_t*)qkv, ld, (const bf16_t*)out, ldo,
                       (const bf16_t*)dout, lddo, lse, (bf16_t*)dqkv, lddq, qs, ks, cosT, sinT, mask, part, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

}  // namespace tm32

// Test / tuning hook: 0 routes T = 32 / 64, head_dim 64, bf16 temporal attention back to the VALU kernels of attn_temporal_fast.hip.
extern "C" int vvae_temporal_attn_mfma32_enable(int on)
{
    tm32::g_enable = on ? 1 : 0;
    return 0;
}

// internal entry points used by attn_temporal_fast.hip's dispatch (declared there)
int tm32_supported(int T, int D, int ld, int ldo, int dtype) { return tm32::shape_ok(T, D, ld, ldo, dtype) ? 1 : 0; }

int tm32_fwd(const void* qkv, int ld, void* out, int ldo, float* lse, const float* qs, const float* ks, const float* cosT, const float* sinT,
             const uint8_t* mask, int mask_div, int inner, int A, int T, int heads, float eps, hipStream_t s)
{
    tm32::Dims d{A, heads, mask_div, inner, T, eps, (long)A * heads};
    if (mask && ((uintptr_t)mask % 4)) return VVAE_ERR_BAD_ARG;
    if (T == 32) return tm32::launch_fwd<1>(qkv, ld, out, ldo, lse, qs, ks, cosT, sinT, mask, d, s);
    return tm32::launch_fwd<2>(qkv, ld, out, ldo, lse, qs, ks, cosT, sinT, mask, d, s);
}

int tm32_bwd(const void* qkv, int ld, const void* out, int ldo, const void* dout, int lddo, const float* lse, void* dqkv, int lddq,
             const float* qs, const float* ks, const float* cosT, const float* sinT, const uint8_t* mask, int mask_div, int inner,
             float* part, int A, int T, int heads, float eps, hipStream_t s)
{
    tm32::Dims d{A, heads, mask_div, inner, T, eps, (long)A * heads};
    if (mask && ((uintptr_t)mask % 4)) return VVAE_ERR_BAD_ARG;
    if (T == 32) return tm32::launch_bwd<1>(qkv, ld, out, ldo, dout, lddo, lse, dqkv, lddq, qs, ks, cosT, sinT, mask, part, d, s);
    return tm32::launch_bwd<2>(qkv, ld, out, ldo, dout, lddo, lse, dqkv, lddq, qs, ks, cosT, sinT, mask, part, d, s);
}
