// Temporal-attention core on the matrix cores: bf16, head_dim 64, T = 16 frames (config C3/C4: B x 3 x 16 x 256 x 256).
//
// Same math and reference lines as attn_temporal_fast.hip (train/layers.py:159-170: q/k LayerNorm, RoPE, masked
// softmax(Q K^T / sqrt(D)) V over the frames of one (patch, head) sequence), which spends ~1000 VALU instructions per
// (sequence, head) on the two 16 x 16 x 64 products.  Here ONE WAVE owns one (sequence, head):
//
//   * lane l = 16 p + t keeps, of frame t's q / k / v rows, the channels [8p, 8p+8) and [32+8p, 32+8p+8): the RoPE rotate-half
//     partner of every channel is in the same lane, the LayerNorm sums finish with two v_permlane swaps (over p), and -- the point of
//     the numbering -- those two 8-channel pieces ARE the A/B fragments of v_mfma_f32_16x16x32_bf16 for k-steps 0 and 1
//     (lane l holds row l & 15, k = 8 (l >> 4) + j): Q', K' never touch LDS for the score products;
//   * scores are computed TRANSPOSED, S^T = K' Q'^T: a lane then holds, for ITS query (column l & 15), the 4 keys 4p..4p+3, so
//     the softmax is 4 registers + two permlane swaps per query, and the exponentiated tile is already the B operand
//     (k = key, column = query) of O^T = V^T P^T on v_mfma_f32_16x16x16_bf16 -- no transpose of P anywhere;
//   * V^T (and, backward, K'^T, Q'^T, dO^T) fragments come out of a row-major LDS image through ds_read_b64_tr_b16 (row pitch
//     160 B: the 8 rows a half-wave reads land on 8 distinct bank octets);  O^T[channel 4p+r][query] leaves as 8-byte stores;
//   * probabilities are normalised in fp32 and rounded to bf16 BEFORE the PV product, exactly where the reference casts them
//     (jax.nn.dot_product_attention: softmax in fp32, probs cast to the value dtype);
//   * backward recomputes both orientations of the score tile (S^T for dQ, S for dK / dV: two extra 16 x 16 x 64 products instead
//     of any transpose through LDS), 20 MFMAs per (sequence, head) in all; the q/k-norm scale gradients of a workgroup's four items
//     are summed over frames with DPP row adds and over the waves through LDS and leave as ONE partial row per workgroup, folded by
//     the caller in fixed order: deterministic.
#include "common.hpp"

namespace tmfma {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4v __attribute__((ext_vector_type(4)));

constexpr int T = 16, D = 64;
constexpr int PITCH = 160;                    // bytes per row of an LDS image (16 rows x 64 bf16 + pad)
constexpr int IMG = T * PITCH;                // 2560 B

struct Dims { int A, heads, mask_div, inner; float eps; long items; };

__device__ __forceinline__ long token_of(const Dims& d, int a, int row) {
    return (long)(a / d.inner) * T * d.inner + (long)row * d.inner + (a % d.inner);
}

__device__ __forceinline__ void unpack8(const uint4& r, float* v) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 pack8(const float* v) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f2bf(v[2 * i]) | ((uint32_t)f2bf(v[2 * i + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ bf16x8 frag8(const uint4& r) { return __builtin_bit_cast(bf16x8, r); }
__device__ __forceinline__ s16x4v pack4(float a, float b, float c, float e) {
    s16x4v r = {(short)f2bf(a), (short)f2bf(b), (short)f2bf(c), (short)f2bf(e)};
    return r;
}
__device__ __forceinline__ s16x4v tr4(const unsigned char* p) {
    typedef __attribute__((address_space(3))) s16x4v* lds_ptr;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p);
}
__device__ __forceinline__ f32x4 mfma32(const bf16x8& a, const bf16x8& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(const s16x4v& a, const s16x4v& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

// sum over the 4 lanes (p = 0..3) that share a frame: lanes t, 16 + t, 32 + t, 48 + t
__device__ __forceinline__ float psum(float v) { return butterfly_sum<32, 16>(v); }
__device__ __forceinline__ float pmax(float v) { v = fmaxf(v, xor_lane<32>(v)); return fmaxf(v, xor_lane<16>(v)); }

// "row layout": x[0..7] = channels 8p .. 8p+7, x[8..15] = channels 32+8p .. 32+8p+7 of frame t.
// x -> xhat in place (bias-free LayerNorm statistics over the row's 64 channels); returns rstd.
__device__ __forceinline__ float xhat16(float (&x)[16], float eps) {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s += x[i]; ss += x[i] * x[i]; }
    s = psum(s); ss = psum(ss);
    const float mean = s * (1.f / D);
    float var = ss * (1.f / D) - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (x[i] - mean) * rstd;
    return rstd;
}
// q/k-norm (y = round(xhat * scale)) followed by RoPE, rounding where the reference rounds (attn_rows.hpp: ln_rope_row)
__device__ __forceinline__ void ln_rope16(float (&x)[16], const float* __restrict__ sc, float eps, const float* __restrict__ cs,
                                          const float* __restrict__ sn) {
    xhat16(x, eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {                        // roundings in pairs: one v_cvt_pk_bf16_f32 per two values
        float lo = x[i] * sc[i], hi = x[i + 8] * sc[i + 8];
        round2<bf16_t>(lo, hi);
        float a = lo * cs[i], b = -hi * sn[i], c = hi * cs[i + 8], e = lo * sn[i + 8];
        round2<bf16_t>(a, b);
        round2<bf16_t>(c, e);
        float y0 = a + b, y1 = c + e;
        round2<bf16_t>(y0, y1);
        x[i] = y0; x[i + 8] = y1;
    }
}
// this lane's 16 table entries (scale, or the cos / sin row of its frame) in row layout
__device__ __forceinline__ void tab16(const float* __restrict__ row, int p, float (&r)[16]) {
    const float4* lo = reinterpret_cast<const float4*>(row + 8 * p);
    const float4* hi = reinterpret_cast<const float4*>(row + 32 + 8 * p);
    const float4 a = lo[0], b = lo[1], c = hi[0], e = hi[1];
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    r[8] = c.x; r[9] = c.y; r[10] = c.z; r[11] = c.w; r[12] = e.x; r[13] = e.y; r[14] = e.z; r[15] = e.w;
}
// load / LN+RoPE one row; returns the two MFMA fragments (k-steps 0 and 1)
__device__ __forceinline__ void norm_row(const bf16_t* __restrict__ g, int p, const float (&sc)[16], float eps, const float (&cs)[16],
                                         const float (&sn)[16], uint4& f0, uint4& f1) {
    float x[16];
    unpack8(*reinterpret_cast<const uint4*>(g + 8 * p), x);
    unpack8(*reinterpret_cast<const uint4*>(g + 32 + 8 * p), x + 8);
    ln_rope16(x, sc, eps, cs, sn);
    f0 = pack8(x); f1 = pack8(x + 8);
}
__device__ __forceinline__ void put_row(unsigned char* img, int t, int p, const uint4& f0, const uint4& f1) {
    *reinterpret_cast<uint4*>(img + t * PITCH + 16 * p) = f0;
    *reinterpret_cast<uint4*>(img + t * PITCH + 64 + 16 * p) = f1;
}
// wave-local LDS hand-off: this wave's ds_writes have landed before its own later reads (no workgroup barrier: images are per wave)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}
// Transposed A fragment of a row-major [16 rows][64 ch] image for output-channel tile `tile`, summing over rows 4p .. 4p+3:
// lane (p, i = l & 15) receives img[row 4p + j][channel 16 tile + i], j = 0..3
__device__ __forceinline__ s16x4v tr_rows(const unsigned char* img, int p, int i, int tile) {
    return tr4(img + (4 * p + (i >> 2)) * PITCH + 32 * tile + 8 * (i & 3));
}

__global__ __launch_bounds__(256) void tattn16_fwd_mfma(const bf16_t* __restrict__ qkv, int ld, bf16_t* __restrict__ out, int ldo,
                                                       float* __restrict__ lse, const float* __restrict__ q_scale,
                                                       const float* __restrict__ k_scale, const float* __restrict__ cosT,
                                                       const float* __restrict__ sinT, const uint8_t* __restrict__ mask, Dims d)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * IMG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = lane & 15, p = lane >> 4;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= d.items) return;                                  // wave-uniform
    const int a = (int)(item / d.heads), h = (int)(item % d.heads);
    const int HD = d.heads * D;
    const long tok = token_of(d, a, t);
    const bf16_t* g = qkv + tok * ld + h * D;
    unsigned char* vimg = smem + wave * IMG;

    const uint4 vlo = *reinterpret_cast<const uint4*>(g + 2 * HD + 8 * p);
    const uint4 vhi = *reinterpret_cast<const uint4*>(g + 2 * HD + 32 + 8 * p);
    float cs[16], sn[16], sc[16];
    tab16(cosT + t * D, p, cs);
    tab16(sinT + t * D, p, sn);
    uint4 q0, q1, k0, k1;
    tab16(q_scale, p, sc);
    norm_row(g, p, sc, d.eps, cs, sn, q0, q1);
    tab16(k_scale, p, sc);
    norm_row(g + HD, p, sc, d.eps, cs, sn, k0, k1);
    put_row(vimg, t, p, vlo, vhi);

    // S^T[key 4p + r][query t]
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    s = mfma32(frag8(k0), frag8(q0), s);
    s = mfma32(frag8(k1), frag8(q1), s);
    const float scale = 0.125f;                                    // 1 / sqrt(64)
    uint32_t mk = 0x01010101u;
    if (mask) mk = *reinterpret_cast<const uint32_t*>(mask + (long)(a / d.mask_div) * T + 4 * p);
    float e[4];
    float m = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        e[r] = ((mk >> (8 * r)) & 0xff) ? s[r] * scale : -3.0e38f;
        m = fmaxf(m, e[r]);
    }
    m = pmax(m);
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { e[r] = ((mk >> (8 * r)) & 0xff) ? __expf(e[r] - m) : 0.f; l += e[r]; }
    l = psum(l);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    const s16x4v pf = pack4(e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv);
    if (p == 0) lse[item * T + t] = l > 0.f ? m + __logf(l) : 0.f;

    wave_lds_fence();
    bf16_t* o = out + tok * ldo + h * D + 4 * p;
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mfma16(tr_rows(vimg, p, t, tile), pf, acc);         // O^T[channel 16 tile + 4p + r][query t]
        const s16x4v ob = pack4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<uint2*>(o + 16 * tile) = __builtin_bit_cast(uint2, ob);
    }
}

// "column layout" of the backward outputs: lane (p, i) holds channels 16 tile + 4p + r (tile, r = 0..3) of frame i.
// g: dy w.r.t. the RoPE output -> dx w.r.t. the raw row, through RoPE^T and the bias-free LayerNorm; returns in `contrib` this row's
// scale-gradient contribution dy_ln * xhat.  raw: the row's raw values in the same layout.
__device__ __forceinline__ void rope_ln_bwd_cols(float (&g)[4][4], const float (&raw)[4][4], int p, int frame, float eps,
                                                 const float* __restrict__ scale, const float* __restrict__ cosT,
                                                 const float* __restrict__ sinT, float (&contrib)[4][4])
{
    float xh[4][4];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int tl = 0; tl < 4; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s += raw[tl][r]; ss += raw[tl][r] * raw[tl][r]; }
    s = psum(s); ss = psum(ss);
    const float mean = s * (1.f / D);
    float var = ss * (1.f / D) - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int tl = 0; tl < 4; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) xh[tl][r] = (raw[tl][r] - mean) * rstd;
    // RoPE transpose on the pair (c, c + 32) = (tile, tile + 2), same r
    const float* cr = cosT + frame * D;
    const float* sr = sinT + frame * D;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const float4 cl = *reinterpret_cast<const float4*>(cr + 16 * tl + 4 * p), ch = *reinterpret_cast<const float4*>(cr + 32 + 16 * tl + 4 * p);
        const float4 sl = *reinterpret_cast<const float4*>(sr + 16 * tl + 4 * p), sh = *reinterpret_cast<const float4*>(sr + 32 + 16 * tl + 4 * p);
        const float c_lo[4] = {cl.x, cl.y, cl.z, cl.w}, c_hi[4] = {ch.x, ch.y, ch.z, ch.w};
        const float s_lo[4] = {sl.x, sl.y, sl.z, sl.w}, s_hi[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lo = g[tl][r], hi = g[tl + 2][r];
            g[tl][r] = lo * c_lo[r] + hi * s_hi[r];
            g[tl + 2][r] = hi * c_hi[r] - lo * s_lo[r];
        }
    }
    float s1 = 0.f, s2 = 0.f;
    float scv[4][4];
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
        const float4 v = *reinterpret_cast<const float4*>(scale + 16 * tl + 4 * p);
        scv[tl][0] = v.x; scv[tl][1] = v.y; scv[tl][2] = v.z; scv[tl][3] = v.w;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float dxh = g[tl][r] * scv[tl][r]; s1 += dxh; s2 += dxh * xh[tl][r]; }
    }
    s1 = psum(s1) * (1.f / D); s2 = psum(s2) * (1.f / D);
#pragma unroll
    for (int tl = 0; tl < 4; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dy = g[tl][r], x = xh[tl][r];
            g[tl][r] = rstd * (dy * scv[tl][r] - s1 - x * s2);
            contrib[tl][r] = dy * x;
        }
}

__device__ __forceinline__ void load_cols(const bf16_t* __restrict__ row, int p, float (&v)[4][4]) {
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
        const uint2 u = *reinterpret_cast<const uint2*>(row + 16 * tl + 4 * p);
        v[tl][0] = __uint_as_float(u.x << 16); v[tl][1] = __uint_as_float(u.x & 0xffff0000u);
        v[tl][2] = __uint_as_float(u.y << 16); v[tl][3] = __uint_as_float(u.y & 0xffff0000u);
    }
}
__device__ __forceinline__ void store_cols(bf16_t* __restrict__ row, int p, const float (&v)[4][4]) {
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
        const s16x4v b = pack4(v[tl][0], v[tl][1], v[tl][2], v[tl][3]);
        *reinterpret_cast<uint2*>(row + 16 * tl + 4 * p) = __builtin_bit_cast(uint2, b);
    }
}

// part: fp32 (gridDim.x, 2 * D): one row per workgroup (4 items), [dq_scale | dk_scale].
__global__ __launch_bounds__(256, 4) void tattn16_bwd_mfma(const bf16_t* __restrict__ qkv, int ld, const bf16_t* __restrict__ out, int ldo,
                                                          const bf16_t* __restrict__ dout, int lddo, const float* __restrict__ lse,
                                                          bf16_t* __restrict__ dqkv, int lddq, const float* __restrict__ q_scale,
                                                          const float* __restrict__ k_scale, const float* __restrict__ cosT,
                                                          const float* __restrict__ sinT, const uint8_t* __restrict__ mask,
                                                          float* __restrict__ part, Dims d)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * (3 * IMG + 64)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = lane & 15, p = lane >> 4;
    unsigned char* base = smem + wave * (3 * IMG + 64);
    unsigned char* kimg = base;                   // K' rows
    unsigned char* qimg = base + IMG;             // Q' rows
    unsigned char* gimg = base + 2 * IMG;         // dO rows
    float* dlds = reinterpret_cast<float*>(base + 3 * IMG);       // delta[16]
    const int HD = d.heads * D;
    const float scale = 0.125f;
    float accq[4][4], acck[4][4];                 // this item's scale-gradient contributions (column layout)
#pragma unroll
    for (int tl = 0; tl < 4; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) { accq[tl][r] = 0.f; acck[tl][r] = 0.f; }

    const long item = (long)blockIdx.x * 4 + wave;
    if (item < d.items) {                                          // wave-uniform: every lane of a working wave is active (tr reads)
        const float* ct = cosT; const float* st_ = sinT; const float* qsp = q_scale; const float* ksp = k_scale;
        const int a = (int)(item / d.heads), h = (int)(item % d.heads);
        const long tok = token_of(d, a, t);
        const bf16_t* g = qkv + tok * ld + h * D;
        uint4 q0, q1, k0, k1;
        {
            float cs[16], sn[16], sc[16];
            tab16(ct + t * D, p, cs);
            tab16(st_ + t * D, p, sn);
            tab16(qsp, p, sc);
            norm_row(g, p, sc, d.eps, cs, sn, q0, q1);
            tab16(ksp, p, sc);
            norm_row(g + HD, p, sc, d.eps, cs, sn, k0, k1);
        }
        const uint4 v0 = *reinterpret_cast<const uint4*>(g + 2 * HD + 8 * p), v1 = *reinterpret_cast<const uint4*>(g + 2 * HD + 32 + 8 * p);
        const bf16_t* go = dout + tok * lddo + h * D;
        const uint4 g0 = *reinterpret_cast<const uint4*>(go + 8 * p), g1 = *reinterpret_cast<const uint4*>(go + 32 + 8 * p);
        const bf16_t* oo = out + tok * ldo + h * D;
        float delta;
        {
            float a16[16], b16[16];
            unpack8(g0, a16); unpack8(g1, a16 + 8);
            unpack8(*reinterpret_cast<const uint4*>(oo + 8 * p), b16); unpack8(*reinterpret_cast<const uint4*>(oo + 32 + 8 * p), b16 + 8);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s += a16[i] * b16[i];
            delta = psum(s);                                       // delta[frame t], in all four lane groups
        }
        wave_lds_fence();                                          // the previous item's reads of the images are done (WAR)
        put_row(kimg, t, p, k0, k1);
        put_row(qimg, t, p, q0, q1);
        put_row(gimg, t, p, g0, g1);
        if (p == 0) dlds[t] = delta;
        const float lse_q = lse[item * T + t];
        uint32_t mk = 0x01010101u, mcol = 1u;
        if (mask) {
            const uint8_t* mrow = mask + (long)(a / d.mask_div) * T;
            mk = *reinterpret_cast<const uint32_t*>(mrow + 4 * p);
            mcol = mrow[t];
        }
        bf16_t* dg = dqkv + tok * lddq + h * D;
        // ---- score tiles in both orientations (8 MFMAs), after which the row fragments are dead:
        //      columns = queries: S^T = K' Q'^T, dP^T = V dO^T -> dS^T;   columns = keys: S = Q' K'^T, dP = dO V^T -> P, dS
        s16x4v dsf, pf, dsf2;
        {
            f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            st = mfma32(frag8(k0), frag8(q0), st);
            st = mfma32(frag8(k1), frag8(q1), st);
            dpt = mfma32(frag8(v0), frag8(g0), dpt);
            dpt = mfma32(frag8(v1), frag8(g1), dpt);
            sq = mfma32(frag8(q0), frag8(k0), sq);
            sq = mfma32(frag8(q1), frag8(k1), sq);
            dp = mfma32(frag8(g0), frag8(v0), dp);
            dp = mfma32(frag8(g1), frag8(v1), dp);
            float dst[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = ((mk >> (8 * r)) & 0xff) ? __expf(st[r] * scale - lse_q) : 0.f;
                dst[r] = pr * (dpt[r] - delta) * scale;            // dS^T[key 4p + r][query t]
            }
            dsf = pack4(dst[0], dst[1], dst[2], dst[3]);
            wave_lds_fence();                                      // images (and delta) written
            const float4 lse4 = *reinterpret_cast<const float4*>(lse + item * T + 4 * p);       // queries 4p .. 4p+3
            const float4 del4 = *reinterpret_cast<const float4*>(dlds + 4 * p);
            const float lq[4] = {lse4.x, lse4.y, lse4.z, lse4.w}, dl[4] = {del4.x, del4.y, del4.z, del4.w};
            float pq[4], dsq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = mcol ? __expf(sq[r] * scale - lq[r]) : 0.f;                    // P[query 4p + r][key t]
                pq[r] = pr;
                dsq[r] = pr * (dp[r] - dl[r]) * scale;
            }
            pf = pack4(pq[0], pq[1], pq[2], pq[3]);
            dsf2 = pack4(dsq[0], dsq[1], dsq[2], dsq[3]);
        }
        // ---- dQ'^T = K'^T dS^T, then through RoPE and q-norm in column layout (the raw row is re-read in that layout: L1 / L2 hit)
        {
            float dq[4][4];
#pragma unroll
            for (int tile = 0; tile < 4; ++tile) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = mfma16(tr_rows(kimg, p, t, tile), dsf, acc);    // dQ'^T[channel 16 tile + 4p + r][query t]
#pragma unroll
                for (int r = 0; r < 4; ++r) dq[tile][r] = acc[r];
            }
            float raw[4][4], contrib[4][4];
            load_cols(g, p, raw);
            rope_ln_bwd_cols(dq, raw, p, t, d.eps, qsp, ct, st_, contrib);
            store_cols(dg, p, dq);
#pragma unroll
            for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                for (int r = 0; r < 4; ++r) accq[tl][r] += contrib[tl][r];
        }
        // ---- dV^T = dO^T P, dK'^T = Q'^T dS
        {
            float dv[4][4];
#pragma unroll
            for (int tile = 0; tile < 4; ++tile) {
                f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
                a1 = mfma16(tr_rows(gimg, p, t, tile), pf, a1);   // dV^T[channel][key t] = sum_q dO[q][channel] P[q][key]
#pragma unroll
                for (int r = 0; r < 4; ++r) dv[tile][r] = a1[r];
            }
            store_cols(dg + 2 * HD, p, dv);
        }
        {
            float dk[4][4];
#pragma unroll
            for (int tile = 0; tile < 4; ++tile) {
                f32x4 a2 = {0.f, 0.f, 0.f, 0.f};
                a2 = mfma16(tr_rows(qimg, p, t, tile), dsf2, a2); // dK'^T[channel][key t] = sum_q Q'[q][channel] dS[q][key]
#pragma unroll
                for (int r = 0; r < 4; ++r) dk[tile][r] = a2[r];
            }
            float raw[4][4], contrib[4][4];
            load_cols(g + HD, p, raw);
            rope_ln_bwd_cols(dk, raw, p, t, d.eps, ksp, ct, st_, contrib);
            store_cols(dg + HD, p, dk);
#pragma unroll
            for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                for (int r = 0; r < 4; ++r) acck[tl][r] += contrib[tl][r];
        }
    }
    // scale gradients: sum over the 16 frames (DPP row adds), then over the workgroup's four items through LDS: one row per workgroup
    __syncthreads();                                               // every wave is done with its images
    float* red = reinterpret_cast<float*>(smem);                   // [wave][2 * D]
#pragma unroll
    for (int tl = 0; tl < 4; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = butterfly_sum<8, 1>(accq[tl][r]), b = butterfly_sum<8, 1>(acck[tl][r]);
            if (t == 0) { red[wave * 2 * D + 16 * tl + 4 * p + r] = a; red[wave * 2 * D + D + 16 * tl + 4 * p + r] = b; }
        }
    __syncthreads();
    if (threadIdx.x < 2 * D)
        part[(long)blockIdx.x * 2 * D + threadIdx.x] = (red[threadIdx.x] + red[2 * D + threadIdx.x]) + (red[4 * D + threadIdx.x] + red[6 * D + threadIdx.x]);
}

int g_enable = 1;

inline int bwd_blocks(long items) { return (int)((items + 3) / 4); }

bool shape_ok(int T_, int D_, int ld, int ldo, int dtype) {
    return g_enable && dtype == VVAE_DT_BF16 && T_ == T && D_ == D && ld % 8 == 0 && ldo % 8 == 0;
}

}  // namespace tmfma

// Test / tuning hook: 0 routes T = 16, head_dim 64, bf16 temporal attention back to the VALU kernels of attn_temporal_fast.hip.
extern "C" int vvae_temporal_attn_mfma_enable(int on)
{
    tmfma::g_enable = on ? 1 : 0;
    return 0;
}

// internal entry points used by attn_temporal_fast.hip's dispatch (declared there)
int tmfma_supported(int T, int D, int ld, int ldo, int dtype) { return tmfma::shape_ok(T, D, ld, ldo, dtype) ? 1 : 0; }
int tmfma_bwd_rows(long items) { return tmfma::bwd_blocks(items); }

int tmfma_fwd(const void* qkv, int ld, void* out, int ldo, float* lse, const float* qs, const float* ks, const float* cosT, const float* sinT,
              const uint8_t* mask, int mask_div, int inner, int A, int heads, float eps, hipStream_t s)
{
    tmfma::Dims d{A, heads, mask_div, inner, eps, (long)A * heads};
    if (mask && ((uintptr_t)mask % 4)) return VVAE_ERR_BAD_ARG;
    hipLaunchKernelGGL(tmfma::tattn16_fwd_mfma, dim3((unsigned)((d.items + 3) / 4)), dim3(256), 0, s, (const bf16_t*)qkv, ld, (bf16_t*)out, ldo, lse,
                       qs, ks, cosT, sinT, mask, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

int tmfma_bwd(const void* qkv, int ld, const void* out, int ldo, const void* dout, int lddo, const float* lse, void* dqkv, int lddq,
              const float* qs, const float* ks, const float* cosT, const float* sinT, const uint8_t* mask, int mask_div, int inner,
              float* part, int A, int heads, float eps, hipStream_t s)
{
    tmfma::Dims d{A, heads, mask_div, inner, eps, (long)A * heads};
    if (mask && ((uintptr_t)mask % 4)) return VVAE_ERR_BAD_ARG;
    hipLaunchKernelGGL(tmfma::tattn16_bwd_mfma, dim3((unsigned)tmfma::bwd_blocks(d.items)), dim3(256), 0, s, (const bf16_t*)qkv, ld,
                       (const bf16_t*)out, ldo, (const bf16_t*)dout, lddo, lse, (bf16_t*)dqkv, lddq, qs, ks, cosT, sinT, mask, part, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}
