// Spatial attention of FactoredAttention (sequence = the h*w patches of one frame, reference train/layers.py:153-170 with the
// mask-free call of :217-221), head_dim 64, S <= 256: q/k-norm + RoPE + softmax(Q K^T / sqrt(D)) V in ONE kernel per direction.
//
// One workgroup (4 waves) per (sequence, head).  K' = rope(k_norm(K)) and V live in LDS for the workgroup's life (64 KB: two
// workgroups per CU); a wave owns 32-query blocks.  Everything between the qkv buffer and the output stays on chip:
//   * S^T = K' Q'^T with v_mfma_f32_32x32x16_bf16: rows = keys, columns = queries, so a lane holds, for ITS query, 16 keys per
//     32-key tile -- the softmax reductions are lane-local plus one cross-lane add, and the exponentiated tile IS the
//     column operand of the next product (O^T = V^T P^T) as it stands: registers 0-7 / 8-15 of a tile are two k16 steps whose
//     key order (0-3, 8-11 | 4-7, 12-15 for the two lane halves) the V^T fragment reads reproduce.  P never touches LDS.
//   * V^T fragments come from the row-major V image through ds_read_b64_tr_b16; K' fragments are plain ds_read_b128.  One XOR
//     swizzle of the 16-byte chunk index, g(row) = ((row>>1)&1)<<2 | ((row>>3)&1)<<1 | ((row>>2)&1), is conflict-free for BOTH
//     read shapes on 128-byte rows (the backward kernel reads every image both ways).
//   * the Q fragment layout keeps a query row in two lanes (l, l^32), each with d = 16 ks + 8 kh + e: the rotate-half partner
//     d + 32 is the same lane's ks + 2, so LayerNorm + RoPE run on the fragment registers (one cross-lane add for the statistics).
// Forward saves the base-2 log-sum-exp per query for the backward pass.
#include "attn_rows.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4v __attribute__((ext_vector_type(4)));

constexpr int SD = 64;                          // head_dim
constexpr int SROW = SD * 2;                    // bytes per LDS row

struct SAttnDims { int A, S, H; float eps; };

// chunk swizzle of an LDS image with 128-byte rows (see header comment)
__device__ __forceinline__ int gsw(int row) { return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1); }

__device__ __forceinline__ float xor32(float v) { return xor_lane<32>(v); }          // v_permlane32_swap: VALU only (common.hpp)

// Fragment-layout row: lane (j = lane & 31, kh = lane >> 5) holds x[ks][e] = channel 16 ks + 8 kh + e of row j.
// q/k-norm (bias-free LayerNorm, y = round(xhat * scale)) followed by RoPE, in place; same rounding points as attn_rows.hpp.
__device__ __forceinline__ void ln_rope_frag(float (&x)[4][8], int kh, const float* __restrict__ scale, float eps,
                                             const float* __restrict__ cosr, const float* __restrict__ sinr)
{
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s += x[ks][e]; ss += x[ks][e] * x[ks][e]; }
    s += xor32(s); ss += xor32(ss);
    const float mean = s / SD;
    float var = ss / SD - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) x[ks][e] = round_to<bf16_t>((x[ks][e] - mean) * rstd * scale[16 * ks + 8 * kh + e]);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int cl = 16 * ks + 8 * kh + e, ch = cl + 32;
            const float lo = x[ks][e], hi = x[ks + 2][e];
            x[ks][e] = round_to<bf16_t>(round_to<bf16_t>(lo * cosr[cl]) + round_to<bf16_t>(-hi * sinr[cl]));
            x[ks + 2][e] = round_to<bf16_t>(round_to<bf16_t>(hi * cosr[ch]) + round_to<bf16_t>(lo * sinr[ch]));
        }
}

__device__ __forceinline__ bf16x8 pack8(const float (&v)[8])
{
    s16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (short)f2bf(v[e]);
    return __builtin_bit_cast(bf16x8, r);
}

// transposed fragment: 8 rows (two groups of 4, `gap` rows apart) x the lane's channel, rows given by the per-lane address
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* p0, const unsigned char* p1)
{
    typedef __attribute__((address_space(3))) s16x4v* lds_ptr;
    const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
    const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p1);
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// byte offset of (row, 16-byte chunk c) in a swizzled image
__device__ __forceinline__ int img_off(int row, int c) { return row * SROW + ((c ^ gsw(row)) << 4); }
// byte offset of the 8 bytes a transposed read takes at (row, 16-channel block cb (0..3), 4-channel group p)
__device__ __forceinline__ int img_off_tr(int row, int cb, int p) { return row * SROW + ((((2 * cb + (p >> 1)) ^ gsw(row))) << 4) + (p & 1) * 8; }

// Lane-constant parts of the fragment addresses (the swizzle term of a row depends on row bits 1..3 only, which a 32-row block
// offset never touches): a row fragment is img + blk * 32 * SROW + row[ks]; a transposed fragment of the 16-row group u of
// block blk and channel tile dt is the pair img + (blk * 32 + 16 u) * SROW + tr[dt][0 | 1] (token rows +0..3 and +8..11).
struct FragAddr {
    int row[4], tr[2][2];
    __device__ __forceinline__ FragAddr(int lane) {
        const int j = lane & 31, kh = lane >> 5, p4 = lane & 3, qr = (lane >> 2) & 3, mh = (lane >> 4) & 1;
        const int gj = gsw(j), gl = (((qr >> 1) & 1) << 2) | kh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) row[ks] = j * SROW + (((2 * ks + kh) ^ gj) << 4);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int x = 0; x < 2; ++x)
                tr[dt][x] = (4 * kh + qr + 8 * x) * SROW + (((2 * (2 * dt + mh) + (p4 >> 1)) ^ (gl | (x << 1))) << 4) + (p4 & 1) * 8;
    }
    __device__ __forceinline__ bf16x8 rowfrag(const unsigned char* img, int blk, int ks) const {
        return *reinterpret_cast<const bf16x8*>(img + blk * 32 * SROW + row[ks]);
    }
    __device__ __forceinline__ bf16x8 trfrag(const unsigned char* img, int blk, int u, int dt) const {
        const unsigned char* b = img + (blk * 32 + 16 * u) * SROW;
        return tr_pair(b + tr[dt][0], b + tr[dt][1]);
    }
};

// Stage rows [0, S) of one head into a swizzled LDS image: K path applies k_norm + RoPE, V path copies.  256 threads, 4 lanes per
// row (attn_rows.hpp slices: lane p owns channels [8p, 8p+8) and [32+8p, 32+8p+8) = chunks p and 4+p).
template <bool NORM>
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ src, long tok0, int ld, unsigned char* img, int S,
                                           const float* __restrict__ scale, float eps, const float* __restrict__ cosT,
                                           const float* __restrict__ sinT)
{
    const int p = threadIdx.x & 3;
    for (int row = threadIdx.x >> 2; row < S; row += (int)(blockDim.x >> 2)) {
        float x[16];
        load_row<bf16_t, SD, 4>(src + (tok0 + row) * ld, p, x);
        if (NORM) ln_rope_row<bf16_t, SD, 4>(x, p, scale, eps, cosT + (long)row * SD, sinT + (long)row * SD);
        float lo[8], hi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { lo[e] = x[e]; hi[e] = x[8 + e]; }
        VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(img + img_off(row, p)), lo);
        VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(img + img_off(row, 4 + p)), hi);
    }
}

// Q fragments of the 32-query block starting at token tok0 + q0: raw row -> q_norm -> RoPE -> bf16 operand registers
__device__ __forceinline__ void load_q_frags(const bf16_t* __restrict__ qrow, int kh, int pos, const float* __restrict__ q_scale, float eps,
                                             const float* __restrict__ cosT, const float* __restrict__ sinT, bf16x8 (&qf)[4])
{
    float x[4][8];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) VecIO<bf16_t, 8>::load(qrow + 16 * ks + 8 * kh, x[ks]);
    ln_rope_frag(x, kh, q_scale, eps, cosT + (long)pos * SD, sinT + (long)pos * SD);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = pack8(x[ks]);
}

template <int NKB>          // key blocks of 32: S = 32 * NKB
__global__ __launch_bounds__(256, 2) void sattn_fwd_kernel(const bf16_t* __restrict__ qkv, int ld, bf16_t* __restrict__ out, int ldo,
                                                           float* __restrict__ lse2, const float* __restrict__ q_scale,
                                                           const float* __restrict__ k_scale, const float* __restrict__ cosT,
                                                           const float* __restrict__ sinT, SAttnDims d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int S = 32 * NKB;
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + S * SROW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a = blockIdx.x / d.H, h = blockIdx.x - a * d.H;
    const int HD = d.H * SD;
    const long tok0 = (long)a * S;
    const bf16_t* base = qkv + h * SD;

    stage_rows<true>(base + HD, tok0, ld, Ks, S, k_scale, d.eps, cosT, sinT);
    stage_rows<false>(base + 2 * HD, tok0, ld, Vs, S, nullptr, 0.f, nullptr, nullptr);
    __syncthreads();

    const int j = lane & 31, kh = lane >> 5;
    const FragAddr fa(lane);
    const float c2 = rsqrtf((float)SD) * 1.44269504088896341f;          // softmax scale in the exp2 domain

    for (int qb = wave; qb < NKB; qb += 4) {
        const int qrow = qb * 32 + j;
        bf16x8 qf[4];
        load_q_frags(base + (tok0 + qrow) * ld, kh, qrow, q_scale, d.eps, cosT, sinT, qf);

        // Online softmax over groups of <= 4 key tiles (64 score registers live at a time).
        // S^T tile: s[g][r] = score(key 32 kb + 8 (r/4) + 4 kh + r%4, query j).  O^T[d][query] = sum over keys V[key][d] P[query][key];
        // the exponentiated tile is consumed at once; fragment u of tile kb covers keys 32 kb + 16 u + {4 kh + 0..3, 8 + 4 kh + 0..3}.
        constexpr int G = NKB < 4 ? NKB : 4;
        float m = -3.0e38f, l = 0.f;
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
#pragma unroll 1
        for (int kb0 = 0; kb0 < NKB; kb0 += G) {
            f32x16 s[G];
            float mg = m;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (kb0 + g < NKB) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[g][e] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        s[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.rowfrag(Ks, kb0 + g, ks), qf[ks], s[g], 0, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) mg = fmaxf(mg, s[g][e]);
                }
            }
            mg = fmaxf(mg, xor32(mg));
            if (kb0 > 0) {                                          // rescale what the earlier groups accumulated
                const float alpha = exp2f((m - mg) * c2);
                l *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
            }
            m = mg;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (kb0 + g < NKB) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        float pv[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) { pv[e] = exp2f((s[g][8 * u + e] - m) * c2); l += pv[e]; }
                        const bf16x8 pf = pack8(pv);                 // the reference multiplies V by probabilities in the value dtype
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
                            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.trfrag(Vs, kb0 + g, u, dt), pf, o[dt], 0, 0, 0);
                    }
                }
            }
        }
        l += xor32(l);
        const float inv = 1.f / l;
        bf16_t* orow = out + (tok0 + qrow) * ldo + h * SD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float v4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = o[dt][rg * 4 + e] * inv;
                VecIO<bf16_t, 4>::store(orow + 32 * dt + 8 * rg + 4 * kh, v4);
            }
        if (kh == 0) lse2[((long)a * d.H + h) * S + qrow] = m * c2 + log2f(l);
    }
}

// ------------------------------------------------------------------------------------------------------------------ backward
// Accumulator-layout row: lane (j = lane & 31, kh = lane >> 5) holds g[dt][r] = channel 32 dt + 8 (r/4) + 4 kh + r%4 of row j
// (what a 32x32 MFMA leaves when rows = channels, columns = tokens).  The rotate-half partner (dt = 0 <-> 1, same r) is in the
// same lane.  g: gradient w.r.t. the rotated row -> gradient w.r.t. the raw row; xh: raw row -> this row's contribution
// dy_ln * xhat to the q/k-norm scale gradient (same algebra and rounding points as rope_ln_bwd_row in attn_rows.hpp).
__device__ __forceinline__ void rope_ln_bwd_acc(float (&g)[2][16], float (&xh)[2][16], int kh, const float* __restrict__ scale, float eps,
                                                const float* __restrict__ cosr, const float* __restrict__ sinr)
{
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s += xh[dt][r]; ss += xh[dt][r] * xh[dt][r]; }
    s += xor32(s); ss += xor32(ss);
    const float mean = s / SD;
    float var = ss / SD - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = rsqrtf(var + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cl = 8 * (r >> 2) + 4 * kh + (r & 3), ch = cl + 32;
        const float lo = g[0][r], hi = g[1][r];
        g[0][r] = lo * cosr[cl] + hi * sinr[ch];
        g[1][r] = hi * cosr[ch] - lo * sinr[cl];
        xh[0][r] = (xh[0][r] - mean) * rstd;
        xh[1][r] = (xh[1][r] - mean) * rstd;
        const float d0 = g[0][r] * scale[cl], d1 = g[1][r] * scale[ch];
        s1 += d0 + d1; s2 += d0 * xh[0][r] + d1 * xh[1][r];
    }
    s1 += xor32(s1); s2 += xor32(s2);
    s1 /= SD; s2 /= SD;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cl = 8 * (r >> 2) + 4 * kh + (r & 3), ch = cl + 32;
        const float dy0 = g[0][r], dy1 = g[1][r], x0 = xh[0][r], x1 = xh[1][r];
        g[0][r] = rstd * (dy0 * scale[cl] - s1 - x0 * s2);
        g[1][r] = rstd * (dy1 * scale[ch] - s1 - x1 * s2);
        xh[0][r] = dy0 * x0;
        xh[1][r] = dy1 * x1;
    }
}

// accumulator-layout row <-> global memory (8-byte pieces: 4 consecutive channels)
__device__ __forceinline__ void load_acc_row(const bf16_t* __restrict__ row, int kh, float (&x)[2][16])
{
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            float v4[4];
            VecIO<bf16_t, 4>::load(row + 32 * dt + 8 * rg + 4 * kh, v4);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[dt][rg * 4 + e] = v4[e];
        }
}
__device__ __forceinline__ void store_acc_row(bf16_t* __restrict__ row, int kh, const float (&x)[2][16])
{
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            float v4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = x[dt][rg * 4 + e];
            VecIO<bf16_t, 4>::store(row + 32 * dt + 8 * rg + 4 * kh, v4);
        }
}

// part: (A*H, 2, 64) fp32 = per-(sequence, head) partial of [dq_scale | dk_scale] (summed by the caller).
// One workgroup of 8 waves per (sequence, head); Q' = rope(q_norm(Q)), K' = rope(k_norm(K)), V and dO are staged ONCE as swizzled LDS
// images (128 KB at S = 256) and every operand of the seven products is a row or a transposed fragment of one of them.
// Phase A: every wave owns a 32-key tile and walks all queries: dV^T, dK^T (scores as [query][key], so the contraction over
// queries is the register index).  Phase B: every wave owns a 32-query tile and walks all keys: dQ^T (scores as [key][query]).
// Recomputing the score tile in both orientations costs two extra products out of seven and saves every transpose through LDS.
template <int NKB>
__global__ __launch_bounds__(512) void sattn_bwd_kernel(const bf16_t* __restrict__ qkv, int ld, const bf16_t* __restrict__ out, int ldo,
                                                           const bf16_t* __restrict__ dout, int lddo, const float* __restrict__ lse2,
                                                           bf16_t* __restrict__ dqkv, int lddq, const float* __restrict__ q_scale,
                                                           const float* __restrict__ k_scale, const float* __restrict__ cosT,
                                                           const float* __restrict__ sinT, float* __restrict__ part, SAttnDims d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int S = 32 * NKB;
    unsigned char* Qs = smem;
    unsigned char* Ks = smem + S * SROW;
    unsigned char* Vs = smem + 2 * S * SROW;
    unsigned char* Gs = smem + 3 * S * SROW;          // dO
    float* lseS = reinterpret_cast<float*>(smem + 4 * S * SROW);
    float* delS = lseS + S;
    float* red = delS + S;                            // [8 waves][2][64] scale-gradient partials
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a = blockIdx.x / d.H, h = blockIdx.x - a * d.H;
    const int HD = d.H * SD;
    const long tok0 = (long)a * S;
    const bf16_t* base = qkv + h * SD;
    const bf16_t* gbase = dout + h * SD;
    const int j = lane & 31, kh = lane >> 5;
    const FragAddr fa(lane);
    const float sm_scale = rsqrtf((float)SD);
    const float c2 = sm_scale * 1.44269504088896341f;

    // ---- staging: Q' and K' (norm + RoPE), V, dO, delta = rowsum(dO * O), lse
    stage_rows<true>(base, tok0, ld, Qs, S, q_scale, d.eps, cosT, sinT);
    stage_rows<true>(base + HD, tok0, ld, Ks, S, k_scale, d.eps, cosT, sinT);
    stage_rows<false>(base + 2 * HD, tok0, ld, Vs, S, nullptr, 0.f, nullptr, nullptr);
    {                                                 // dO image + delta in one pass over dO
        const int p = threadIdx.x & 3;
        for (int row = threadIdx.x >> 2; row < S; row += 128) {
            float go[16], oo[16];
            load_row<bf16_t, SD, 4>(gbase + (tok0 + row) * lddo, p, go);
            load_row<bf16_t, SD, 4>(out + (tok0 + row) * ldo + h * SD, p, oo);
            float dl = 0.f, lo[8], hi[8];
#pragma unroll
            for (int i = 0; i < 16; ++i) dl += go[i] * oo[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) { lo[e] = go[e]; hi[e] = go[8 + e]; }
            VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(Gs + img_off(row, p)), lo);
            VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(Gs + img_off(row, 4 + p)), hi);
            dl = lpr_sum<4>(dl);
            if (p == 0) { delS[row] = dl; lseS[row] = lse2[((long)a * d.H + h) * S + row]; }
        }
    }
    // scale-gradient contributions (accumulator layout: 32 channels per lane) are summed over the 32 rows a half-wave holds with
    // DPP butterflies (xor 1, 2, then the half-row and row mirrors; one cross-row shuffle) at the end of each tile and added to
    // this wave's slots in LDS: nothing of it stays live across the tile loops
    auto add_scale_grad = [&](const float (&ds)[2][16], int which) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = ds[dt][r];
                t += dpp_xor1(t);
                t += dpp_xor2(t);
                t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0x141, 0xf, 0xf, true));      // row_half_mirror
                t += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0x140, 0xf, 0xf, true));      // row_mirror
                t += xor_lane<16>(t);
                if (j == 0) red[(wave * 2 + which) * SD + 32 * dt + 8 * (r >> 2) + 4 * kh + (r & 3)] += t;
            }
    };
    for (int i = threadIdx.x; i < 16 * SD; i += 512) red[i] = 0.f;
    __syncthreads();

    // ---- phase A: dV, dK for the wave's key tiles
    for (int kt = wave; kt < NKB; kt += 8) {
        const int key = kt * 32 + j;
        bf16x8 kc[4], vc[4];                          // the tile's own K' / V rows: column operands, constant over the query loop
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kc[ks] = fa.rowfrag(Ks, kt, ks); vc[ks] = fa.rowfrag(Vs, kt, ks); }
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
            // register r <-> query 32 qb + 8 (r/4) + 4 kh + r%4; the two register halves are the two k16 steps of the next products
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float p8[8], s8[8];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const float4 l4 = *reinterpret_cast<const float4*>(lseS + qb * 32 + 8 * (2 * u + hh) + 4 * kh);
                    const float4 d4 = *reinterpret_cast<const float4*>(delS + qb * 32 + 8 * (2 * u + hh) + 4 * kh);
                    const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = exp2f(s[8 * u + 4 * hh + e] * c2 - lq[e]);
                        p8[4 * hh + e] = pe;
                        s8[4 * hh + e] = pe * (dp[8 * u + 4 * hh + e] - dq4[e]);
                    }
                }
                const bf16x8 pf = pack8(p8), sf = pack8(s8);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.trfrag(Gs, qb, u, dt), pf, dv[dt], 0, 0, 0);   // dV^T[d][key] += dO^T[d][q] P[q][key]
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.trfrag(Qs, qb, u, dt), sf, dk[dt], 0, 0, 0);   // dK^T[d][key] += Q'^T[d][q] dS[q][key]
                }
            }
        }
        // lane <-> key, registers <-> channels: dv leaves as it is; dk goes back through RoPE and k_norm
        float g[2][16], xh[2][16];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[dt][r] = dv[dt][r];
        store_acc_row(dqkv + (tok0 + key) * lddq + 2 * HD + h * SD, kh, g);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[dt][r] = dk[dt][r] * sm_scale;
        load_acc_row(base + HD + (tok0 + key) * ld, kh, xh);
        rope_ln_bwd_acc(g, xh, kh, k_scale, d.eps, cosT + (long)key * SD, sinT + (long)key * SD);
        store_acc_row(dqkv + (tok0 + key) * lddq + HD + h * SD, kh, g);
        add_scale_grad(xh, 1);
    }

    // ---- phase B: dQ for the wave's query tiles
    for (int qt = wave; qt < NKB; qt += 8) {
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
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float s8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) s8[e] = exp2f(s[8 * u + e] * c2 - lq) * (dp[8 * u + e] - dl);
                const bf16x8 sf = pack8(s8);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.trfrag(Ks, kb, u, dt), sf, dq[dt], 0, 0, 0);   // dQ^T[d][query] += K'^T[d][key] dS^T[key][query]
            }
        }
        float g[2][16], xh[2][16];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[dt][r] = dq[dt][r] * sm_scale;
        load_acc_row(base + (tok0 + qrow) * ld, kh, xh);
        rope_ln_bwd_acc(g, xh, kh, q_scale, d.eps, cosT + (long)qrow * SD, sinT + (long)qrow * SD);
        store_acc_row(dqkv + (tok0 + qrow) * lddq + h * SD, kh, g);
        add_scale_grad(xh, 0);
    }

    __syncthreads();
    if (threadIdx.x < 2 * SD) {
        const int t = threadIdx.x;
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += red[w * 2 * SD + t];
        part[(long)blockIdx.x * 2 * SD + t] = tot;
    }
}

bool sattn_ok(int S, int D, int dtype) { return D == SD && dtype == VVAE_DT_BF16 && S >= 32 && S <= 256 && S % 32 == 0; }

template <int NKB>
int launch_sattn_bwd(const void* qkv, int ld, const void* out, int ldo, const void* dout, int lddo, const float* lse2, void* dqkv, int lddq,
                     const float* qs, const float* ks, const float* cosT, const float* sinT, float* part, SAttnDims d, hipStream_t s)
{
    constexpr int lds = 4 * 32 * NKB * SROW + 2 * 32 * NKB * 4 + 16 * SD * 4;
    auto k = sattn_bwd_kernel<NKB>;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(d.A * d.H), dim3(512), lds, s, (const bf16_t*)qkv, ld, (const bf16_t*)out, ldo, (const bf16_t*)dout, lddo, lse2,
                       (bf16_t*)dqkv, lddq, qs, ks, cosT, sinT, part, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

template <int NKB>
int launch_sattn_fwd(const void* qkv, int ld, void* out, int ldo, float* lse2, const float* qs, const float* ks, const float* cosT,
                     const float* sinT, SAttnDims d, hipStream_t s)
{
    constexpr int lds = 2 * 32 * NKB * SROW;
    auto k = sattn_fwd_kernel<NKB>;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(d.A * d.H), dim3(256), lds, s, (const bf16_t*)qkv, ld, (bf16_t*)out, ldo, lse2, qs, ks, cosT, sinT, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

}  // namespace

#define SATTN_DISPATCH(FN, ...)                                                         \
    switch (S / 32) {                                                                   \
        case 1: return FN<1>(__VA_ARGS__); case 2: return FN<2>(__VA_ARGS__);           \
        case 3: return FN<3>(__VA_ARGS__); case 4: return FN<4>(__VA_ARGS__);           \
        case 5: return FN<5>(__VA_ARGS__); case 6: return FN<6>(__VA_ARGS__);           \
        case 7: return FN<7>(__VA_ARGS__); default: return FN<8>(__VA_ARGS__);          \
    }

// 1 if the fused spatial-attention kernels take this shape (bf16, head_dim 64, S a multiple of 32 up to 256).
extern "C" int vvae_spatial_attn_supported(int S, int D, int dtype) { return sattn_ok(S, D, dtype) ? 1 : 0; }

// qkv: (A*S, >= 3*heads*D) bf16 row pitch ld, [q heads | k heads | v heads]; out: (A*S, >= heads*D) row pitch ldo.
// lse2: fp32 (A*heads, S) written (base-2 log-sum-exp of the scaled scores, for the backward pass).
// q_scale / k_scale fp32 (D); cos / sin fp32 (>= S, D) RoPE tables (position = index in the sequence).
extern "C" int vvae_spatial_attn_fwd(const void* qkv, int ld, void* out, int ldo, float* lse2, const float* q_scale, const float* k_scale,
                                     const float* cos_table, const float* sin_table, int A, int S, int heads, int D, float eps, int dtype,
                                     void* stream)
{
    if (!qkv || !out || !lse2 || !q_scale || !k_scale || !cos_table || !sin_table || A <= 0 || heads <= 0 || !sattn_ok(S, D, dtype) ||
        ld < 3 * heads * D || ldo < heads * D || ld % 8 || ldo % 8 || ((uintptr_t)qkv % 16) || ((uintptr_t)out % 16))
        return VVAE_ERR_BAD_ARG;
    SAttnDims d{A, S, heads, eps};
    hipStream_t s = (hipStream_t)stream;
    SATTN_DISPATCH(launch_sattn_fwd, qkv, ld, out, ldo, lse2, q_scale, k_scale, cos_table, sin_table, d, s);
}

// out, lse2: forward results; dout: gradient of out (row pitch lddo).  dqkv (A*S, >= 3*heads*D) row pitch lddq: all three sections
// written.  part: fp32 (A*heads, 2, D) written: per-(sequence, head) partials of [dq_scale | dk_scale], summed by the caller.
extern "C" int vvae_spatial_attn_bwd(const void* qkv, int ld, const void* out, int ldo, const void* dout, int lddo, const float* lse2,
                                     void* dqkv, int lddq, const float* q_scale, const float* k_scale, const float* cos_table,
                                     const float* sin_table, float* part, int A, int S, int heads, int D, float eps, int dtype, void* stream)
{
    if (!qkv || !out || !dout || !lse2 || !dqkv || !part || !q_scale || !k_scale || !cos_table || !sin_table || A <= 0 || heads <= 0 ||
        !sattn_ok(S, D, dtype) || ld < 3 * heads * D || lddq < 3 * heads * D || ldo < heads * D || lddo < heads * D || ld % 8 || ldo % 8 ||
        lddo % 8 || lddq % 8 || ((uintptr_t)qkv % 16) || ((uintptr_t)out % 16) || ((uintptr_t)dout % 16) || ((uintptr_t)dqkv % 16))
        return VVAE_ERR_BAD_ARG;
    SAttnDims d{A, S, heads, eps};
    hipStream_t s = (hipStream_t)stream;
    SATTN_DISPATCH(launch_sattn_bwd, qkv, ld, out, ldo, dout, lddo, lse2, dqkv, lddq, q_scale, k_scale, cos_table, sin_table, part, d, s);
}
