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
//   * global memory is only touched in ROW layout (4 lanes per token row, 64 contiguous bytes per lane quad, attn_rows.hpp): the raw
//     rows, the RoPE tables and the scale slices a phase needs are all requested before its first dependent instruction; MFMA
//     fragments and accumulator tiles (lane = token, registers = channels: 8-byte pieces of 64 different rows per instruction if
//     sent to memory as they are -- measured at a third of the kernel's time) cross over through a per-wave 2 KB scratch image.
//   * the tile loops are software-pipelined by hand: the next tile's row fragments and this tile's transposed fragments are requested
//     before the exponentials start; exp2 is the bare v_exp_f32 (arguments are <= 0).
// Forward saves the base-2 log-sum-exp per query for the backward pass.
// Where the time goes (tools/sattn_probe.py, A = 64, S = 256, 8 heads): every workgroup of the launch is resident at once, so the
// phases run chip-wide in lock step -- staging and the tile epilogues at HBM rate, the tile loops with HBM idle.  (Starting every
// other workgroup a few microseconds late, so that one half's memory phases meet the other half's tile loops, measured the same.)
#include "attn_tile32.hpp"

#ifndef SATTN_PROBE          // tools/sattn_probe.py builds timing-only variants with phases cut out (results are garbage there)
#define SATTN_PROBE 0
#endif

namespace {

struct SAttnDims { int A, S, H; float eps; };

// Staging of one head's rows into swizzled LDS images, 4 lanes per row (attn_rows.hpp slices: lane p owns channels [8p, 8p+8) and
// [32+8p, 32+8p+8) = 16-byte chunks p and 4+p).  In two steps so that EVERY global load of the workgroup's staging phase is in
// flight before the first dependent instruction: RawRows::fetch (the loads), then one of the put_* (unpack / norm + RoPE / LDS store).
template <int NT, int S_>
struct RawRows {
    static constexpr int RPP = NT / 4;                       // rows per pass
    static constexpr int PASSES = (S_ + RPP - 1) / RPP;
    uint4 v[PASSES][2];
    __device__ __forceinline__ void fetch(const bf16_t* __restrict__ src, long tok0, int ld) {
        const int p = threadIdx.x & 3;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int row = (threadIdx.x >> 2) + i * RPP;
            if (S_ % RPP == 0 || row < S_) {
                const bf16_t* r = src + (tok0 + row) * ld;
                v[i][0] = *reinterpret_cast<const uint4*>(r + 8 * p);
                v[i][1] = *reinterpret_cast<const uint4*>(r + 32 + 8 * p);
            } else {
                v[i][0] = make_uint4(0u, 0u, 0u, 0u);
                v[i][1] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    static __device__ __forceinline__ void unpack(const uint4 (&u)[2], float (&x)[16]) {
        const uint32_t w[8] = {u[0].x, u[0].y, u[0].z, u[0].w, u[1].x, u[1].y, u[1].z, u[1].w};
#pragma unroll
        for (int i = 0; i < 8; ++i) { x[2 * i] = __uint_as_float(w[i] << 16); x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    // plain copy (V, dO)
    __device__ __forceinline__ void put_raw(unsigned char* img) const {
        const int p = threadIdx.x & 3;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int row = (threadIdx.x >> 2) + i * RPP;
            if (S_ % RPP == 0 || row < S_) {
                *reinterpret_cast<uint4*>(img + img_off(row, p)) = v[i][0];
                *reinterpret_cast<uint4*>(img + img_off(row, 4 + p)) = v[i][1];
            }
        }
    }
    // q/k-norm + RoPE (position = row index) with the tables already in registers
    template <class Tabs>
    __device__ __forceinline__ void put_norm(unsigned char* img, const float (&sc)[16], float eps, const Tabs& t) const {
        const int p = threadIdx.x & 3;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int row = (threadIdx.x >> 2) + i * RPP;
            if (S_ % RPP == 0 || row < S_) {
                float x[16];
                unpack(v[i], x);
                ln_rope_row_reg<bf16_t, SD, 4>(x, eps, sc, t.cs[i], t.sn[i]);
                float lo[8], hi[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { lo[e] = x[e]; hi[e] = x[8 + e]; }
                VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(img + img_off(row, p)), lo);
                VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(img + img_off(row, 4 + p)), hi);
            }
        }
    }
};
// the RoPE table slices of the rows a thread stages (Q' and K' rows of one position share them)
template <int NT, int S_>
struct RowTabs {
    static constexpr int RPP = NT / 4, PASSES = (S_ + RPP - 1) / RPP;
    float cs[PASSES][16], sn[PASSES][16];
    __device__ __forceinline__ void fetch(const float* __restrict__ cosT, const float* __restrict__ sinT) {
        const int p = threadIdx.x & 3;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            int row = (threadIdx.x >> 2) + i * RPP;
            if (S_ % RPP != 0 && row >= S_) row = 0;
            load_tab<bf16_t, SD, 4>(cosT + (long)row * SD, p, cs[i]);
            load_tab<bf16_t, SD, 4>(sinT + (long)row * SD, p, sn[i]);
        }
    }
};

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
    unsigned char* scr = smem + 2 * S * SROW + wave * SCR_BYTES;     // this wave's scratch image
    const int a = blockIdx.x / d.H, h = blockIdx.x - a * d.H;
    const int HD = d.H * SD;
    const long tok0 = (long)a * S;
    const bf16_t* base = qkv + h * SD;

    if (!(SATTN_PROBE & 1)) {
        RawRows<256, S> rk, rv;
        RowTabs<256, S> tabs;
        float sck[16];
        rk.fetch(base + HD, tok0, ld);                             // the rows with VALU work first: V streams in behind the k-norm
        tabs.fetch(cosT, sinT);
        load_tab<bf16_t, SD, 4>(k_scale, threadIdx.x & 3, sck);
        rv.fetch(base + 2 * HD, tok0, ld);
        rk.put_norm(Ks, sck, d.eps, tabs);
        rv.put_raw(Vs);
    }
    __syncthreads();

    const int j = lane & 31, kh = lane >> 5;
    const FragAddr fa(lane);
    const float c2 = rsqrtf((float)SD) * 1.44269504088896341f;          // softmax scale in the exp2 domain

    for (int qb = wave; qb < NKB && !(SATTN_PROBE & 2); qb += 4) {
        const int qrow = qb * 32 + j;
        // Q' fragments: raw rows in row layout (coalesced; every load of both halves issued first) -> q_norm -> RoPE -> scratch image
        // -> operand registers
        bf16x8 qf[4];
        {
            float x[2][16], cs[2][16], sn[2][16], sc[16];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const long rr = qb * 32 + 16 * hf + (lane >> 2);
                load_row<bf16_t, SD, 4>(base + (tok0 + rr) * ld, lane & 3, x[hf]);
                load_tab<bf16_t, SD, 4>(cosT + rr * SD, lane & 3, cs[hf]);
                load_tab<bf16_t, SD, 4>(sinT + rr * SD, lane & 3, sn[hf]);
            }
            load_tab<bf16_t, SD, 4>(q_scale, lane & 3, sc);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                ln_rope_row_reg<bf16_t, SD, 4>(x[hf], d.eps, sc, cs[hf], sn[hf]);
                wave_lds_fence();                                   // earlier reads of the scratch image are done
                rows_put(scr, lane, x[hf]);
                wave_lds_fence();
                if ((j >> 4) == hf) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        qf[ks] = *reinterpret_cast<const bf16x8*>(scr + (j & 15) * SROW + (((2 * ks + kh) ^ gsw(j & 15)) << 4));
                }
            }
        }

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
                const float alpha = exp2_fast((m - mg) * c2);
                l *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
            }
            m = mg;
            const float mc2 = m * c2;                                // exp2((s - m) c2) as exp2(fma(s, c2, -m c2)): one VALU op per score less in a VALU-bound loop
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (kb0 + g < NKB) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        float pv[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) { pv[e] = exp2_fast(__builtin_fmaf(s[g][8 * u + e], c2, -mc2)); l += pv[e]; }
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
        float og[2][16];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) og[dt][e] = o[dt][e] * inv;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {                            // O rows leave through the scratch image: whole 64-byte row segments
            wave_lds_fence();
            acc_half_put(scr, j, kh, hf, og);
            wave_lds_fence();
            float x[16];
            rows_get(scr, lane, x);
            store_row<bf16_t, SD, 4>(out + (tok0 + qb * 32 + 16 * hf + (lane >> 2)) * ldo + h * SD, lane & 3, x);
        }
        if (kh == 0) lse2[((long)a * d.H + h) * S + qrow] = m * c2 + log2f(l);
    }
}

// ------------------------------------------------------------------------------------------------------------------ backward
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
    unsigned char* scr = reinterpret_cast<unsigned char*>(red + 16 * SD) + wave * SCR_BYTES;      // this wave's scratch image
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
    if (!(SATTN_PROBE & 1)) {
        typedef RawRows<512, S> RR;
        RR rv, rg, ro, rq, rk;
        RowTabs<512, S> tabs;
        float scq[16], sck[16];
        rq.fetch(base, tok0, ld);                                  // the rows with VALU work (q/k-norm + RoPE) first: the rest streams
        rk.fetch(base + HD, tok0, ld);                             // in behind them
        tabs.fetch(cosT, sinT);
        load_tab<bf16_t, SD, 4>(q_scale, threadIdx.x & 3, scq);
        load_tab<bf16_t, SD, 4>(k_scale, threadIdx.x & 3, sck);
        rv.fetch(base + 2 * HD, tok0, ld);
        rg.fetch(gbase, tok0, lddo);
        ro.fetch(out + h * SD, tok0, ldo);
        float lse_r[RR::PASSES];
#pragma unroll
        for (int i = 0; i < RR::PASSES; ++i) {
            const int row = (threadIdx.x >> 2) + i * RR::RPP;
            lse_r[i] = row < S ? lse2[((long)a * d.H + h) * S + row] : 0.f;
        }
        rq.put_norm(Qs, scq, d.eps, tabs);
        rk.put_norm(Ks, sck, d.eps, tabs);
        rv.put_raw(Vs);
        rg.put_raw(Gs);
#pragma unroll
        for (int i = 0; i < RR::PASSES; ++i) {                // delta = rowsum(dO * O)
            const int row = (threadIdx.x >> 2) + i * RR::RPP;
            float go[16], oo[16];
            RR::unpack(rg.v[i], go);
            RR::unpack(ro.v[i], oo);
            float dl = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) dl += go[e] * oo[e];
            dl = lpr_sum<4>(dl);
            if ((threadIdx.x & 3) == 0 && row < S) { delS[row] = dl; lseS[row] = lse_r[i]; }
        }
    }
    // A finished 32-row gradient tile (accumulator layout, fp32) leaves through the scratch image in two 16-row halves, in row layout:
    // whole 64-byte row segments per lane quad on the way out, and the raw rows / RoPE tables the q/k-norm backward needs come in
    // the same way.  The tile is rounded to bf16 on the way (the reference's attention hands bf16 gradients to its RoPE backward).
    auto put_v_tile = [&](const float (&g)[2][16], int row0) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            wave_lds_fence();
            acc_half_put(scr, j, kh, hf, g);
            wave_lds_fence();
            float x[16];
            rows_get(scr, lane, x);
            store_row<bf16_t, SD, 4>(dqkv + (tok0 + row0 + 16 * hf + (lane >> 2)) * lddq + 2 * HD + h * SD, lane & 3, x);
        }
    };
    // which: 0 = q section, 1 = k section.  Also leaves this wave's scale-gradient partial in its slot of red[].
    // q / k tiles (which: 0 = q section, 1 = k section) in two steps, so that the loads fly while something else runs:
    // qk_loads issues every global load of the tile -- the raw rows (for xhat) and the RoPE tables of both halves, the scale slice;
    // qk_finish transposes, runs RoPE / q-k-norm backward, stores, and leaves this wave's scale-gradient partial in its slot of red[].
    struct RowCtx { float xr[2][16], cs[2][16], sn[2][16], sc[16]; };
    auto qk_loads = [&](RowCtx& c, int row0, int which, const float* __restrict__ scale) {
        const int p = lane & 3;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const long rr = row0 + 16 * hf + (lane >> 2);
            load_row<bf16_t, SD, 4>(base + which * HD + (tok0 + rr) * ld, p, c.xr[hf]);
            load_tab<bf16_t, SD, 4>(cosT + rr * SD, p, c.cs[hf]);
            load_tab<bf16_t, SD, 4>(sinT + rr * SD, p, c.sn[hf]);
        }
        load_tab<bf16_t, SD, 4>(scale, p, c.sc);
    };
    auto qk_finish = [&](RowCtx& c, const float (&g)[2][16], int row0, int which) {
        const int p = lane & 3;
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
            store_row<bf16_t, SD, 4>(dqkv + (tok0 + rr) * lddq + which * HD + h * SD, p, gx);
#pragma unroll
            for (int i = 0; i < 16; ++i) ds[i] = hf ? ds[i] + c.xr[hf][i] : c.xr[hf][i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float t = frames_sum<4>(ds[i]);                   // over the 16 lanes that hold the same channels
            if (lane < 4) red[(wave * 2 + which) * SD + Slice<bf16_t, SD, 4>::ch(i, p)] = t;
        }
    };
    static_assert(NKB <= 8, "one key tile and one query tile per wave: the red[] slots are written once");
    for (int i = threadIdx.x; i < 16 * SD; i += 512) red[i] = 0.f;
    __syncthreads();

    // ---- phase A: dV, dK for the wave's key tiles
    for (int kt = wave; kt < NKB && !(SATTN_PROBE & 2); kt += 8) {
        bf16x8 kc[4], vc[4];                          // the tile's own K' / V rows: column operands, constant over the query loop
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kc[ks] = fa.rowfrag(Ks, kt, ks); vc[ks] = fa.rowfrag(Vs, kt, ks); }
        f32x16 dv[2], dk[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dv[dt][e] = 0.f; dk[dt][e] = 0.f; }
        // Software pipeline: the row fragments of query block qb + 1 and the transposed fragments / lse / delta of block qb are
        // requested before the exponentials of block qb start, so LDS latency runs under VALU work instead of in front of the MFMAs.
        bf16x8 qr[4], gr[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qr[ks] = fa.rowfrag(Qs, 0, ks); gr[ks] = fa.rowfrag(Gs, 0, ks); }
#pragma unroll 1
        for (int qb = 0; qb < NKB; ++qb) {
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qr[ks], kc[ks], s, 0, 0, 0);        // [query][key]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gr[ks], vc[ks], dp, 0, 0, 0);
            }
            bf16x8 gt[2][2], qt[2][2];
            float4 l4[4], d4[4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) { gt[u][dt] = fa.trfrag(Gs, qb, u, dt); qt[u][dt] = fa.trfrag(Qs, qb, u, dt); }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                l4[i] = *reinterpret_cast<const float4*>(lseS + qb * 32 + 8 * i + 4 * kh);
                d4[i] = *reinterpret_cast<const float4*>(delS + qb * 32 + 8 * i + 4 * kh);
            }
            {
                const int nb = qb + 1 < NKB ? qb + 1 : qb;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { qr[ks] = fa.rowfrag(Qs, nb, ks); gr[ks] = fa.rowfrag(Gs, nb, ks); }
            }
            // register r <-> query 32 qb + 8 (r/4) + 4 kh + r%4; the two register halves are the two k16 steps of the next products
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float p8[8], s8[8];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const float4 lv = l4[2 * u + hh], dv4 = d4[2 * u + hh];
                    const float lq[4] = {lv.x, lv.y, lv.z, lv.w}, dq4[4] = {dv4.x, dv4.y, dv4.z, dv4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = exp2_fast(s[8 * u + 4 * hh + e] * c2 - lq[e]);
                        p8[4 * hh + e] = pe;
                        s8[4 * hh + e] = pe * (dp[8 * u + 4 * hh + e] - dq4[e]);
                    }
                }
                const bf16x8 pf = pack8(p8), sf = pack8(s8);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gt[u][dt], pf, dv[dt], 0, 0, 0);   // dV^T[d][key] += dO^T[d][q] P[q][key]
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt[u][dt], sf, dk[dt], 0, 0, 0);   // dK^T[d][key] += Q'^T[d][q] dS[q][key]
                }
            }
        }
        // lane <-> key, registers <-> channels: dv leaves as it is; dk goes back through RoPE and k_norm
        RowCtx c;
        qk_loads(c, kt * 32, 1, k_scale);                           // in flight while dV goes out
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
        qk_finish(c, g, kt * 32, 1);
    }

    // ---- phase B: dQ for the wave's query tiles
    for (int qt = wave; qt < NKB && !(SATTN_PROBE & 4); qt += 8) {
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
        bf16x8 kr[4], vr[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kr[ks] = fa.rowfrag(Ks, 0, ks); vr[ks] = fa.rowfrag(Vs, 0, ks); }
#pragma unroll 1
        for (int kb = 0; kb < NKB; ++kb) {
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[ks], qc[ks], s, 0, 0, 0);        // [key][query]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[ks], gc[ks], dp, 0, 0, 0);
            }
            bf16x8 kt2[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) kt2[u][dt] = fa.trfrag(Ks, kb, u, dt);
            {
                const int nb = kb + 1 < NKB ? kb + 1 : kb;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { kr[ks] = fa.rowfrag(Ks, nb, ks); vr[ks] = fa.rowfrag(Vs, nb, ks); }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float s8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) s8[e] = exp2_fast(s[8 * u + e] * c2 - lq) * (dp[8 * u + e] - dl);
                const bf16x8 sf = pack8(s8);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt2[u][dt], sf, dq[dt], 0, 0, 0);   // dQ^T[d][query] += K'^T[d][key] dS^T[key][query]
            }
        }
        float g[2][16];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[dt][r] = dq[dt][r] * sm_scale;
        RowCtx c;
        qk_loads(c, qt * 32, 0, q_scale);
        qk_finish(c, g, qt * 32, 0);
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
    constexpr int lds = 4 * 32 * NKB * SROW + 2 * 32 * NKB * 4 + 16 * SD * 4 + 8 * SCR_BYTES;
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
    constexpr int lds = 2 * 32 * NKB * SROW + 4 * SCR_BYTES;
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
