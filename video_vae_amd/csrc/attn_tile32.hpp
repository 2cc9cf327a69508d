// Shared pieces of the matrix-core attention kernels that work on 32 x 32 score tiles with head_dim 64 (attn_spatial.hip: sequences of
// h*w patches; attn_temporal_mfma32.hip: sequences of 32 / 64 frames): the swizzled 128-byte-row LDS image and its two fragment read
// shapes, and the accumulator-layout <-> row-layout hand-off through a per-wave scratch image.
#pragma once
#include "attn_rows.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4v __attribute__((ext_vector_type(4)));

constexpr int SD = 64;                          // head_dim
constexpr int SROW = SD * 2;                    // bytes per LDS row

// chunk swizzle of an LDS image with 128-byte rows (see header comment)
__device__ __forceinline__ int gsw(int row) { return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1); }

__device__ __forceinline__ float xor32(float v) { return xor_lane<32>(v); }          // v_permlane32_swap: VALU only (common.hpp)

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

// ---- accumulator layout <-> row layout through a per-wave 16-row scratch image (2 KB, same swizzle as the big images) ----------
// acc layout: lane (j = lane & 31, kh = lane >> 5) holds channels 32 dt + 8 rg + 4 kh + e (e < 4) of row j of a 32-row tile (what a
// 32x32 MFMA leaves with rows = channels): written or read straight to memory that is 8-byte pieces of 64 different rows per
// instruction -- one cache line each.  row layout (attn_rows.hpp, 4 lanes per row): lane (row = lane >> 2, p = lane & 3) holds
// channels [8p, 8p+8) and [32+8p, 32+8p+8): 16 rows x 64 contiguous bytes per instruction.  A tile goes through the scratch image
// one 16-row half at a time; writer and reader are the same wave (LDS operations of a wave complete in order: wave_lds_fence).
constexpr int SCR_BYTES = 16 * SROW;

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}

// the lanes that own rows [16 hf, 16 hf + 16) of the tile write their 8-byte pieces (values rounded to bf16)
__device__ __forceinline__ void acc_half_put(unsigned char* scr, int j, int kh, int hf, const float (&g)[2][16])
{
    if ((j >> 4) == hf) {
        const int row = j & 15, gs = gsw(row);
        unsigned char* r = scr + row * SROW + kh * 8;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                uint2 o;
                o.x = (uint32_t)f2bf(g[dt][4 * rg]) | ((uint32_t)f2bf(g[dt][4 * rg + 1]) << 16);
                o.y = (uint32_t)f2bf(g[dt][4 * rg + 2]) | ((uint32_t)f2bf(g[dt][4 * rg + 3]) << 16);
                *reinterpret_cast<uint2*>(r + (((4 * dt + rg) ^ gs) << 4)) = o;
            }
    }
}
// row-layout write / read of the scratch image: lane (row = lane >> 2, p = lane & 3), chunks p and 4 + p
__device__ __forceinline__ void rows_put(unsigned char* scr, int lane, const float (&x)[16])
{
    const int row = lane >> 2, p = lane & 3;
    float lo[8], hi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { lo[e] = x[e]; hi[e] = x[8 + e]; }
    VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(scr + img_off(row, p)), lo);
    VecIO<bf16_t, 8>::store(reinterpret_cast<bf16_t*>(scr + img_off(row, 4 + p)), hi);
}
__device__ __forceinline__ void rows_get(const unsigned char* scr, int lane, float (&x)[16])
{
    const int row = lane >> 2, p = lane & 3;
    float lo[8], hi[8];
    VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(scr + img_off(row, p)), lo);
    VecIO<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(scr + img_off(row, 4 + p)), hi);
#pragma unroll
    for (int e = 0; e < 8; ++e) { x[e] = lo[e]; x[8 + e] = hi[e]; }
}

// exp2 of a non-positive argument: the bare v_exp_f32 (results below 2^-126 flush to zero; exp2f() wraps the instruction in a
// range-scaling sequence of five more VALU operations per element for denormal results that a probability rounded to bf16 never needs)
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

}  // namespace
