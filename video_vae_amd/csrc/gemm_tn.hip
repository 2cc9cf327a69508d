// Weight-gradient GEMM of the dense layers: C[M][N] (fp32) = sum_k A[k][m] * B[k][n], optional db[n] = sum_k B[k][n].
//
// A = layer input (tokens, in_features), B = output gradient (tokens, out_features), both bf16 row-major (token-major), i.e.
// the reduction runs over the ROWS of both operands: K = 16 384 tokens against M x N = 768 x 1536 at the production config
// (126 such products per training step; the Linear layers of train/layers.py:15,142-151,179-189 under autodiff).  A BLAS
// "TN" GEMM of this shape has only (M/128)(N/128) = 72 output tiles for 256 CUs; this kernel splits K across workgroups
// instead and reuses the conv-wgrad recipe:
//   * both operands are K-major in memory, so MFMA fragments (8 consecutive k per lane) come from LDS through
//     ds_read_b64_tr_b16 (hardware 4x16 transpose); token rows are padded to 288 bytes (256 + 32) so the 8 rows a half-wave
//     touches fall in 8 different 32-byte bank slots (conflict-free);
//   * 128x128 tile per workgroup, 4 waves x (64x64 = 4x4 MFMA tiles), 32-token k-steps, register-staged double-buffered LDS
//     with ONE barrier per k-step (global loads of step i+1 are in flight while step i computes);
//   * split-K partial tiles go to an fp32 slab, summed by a second kernel in fixed order: deterministic, no float atomics;
//   * the bias gradient rides along as an all-ones A fragment in the m-block-0 workgroups (no separate column-sum pass);
//   * M, N need only be multiples of 8 (round 3): the columns of a tile past M / N are staged as zeros and not stored -- the 96-wide latent
//     heads (768 x 96, 96 x 768) took a batched library product + a framework sum + a two-kernel column sum each (4 launches, ~39 us).
#include "common.hpp"

namespace tn256 {            // gemm_tn256.hip: 256 x 256 tiles for M, N multiples of 256
bool supported(int M, int N, int K, int lda, int ldb);
int pick_splits(int M, int N, int K);
int launch(const void* A, int lda, const void* B, int ldb, float* slab, float* slab_db, int M, int N, int K, int splits, int* used,
           hipStream_t s);
}

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4v __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, KS = 32;
constexpr int PITCH = 288;                       // bytes per token row of a 128-channel LDS tile
constexpr int TILE_BYTES = KS * PITCH;           // 9216
constexpr int LDS_BYTES = 4 * TILE_BYTES;        // 2 buffers x (A tile + B tile)

struct GemmDims { int M, N, K, lda, ldb, klen; };      // klen = tokens per split (multiple of KS)

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p)
{
    typedef __attribute__((address_space(3))) s16x4v* lds_ptr;
    const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 16 * PITCH));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* __restrict__ slab,
                                                           float* __restrict__ slab_db, GemmDims d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, sp = blockIdx.z;
    const int k_beg = sp * d.klen;
    int k_end = k_beg + d.klen;
    if (k_end > d.K) k_end = d.K;
    // staging: a 32-token x 128-channel tile = 512 sixteen-byte items per operand, two per thread
    const int tok0 = tid >> 4, part = tid & 15;                 // items tid and tid + 256: tokens tok0 and tok0 + 16
    const bf16_t* ga = A + (long)m0 + part * 8;
    const bf16_t* gb = B + (long)n0 + part * 8;
    const int lds_item = tok0 * PITCH + part * 16;
    const bool a_in = m0 + part * 8 < d.M, b_in = n0 + part * 8 < d.N;      // edge tiles (M, N multiples of 8, not of 128): zero columns
    // transposed-read lane offset: lane (g = l>>4, q = (l>>2)&3, p = l&3) -> token row 4g+q, channels 4p..4p+3
    const int loff = (4 * (lane >> 4) + ((lane >> 2) & 3)) * PITCH + 8 * (lane & 3);
    const bool do_bias = slab_db != nullptr && blockIdx.x == 0 && wm == 0;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const s16x8 ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);

    uint4 ra[2], rb[2];
    auto fetch = [&](int k) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kt = k + tok0 + 16 * i;
            ra[i] = make_uint4(0, 0, 0, 0); rb[i] = make_uint4(0, 0, 0, 0);
            if (kt < k_end) {
                if (a_in) ra[i] = *reinterpret_cast<const uint4*>(ga + (long)kt * d.lda);
                if (b_in) rb[i] = *reinterpret_cast<const uint4*>(gb + (long)kt * d.ldb);
            }
        }
    };
    fetch(k_beg);
    int buf = 0;
    for (int k = k_beg; k < k_end; k += KS) {
        unsigned char* As = smem + buf * 2 * TILE_BYTES;
        unsigned char* Bs = As + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<uint4*>(As + lds_item + 16 * i * PITCH) = ra[i];
            *reinterpret_cast<uint4*>(Bs + lds_item + 16 * i * PITCH) = rb[i];
        }
        __syncthreads();
        if (k + KS < k_end) fetch(k + KS);
        bf16x8 afr[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) afr[i] = tr_frag(As + (wm * 64 + i * 16) * 2 + loff);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = tr_frag(Bs + (wn * 64 + j * 16) * 2 + loff);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[i], bfr[j], acc[i][j], 0, 0, 0);
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bfr[j], accb[j], 0, 0, 0);
        }
        buf ^= 1;
    }
    // D[row = m (4g+e)][col = n (lane&15)]
    const int col = lane & 15, rg = lane >> 4;
    float* out = slab + (long)sp * d.M * d.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + wm * 64 + i * 16 + rg * 4 + e, n = n0 + wn * 64 + j * 16 + col;
                if (m < d.M && n < d.N) out[(long)m * d.N + n] = acc[i][j][e];
            }
    if (do_bias && rg == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + col;
            if (n < d.N) slab_db[(long)sp * d.N + n] = accb[j][0];
        }
    }
}

// C[i] = sum_s slab[s][i]  (float4 lanes, fixed order);  db likewise
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ slab_db,
                                                             float* __restrict__ C, float* __restrict__ db, long MN, int N, int splits)
{
    const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
    const long n4 = MN / 4;
    if (i4 < n4) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sp = 0; sp < splits; ++sp) {
            const float4 t = reinterpret_cast<const float4*>(slab + (long)sp * MN)[i4];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        reinterpret_cast<float4*>(C)[i4] = s;
    } else if (db && i4 < n4 + N) {
        const int n = (int)(i4 - n4);
        float s = 0.f;
        for (int sp = 0; sp < splits; ++sp) s += slab_db[(long)sp * N + n];
        db[n] = s;
    }
}

bool g_tn_big = true;

inline int pick_splits(int M, int N, int K)
{
    const int tiles = ceil_div(M, BM) * ceil_div(N, BN);
    int s = (640 + tiles - 1) / tiles;                 // ~2.5 workgroups per CU over the chip
    const int max_s = (K + 4 * KS - 1) / (4 * KS);     // at least 4 k-steps per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return s;
}

}  // namespace

// 1 if vvae_gemm_tn_bf16 takes this shape.
extern "C" int vvae_gemm_tn_supported(int M, int N, int K, int lda, int ldb)
{
    return (M > 0 && N > 0 && K > 0 && M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N) ? 1 : 0;
}

// Scratch bytes for vvae_gemm_tn_bf16 (split-K slabs).
extern "C" size_t vvae_gemm_tn_ws_bytes(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || M % 8 || N % 8) return 0;
    const int s = (g_tn_big && tn256::supported(M, N, K, M, N)) ? tn256::pick_splits(M, N, K) : pick_splits(M, N, K);
    return ((size_t)s * M * N + (size_t)s * N) * sizeof(float);
}

// A: (K, M) bf16 row pitch lda; B: (K, N) bf16 row pitch ldb; C: (M, N) fp32 contiguous, overwritten; db: (N) fp32 or NULL.
extern "C" int vvae_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, float* C, float* db, int M, int N, int K,
                                 void* ws, size_t ws_bytes, void* stream)
{
    if (!A || !B || !C || !vvae_gemm_tn_supported(M, N, K, lda, ldb) || ((uintptr_t)A % 16) || ((uintptr_t)B % 16) ||
        ((uintptr_t)C % 16)) return VVAE_ERR_BAD_ARG;
    if (!ws || ws_bytes < vvae_gemm_tn_ws_bytes(M, N, K) || ((uintptr_t)ws % 16)) return VVAE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const long MN = (long)M * N;
    if (g_tn_big && tn256::supported(M, N, K, lda, ldb)) {
        const int splits = tn256::pick_splits(M, N, K);
        float* slab = (float*)ws;
        float* slab_db = db ? slab + (size_t)splits * M * N : nullptr;
        int used = 0;
        const int rc = tn256::launch(A, lda, B, ldb, slab, slab_db, M, N, K, splits, &used, s);
        if (rc) return rc;
        hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(ceil_div(MN / 4 + N, 256)), dim3(256), 0, s, slab, slab_db, C, db, MN, N, used);
        VVAE_LAUNCH_CHECK();
        return 0;
    }
    const int splits = pick_splits(M, N, K);
    int klen = (K + splits - 1) / splits;
    klen = (klen + KS - 1) / KS * KS;
    const int used = (K + klen - 1) / klen;               // splits that own at least one token
    GemmDims d{M, N, K, lda, ldb, klen};
    float* slab = (float*)ws;
    float* slab_db = db ? slab + (size_t)splits * M * N : nullptr;
    hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3(ceil_div(M, BM), ceil_div(N, BN), used), dim3(256), LDS_BYTES, s, (const bf16_t*)A, (const bf16_t*)B, slab,
                       slab_db, d);
    VVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(ceil_div(MN / 4 + N, 256)), dim3(256), 0, s, slab, slab_db, C, db, MN, N, used);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// Test / bench hook: 0 routes every shape through the 128 x 128 kernel (default 1: 256 x 256 tiles where M, N allow).
extern "C" int vvae_gemm_tn_use_big_tiles(int on) { g_tn_big = on != 0; return 0; }
