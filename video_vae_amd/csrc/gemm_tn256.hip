// Weight-gradient GEMM, large-tile form: slab[split][M][N] (fp32) = sum over the split's tokens k of A[k][m] * B[k][n]
// (A = layer input (tokens, in), B = output gradient (tokens, out), both bf16 token-major), M and N multiples of 256.
// Called through vvae_gemm_tn_bf16 (gemm_tn.hip), which also runs the fixed-order slab reduction.
//
// Why a second kernel: the 128 x 128 tile kernel moves (128 + 128) * 2 B per 128 * 128 MACs through L2 -> LDS (600 MB for the
// 768 x 1536 x 16 384 product) and measures at the ~11-13 TB/s this chip sustains on that path, not at its matrix rate.
// A 256 x 256 tile halves the bytes per MAC; the loop then needs ~32 B/clk per CU and is paced by the matrix pipe.
//
//   * 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 of the tile = 8 x 4 v_mfma_f32_16x16x32_bf16 accumulators (128 VGPRs).
//   * operands are K-major in memory (rows = tokens), so fragments (8 consecutive k per lane) come out of LDS through
//     ds_read_b64_tr_b16; the LDS image is [32 tokens][256 channels] bf16 = 512-byte rows filled by LDS-DMA
//     (global_load_lds_dwordx4, one piece = 2 whole token rows: full 128-byte lines on the L2 side, no VGPR round trip).
//     Chunk swizzle: the 16-byte chunk c of token row r lives in slot c ^ ((r & 7) << 1) -- applied on the per-lane SOURCE address
//     of the DMA and on the read address -- so the 8 rows x 32 bytes a half-wave transposes cover all 64 banks.
//   * ring of 4 stages (32 tokens each, 128 KB), tiles issued three steps ahead, counted s_waitcnt vmcnt(8), raw s_barrier;
//     waves 4-7 run one segment behind waves 0-3, so on every SIMD one wave multiplies (32 MFMAs) while its partner reads the
//     next step's fragments, issues its DMA pieces and waits (the structure of gemm_nt.hip).  32 MFMAs per wave and stage.
//   * split-K over workgroups (tiles x splits ~ one workgroup per CU); partial tiles go to fp32 slabs with 64-byte row
//     segments per store, summed in fixed order by gemm_tn_reduce_kernel: deterministic, no float atomics.
//   * the bias gradient (column sums of B) rides along as an all-ones row operand in the m-tile-0 workgroups.
#include "common.hpp"

#ifndef TN_ABL            // timing-only builds (tools/tn_ablation.py): bit 0 no DMA behind the prologue, bit 1 no fragment reads behind the first, bit 2 no MFMAs
#define TN_ABL 0
#endif

namespace tn256 {

#ifdef TN_STAMPS
__device__ unsigned long long g_tn_stamps[256 * 4];
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4v __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, KS = 32, S = 4;
constexpr int ROW = 512;                       // bytes per token row of one operand tile
constexpr int OPB = KS * ROW;                  // 16 KB per operand and stage
constexpr int STAGE = 2 * OPB;
constexpr int LDS_BYTES = S * STAGE;           // 128 KB
constexpr int P = 4;                           // DMA pieces per wave and stage (2 A + 2 B)
#ifndef TN_ALATE
#define TN_ALATE 3
#endif
constexpr int ALATE = TN_ALATE;                // A fragments (of 8) a wave reads in its READ phase; the others BETWEEN the products of its MFMA phase

struct Dims { int M, N, K, lda, ldb, klen, tiles_m, tiles_n; };

template <int N> __device__ __forceinline__ void wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
}

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// One 256 x 256 output tile over tokens [k_beg, k_beg + nk * 32): out[m][n] (row pitch ld_out, already offset to the tile's matrix)
// is WRITTEN: a split-K slab or, with the whole K range, the final gradient.  out_db (or NULL): column sums of B for this tile's
// columns (only meaningful for m-tile 0).
// v_mfma_f32_16x16x32_bf16: one 32-token stage = one k-step of 8 x 4 products per wave.  (Round 1 multiplied with 32x32x16: the same
// LDS bytes and the same cycles per FLOP, but on random data the chip holds a higher clock under the 16x16x32 loop --
// MI355X_MICROARCH.md, DVFS give-back item 7: 967 -> 875 us per grouped launch.)  Fragment of a 16-channel block: lane (g = l >> 4, q = (l >> 2) & 3, p = l & 3)
// reads token rows 4g + q and 4g + q + 16, channels 4p .. 4p + 3 (two ds_read_b64_tr_b16); 16-byte chunk c of token row r lives in
// slot c ^ ((r & 7) << 1) -- on the DMA's per-lane SOURCE address and on the read address -- so the 8 rows x 32 bytes a half-wave
// reads cover all 64 banks.
__device__ __forceinline__ void tn256_tile(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, int lda, int ldb, int m0, int n0, int k_beg,
                                             int nk, float* __restrict__ out, int ld_out, float* __restrict__ out_db)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, grp = wave >> 2;

    const bf16_t* ga[2];
    const bf16_t* gb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 2 * (2 * wave + i) + (lane >> 5);
        const int chunk = (lane & 31) ^ ((row & 7) << 1);
        ga[i] = A + (long)(k_beg + row) * lda + m0 + chunk * 8;
        gb[i] = B + (long)(k_beg + row) * ldb + n0 + chunk * 8;
    }
    auto issue = [&](int t, int stage) {
        if ((TN_ABL & 1) && t >= S) return;
        unsigned char* sb = smem + stage * STAGE + (2 * wave) * 1024;
        const long ka = (long)t * KS * lda, kb = (long)t * KS * ldb;
        glds16(ga[0] + ka, sb);
        glds16(ga[1] + ka, sb + 1024);
        glds16(gb[0] + kb, sb + OPB);
        glds16(gb[1] + kb, sb + OPB + 1024);
    };
    auto wait_tiles = [&](int tiles_in_flight) {
        if (tiles_in_flight >= 2) wait_vm<2 * P>(); else if (tiles_in_flight == 1) wait_vm<P>(); else wait_vm<0>();
    };

    const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const int r1 = 4 * g4 + q4, sw = (r1 & 7) << 1;
    int aoff[8], boff[4];
#pragma unroll
    for (int ib = 0; ib < 8; ++ib) aoff[ib] = r1 * ROW + ((((8 * wm + ib) * 2 + (p4 >> 1)) ^ sw) << 4) + (p4 & 1) * 8;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) boff[jb] = OPB + r1 * ROW + ((((4 * wn + jb) * 2 + (p4 >> 1)) ^ sw) << 4) + (p4 & 1) * 8;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = out_db != nullptr && wm == 0;                       // wave-uniform
    const s16x8 ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);

    auto frag = [&](const unsigned char* p) {
        typedef __attribute__((address_space(3))) s16x4v* lds_ptr;
        const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
        const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 16 * ROW));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    const int npro = nk < S ? nk : S;
    for (int t = 0; t < npro; ++t) issue(t, t);
    wait_tiles(npro - 1);
    vvae_phase_barrier();
    if (grp) vvae_phase_barrier();
    int st = 0;
    bf16x8 af[8], bfr[4];
#ifdef TN_STAMPS
    if (tid == 0 && blockIdx.x < 256) { g_tn_stamps[blockIdx.x * 4] = __builtin_amdgcn_s_memtime(); g_tn_stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    // Round 4 (tools/tn_ablation.py, profiles/r04_tn_ablation.txt): a stage took ~1 520 cycles where its two MFMA phases are 1 152; with the DMA
    // OR the fragment reads switched off 1 170-1 200, the DMA alone 909.  A wave's read phase is serial -- 24 transposed reads to issue, then 4 DMA
    // pieces that each hold the wave until the texture path takes them (~113 cycles a piece with four waves pulling) -- and outlasted the partner's
    // 36 products.  Reads parked in FRONT of the wave's own products only moved the issue time (1 518-1 537; 1 792 with six of eight A fragments
    // there).  So: the read phase fetches what the first products need (A fragments 0 .. ALATE-1, the four B fragments); every other A fragment is
    // requested in the issue slots BETWEEN the products of the MFMA phase (a product holds the pipe 16 cycles and issues in 4), two block rows
    // ahead of its use.  A stage is then still being read one segment later than before, so its slot is refilled one step later: two stages in
    // flight instead of three.
    for (int t = 0; t < nk; ++t) {
        const unsigned char* cur = smem + st * STAGE;
        const bool reads = !((TN_ABL & 2) && t > 0);
        if (reads) {
#pragma unroll
            for (int ib = 0; ib < ALATE; ++ib) af[ib] = frag(cur + aoff[ib]);
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) bfr[jb] = frag(cur + boff[jb]);
        } else {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int ib = 0; ib < 8; ++ib) { u32x4 u = __builtin_bit_cast(u32x4, af[ib]); asm volatile("" : "+v"(u)); af[ib] = __builtin_bit_cast(bf16x8, u); }
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) { u32x4 u = __builtin_bit_cast(u32x4, bfr[jb]); asm volatile("" : "+v"(u)); bfr[jb] = __builtin_bit_cast(bf16x8, u); }
        }
        // the slot of stage t - 2 is free: its last readers were the other half's products of step t - 2, two segments ago
        if (t >= 2 && t - 2 + S < nk) issue(t - 2 + S, (st + S - 2) % S);
        const int issued = t >= 2 ? (t + S - 1 < nk ? t + S - 1 : nk) : npro;
        wait_tiles(issued - t - 2);
        vvae_phase_barrier();
#pragma unroll
        for (int ib = 0; ib < 8; ++ib) {
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                if (!(TN_ABL & 4)) acc[ib][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ib], bfr[jb], acc[ib][jb], 0, 0, 0);
                else { typedef unsigned u32x4 __attribute__((ext_vector_type(4))); const u32x4 ua = __builtin_bit_cast(u32x4, af[ib]), ub = __builtin_bit_cast(u32x4, bfr[jb]); asm volatile("" :: "v"(ua), "v"(ub)); }
                if (jb == 0 && ib + ALATE < 8 && reads) af[ib + ALATE] = frag(cur + aoff[ib + ALATE]);
            }
        }
        if (!(TN_ABL & 6)) {
            // pin the order above: product, the two reads of a late fragment, three products -- per block row that still has a fragment to request
#pragma unroll
            for (int ib = 0; ib < 8; ++ib) {
                if (ib + ALATE < 8) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
            }
        }
        if (do_bias && !(TN_ABL & 4)) {
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) accb[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bfr[jb], accb[jb], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every late read has its data (its products waited on it): the stage may be refilled
        vvae_phase_barrier();
        st = st + 1 == S ? 0 : st + 1;
    }
    if (!grp) vvae_phase_barrier();
#ifdef TN_STAMPS
    if (tid == 0 && blockIdx.x < 256) { g_tn_stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime(); g_tn_stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
#endif

    // ---- tile -> memory.  acc[ib][jb][j]: m = 128 wm + 16 ib + 4 g + j, n = 64 wn + 16 jb + (lane & 15)
    const int nl = lane & 15;
#pragma unroll
    for (int ib = 0; ib < 8; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + 128 * wm + 16 * ib + 4 * g4 + j;
                out[(long)m * ld_out + n0 + 64 * wn + 16 * jb + nl] = acc[ib][jb][j];
            }
    if (do_bias && g4 == 0) {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) out_db[n0 + 64 * wn + 16 * jb + nl] = accb[jb][0];
    }
}

// split-K form: slab[split][M][N], slab_db[split][N]
__global__ __launch_bounds__(512, 1) void gemm_tn256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* __restrict__ slab,
                                                            float* __restrict__ slab_db, Dims d)
{
    // workgroup -> (split, tile): consecutive logical ids (tiles of one split are neighbours: they share the split's operand
    // panels) stay on one XCD.  Bijective form of the XCD remap (grid size need not be a multiple of 8).
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int tiles = d.tiles_m * d.tiles_n;
    const int sp = L / tiles, tile = L - sp * tiles;
    const int tm = tile / d.tiles_n, tn = tile - tm * d.tiles_n;
    const int k_beg = sp * d.klen;
    int k_end = k_beg + d.klen;
    if (k_end > d.K) k_end = d.K;
    tn256_tile(A, B, d.lda, d.ldb, tm * BM, tn * BN, k_beg, (k_end - k_beg) / KS, slab + (long)sp * d.M * d.N, d.N,
               (slab_db && tm == 0) ? slab_db + (long)sp * d.N : nullptr);
}

// grouped form: up to 64 weight-gradient products that share K (the tokens of one step) in ONE launch, one whole-K tile per
// workgroup -- with >= 256 tiles in flight nothing is split, so there are no slabs and no reduction pass: every tile is written
// once, straight into its parameter's slot of the optimizer's flat gradient buffer.
struct GroupProb { const bf16_t* A; const bf16_t* B; float* C; float* db; int M, N, lda, ldb, tiles_m, tiles_n, tile_start; };
constexpr int GROUP_MAX = 64;
struct GroupArgs { GroupProb p[GROUP_MAX]; int nprob, K; };        // 3 KB of kernel arguments

// Which tiles share an L2.  An operand panel (256 channels x all K tokens = 8 MB at K = 16 384) is far larger than an XCD's 4 MB L2,
// so two tiles share a panel's fetches only while they walk K side by side on the same XCD.  Consecutive logical tiles go to one
// XCD (64 of a 512-tile launch, run in two rounds of 32); a product cut by an XCD or round boundary fetches the panels both parts
// touch twice.  Inside a product the SHORTER tile dimension runs fastest, so such a cut re-fetches min(tiles_m, tiles_n) (+1)
// panels instead of max(...) (round 1 ran tm-major: 5-6 of a 3 x 6 product's 9 panels per cut, 1.56x the algorithmic bytes).
// Tried and dropped: dealing WHOLE products to the XCDs (shares of 18 + 18 + 6 + 18 = 60 tiles): 1.8 % faster per round, but the
// two rounds then carry 480 tiles instead of 512 -- a net 4 % loss, plus a 171-tile remainder launch per step.
__global__ __launch_bounds__(512, 1) void gemm_tn256_grouped_kernel(GroupArgs g)
{
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    int pi = 0;
    for (int i = 1; i < g.nprob; ++i) pi = L >= g.p[i].tile_start ? i : pi;      // uniform scan (tile_start ascending)
    const GroupProb& P = g.p[pi];
    const int tile = L - P.tile_start;
    int tm, tn;
    if (P.tiles_m <= P.tiles_n) { tn = tile / P.tiles_m; tm = tile - tn * P.tiles_m; }
    else { tm = tile / P.tiles_n; tn = tile - tm * P.tiles_n; }
    tn256_tile(P.A, P.B, P.lda, P.ldb, tm * BM, tn * BN, 0, g.K / KS, P.C, P.N, (P.db && tm == 0) ? P.db : nullptr);
}

// splits so that tiles x splits ~ one workgroup per CU, each with at least 8 k-steps
int pick_splits(int M, int N, int K)
{
    const int tiles = (M / BM) * (N / BN);
    int s = 256 / tiles;
    const int max_s = K / (8 * KS);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return s;
}

bool supported(int M, int N, int K, int lda, int ldb)
{
    return M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0 && K % KS == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N;
}

// launches the partial-tile kernel; returns the number of slabs written through *used (0 on error) and a status
int launch(const void* A, int lda, const void* B, int ldb, float* slab, float* slab_db, int M, int N, int K, int splits, int* used,
           hipStream_t s)
{
    int klen = (K + splits - 1) / splits;
    klen = (klen + KS - 1) / KS * KS;
    *used = (K + klen - 1) / klen;
    Dims d{M, N, K, lda, ldb, klen, M / BM, N / BN};
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tn256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(gemm_tn256_kernel, dim3(d.tiles_m * d.tiles_n * *used), dim3(512), LDS_BYTES, s, (const bf16_t*)A, (const bf16_t*)B,
                       slab, slab_db, d);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int launch_grouped(const GroupArgs& g, int total_tiles, hipStream_t s)
{
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tn256_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(gemm_tn256_grouped_kernel, dim3(total_tiles), dim3(512), LDS_BYTES, s, g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace tn256

#ifdef TN_STAMPS
extern "C" int vvae_gemm_tn_stamps(unsigned long long* host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tn256::g_tn_stamps), sizeof(unsigned long long) * 256 * 4, 0, hipMemcpyDeviceToHost);
}
#endif

// Grouped dense weight gradients: for i < n: C_i (M_i, N_i) fp32 (contiguous, overwritten) = A_i^T B_i, db_i (N_i) fp32 or NULL =
// column sums of B_i, with A_i (K, M_i) / B_i (K, N_i) bf16 token-major (row pitches lda_i / ldb_i), all sharing K.
// n <= 64, M_i and N_i multiples of 256, K a multiple of 32.  One launch, one whole-K 256 x 256 tile per workgroup: no split-K slabs
// and no reduction pass (meant for >= ~200 tiles per call so the chip is full without splitting K).
extern "C" int vvae_gemm_tn_grouped_bf16(const void* const* A, const int* lda, const void* const* B, const int* ldb, float* const* C,
                                         float* const* db, const int* M, const int* N, int n, int K, void* stream)
{
    if (!A || !lda || !B || !ldb || !C || !db || !M || !N || n <= 0 || n > tn256::GROUP_MAX || K <= 0 || K % tn256::KS) return VVAE_ERR_BAD_ARG;
    tn256::GroupArgs g;
    g.nprob = n; g.K = K;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (!A[i] || !B[i] || !C[i] || !tn256::supported(M[i], N[i], K, lda[i], ldb[i]) || ((uintptr_t)A[i] % 16) || ((uintptr_t)B[i] % 16) ||
            ((uintptr_t)C[i] % 16)) return VVAE_ERR_BAD_ARG;
        g.p[i] = tn256::GroupProb{(const bf16_t*)A[i], (const bf16_t*)B[i], C[i], db[i], M[i], N[i], lda[i], ldb[i], M[i] / tn256::BM,
                                  N[i] / tn256::BN, tiles};
        tiles += (M[i] / tn256::BM) * (N[i] / tn256::BN);
    }
    return tn256::launch_grouped(g, tiles, (hipStream_t)stream);
}

