// Dense-layer GEMM of the transformer trunk: C[M][N] (bf16) = epi(A[M][K] . B[N][K]^T + bias[N]), fp32 accumulation.
//
// Both operands are K-contiguous ("NT"): the forward pass of a Linear layer multiplies tokens (M, in) by the transposed bf16
// weight shadow (out, in); its input gradient multiplies dY (M, out) by the weight itself (in, out) -- reference
// train/layers.py:15,142-151,179-189 (nnx.Linear under autodiff), 16 such products per FactoredAttention layer and step.
//
// Shape of the problem on MI355X: M = 16 384 tokens, N and K in {512, 768, 1536}.  A BLAS tile of 192 x 256 cuts M into 85.3
// row blocks, i.e. 258 or 516 workgroups for 256 CUs -- a third round for 2 % of the work.  The tiles here divide the problem
// exactly: 256 x 192 (N = 768: 256 workgroups, N = 1536: 512) and 256 x 128 (N = 512: 256), one workgroup per CU and round.
//
//   * 8 waves as 4 (M) x 2 (N); a wave owns 64 x 96 (or 64 x 64) of C = 2 x 3 (2 x 2) v_mfma_f32_32x32x16_bf16 tiles.  The
//     WEIGHT rows are the MFMA's row operand, so a lane's 4 consecutive accumulator registers are 4 consecutive output
//     channels of one token (packed 8-byte writes in the epilogue).
//   * K advances in 32-element tiles staged by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write) into a
//     ring of S = 5 LDS stages.  A CU can take ~64 B/clk from L2 and the loop needs ~70 B/clk to keep the matrix pipe
//     full, so the loop lives on the L2 -> LDS stream and what matters is how many bytes are in flight: tiles are issued
//     four steps ahead (112 KB per CU) and retired with COUNTED s_waitcnt vmcnt(n) -- never 0 inside the loop -- behind one
//     raw s_barrier per step.  The wait at step t retires tile t+1, so after the barrier every wave may pre-read the first
//     fragments of tile t+1 while tile t is still being multiplied: the matrix pipe does not idle across the barrier.
//   * LDS image: [row][32 k] bf16 = 64-byte rows, written lane-linearly by the DMA (a 1 KiB piece = 16 rows).  The XOR swizzle
//     that makes the 32-row ds_read_b128 fragment reads conflict-free (16-byte chunk c of row r lives in slot c ^ ((r >> 2) & 3))
//     is applied to the per-lane SOURCE address and to the read address (the same involution on both sides).
//   * epilogue through LDS: accumulators (+bias, rounded to bf16) are parked as a [256][BN] image, then stored with coalesced
//     16-byte rows; the residual add / SiLU / SiLU-derivative variants run in that second pass on the rounded values, so the
//     fused result is bit-identical to Linear followed by the separate elementwise op.
//   * workgroup -> tile map keeps the tiles of one row block (which share the A panel) on one XCD (private L2).
#include "common.hpp"
#include <cstdlib>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;                         // K elements per staged tile (64-byte LDS rows)
constexpr int ROWB = BK * 2;

enum { EPI_NONE = 0, EPI_RES = 1, EPI_SILU = 2, EPI_MUL_DSILU = 3 };

struct NtDims { int M, N, K, lda, ldb, ldc, ldr, ldc2, epi, dbg; };

template <int BM_, int BN_, int WM_, int WN_, int S_>
struct NtCfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, S = S_, D = S_ - 1;
    static constexpr int NWAVES = WM * WN, NT = NWAVES * 64;
    static constexpr int WTM = BM / WM, WTN = BN / WN;          // wave tile
    static constexpr int MB = WTM / 32, NB = WTN / 32;          // 32x32 MFMA tiles per wave
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    static constexpr int A_PIECES = BM / 16, B_PIECES = BN / 16; // 1 KiB DMA pieces (16 rows) per tile
    static constexpr int PA = A_PIECES / NWAVES;                 // A pieces per wave
    static constexpr int PB_MAX = (B_PIECES + NWAVES - 1) / NWAVES, PB_MIN = B_PIECES / NWAVES;
    static constexpr int SPLIT = B_PIECES % NWAVES;              // waves < SPLIT carry PB_MAX B pieces (0: all carry PB_MIN)
    static constexpr int CP = BN * 2 + 16;                      // epilogue image row pitch (bytes)
    static constexpr int LDS = S * STAGE > BM * CP ? S * STAGE : BM * CP;
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && A_PIECES % NWAVES == 0 && NWAVES == 8, "tile / wave layout");
    static_assert((BN * 2 / 16) * BM % NT == 0, "epilogue chunks per thread");
    static_assert(D >= 2 && LDS <= 160 * 1024, "ring depth");
};

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// s_waitcnt vmcnt(N) only (expcnt / lgkmcnt left at their maxima): gfx9 encoding vmcnt = imm[3:0] | imm[15:14]
template <int N> __device__ __forceinline__ void wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
// d/dx silu(x) = s (1 + x (1 - s)), s = sigmoid(x)
__device__ __forceinline__ float dsilu_f(float x) { const float s = 1.f / (1.f + __expf(-x)); return s * (1.f + x * (1.f - s)); }

template <typename C>
__global__ __launch_bounds__(C::NT, 1) void gemm_nt_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ Cout,
                                                           const float* __restrict__ bias, const bf16_t* __restrict__ res,
                                                           bf16_t* __restrict__ C2, NtDims d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;

    // XCD-aware tile map: blockIdx round-robins over the 8 XCDs; give each XCD a contiguous run of tiles (n fastest), so the
    // tiles that share an A row panel hit the same L2
    const int tn_count = d.N / C::BN;
    const int ntiles = gridDim.x;
    int t = blockIdx.x;
    if ((ntiles & 7) == 0) t = (blockIdx.x & 7) * (ntiles >> 3) + (blockIdx.x >> 3);
    const int m0 = (t / tn_count) * C::BM, n0 = (t % tn_count) * C::BN;

    // ---- staging: a DMA piece = 16 rows x 64 bytes; lane -> row lane >> 2, slot lane & 3; the slot holds source chunk
    //      slot ^ ((row >> 2) & 3) = slot ^ ((lane >> 4) & 3).  Wave w issues A pieces w*PA .. and B pieces w, w + 8.
    const int prow = lane >> 2, pkc = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    const bf16_t* ga[C::PA];
    const bf16_t* gb[C::PB_MAX];
#pragma unroll
    for (int i = 0; i < C::PA; ++i) ga[i] = A + (long)(m0 + (wave * C::PA + i) * 16 + prow) * d.lda + pkc;
#pragma unroll
    for (int i = 0; i < C::PB_MAX; ++i) {
        int pb = wave + 8 * i;
        if (pb >= C::B_PIECES) pb = wave;                        // never issued (see nb below); keeps the address valid
        gb[i] = B + (long)(n0 + pb * 16 + prow) * d.ldb + pkc;
    }
    const int nb = (C::SPLIT == 0 || wave < C::SPLIT) ? C::PB_MAX : C::PB_MIN;   // wave-uniform
    auto issue = [&](int kt, int stage) {
        unsigned char* sb = smem + stage * C::STAGE;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < C::PA; ++i) glds16(ga[i] + k0, sb + (wave * C::PA + i) * 1024);
#pragma unroll
        for (int i = 0; i < C::PB_MAX; ++i)
            if (i < nb) glds16(gb[i] + k0, sb + C::A_BYTES + (wave + 8 * i) * 1024);
    };
    // allow `tiles` whole tiles of this wave's pieces to stay in flight
    auto wait_tiles = [&](int tiles) {
        if (C::SPLIT == 0 || wave < C::SPLIT) {
            if (tiles >= 3) wait_vm<3 * (C::PA + C::PB_MAX)>(); else if (tiles == 2) wait_vm<2 * (C::PA + C::PB_MAX)>();
            else if (tiles == 1) wait_vm<C::PA + C::PB_MAX>(); else wait_vm<0>();
        } else {
            if (tiles >= 3) wait_vm<3 * (C::PA + C::PB_MIN)>(); else if (tiles == 2) wait_vm<2 * (C::PA + C::PB_MIN)>();
            else if (tiles == 1) wait_vm<C::PA + C::PB_MIN>(); else wait_vm<0>();
        }
    };

    // ---- fragment reads: 32 rows x 16 k per ds_read_b128; lane (row = lane & 31, kh = lane >> 5)
    const int fr = lane & 31, kh = lane >> 5, sw = (fr >> 2) & 3;
    int koff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) koff[ks] = ((2 * ks + kh) ^ sw) << 4;
    const int a_row = (wm * C::WTM + fr) * ROWB;                  // token rows of this wave (MFMA column operand)
    const int b_row = C::A_BYTES + (wn * C::WTN + fr) * ROWB;     // weight rows (MFMA row operand)
    auto read_frags = [&](const unsigned char* base, int ks, bf16x8 (&tf)[C::MB], bf16x8 (&wf)[C::NB]) {
#pragma unroll
        for (int j = 0; j < C::MB; ++j) tf[j] = *reinterpret_cast<const bf16x8*>(base + a_row + j * 32 * ROWB + koff[ks]);
#pragma unroll
        for (int i = 0; i < C::NB; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(base + b_row + i * 32 * ROWB + koff[ks]);
    };
    auto mma = [&](f32x16 (&acc)[C::NB][C::MB], const bf16x8 (&tf)[C::MB], const bf16x8 (&wf)[C::NB]) {
#pragma unroll
        for (int i = 0; i < C::NB; ++i)
#pragma unroll
            for (int j = 0; j < C::MB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], tf[j], acc[i][j], 0, 0, 0);
    };

    f32x16 acc[C::NB][C::MB];
#pragma unroll
    for (int i = 0; i < C::NB; ++i)
#pragma unroll
        for (int j = 0; j < C::MB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- main loop: two waves share a SIMD (w and w + 4).  Waves 4-7 run ONE SEGMENT behind waves 0-3, so on every SIMD one
    //      wave multiplies (12 MFMAs on fragments it already holds) while its partner reads the next tile's fragments from
    //      LDS, issues its DMA pieces and sits out its counted vmcnt wait: the matrix pipe never waits for an LDS read.
    //          waves 0-3:  L0 | C0 | L1 | C1 | L2 | ...        (| = s_barrier)
    //          waves 4-7:  -- | L0 | C0 | L1 | C1 | ...
    //      L_t: read the fragments of tile t; refill the stage tile t-1 used (both halves finished reading it before the
    //      barrier that opened this segment); wait until tile t+1 has landed, leaving S-2 tiles in flight.
    const int nk = d.K / BK;
    const int grp = wave >> 2;
    const int npro = nk < C::S ? nk : C::S;   // the whole ring
    for (int t = 0; t < npro; ++t) issue(t, t);
    wait_tiles(npro - 1);                     // tile 0 has landed (this wave's pieces) ...
    __builtin_amdgcn_s_barrier();             // ... and everybody's
    if (grp) __builtin_amdgcn_s_barrier();    // the stagger
    bf16x8 tf0[C::MB], wf0[C::NB], tf1[C::MB], wf1[C::NB];
    int st = 0;                               // stage of tile t
    for (int t = 0; t < nk; ++t) {
        const unsigned char* cur = smem + st * C::STAGE;
        if (!(d.dbg & 4)) {
            read_frags(cur, 0, tf0, wf0);
            read_frags(cur, 1, tf1, wf1);
        }
        if (t >= 1 && t - 1 + C::S < nk && !(d.dbg & 1)) issue(t - 1 + C::S, st == 0 ? C::S - 1 : st - 1);
        // tiles issued so far: min(nk, t + S); everything beyond tile t+1 may stay in flight
        const int issued = t + C::S < nk ? t + C::S : nk;
        wait_tiles(issued - t - 2);
        __builtin_amdgcn_s_barrier();
        if (!(d.dbg & 2)) { mma(acc, tf0, wf0); mma(acc, tf1, wf1); }
        __builtin_amdgcn_s_barrier();
        st = st + 1 == C::S ? 0 : st + 1;
    }
    if (!grp) __builtin_amdgcn_s_barrier();
    __syncthreads();                          // every wave is done with the operand buffers: reuse them for the C image

    // ---- epilogue pass 1: acc (+bias) -> bf16 image [BM][BN], pitch CP.  acc[i][j][r]: channel 32 i + 8 (r/4) + 4 kh + r%4,
    //      token 32 j + fr
#pragma unroll
    for (int i = 0; i < C::NB; ++i) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int n = wn * C::WTN + i * 32 + rg * 8 + kh * 4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) { const float* bp = bias + n0 + n; bv = make_float4(bp[0], bp[1], bp[2], bp[3]); }
#pragma unroll
            for (int j = 0; j < C::MB; ++j) {
                const int m = wm * C::WTM + j * 32 + fr;
                uint2 pk;
                pk.x = (uint32_t)f2bf(acc[i][j][rg * 4 + 0] + bv.x) | ((uint32_t)f2bf(acc[i][j][rg * 4 + 1] + bv.y) << 16);
                pk.y = (uint32_t)f2bf(acc[i][j][rg * 4 + 2] + bv.z) | ((uint32_t)f2bf(acc[i][j][rg * 4 + 3] + bv.w) << 16);
                *reinterpret_cast<uint2*>(smem + m * C::CP + n * 2) = pk;
            }
        }
    }
    __syncthreads();
    // ---- epilogue pass 2: coalesced 16-byte rows, fused elementwise tail on the ROUNDED linear output
    constexpr int CPR = C::BN * 2 / 16;                          // chunks per row
#pragma unroll 4
    for (int q = tid; q < C::BM * CPR; q += C::NT) {
        const int row = q / CPR, cc = q % CPR;
        uint4 v = *reinterpret_cast<const uint4*>(smem + row * C::CP + cc * 16);
        const long gm = m0 + row;
        const int gn = n0 + cc * 8;
        if (d.epi != EPI_NONE) {
            float x[8];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[2 * e] = __uint_as_float(w[e] << 16); x[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
            float y[8];
            if (d.epi == EPI_SILU) {
                *reinterpret_cast<uint4*>(C2 + gm * d.ldc2 + gn) = v;         // pre-activation, kept for backward
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = silu_f(x[e]);
            } else {
                float r[8];
                VecIO<bf16_t, 8>::load(res + gm * d.ldr + gn, r);
                if (d.epi == EPI_RES) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = x[e] + r[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = x[e] * dsilu_f(r[e]);
                }
            }
            VecIO<bf16_t, 8>::store(Cout + gm * d.ldc + gn, y);
        } else {
            *reinterpret_cast<uint4*>(Cout + gm * d.ldc + gn) = v;
        }
    }
}

typedef NtCfg<256, 192, 4, 2, 5> Nt192;
typedef NtCfg<256, 128, 4, 2, 5> Nt128;

inline int nt_pick(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0 || M % 256 || K % BK) return 0;
    if (N % 192 == 0) return 192;
    if (N % 128 == 0) return 128;
    return 0;
}

template <typename C>
int launch_nt(const void* A, const void* B, void* Cout, const float* bias, const void* res, void* C2, const NtDims& d, hipStream_t s)
{
    auto k = gemm_nt_kernel<C>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = (d.M / C::BM) * (d.N / C::BN);
    hipLaunchKernelGGL(k, dim3(tiles), dim3(C::NT), C::LDS, s, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)Cout, bias, (const bf16_t*)res,
                       (bf16_t*)C2, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// 1 if vvae_gemm_nt_bf16 takes this shape (M % 256 == 0, K % 32 == 0, N % 192 == 0 or N % 128 == 0, 16-byte aligned pitches).
extern "C" int vvae_gemm_nt_supported(int M, int N, int K, int lda, int ldb, int ldc)
{
    return (nt_pick(M, N, K) && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && lda >= K && ldb >= K && ldc >= N) ? 1 : 0;
}

// C (M, N) bf16 = epi(A (M, K) . B (N, K)^T + bias).  bias fp32 (N) or NULL.
// epi 0: none.  1: + res (M, N) bf16 (residual add on the rounded linear output).  2: SiLU; the rounded pre-activation goes
// to C2 (M, N).  3: * silu'(res) (res = the saved pre-activation: the input gradient of a SiLU layer).
extern "C" int vvae_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias, const void* res,
                                 int ldr, void* C2, int ldc2, int epi, int M, int N, int K, void* stream)
{
    if (!A || !B || !C || !vvae_gemm_nt_supported(M, N, K, lda, ldb, ldc) || epi < 0 || epi > 3 || ((uintptr_t)A % 16) ||
        ((uintptr_t)B % 16) || ((uintptr_t)C % 16)) return VVAE_ERR_BAD_ARG;
    if ((epi == EPI_RES || epi == EPI_MUL_DSILU) && (!res || ldr % 8 || ldr < N || ((uintptr_t)res % 16))) return VVAE_ERR_BAD_ARG;
    if (epi == EPI_SILU && (!C2 || ldc2 % 8 || ldc2 < N || ((uintptr_t)C2 % 16))) return VVAE_ERR_BAD_ARG;
    static const int dbg = getenv("VVAE_NT_DBG") ? atoi(getenv("VVAE_NT_DBG")) : 0;   // ablation hook (bench only)
    NtDims d{M, N, K, lda, ldb, ldc, ldr, ldc2, epi, dbg};
    hipStream_t s = (hipStream_t)stream;
    if (nt_pick(M, N, K) == 192) return launch_nt<Nt192>(A, B, C, bias, res, C2, d, s);
    return launch_nt<Nt128>(A, B, C, bias, res, C2, d, s);
}
