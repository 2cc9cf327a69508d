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
//   * 8 waves as 4 (M) x 2 (N); a wave owns 64 x 96 (or 64 x 64) of C = 4 x 6 (4 x 4) v_mfma_f32_16x16x32_bf16 tiles.  The
//     WEIGHT rows are the MFMA's row operand, so a lane's 4 consecutive accumulator registers are 4 consecutive output
//     channels of one token (packed 8-byte writes in the epilogue).
//   * K advances in 64-element tiles staged by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write) into two
//     LDS stages.  The loop lives on the L2 -> LDS stream, and that stream runs at full rate only for whole 128-byte lines: a
//     piece is 8 rows x 128 B (with a 32-element tile, i.e. 64-byte half lines, the DMA alone measured 13.5 TB/s chip-wide
//     against a ~17 TB/s ceiling for half-line requests -- slower than the matrix pipe needs).
//   * LDS image: [row][64 k] bf16 = 128-byte rows, written lane-linearly by the DMA.  The XOR swizzle that makes the 32-row
//     ds_read_b128 fragment reads conflict-free (16-byte chunk c of row r lives in slot c ^ ((r >> 1) & 7)) is applied to the
//     per-lane SOURCE address and to the read address (the same involution on both sides).
//   * waves 4-7 run ONE SEGMENT behind waves 0-3 (two waves share a SIMD: w and w + 4): on every SIMD one wave multiplies
//     (48 MFMAs on fragments it already holds) while its partner reads the next tile's fragments.  Both halves issue their
//     DMA pieces of tile t+1 in the same segment -- the first in which the stage tile t-1 used is free -- and wait for them one
//     segment before the first read, so a tile has a whole step (two segments) to land:
//          waves 0-3:  L0 | C0 | L1 | C1 | ...     L_t: read fragments of tile t, issue own pieces of tile t+1
//          waves 4-7:  -- | L0 | C0 | L1 | ...     C_t: issue own pieces of tile t+2, multiply tile t           (| = s_barrier)
//   * epilogue through LDS: accumulators (+bias, rounded to bf16) are parked as a [256][BN] image, then stored with coalesced
//     16-byte rows; the residual add / SiLU / SiLU-derivative variants run in that second pass on the rounded values, so the
//     fused result is bit-identical to Linear followed by the separate elementwise op.
//   * workgroup -> tile map keeps the tiles of one row block (which share the A panel) on one XCD (private L2).
#include "common.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 64;                         // K elements per staged tile (128-byte LDS rows)
constexpr int ROWB = BK * 2;

enum { EPI_NONE = 0, EPI_RES = 1, EPI_SILU = 2, EPI_MUL_DSILU = 3 };

struct NtDims { int M, N, K, lda, ldb, ldc, ldr, ldc2, epi, stagger, tiles, prefetch; };

template <int BM_, int BN_, int WM_, int WN_>
struct NtCfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr int NWAVES = WM * WN, NT = NWAVES * 64;
    static constexpr int WTM = BM / WM, WTN = BN / WN;          // wave tile
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    static constexpr int PA = BM / 8 / NWAVES, PB = BN / 8 / NWAVES;   // 1 KiB DMA pieces (8 rows) per wave and tile
    static constexpr int CP = BN * 2 + 16;                      // epilogue image row pitch (bytes)
    static constexpr int LDS = 2 * STAGE + 8 * 256;             // the C image of a half tile reuses stage 1; 256 B per wave of prefetch sink
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && (BM / 8) % NWAVES == 0 && (BN / 8) % NWAVES == 0 && NWAVES == 8, "tile / wave layout");
    static_assert((BN * 2 / 16) * BM % NT == 0 && LDS <= 160 * 1024, "epilogue chunks per thread / LDS");
};

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// s_waitcnt vmcnt(0) only (expcnt / lgkmcnt left at their maxima)
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }

__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }      // v_rcp_f32: 1 ulp
// d/dx silu(x) = s (1 + x (1 - s)), s = sigmoid(x)
__device__ __forceinline__ float dsilu_f(float x) { const float s = __builtin_amdgcn_rcpf(1.f + __expf(-x)); return s * (1.f + x * (1.f - s)); }

#ifndef NT_EXP            // tools/nt_epilogue_probe.py builds timing-only variants: 1 = no global stores in the epilogue, 2 = no epilogue at all
#define NT_EXP 0
#endif
#ifdef NT_STAMPS
// diagnostic build (tools/nt_timeline_probe.py): wave 0 of every workgroup stamps s_memrealtime (100 MHz) at the phase boundaries of its tiles
__device__ unsigned long long g_nt_stamps[256 * 16];
#define NT_STAMP(i) do { if (tid == 0 && (i) < 16) g_nt_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define NT_STAMP(i) do {} while (0)
#endif

template <int N_>
__device__ __forceinline__ void wait_vm()                       // s_waitcnt vmcnt(N_) only: at most the N_ youngest vector-memory ops outstanding
{
    static_assert(N_ >= 0 && N_ < 64, "vmcnt is six bits");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N_ & 15) | ((N_ >> 4) << 14));
}

// PERSISTENT over the tiles blockIdx.x, blockIdx.x + gridDim.x, ... (the launch takes min(tiles, 256) workgroups): at N = 1536 every CU
// multiplies two tiles, and what used to lie between them -- the first tile's stores draining before its workgroup could retire, the launch
// of the next workgroup, the first DMA of its operands, ~8 us of a ~60 us product -- now overlaps: the next tile's first k-tile is requested
// in the current tile's last k-step, and the current tile's stores drain under the next tile's main loop.
template <typename C, int EPI>
__global__ __launch_bounds__(C::NT, 1) void gemm_nt_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ Cout,
                                                           const float* __restrict__ bias, const bf16_t* __restrict__ res,
                                                           bf16_t* __restrict__ C2, NtDims d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN, grp = wave >> 2;

    // XCD-aware tile map: blockIdx round-robins over the 8 XCDs; give each XCD a contiguous run of tiles (n fastest), so the
    // tiles that share an A row panel hit the same L2
    const int tn_count = d.N / C::BN;
    const int ntiles = d.tiles;
    auto origin = [&](int b, int& m0, int& n0) {
        int t0 = b;
        if ((ntiles & 7) == 0) t0 = (b & 7) * (ntiles >> 3) + (b >> 3);
        m0 = (t0 / tn_count) * C::BM;
        n0 = (t0 % tn_count) * C::BN;
    };

    const int nk_ = d.K / BK;
    const int pfd = d.prefetch;
    // ---- staging: a DMA piece = 8 rows x 128 bytes; lane -> row lane >> 3, slot lane & 7; the slot holds source chunk
    //      slot ^ ((row >> 1) & 7).  Wave w issues A pieces w*PA .. and B pieces w*PB ..
    const bf16_t* ga[C::PA];
    const bf16_t* gb[C::PB];
    auto setup = [&](int m0, int n0) {
        int ln = lane;
        asm volatile("" : "+v"(ln));              // recomputed per tile: hoisted out of the tile loop these 14 offsets were spilled
#pragma unroll
        for (int i = 0; i < C::PA; ++i) {
            const int row = (wave * C::PA + i) * 8 + (ln >> 3);
            ga[i] = A + (long)(m0 + row) * d.lda + (((ln & 7) ^ ((row >> 1) & 7)) << 3);
        }
#pragma unroll
        for (int i = 0; i < C::PB; ++i) {
            const int row = (wave * C::PB + i) * 8 + (ln >> 3);
            gb[i] = B + (long)(n0 + row) * d.ldb + (((ln & 7) ^ ((row >> 1) & 7)) << 3);
        }
    };
    auto issue = [&](int kt, int stage) {
        unsigned char* sb = smem + stage * C::STAGE;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < C::PA; ++i) glds16(ga[i] + k0, sb + (wave * C::PA + i) * 1024);
#pragma unroll
        for (int i = 0; i < C::PB; ++i) glds16(gb[i] + k0, sb + C::A_BYTES + (wave * C::PB + i) * 1024);
    };

    // L2 prefetch of the A panel (vvae_gemm_nt_prefetch, distance in k-tiles, default 3; 0 = off; -0.1 ... -0.35 ms per train step, tools/ab_hook.py): one 4-byte LDS-DMA per lane into a sink, lanes 0-31 the
    // 32 rows this wave stages of k-tile kt, lanes 32-63 of k-tile kt + 1 -- a row's k-tile is exactly one 128-byte line.  The eight column
    // tiles of a row block run in lockstep on one XCD, so without it all eight take the L2 MISS latency for every A piece, and with two LDS
    // stages the k-step cannot be shorter than that latency.
    unsigned char* sink = smem + 2 * C::STAGE + wave * 256;
    const int pf_row = (wave * C::PA + ((lane & 31) >> 3)) * 8 + (lane & 7);
    auto prefetch = [&](int kt, int m0_, int nm0_, bool more_) {
        kt += lane >> 5;
        int mrow = m0_;
        if (kt >= nk_) { if (more_) { kt -= nk_; mrow = nm0_; } if (kt >= nk_) kt = nk_ - 1; }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (long)(mrow + pf_row) * d.lda + kt * BK),
                                         (__attribute__((address_space(3))) void*)sink, 4, 0, 0);
    };

    // Start-time stagger (OFF by default, vvae_gemm_nt_stagger): every other workgroup of an XCD sleeps
    // d.stagger x 2048 cycles before its first load.  As a graph of 20 back-to-back launches two cohorts ~2 us apart measured 55.9 -> 47.4 us
    // on the plain 16384 x 1536 x 768 product and 59.5 -> 53.1 with the SiLU pair of outputs (tools/nt_stagger_probe.py) -- but inside the
    // train step, between LayerNorm and attention kernels, the same setting changes nothing (35.43 vs 35.48 ms per step, tools/ab_hook.py):
    // back-to-back copies of one GEMM are not the condition the kernel runs in.
    if (blockIdx.x < 256 && ((blockIdx.x >> 3) & 1))
        for (int i = 0; i < d.stagger; ++i) __builtin_amdgcn_s_sleep(32);

    // ---- fragment reads for v_mfma_f32_16x16x32_bf16: 16 rows x 32 k per ds_read_b128; lane (row = lane & 15, k-group kg = lane >> 4).
    //      (Round 1 multiplied with 32x32x16: same LDS bytes and cycles per FLOP, lower sustained clock on random data.)
    constexpr int MB16 = C::WTM / 16, NB16 = C::WTN / 16;
    const int fr = lane & 15, kg = lane >> 4, sw = (fr >> 1) & 7;      // rows fr and fr + 16 j share (row >> 1) & 7
    int koff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) koff[ks] = ((4 * ks + kg) ^ sw) << 4;
    const int a_row = (wm * C::WTM + fr) * ROWB;                  // token rows of this wave (MFMA column operand)
    const int b_row = C::A_BYTES + (wn * C::WTN + fr) * ROWB;     // weight rows (MFMA row operand)
    constexpr int CPR = C::BN * 2 / 16;                          // 16-byte chunks per row of the C tile
    constexpr int NIT = C::BM * CPR / C::NT;                     // chunks per thread and tile
    constexpr int HIT = NIT / 2;                                 // ... and half tile
    static_assert(C::WM == 4 && C::BM == 256 && NIT % 2 == 0 && (C::BM / 2) * C::CP <= C::STAGE, "half-tile C image behind stage 0");
    unsigned char* img = smem + C::STAGE;                        // the C image of a HALF tile (128 rows) lives in stage 1's bytes
    constexpr bool has_res = EPI == EPI_RES || EPI == EPI_MUL_DSILU;
    const int nk = d.K / BK;

    int tb = blockIdx.x, m0, n0;
    origin(tb, m0, n0);
    setup(m0, n0);
    NT_STAMP(0);
    issue(0, 0);
    bool first = true;
    [[maybe_unused]] int stamp = 1;
    while (true) {
        const int nb = tb + (int)gridDim.x;
        const bool more = nb < ntiles;
        int nm0 = 0, nn0 = 0;
        if (more) origin(nb, nm0, nn0);
        f32x4 acc[NB16][MB16];
#pragma unroll
        for (int i = 0; i < NB16; ++i)
#pragma unroll
            for (int j = 0; j < MB16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // k-tile 0 of this tile has landed.  After the first tile it was requested during the previous tile's last k-step, ahead of
        // that tile's stores: those (NIT per thread, twice that with the SiLU pair) may stay in flight.
        if (first) wait_vm0();
        else wait_vm<(EPI == EPI_SILU ? 2 : 1) * NIT>();
        vvae_phase_barrier();             // tile 0 visible to all; every wave is done with the C image
        NT_STAMP(stamp); ++stamp;                 // k-tile 0 landed
        if (grp) {
            if (nk > 1) issue(1, 1);              // this half's pieces of tile 1 (the other half issues its own in L0)
            vvae_phase_barrier();         // the stagger
        }
        for (int t = 0; t < nk; ++t) {
            const unsigned char* cur = smem + (t & 1) * C::STAGE;
            // ---- L_t
            bf16x8 tf[2][MB16], wf[2][NB16];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = 0; j < MB16; ++j) tf[ks][j] = *reinterpret_cast<const bf16x8*>(cur + a_row + j * 16 * ROWB + koff[ks]);
#pragma unroll
                for (int i = 0; i < NB16; ++i) wf[ks][i] = *reinterpret_cast<const bf16x8*>(cur + b_row + i * 16 * ROWB + koff[ks]);
            }
            // the last k-step has nothing of this tile left to request: the NEXT tile's first k-tile goes out instead (nk is even, so
            // it lands in stage 0, free since k-step nk - 2), a k-step and the whole epilogue ahead of its first use
            const bool ahead = more && t + 1 == nk;
            const bool pf_now = pfd > 0 && (t & 1) == 0;          // a prefetch follows this k-step's requests (never in the last k-step)
            if (!grp) {
                if (t + 1 < nk) issue(t + 1, (t + 1) & 1);
                else if (ahead) { setup(nm0, nn0); issue(0, 0); }
                if (pf_now) prefetch(t + pfd, m0, nm0, more);
            } else if (pfd > 0 && (t & 1)) wait_vm<1>();          // waves 4-7: their pieces of tile t+1 (issued a step ago) have landed;
            else wait_vm0();                                      //            the prefetch issued behind them may still be in flight
            vvae_phase_barrier();
            // ---- C_t
            if (grp) {
                if (t + 2 < nk) issue(t + 2, t & 1);
                else if (ahead) { setup(nm0, nn0); issue(0, 0); }
                if (pf_now) prefetch(t + 1 + pfd, m0, nm0, more);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < NB16; ++i)
#pragma unroll
                    for (int j = 0; j < MB16; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], tf[ks][j], acc[i][j], 0, 0, 0);
            if (!grp && t + 1 < nk) { if (pf_now) wait_vm<1>(); else wait_vm0(); }   // waves 0-3: their pieces of tile t+1 have landed
            vvae_phase_barrier();
        }
        if (!grp) vvae_phase_barrier();
        __syncthreads();                          // every wave is done with the operand buffers
        NT_STAMP(stamp); ++stamp;                 // main loop done

        if (NT_EXP == 2) {                        // timing-only: the accumulators stay live through a store no launch takes
            if (d.M < 0) { for (int i = 0; i < NB16; ++i) for (int j = 0; j < MB16; ++j) Cout[i * MB16 + j] = (bf16_t)acc[i][j][0]; }
            if (!more) break;
            tb = nb; m0 = nm0; n0 = nn0; first = false;
            continue;
        }
        // ---- acc (+bias) -> packed bf16, in registers (48 instead of 96: the half of the waves that parks second holds them through the
        //      first half's stores).  acc[i][j][r]: channel 16 i + 4 kg + r, token 16 j + fr
        uint2 pk[NB16][MB16];
#pragma unroll
        for (int i = 0; i < NB16; ++i) {
            const int n = wn * C::WTN + i * 16 + kg * 4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) { const float* bp = bias + n0 + n; bv = make_float4(bp[0], bp[1], bp[2], bp[3]); }
#pragma unroll
            for (int j = 0; j < MB16; ++j) {
                pk[i][j].x = (uint32_t)f2bf(acc[i][j][0] + bv.x) | ((uint32_t)f2bf(acc[i][j][1] + bv.y) << 16);
                pk[i][j].y = (uint32_t)f2bf(acc[i][j][2] + bv.z) | ((uint32_t)f2bf(acc[i][j][3] + bv.w) << 16);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // the second operand of the tail (residual / saved pre-activation) does not depend on the product: request this half's
            // chunks now, so they arrive while the image is written (they were three exposed memory latencies per tile)
            uint4 rpre[HIT];
            if (has_res) {
#pragma unroll
                for (int it = 0; it < HIT; ++it) {
                    const int q = tid + it * C::NT;
                    rpre[it] = *reinterpret_cast<const uint4*>(res + (long)(m0 + 128 * h + q / CPR) * d.ldr + n0 + (q % CPR) * 8);
                }
            }
            // ---- pass 1: the waves that own rows [128 h, 128 h + 128) -> bf16 image [128][BN], pitch CP
            if ((wm >> 1) == h) {
#pragma unroll
                for (int i = 0; i < NB16; ++i) {
                    const int n = wn * C::WTN + i * 16 + kg * 4;
#pragma unroll
                    for (int j = 0; j < MB16; ++j) {
                        const int m = (wm & 1) * C::WTM + j * 16 + fr;
                        *reinterpret_cast<uint2*>(img + m * C::CP + n * 2) = pk[i][j];
                    }
                }
            }
            __syncthreads();
            // ---- pass 2: coalesced 16-byte rows, fused elementwise tail on the ROUNDED linear output
            uint4 v[HIT];
#pragma unroll
            for (int it = 0; it < HIT; ++it) {
                const int q = tid + it * C::NT;
                v[it] = *reinterpret_cast<const uint4*>(img + (q / CPR) * C::CP + (q % CPR) * 16);
            }
#pragma unroll
            for (int it = 0; it < HIT; ++it) {
                const int q = tid + it * C::NT;
                const int row = 128 * h + q / CPR, cc = q % CPR;
                const long gm = m0 + row;
                const int gn = n0 + cc * 8;
                if (EPI != EPI_NONE) {
                    float x[8];
                    const uint32_t w[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { x[2 * e] = __uint_as_float(w[e] << 16); x[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
                    float y[8];
                    if (EPI == EPI_SILU) {
                        if (NT_EXP != 1 || d.M < 0) *reinterpret_cast<uint4*>(C2 + gm * d.ldc2 + gn) = v[it];     // pre-activation, kept for backward
#pragma unroll
                        for (int e = 0; e < 8; ++e) y[e] = silu_f(x[e]);
                    } else {
                        float r[8];
                        {
                            const uint4 rp = rpre[it];
                            const uint32_t rw[4] = {rp.x, rp.y, rp.z, rp.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) { r[2 * e] = __uint_as_float(rw[e] << 16); r[2 * e + 1] = __uint_as_float(rw[e] & 0xffff0000u); }
                        }
                        if (EPI == EPI_RES) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) y[e] = x[e] + r[e];
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) y[e] = x[e] * dsilu_f(r[e]);
                        }
                    }
                    if (NT_EXP != 1 || d.M < 0) VecIO<bf16_t, 8>::store(Cout + gm * d.ldc + gn, y);
                } else {
                    if (NT_EXP != 1 || d.M < 0) *reinterpret_cast<uint4*>(Cout + gm * d.ldc + gn) = v[it];
                }
            }
            if (h == 0) __syncthreads();          // the image is read before the second half overwrites it
        }
        NT_STAMP(stamp); ++stamp;                 // epilogue issued
        if (!more) break;
        tb = nb; m0 = nm0; n0 = nn0;
        first = false;
    }
}

typedef NtCfg<256, 192, 4, 2> Nt192;
typedef NtCfg<256, 128, 4, 2> Nt128;

inline int nt_pick(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0 || M % 256 || K % BK) return 0;
    if (N % 192 == 0) return 192;
    if (N % 128 == 0) return 128;
    return 0;
}

int g_nt_stagger = 0, g_nt_persistent = 1, g_nt_prefetch = 3, g_nt_prefetch_mask = 5;      // plain and SiLU-pair products prefetch; with a residual / silu' operand it lost 0.1 ms per step

template <typename C, int EPI>
int launch_nt_epi(const void* A, const void* B, void* Cout, const float* bias, const void* res, void* C2, const NtDims& d, hipStream_t s)
{
    auto k = gemm_nt_kernel<C, EPI>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = (d.M / C::BM) * (d.N / C::BN);
    NtDims dd = d;
    dd.tiles = tiles;
    // persistent: one workgroup per CU walks tiles b, b + 256, ... -- when they divide evenly (else one tile per workgroup, as before)
    //             and K is an even number of k-tiles (the next tile's first k-tile is requested into stage 0 during k-step nk - 1)
    const int grid = (g_nt_persistent && tiles > 256 && tiles % 256 == 0 && (d.K / BK) % 2 == 0) ? 256 : tiles;
    hipLaunchKernelGGL(k, dim3(grid), dim3(C::NT), C::LDS, s, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)Cout, bias, (const bf16_t*)res,
                       (bf16_t*)C2, dd);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// the epilogue kind is a template parameter: straight-line tails, so the compiler's own s_waitcnt are counted ones
template <typename C>
int launch_nt(const void* A, const void* B, void* Cout, const float* bias, const void* res, void* C2, const NtDims& d, hipStream_t s)
{
    switch (d.epi) {
    case EPI_NONE: return launch_nt_epi<C, EPI_NONE>(A, B, Cout, bias, res, C2, d, s);
    case EPI_RES: return launch_nt_epi<C, EPI_RES>(A, B, Cout, bias, res, C2, d, s);
    case EPI_SILU: return launch_nt_epi<C, EPI_SILU>(A, B, Cout, bias, res, C2, d, s);
    default: return launch_nt_epi<C, EPI_MUL_DSILU>(A, B, Cout, bias, res, C2, d, s);
    }
}

}  // namespace

// Test / tuning hook: the start-time stagger of vvae_gemm_nt_bf16 in units of 2048 cycles (default 0 = every workgroup starts at once).
extern "C" int vvae_gemm_nt_stagger(int units)
{
    if (units < 0 || units > 64) return VVAE_ERR_BAD_ARG;
    g_nt_stagger = units;
    return 0;
}

// Test / tuning hook: L2 prefetch distance of the A panel in k-tiles (default 3; 0 = off).
extern "C" int vvae_gemm_nt_prefetch(int dist)
{
    if (dist < 0 || dist > 8) return VVAE_ERR_BAD_ARG;
    g_nt_prefetch = dist;
    return 0;
}

// Test / tuning hook: which epilogue kinds prefetch (bit e = kind e; default all).
extern "C" int vvae_gemm_nt_prefetch_mask(int mask)
{
    g_nt_prefetch_mask = mask & 15;
    return 0;
}

// Test / tuning hook: 0 = one tile per workgroup (round 1's launch form), 1 = persistent workgroups (default).
extern "C" int vvae_gemm_nt_persistent(int on)
{
    g_nt_persistent = on ? 1 : 0;
    return 0;
}

// 1 if vvae_gemm_nt_bf16 takes this shape (M % 256 == 0, K % 64 == 0, N % 192 == 0 or N % 128 == 0, 16-byte aligned pitches).
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
    NtDims d{M, N, K, lda, ldb, ldc, ldr, ldc2, epi, g_nt_stagger, 0, ((g_nt_prefetch_mask >> epi) & 1) ? g_nt_prefetch : 0};
    hipStream_t s = (hipStream_t)stream;
    if (nt_pick(M, N, K) == 192) return launch_nt<Nt192>(A, B, C, bias, res, C2, d, s);
    return launch_nt<Nt128>(A, B, C, bias, res, C2, d, s);
}

#ifdef NT_STAMPS
extern "C" int vvae_nt_stamps_copy(unsigned long long* host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nt_stamps), sizeof(unsigned long long) * 256 * 16, 0, hipMemcpyDeviceToHost);
}
#endif
