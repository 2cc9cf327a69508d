// Dense-layer GEMM of the transformer trunk, second form ("ping-pong with split staging"): C[M][N] (bf16) = epi(A[M][K] . B[N][K]^T + bias[N]),
// fp32 accumulation, both operands K-contiguous -- the forward product of a Linear layer (tokens x transposed bf16 weight shadow) and its
// input gradient (dY x weight as stored); reference train/layers.py:15,142-151,158-160,179-196 (nnx.Linear under autodiff).
//
// Same tile, wave layout, LDS image and MFMA shape as gemm_nt.hip (256 x 192 / 256 x 128 tile, 8 waves as 4 (M) x 2 (N), 64-deep k-tiles of
// 128-byte rows staged by LDS-DMA into two stages with the XOR chunk swizzle on the source and the read address, v_mfma_f32_16x16x32_bf16,
// waves 4-7 one segment behind waves 0-3 so that on every SIMD one wave multiplies while its partner reads).  What is different is WHEN the
// DMA pieces are issued and what happens between two tiles -- the two things the ablation of gemm_nt.hip priced (per 768-deep tile: DMA alone
// 12.7 us, MFMA alone 9.2 us, reads + barriers 6.3 us, and a k-step of 1.5 us where the MFMAs need 0.77):
//
//   * gemm_nt issues all 56 pieces of k-tile t+2 at the start of ONE segment -- waves 0-3 in their read phase, waves 4-7 in front of their
//     MFMAs -- and nothing in the other segment.  An LDS-DMA instruction costs the issuing wave 60-180 cycles (MI355X_MICROARCH.md), seven
//     of them are as long as the 48 MFMAs they stand in front of, so the read phase (20 ds_read_b128 + 7 DMA) outlasts the partner's MFMA
//     phase and the matrix pipe idles half the time.  Here the pieces are spread and sit BETWEEN the MFMAs of the issuing wave (one piece per
//     ~7 MFMAs: the pipe keeps draining its queue while the wave stands at the DMA), and each half stages its own token rows in its own MFMA
//     phase -- legal because the token rows 0-127 of a stage are read by waves 0-3 only (rows 128-255 by waves 4-7): they are free as soon
//     as that half has its fragments in registers, one segment before the weight rows are.
//            waves 0-3:   L(g): read k-tile g, issue own WEIGHT pieces of g+1   |  C(g): 48 MFMA  +  own TOKEN pieces of g+2 in between
//            waves 4-7:   C(g-1): 48 MFMA + own token AND weight pieces of g+1  |  L(g): read k-tile g                       (| = s_barrier)
//   * the k-tiles of all the tiles a workgroup walks form ONE stream (g = tile * nk + kt): the staging of the next tile's first k-tiles is
//     just the next issue of the stream, no drain and no restart between tiles.
//   * epilogue straight from the accumulators, per wave, no LDS image and no workgroup barrier: bias (staged once per launch into LDS by
//     LDS-DMA, so the main loop's counted vmcnt waits see no register-destination load), bf16 rounding, the residual / SiLU / SiLU'
//     tails on the ROUNDED value (bit-identical to gemm_nt.hip and to Linear followed by the elementwise op), 8-byte stores of a lane's four
//     consecutive channels.  It runs in the NEXT tile's first read phase, between the issue of that phase's ds_reads and their wait, while the
//     partner wave multiplies: nothing of it stands between two MFMA phases except what outlasts the partner's 48 MFMAs.
#include "common.hpp"

namespace pp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64, ROWB = BK * 2;
enum { EPI_NONE = 0, EPI_RES = 1, EPI_SILU = 2, EPI_MUL_DSILU = 3 };
constexpr int BIAS_MAX_N = 2048;               // the launch's whole bias vector lives in LDS (8 KB)

struct Dims { int M, N, K, lda, ldb, ldc, ldr, ldc2, tiles; };

template <int BM_, int BN_>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, WM = 4, WN = 2, NWAVES = 8, NT = 512;
    static constexpr int WTM = BM / WM, WTN = BN / WN;
    static constexpr int MB16 = WTM / 16, NB16 = WTN / 16, NMFMA = 2 * MB16 * NB16;
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    static constexpr int PA = BM / 8 / NWAVES, PB = BN / 8 / NWAVES;       // 1 KiB pieces (8 rows x 128 B) per wave and k-tile
    static constexpr int BIAS_OFF = 2 * STAGE, LDS = BIAS_OFF + BIAS_MAX_N * 4;
    static_assert(BM == 256 && WTM == 64 && WTN % 16 == 0 && (BN / 8) % NWAVES == 0 && LDS <= 160 * 1024, "tile / wave layout");
};

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N_> __device__ __forceinline__ void wait_vm()     // s_waitcnt vmcnt(N_) only
{
    static_assert(N_ >= 0 && N_ < 64, "vmcnt is six bits");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N_ & 15) | ((N_ >> 4) << 14));
}
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }      // lgkmcnt(0), vmcnt / expcnt left alone

__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) { const float s = __builtin_amdgcn_rcpf(1.f + __expf(-x)); return s * (1.f + x * (1.f - s)); }
__device__ __forceinline__ float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack2(float a, float b) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }

int g_pp_spread = 1;      // vvae_gemm_pp_spread: 1 = DMA pieces between the MFMAs (default), 0 = in front of them (timing A/B only)

template <typename C, int EPI, int ABL = 0>
__global__ __launch_bounds__(C::NT, 1) void gemm_pp_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ Cout,
                                                           const float* __restrict__ bias, const bf16_t* __restrict__ res, bf16_t* __restrict__ C2,
                                                           Dims d, int spread)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN, grp = wave >> 2;
    constexpr int MB16 = C::MB16, NB16 = C::NB16, PA = C::PA, PB = C::PB;
    constexpr int abl = ABL;            // timing-only ablation bits (builds with -DPP_ABLATION, vvae_gemm_pp_spread bits 1-3): 1 no DMA behind the prologue, 2 no fragment reads, 4 no MFMAs

    // tiles of one row block (they share the token panel) on one XCD: blockIdx round-robins over the 8 XCDs, each gets a contiguous run
    const int tn_count = d.N / C::BN, ntiles = d.tiles;
    auto origin = [&](int b, int& m0, int& n0) {
        int t0 = b;
        if ((ntiles & 7) == 0) t0 = (b & 7) * (ntiles >> 3) + (b >> 3);
        m0 = (t0 / tn_count) * C::BM;
        n0 = (t0 % tn_count) * C::BN;
    };
    const int nk = d.K / BK;
    const int ntl = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // tiles this workgroup walks
    const int G = ntl * nk;                                                                  // its stream of k-tiles

    // ---- staging: a piece = 8 rows x 128 bytes; lane -> row lane >> 3, slot lane & 7, source chunk slot ^ ((row >> 1) & 7).
    //      Wave w stages token pieces w*PA .. (rows 32 w ..: waves 0-3 rows 0-127, waves 4-7 rows 128-255) and weight pieces w*PB ..
    int aoff[PA], boff[PB];                                  // per-lane element offsets inside a tile's operand panels
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (wave * PA + i) * 8 + (lane >> 3);
        aoff[i] = row * d.lda + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int row = (wave * PB + i) * 8 + (lane >> 3);
        boff[i] = row * d.ldb + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
    // two issue cursors (token pieces run ahead of weight pieces in waves 0-3): the k-tile of the stream each will stage next
    int ga_tile = blockIdx.x, ga_kt = 0, gb_tile = blockIdx.x, gb_kt = 0;
    const bf16_t* abase;                                     // wave-uniform: start of the cursor's token panel / weight panel at k-tile 0
    const bf16_t* bbase;
    {
        int m0, n0;
        origin(blockIdx.x, m0, n0);
        abase = A + (long)m0 * d.lda;
        bbase = B + (long)n0 * d.ldb;
    }
    int ia = 0, ib = 0;                                      // stream indices of the next token / weight k-tile to stage
    auto issue_a_piece = [&](int i) {
        if ((abl & 1) && ia > 1) return;
        glds16(abase + ga_kt * BK + aoff[i], smem + (ia & 1) * C::STAGE + (wave * PA + i) * 1024);
    };
    auto issue_b_piece = [&](int i) {
        if ((abl & 1) && ib > 1) return;
        glds16(bbase + gb_kt * BK + boff[i], smem + (ib & 1) * C::STAGE + C::A_BYTES + (wave * PB + i) * 1024);
    };
    auto advance_a = [&]() {
        ++ia;
        if (++ga_kt == nk) {
            ga_kt = 0; ga_tile += gridDim.x;
            if (ga_tile < ntiles) { int m0, n0; origin(ga_tile, m0, n0); abase = A + (long)m0 * d.lda; }
        }
    };
    auto advance_b = [&]() {
        ++ib;
        if (++gb_kt == nk) {
            gb_kt = 0; gb_tile += gridDim.x;
            if (gb_tile < ntiles) { int m0, n0; origin(gb_tile, m0, n0); bbase = B + (long)n0 * d.ldb; }
        }
    };
    auto issue_a_all = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) issue_a_piece(i);
        advance_a();
    };
    auto issue_b_all = [&]() {
#pragma unroll
        for (int i = 0; i < PB; ++i) issue_b_piece(i);
        advance_b();
    };

    // ---- fragment reads for v_mfma_f32_16x16x32_bf16: 16 rows x 32 k per ds_read_b128; lane (row fr = lane & 15, k-group kg = lane >> 4)
    const int fr = lane & 15, kg = lane >> 4, sw = (fr >> 1) & 7;
    int koff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) koff[ks] = ((4 * ks + kg) ^ sw) << 4;
    const int a_row = (wm * C::WTM + fr) * ROWB;                       // token rows of this wave (MFMA column operand)
    const int b_row = C::A_BYTES + (wn * C::WTN + fr) * ROWB;          // weight rows (MFMA row operand): acc register r = channel 4 kg + r

    f32x4 acc[NB16][MB16];
    bf16x8 tf[2][MB16], wf[2][NB16];
    auto read_frags = [&](int g) {
        if (abl & 2) {                                       // timing-only: fragments of unknown content the compiler cannot fold away
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = 0; j < MB16; ++j) { u32x4 u; asm volatile("" : "=v"(u)); tf[ks][j] = __builtin_bit_cast(bf16x8, u); }
#pragma unroll
                for (int i = 0; i < NB16; ++i) { u32x4 u; asm volatile("" : "=v"(u)); wf[ks][i] = __builtin_bit_cast(bf16x8, u); }
            }
            return;
        }
        const unsigned char* cur = smem + (g & 1) * C::STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < MB16; ++j) tf[ks][j] = *reinterpret_cast<const bf16x8*>(cur + a_row + j * 16 * ROWB + koff[ks]);
#pragma unroll
            for (int i = 0; i < NB16; ++i) wf[ks][i] = *reinterpret_cast<const bf16x8*>(cur + b_row + i * 16 * ROWB + koff[ks]);
        }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < NB16; ++i)
#pragma unroll
            for (int j = 0; j < MB16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // the MFMAs of one k-tile with NP DMA pieces between them (WITH_B: token pieces first, then weight pieces)
    auto mfma_phase = [&](auto np_tag, auto with_b_tag, bool do_issue, bool in_front) {
        constexpr int NP = decltype(np_tag)::value;
        constexpr bool WITH_B = decltype(with_b_tag)::value;
        auto piece = [&](int p) {
            if (p < PA) issue_a_piece(p);
            else if (WITH_B) issue_b_piece(p - PA);
        };
        if (do_issue && in_front) {
#pragma unroll
            for (int p = 0; p < NP; ++p) piece(p);
        }
        int n = 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < NB16; ++i)
#pragma unroll
                for (int j = 0; j < MB16; ++j) {
                    if (!(abl & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], tf[ks][j], acc[i][j], 0, 0, 0);
                    else { const u32x4 ua = __builtin_bit_cast(u32x4, wf[ks][i]), ub = __builtin_bit_cast(u32x4, tf[ks][j]); asm volatile("" :: "v"(ua), "v"(ub)); }   // keeps the reads alive
                    // piece p goes out behind MFMA number (2 p + 1) NMFMA / (2 NP) - 1: evenly spread, none behind the last MFMAs
                    if (NP > 0) {
#pragma unroll
                        for (int p = 0; p < NP; ++p)
                            if (n == ((2 * p + 1) * C::NMFMA) / (2 * NP) - 1 && do_issue && !in_front) {
                                piece(p);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    }
                    ++n;
                }
        if (do_issue) {
            advance_a();
            if (WITH_B) advance_b();
        }
    };

    // ---- epilogue of the tile whose accumulators the wave holds: straight from registers, 8-byte stores (4 consecutive channels of one token)
    const float* bias_lds = reinterpret_cast<const float*>(smem + C::BIAS_OFF);
    constexpr bool has_res = EPI == EPI_RES || EPI == EPI_MUL_DSILU;
    auto epilogue = [&](int tile) {
        int m0, n0;
        origin(tile, m0, n0);
        const long row0 = (long)(m0 + wm * C::WTM + fr);
        const int col0 = n0 + wn * C::WTN + kg * 4;
#pragma unroll
        for (int i = 0; i < NB16; ++i) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) bv = *reinterpret_cast<const float4*>(bias_lds + col0 + i * 16);
            uint2 rr[MB16];
            if (has_res) {
#pragma unroll
                for (int j = 0; j < MB16; ++j) rr[j] = *reinterpret_cast<const uint2*>(res + (row0 + j * 16) * d.ldr + col0 + i * 16);
            }
#pragma unroll
            for (int j = 0; j < MB16; ++j) {
                uint2 pk;
                pk.x = pack2(acc[i][j][0] + bv.x, acc[i][j][1] + bv.y);
                pk.y = pack2(acc[i][j][2] + bv.z, acc[i][j][3] + bv.w);
                const long o = (row0 + j * 16) * d.ldc + col0 + i * 16;
                if (EPI == EPI_NONE) {
                    *reinterpret_cast<uint2*>(Cout + o) = pk;
                } else {
                    const float x[4] = {lo_bf(pk.x), hi_bf(pk.x), lo_bf(pk.y), hi_bf(pk.y)};
                    float y[4];
                    if (EPI == EPI_SILU) {
                        *reinterpret_cast<uint2*>(C2 + (row0 + j * 16) * d.ldc2 + col0 + i * 16) = pk;       // the rounded pre-activation, kept for backward
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = silu_f(x[e]);
                    } else {
                        const float r[4] = {lo_bf(rr[j].x), hi_bf(rr[j].x), lo_bf(rr[j].y), hi_bf(rr[j].y)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = EPI == EPI_RES ? x[e] + r[e] : x[e] * dsilu_f(r[e]);
                    }
                    uint2 yo;
                    yo.x = pack2(y[0], y[1]);
                    yo.y = pack2(y[2], y[3]);
                    *reinterpret_cast<uint2*>(Cout + o) = yo;
                }
            }
        }
    };
    constexpr int NST = (EPI == EPI_SILU ? 2 : 1) * NB16 * MB16;       // stores a wave's epilogue leaves in flight
    static_assert(NST + PA + PB < 64, "the counted waits behind an epilogue must fit vmcnt");

    // ---- prologue: the launch's bias vector (waves 0 .. N/256-1, one 1-KiB piece each), k-tile 0, and the first pieces of k-tile 1
    if (bias && wave * 256 < d.N) {
        if (wave * 256 + lane * 4 < d.N) glds16(bias + wave * 256 + lane * 4, smem + C::BIAS_OFF + wave * 1024);
    }
    issue_a_all();
    issue_b_all();
    using IC0 = std::integral_constant<int, 0>;
    using ICA = std::integral_constant<int, PA>;
    using ICAB = std::integral_constant<int, PA + PB>;
    const bool front = spread == 0;
    if (!grp) {
        // ================= waves 0-3 =================
        if (G > 1) issue_a_all();                            // token pieces of k-tile 1 (normally issued in C(-1))
        if (G > 1) wait_vm<PA>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        int tile = blockIdx.x, kt = 0;
        for (int g = 0; g < G; ++g) {
            // ---- L(g)
            read_frags(g);
            const bool ep = kt == 0 && g > 0;                // the previous tile's accumulators are still in the registers
            if (g + 1 < G) issue_b_all();                    // own weight pieces of k-tile g + 1 (ahead of the epilogue's stores: see the waits)
            if (ep) epilogue(tile - (int)gridDim.x);
            wait_lgkm0();                                    // fragments in registers: the stage may be overwritten behind the barrier
            __builtin_amdgcn_s_barrier();
            // ---- C(g): own token pieces of k-tile g + 2 between the MFMAs
            if (kt == 0) zero_acc();
            mfma_phase(ICA{}, std::false_type{}, g + 2 < G, front);
            // k-tile g + 1 complete on this wave's side: token pieces from C(g-1), weight pieces from L(g); younger: this tile's epilogue stores
            // (if any) and the token pieces just issued
            if (g + 2 < G) { if (ep) wait_vm<NST + PA>(); else wait_vm<PA>(); }
            else { if (ep) wait_vm<NST>(); else wait_vm<0>(); }
            __builtin_amdgcn_s_barrier();
            if (++kt == nk) { kt = 0; tile += gridDim.x; }
        }
        epilogue(tile - (int)gridDim.x);
        __builtin_amdgcn_s_barrier();                        // waves 4-7 run one segment longer
    } else {
        // ================= waves 4-7, one segment behind =================
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        // "C(-1)": own pieces of k-tile 1
        if (G > 1) { issue_a_all(); issue_b_all(); }
        __builtin_amdgcn_s_barrier();
        int tile = blockIdx.x, kt = 0;
        for (int g = 0; g < G; ++g) {
            // ---- L(g)
            read_frags(g);
            const bool ep = kt == 0 && g > 0;
            if (ep) epilogue(tile - (int)gridDim.x);
            if (ep) wait_vm<NST>(); else wait_vm<0>();       // own pieces of k-tile g + 1 (issued in C(g-1), ahead of the epilogue's stores) have landed
            wait_lgkm0();
            __builtin_amdgcn_s_barrier();
            // ---- C(g): own token and weight pieces of k-tile g + 2 between the MFMAs
            if (kt == 0) zero_acc();
            mfma_phase(ICAB{}, std::true_type{}, g + 2 < G, front);
            __builtin_amdgcn_s_barrier();
            if (++kt == nk) { kt = 0; tile += gridDim.x; }
        }
        epilogue(tile - (int)gridDim.x);
    }
}

typedef Cfg<256, 192> Pp192;
typedef Cfg<256, 128> Pp128;

inline int pick(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K < 2 * BK || M % 256 || K % BK || N > BIAS_MAX_N) return 0;
    if (N % 192 == 0) return 192;
    if (N % 128 == 0) return 128;
    return 0;
}

template <typename C, int EPI>
int launch_epi(const void* A, const void* B, void* Cout, const float* bias, const void* res, void* C2, const Dims& d, hipStream_t s)
{
    void (*k)(const bf16_t*, const bf16_t*, bf16_t*, const float*, const bf16_t*, bf16_t*, Dims, int) = gemm_pp_kernel<C, EPI>;
#ifdef PP_ABLATION
    if (EPI == EPI_NONE) {
        switch (g_pp_spread >> 1) {
        case 1: k = gemm_pp_kernel<C, EPI_NONE, 1>; break;
        case 2: k = gemm_pp_kernel<C, EPI_NONE, 2>; break;
        case 3: k = gemm_pp_kernel<C, EPI_NONE, 3>; break;
        case 4: k = gemm_pp_kernel<C, EPI_NONE, 4>; break;
        case 5: k = gemm_pp_kernel<C, EPI_NONE, 5>; break;
        case 6: k = gemm_pp_kernel<C, EPI_NONE, 6>; break;
        case 7: k = gemm_pp_kernel<C, EPI_NONE, 7>; break;
        default: break;
        }
    }
#endif
    {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return (int)e;
    }
    Dims dd = d;
    dd.tiles = (d.M / C::BM) * (d.N / C::BN);
    // one workgroup per CU walking tiles b, b + 256, ... when they divide evenly, else one tile per workgroup
    const int grid = (dd.tiles > 256 && dd.tiles % 256 == 0) ? 256 : dd.tiles;
    hipLaunchKernelGGL(k, dim3(grid), dim3(C::NT), C::LDS, s, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)Cout, bias, (const bf16_t*)res, (bf16_t*)C2, dd,
                       g_pp_spread & 1);
    VVAE_LAUNCH_CHECK();
    return 0;
}

template <typename C>
int launch(const void* A, const void* B, void* Cout, const float* bias, const void* res, void* C2, const Dims& d, int epi, hipStream_t s)
{
    switch (epi) {
    case EPI_NONE: return launch_epi<C, EPI_NONE>(A, B, Cout, bias, res, C2, d, s);
    case EPI_RES: return launch_epi<C, EPI_RES>(A, B, Cout, bias, res, C2, d, s);
    case EPI_SILU: return launch_epi<C, EPI_SILU>(A, B, Cout, bias, res, C2, d, s);
    default: return launch_epi<C, EPI_MUL_DSILU>(A, B, Cout, bias, res, C2, d, s);
    }
}

}  // namespace pp

// Timing A/B hook: 1 = DMA pieces between the MFMAs (default), 0 = all in front of them; bits 1-3 switch the DMA, the fragment reads and the
// MFMAs of the main loop off (tools/pp_ablation.py: which two of the three serialise).
extern "C" int vvae_gemm_pp_spread(int on)
{
    pp::g_pp_spread = on;               // bit 0: spread; bits 1-3: timing-only ablation (no DMA / no fragment reads / no MFMAs): wrong results
    return 0;
}

// 1 if vvae_gemm_pp_bf16 takes this shape (M % 256 == 0, K % 64 == 0, K >= 128, N % 192 == 0 or N % 128 == 0, N <= 2048, 16-byte aligned
// pitches that keep a tile's panels inside 32-bit element offsets).
extern "C" int vvae_gemm_pp_supported(int M, int N, int K, int lda, int ldb, int ldc)
{
    return (pp::pick(M, N, K) && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && lda >= K && ldb >= K && ldc >= N && (long)256 * lda < (1L << 31) &&
            (long)192 * ldb < (1L << 31)) ? 1 : 0;
}

// C (M, N) bf16 = epi(A (M, K) . B (N, K)^T + bias): the arguments of vvae_gemm_nt_bf16 (gemm_nt.hip), the same results bit for bit.
extern "C" int vvae_gemm_pp_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias, const void* res,
                                 int ldr, void* C2, int ldc2, int epi, int M, int N, int K, void* stream)
{
    if (!A || !B || !C || !vvae_gemm_pp_supported(M, N, K, lda, ldb, ldc) || epi < 0 || epi > 3 || ((uintptr_t)A % 16) || ((uintptr_t)B % 16) ||
        ((uintptr_t)C % 8) || (bias && ((uintptr_t)bias % 16))) return VVAE_ERR_BAD_ARG;
    if ((epi == pp::EPI_RES || epi == pp::EPI_MUL_DSILU) && (!res || ldr % 4 || ldr < N || ((uintptr_t)res % 8))) return VVAE_ERR_BAD_ARG;
    if (epi == pp::EPI_SILU && (!C2 || ldc2 % 4 || ldc2 < N || ((uintptr_t)C2 % 8))) return VVAE_ERR_BAD_ARG;
    pp::Dims d{M, N, K, lda, ldb, ldc, ldr, ldc2, 0};
    hipStream_t s = (hipStream_t)stream;
    if (pp::pick(M, N, K) == 192) return pp::launch<pp::Pp192>(A, B, C, bias, res, C2, d, epi, s);
    return pp::launch<pp::Pp128>(A, B, C, bias, res, C2, d, epi, s);
}
