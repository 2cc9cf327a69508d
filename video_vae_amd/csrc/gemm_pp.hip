// Dense-layer GEMM of the transformer trunk, second form ("ping-pong, staging in the read phases"): C[M][N] (bf16) = epi(A[M][K] . B[N][K]^T +
// bias[N]), fp32 accumulation, both operands K-contiguous -- the forward product of a Linear layer (tokens x transposed bf16 weight shadow) and
// its input gradient (dY x weight as stored); reference train/layers.py:15,142-151,158-160,179-196 (nnx.Linear under autodiff).
//
// Same tile, wave layout, LDS row image and MFMA shape as gemm_nt.hip (256 x 192 / 256 x 128 tile, 8 waves as 4 (M) x 2 (N), 64-deep k-tiles of
// 128-byte rows staged by LDS-DMA with the XOR chunk swizzle on the source and the read address, v_mfma_f32_16x16x32_bf16, waves 4-7 one segment
// behind waves 0-3 so that on every SIMD one wave multiplies while its partner reads).  What differs follows from an ablation of the main loop
// (tools/pp_ablation.py, profiles/r04_pp_ablation.txt: the loop with its DMA, its fragment reads and its MFMAs switched off one by one and in
// pairs): the three parts ADD UP instead of overlapping -- MFMAs alone 0.54 us per k-step, DMA alone 0.33, both 0.80; fragment reads alone
// cost nothing.  An LDS-DMA instruction stalls the issuing wave for 60-100 cycles, and a wave that issues its pieces in front of or between its
// own MFMAs keeps the matrix pipe idle for that long, because its partner on the SIMD is in its read phase and has no MFMA to offer.  So:
//
//   * ALL staging is issued from READ phases, where the partner wave is multiplying: the MFMA phases are 48 bare MFMAs.
//            waves 0-3:   L(g): read k-tile g, stage ALL weight rows of g+1   |  C(g): 48 MFMA
//            waves 4-7:   C(g-1): 48 MFMA                                     |  L(g): read k-tile g, stage ALL token rows of g+2     (| = s_barrier)
//     A read phase of waves 4-7 comes one segment too late to stage anything the other half needs next into a two-slot ring -- so the TOKEN
//     operand has a ring of THREE slots (96 KB; 48 KB for the weights' two: 144 KB) and its pieces go out a k-step and a half before their first
//     read; the weight pieces (waves 0-3, two slots) one k-step before, as in gemm_nt.  Up to 88 KB are in flight per CU instead of 56.
//   * the k-tiles of all the tiles a workgroup walks form ONE stream (g = tile * nk + kt): the staging of the next tile's first k-tiles is
//     just the next issue of the stream, no drain and no restart between tiles.
//   * epilogue per WAVE, no workgroup barrier and none of the operand stages: a wave rounds a 16-token x 32-channel piece of its accumulators
//     (+ bias, staged once per launch into LDS by LDS-DMA so that no register-destination load sits among the counted vmcnt waits), passes it
//     through 1.25 KB of LDS of its own (two ds_write_b64, one ds_read_b128: a lane then holds 8 consecutive channels of one token) and stores
//     16 bytes per lane, 64-byte row segments; the residual / SiLU / SiLU' tails run on the ROUNDED value (bit-identical to gemm_nt.hip and
//     to Linear followed by the elementwise op).  It runs in the NEXT tile's first read phase, behind that phase's reads and DMA issue, under
//     the partner's MFMA phase.  (Stores of a lane's 4 channels straight from the accumulator layout -- 32-byte row segments -- were tried first:
//     21 us of a 43 us product.)
#include "common.hpp"

namespace pp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64, ROWB = BK * 2;
enum { EPI_NONE = 0, EPI_RES = 1, EPI_SILU = 2, EPI_MUL_DSILU = 3 };

struct Dims { int M, N, K, lda, ldb, ldc, ldr, ldc2, tiles, final_ring; };

template <int BM_, int BN_>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, WM = 4, WN = 2, NWAVES = 8, NT = 512;
    static constexpr int WTM = BM / WM, WTN = BN / WN;
    static constexpr int MB16 = WTM / 16, NB16 = WTN / 16, NMFMA = 2 * MB16 * NB16;
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    static constexpr int PA = BM / 8 / 4, PB = BN / 8 / 4;                 // 1 KiB pieces (8 rows x 128 B) per staging wave and k-tile: waves 4-7 tokens, 0-3 weights
    static constexpr int B_OFF = 3 * A_BYTES;                              // three token slots, then two weight slots
    static constexpr int SCR_PITCH = 80, SCR_WAVE = 16 * SCR_PITCH;        // epilogue scratch of a wave: 16 tokens x 32 channels, rows padded by 16 B
    static constexpr int SCR_OFF = B_OFF + 2 * B_BYTES, BIAS_OFF = SCR_OFF + NWAVES * SCR_WAVE;
    static constexpr int BIAS_MAX_N = (160 * 1024 - BIAS_OFF) / 4 / 256 * 256;    // the launch's whole bias vector lives in LDS
    static constexpr int LDS = BIAS_OFF + BIAS_MAX_N * 4;
    static_assert(BM == 256 && WTM == 64 && NB16 % 2 == 0 && (BN / 8) % 4 == 0 && LDS <= 160 * 1024 && BIAS_MAX_N >= 1536, "tile / wave layout");
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
// s_barrier that the compiler's scheduler may not move instructions across.  The MFMAs are register-only and nothing in the language orders
// them against a barrier: without the fences hipcc sank 44 of a phase's 48 MFMAs BELOW the barrier that ends the phase (4 above, 44 below, in
// both loops), i.e. into the wave's own read phase and next to the partner wave's MFMA phase -- both halves' MFMAs on the matrix pipe in one
// segment, nothing in the other: the ping-pong ran as a sum of its parts (tools/pp_ablation.py).
__device__ __forceinline__ void phase_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }      // lgkmcnt(0), vmcnt / expcnt left alone

#ifndef PP_NT
#define PP_NT 0
#endif
__device__ __forceinline__ void st16(bf16_t* p, const uint4 v)       // one lane's 16 bytes of an output row
{
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    const u32x4v w = {v.x, v.y, v.z, v.w};
    if (PP_NT) __builtin_nontemporal_store(w, reinterpret_cast<u32x4v*>(p));
    else *reinterpret_cast<u32x4v*>(p) = w;
}
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) { const float s = __builtin_amdgcn_rcpf(1.f + __expf(-x)); return s * (1.f + x * (1.f - s)); }
__device__ __forceinline__ uint32_t pack2(float a, float b) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }

#ifdef PP_ABLATION
// diagnostic build only: shader-clock and 100 MHz wall stamps around the main loop of wave 0 of every workgroup (MI355X_MICROARCH.md, DVFS
// give-back item 6: in-kernel clock = d s_memtime / d s_memrealtime x 100 MHz); the values go to a buffer nothing else reads
__device__ unsigned long long g_pp_stamps[256 * 4];
__device__ unsigned long long g_pp_tl[256 * 8];       // 100 MHz wall stamps of one launch's phases per workgroup (tools/pp_timeline.py)
#define PP_TL(slot, who) do { if (tid == (who) && blockIdx.x < 256) g_pp_tl[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PP_TL(slot, who) do { } while (0)
#endif
int g_pp_final_ring = 1;  // vvae_gemm_pp_final_ring: the last epilogue of a launch through the idle operand rings (1) or through the wave's 1.25 KB (0)
int g_pp_ablate = 0;      // builds with -DPP_ABLATION only (tools/pp_ablation.py): 1 no DMA behind the prologue, 2 no fragment reads, 4 no MFMAs

template <typename C, int EPI, int ABL = 0>
__global__ __launch_bounds__(C::NT, 1) void gemm_pp_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ Cout,
                                                           const float* __restrict__ bias, const bf16_t* __restrict__ res, bf16_t* __restrict__ C2,
                                                           Dims d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform: scalar branches, SGPR address parts
    PP_TL(0, 0);
    const int wm = wave / C::WN, wn = wave % C::WN, grp = wave >> 2, wq = wave & 3;
    constexpr int MB16 = C::MB16, NB16 = C::NB16, PA = C::PA, PB = C::PB;
    constexpr int abl = ABL;

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
    //      Waves 4-7 stage the token rows (wave 4 + q: pieces q*PA ..), waves 0-3 the weight rows (wave q: pieces q*PB ..).
    // per-lane element offset of this wave's piece i inside a tile's operand panel: piece i lies 8 i rows below piece 0, and its swizzle term
    // ((row >> 1) & 7) is piece 0's plus 4 i modulo 8, i.e. XOR 4 for odd i: two lane offsets serve all pieces
    const int ld_s = grp ? d.lda : d.ldb;
    const int row_s = wq * (grp ? PA : PB) * 8 + (lane >> 3);
    const int chunk0 = (lane & 7) ^ ((row_s >> 1) & 7);
    const int poff_e = row_s * ld_s + (chunk0 << 3), poff_o = row_s * ld_s + ((chunk0 ^ 4) << 3);
    int s_tile = blockIdx.x, s_kt = 0, s_slot = 0, s_g = 0;  // the staging cursor: next k-tile of the stream this wave stages, and its ring slot
    const bf16_t* sbase;                                     // wave-uniform: the cursor's token panel (waves 4-7) / weight panel (waves 0-3) at k = 0
    {
        int m0, n0;
        origin(blockIdx.x, m0, n0);
        sbase = grp ? A + (long)m0 * d.lda : B + (long)n0 * d.ldb;
    }
    auto stage_next = [&]() {                                // this wave's pieces of k-tile s_g of the stream
        if (!((abl & 1) && s_g > 1)) {
            if (grp) {
#pragma unroll
                for (int i = 0; i < PA; ++i) glds16(sbase + s_kt * BK + ((i & 1) ? poff_o : poff_e) + i * 8 * ld_s, smem + s_slot * C::A_BYTES + (wq * PA + i) * 1024);
            } else {
#pragma unroll
                for (int i = 0; i < PB; ++i) glds16(sbase + s_kt * BK + ((i & 1) ? poff_o : poff_e) + i * 8 * ld_s, smem + C::B_OFF + s_slot * C::B_BYTES + (wq * PB + i) * 1024);
            }
        }
        ++s_g;
        if (++s_slot == (grp ? 3 : 2)) s_slot = 0;
        if (++s_kt == nk) {
            s_kt = 0; s_tile += gridDim.x;
            if (s_tile < ntiles) {
                int m0, n0;
                origin(s_tile, m0, n0);
                sbase = grp ? A + (long)m0 * d.lda : B + (long)n0 * d.ldb;
            }
        }
    };

    // ---- fragment reads for v_mfma_f32_16x16x32_bf16: 16 rows x 32 k per ds_read_b128; lane (row fr = lane & 15, k-group kg = lane >> 4)
    const int fr = lane & 15, kg = lane >> 4, sw = (fr >> 1) & 7;
    int koff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) koff[ks] = ((4 * ks + kg) ^ sw) << 4;
    const int a_row = (wm * C::WTM + fr) * ROWB;                       // token rows of this wave (MFMA column operand)
    const int b_row = C::B_OFF + (wn * C::WTN + fr) * ROWB;            // weight rows (MFMA row operand): acc register r = channel 4 kg + r

    f32x4 acc[NB16][MB16];
    bf16x8 tf[2][MB16], wf[2][NB16];
    int ra = 0, rb = 0;                                      // ring slots of the k-tile to read next
    [[maybe_unused]] int nread = 0;
    auto read_frags = [&]() {
        if ((abl & 2) && nread > 0) {                        // timing-only: the fragments of k-tile 0 (real data) stay in the registers for every k-tile
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = 0; j < MB16; ++j) { u32x4 u = __builtin_bit_cast(u32x4, tf[ks][j]); asm volatile("" : "+v"(u)); tf[ks][j] = __builtin_bit_cast(bf16x8, u); }
#pragma unroll
                for (int i = 0; i < NB16; ++i) { u32x4 u = __builtin_bit_cast(u32x4, wf[ks][i]); asm volatile("" : "+v"(u)); wf[ks][i] = __builtin_bit_cast(bf16x8, u); }
            }
        } else {
            const unsigned char* ca = smem + ra * C::A_BYTES;
            const unsigned char* cb = smem + rb * C::B_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = 0; j < MB16; ++j) tf[ks][j] = *reinterpret_cast<const bf16x8*>(ca + a_row + j * 16 * ROWB + koff[ks]);
#pragma unroll
                for (int i = 0; i < NB16; ++i) wf[ks][i] = *reinterpret_cast<const bf16x8*>(cb + b_row + i * 16 * ROWB + koff[ks]);
            }
        }
        if (++ra == 3) ra = 0;
        rb ^= 1;
        ++nread;
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < NB16; ++i)
#pragma unroll
            for (int j = 0; j < MB16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto mfma_phase = [&]() {
        __builtin_amdgcn_s_setprio(1);                       // the multiplying wave outranks its reading partner at the SIMD's issue port
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < NB16; ++i)
#pragma unroll
                for (int j = 0; j < MB16; ++j) {
                    if (!(abl & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], tf[ks][j], acc[i][j], 0, 0, 0);
                    else { const u32x4 ua = __builtin_bit_cast(u32x4, wf[ks][i]), ub = __builtin_bit_cast(u32x4, tf[ks][j]); asm volatile("" :: "v"(ua), "v"(ub)); }
                }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- epilogue of the tile whose accumulators the wave holds.  Unit = 16 tokens (j) x 32 channels (i pair): two ds_write_b64 per lane put the
    //      rounded values into the wave's own scratch rows (80-byte pitch: the 16 lanes of a write group hit 16 different bank pairs), one
    //      ds_read_b128 gives lane l the 8 consecutive channels 8 (l & 3) .. of token l >> 2; a wave's LDS operations execute in issue order, so
    //      consecutive units need no wait between them beyond the data dependences the compiler tracks.
    unsigned char* scr = smem + C::SCR_OFF + wave * C::SCR_WAVE;
    const float* bias_lds = reinterpret_cast<const float*>(smem + C::BIAS_OFF);
    constexpr bool has_res = EPI == EPI_RES || EPI == EPI_MUL_DSILU;
    const int w_off = fr * C::SCR_PITCH + kg * 8, r_off = (lane >> 2) * C::SCR_PITCH + (lane & 3) * 16;
    // ``final_tag``: the LAST epilogue of the launch.  The operand rings are quiet then (every staged k-tile has landed and been read), so each
    // wave takes 16 KB of them as scratch for ALL its units at once: all writes, one wait, all reads, all stores -- instead of twelve dependent
    // LDS round trips through the 1.25 KB it owns while the rings are live.
    auto epilogue = [&](int tile, auto final_tag) {
        constexpr bool FINAL = decltype(final_tag)::value;
        constexpr int NU = (NB16 / 2) * MB16;                // units of a wave
        static_assert(NU * C::SCR_WAVE <= 16 * 1024 && 8 * 16 * 1024 <= C::SCR_OFF, "final scratch inside the rings");
        unsigned char* sbase = FINAL ? smem + wave * 16 * 1024 : scr;
        int m0, n0;
        origin(tile, m0, n0);
        const long row0 = (long)(m0 + wm * C::WTM + (lane >> 2));
        const int col0 = n0 + wn * C::WTN + (lane & 3) * 8;
        const int bcol = n0 + wn * C::WTN + kg * 4;
        // the second operand of the tail (residual / saved pre-activation) does not depend on the product: all of the wave's units are requested
        // at once, before anything else (one memory latency per tile, not one per unit; the fragments of the next k-tile are read behind the
        // epilogue in these variants, so the 48 registers are free)
        uint4 rq[has_res ? NB16 / 2 : 1][MB16];
        if (has_res) {
#pragma unroll
            for (int ip = 0; ip < NB16 / 2; ++ip)
#pragma unroll
                for (int j = 0; j < MB16; ++j) rq[ip][j] = *reinterpret_cast<const uint4*>(res + (row0 + j * 16) * d.ldr + col0 + ip * 32);
        }
        auto park = [&](int ip, int j, unsigned char* dst) {     // acc (+ bias) of one unit, rounded, into scratch rows
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 bv = bias ? *reinterpret_cast<const float4*>(bias_lds + bcol + (2 * ip + h) * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
                const f32x4 a = acc[2 * ip + h][j];
                uint2 pk;
                pk.x = pack2(a[0] + bv.x, a[1] + bv.y);
                pk.y = pack2(a[2] + bv.z, a[3] + bv.w);
                *reinterpret_cast<uint2*>(dst + w_off + h * 32) = pk;
            }
        };
        auto finish = [&](int ip, int j, const uint4 v) {        // a lane's 8 consecutive channels of one token: tail, 16-byte store
            const long gm = row0 + j * 16;
            const int gn = col0 + ip * 32;
            if (EPI == EPI_NONE) {
                st16(Cout + gm * d.ldc + gn, v);
            } else {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                float x[8], y[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[2 * e] = __uint_as_float(w[e] << 16); x[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
                if (EPI == EPI_SILU) {
                    st16(C2 + gm * d.ldc2 + gn, v);                                    // the rounded pre-activation, kept for backward
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = silu_f(x[e]);
                } else {
                    const uint4 rp = rq[has_res ? ip : 0][j];
                    const uint32_t rw[4] = {rp.x, rp.y, rp.z, rp.w};
                    float r[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { r[2 * e] = __uint_as_float(rw[e] << 16); r[2 * e + 1] = __uint_as_float(rw[e] & 0xffff0000u); }
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = EPI == EPI_RES ? x[e] + r[e] : x[e] * dsilu_f(r[e]);
                }
                uint4 o;
                o.x = pack2(y[0], y[1]); o.y = pack2(y[2], y[3]); o.z = pack2(y[4], y[5]); o.w = pack2(y[6], y[7]);
                st16(Cout + gm * d.ldc + gn, o);
            }
        };
        if (FINAL) {
            uint4 v[NU];
#pragma unroll
            for (int ip = 0; ip < NB16 / 2; ++ip)
#pragma unroll
                for (int j = 0; j < MB16; ++j) park(ip, j, sbase + (ip * MB16 + j) * C::SCR_WAVE);
#pragma unroll
            for (int u = 0; u < NU; ++u) v[u] = *reinterpret_cast<const uint4*>(sbase + u * C::SCR_WAVE + r_off);
#pragma unroll
            for (int ip = 0; ip < NB16 / 2; ++ip)
#pragma unroll
                for (int j = 0; j < MB16; ++j) finish(ip, j, v[ip * MB16 + j]);
        } else {
#pragma unroll
            for (int ip = 0; ip < NB16 / 2; ++ip)
#pragma unroll
                for (int j = 0; j < MB16; ++j) {
                    park(ip, j, sbase);
                    finish(ip, j, *reinterpret_cast<const uint4*>(sbase + r_off));
                }
        }
    };
    constexpr int NST = (EPI == EPI_SILU ? 2 : 1) * (NB16 / 2) * MB16;     // stores a wave's epilogue leaves in flight
    static_assert(NST + PA < 64 && NST + PB < 64, "the counted waits behind an epilogue must fit vmcnt");

    // ---- prologue: the launch's bias vector (waves 0 .. N/256-1, one 1-KiB piece each); weights of k-tile 0 (waves 0-3); tokens of k-tiles 0, 1
    if (bias && wave * 256 + lane * 4 < d.N) glds16(bias + wave * 256 + lane * 4, smem + C::BIAS_OFF + wave * 1024);
    stage_next();
    if (grp && G > 1) { stage_next(); wait_vm<PA>(); }       // k-tile 0 has landed; the token pieces of k-tile 1 may still be in flight
    else wait_vm<0>();
    phase_barrier();
    PP_TL(1, 0);
#ifdef PP_ABLATION
    if (tid == 0 && blockIdx.x < 256) { g_pp_stamps[blockIdx.x * 4] = __builtin_amdgcn_s_memtime(); g_pp_stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    if (!grp) {
        // ================= waves 0-3: L(g) at segment 2g, C(g) at 2g + 1; they stage the weight rows of k-tile g + 1 in L(g) =================
        int tile = blockIdx.x, kt = 0;
        for (int g = 0; g < G; ++g) {
            const bool ep = kt == 0 && g > 0;                // the previous tile's accumulators are still in the registers
            if (!has_res) read_frags();
            if (g + 1 < G) stage_next();                     // ahead of the epilogue's stores: see the wait below
            if (ep && !(abl & 8)) epilogue(tile - (int)gridDim.x, std::false_type{});
            if (has_res) read_frags();                       // tails with a second operand: its prefetch registers and the fragments are not live together
            wait_lgkm0();                                    // fragments in registers: the slots may be overwritten behind the barrier
            phase_barrier();
            if (kt == 0) zero_acc();
            mfma_phase();
            if (ep) wait_vm<NST>(); else wait_vm<0>();       // weights of k-tile g + 1 have landed (only this tile's epilogue stores may be younger)
            phase_barrier();
            if (++kt == nk) { kt = 0; tile += gridDim.x; }
        }
#ifdef PP_ABLATION
        if (tid == 0 && blockIdx.x < 256) { g_pp_stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime(); g_pp_stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
#endif
        PP_TL(2, 0);
        if (d.final_ring) epilogue(tile - (int)gridDim.x, std::true_type{}); else epilogue(tile - (int)gridDim.x, std::false_type{});
        PP_TL(3, 0);
#ifdef PP_ABLATION
        wait_vm<0>();
#endif
        PP_TL(4, 0);
        // no barrier behind it: waves 4-7 run one segment longer, and nothing they still do touches what these waves touch -- their last MFMA phase
        // works from registers, and every wave's final epilogue stays inside its own 16 KB of the (by now quiet) rings.  (Round 4 first had a
        // closing barrier here and at the end of the last MFMA phase of waves 4-7: their final epilogue waited for this one to finish -- 3 us
        // of a 27 us launch, tools/pp_timeline.py.)
    } else {
        // ================= waves 4-7: L(g) at segment 2g + 1, C(g) at 2g + 2; they stage the token rows of k-tile g + 2 in L(g) =================
        phase_barrier();                        // segment 0: nothing to do yet
        int tile = blockIdx.x, kt = 0;
        bool ep_prev = false;
        for (int g = 0; g < G; ++g) {
            const bool ep = kt == 0 && g > 0;
            const bool st = g + 2 < G;
            if (!has_res) read_frags();
            if (st) stage_next();
            if (ep && !(abl & 8)) epilogue(tile - (int)gridDim.x, std::false_type{});
            if (has_res) read_frags();
            // tokens of k-tile g + 1 (staged in L(g-1), or the prologue) have landed; younger: an epilogue's stores of L(g-1) or of this phase
            // (never both: a tile has at least two k-tiles) and the pieces just staged
            if (ep || ep_prev) { if (st) wait_vm<NST + PA>(); else wait_vm<NST>(); }
            else { if (st) wait_vm<PA>(); else wait_vm<0>(); }
            ep_prev = ep;
            wait_lgkm0();
            phase_barrier();
            if (kt == 0) zero_acc();
            mfma_phase();
            if (g + 1 < G) phase_barrier();     // (the last one would only wait for waves 0-3 to finish their final epilogue)
            if (++kt == nk) { kt = 0; tile += gridDim.x; }
        }
        PP_TL(5, 256);
        if (d.final_ring) epilogue(tile - (int)gridDim.x, std::true_type{}); else epilogue(tile - (int)gridDim.x, std::false_type{});
        PP_TL(6, 256);
#ifdef PP_ABLATION
        wait_vm<0>();
#endif
        PP_TL(7, 256);
    }
}

typedef Cfg<256, 192> Pp192;
typedef Cfg<256, 128> Pp128;

inline int pick(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K < 2 * BK || M % 256 || K % BK) return 0;
    // the bias vector: one 1-KiB DMA piece per wave (8 x 256 floats) into the LDS left over by the rings
    if (N % 192 == 0) return N <= (Pp192::BIAS_MAX_N < 2048 ? Pp192::BIAS_MAX_N : 2048) ? 192 : 0;
    if (N % 128 == 0) return N <= (Pp128::BIAS_MAX_N < 2048 ? Pp128::BIAS_MAX_N : 2048) ? 128 : 0;
    return 0;
}

template <typename C, int EPI>
int launch_epi(const void* A, const void* B, void* Cout, const float* bias, const void* res, void* C2, const Dims& d, hipStream_t s)
{
    void (*k)(const bf16_t*, const bf16_t*, bf16_t*, const float*, const bf16_t*, bf16_t*, Dims) = gemm_pp_kernel<C, EPI>;
#ifdef PP_ABLATION
    if (EPI == EPI_NONE) {
        switch (g_pp_ablate) {
        case 1: k = gemm_pp_kernel<C, EPI_NONE, 1>; break;
        case 2: k = gemm_pp_kernel<C, EPI_NONE, 2>; break;
        case 3: k = gemm_pp_kernel<C, EPI_NONE, 3>; break;
        case 4: k = gemm_pp_kernel<C, EPI_NONE, 4>; break;
        case 5: k = gemm_pp_kernel<C, EPI_NONE, 5>; break;
        case 6: k = gemm_pp_kernel<C, EPI_NONE, 6>; break;
        case 7: k = gemm_pp_kernel<C, EPI_NONE, 7>; break;
        case 8: k = gemm_pp_kernel<C, EPI_NONE, 8>; break;          // no epilogue for tiles in mid-launch (their output is never written): tools/pp_mid_epilogue.py
        default: break;
        }
    }
#endif
    static bool attr_done = false;
#ifdef PP_ABLATION
    attr_done = false;
#endif
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    Dims dd = d;
    dd.tiles = (d.M / C::BM) * (d.N / C::BN);
    dd.final_ring = g_pp_final_ring;
    // one workgroup per CU walking tiles b, b + 256, ... when they divide evenly, else one tile per workgroup
    const int grid = (dd.tiles > 256 && dd.tiles % 256 == 0) ? 256 : dd.tiles;
    hipLaunchKernelGGL(k, dim3(grid), dim3(C::NT), C::LDS, s, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)Cout, bias, (const bf16_t*)res, (bf16_t*)C2, dd);
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

// Tuning hook: 1 (default) = the last epilogue of a launch uses the idle operand rings as scratch for all of a wave's units at once; 0 = unit by unit.
extern "C" int vvae_gemm_pp_final_ring(int on)
{
    pp::g_pp_final_ring = on ? 1 : 0;
    return 0;
}

// Timing-only hook of the -DPP_ABLATION build (tools/pp_ablation.py): bit 0 no DMA behind the prologue, bit 1 no fragment reads, bit 2 no MFMAs
// in the main loop of the plain product (wrong results).  The shipped library ignores it.
extern "C" int vvae_gemm_pp_ablate(int bits)
{
    pp::g_pp_ablate = bits & 15;
    return 0;
}

#ifdef PP_ABLATION
extern "C" int vvae_gemm_pp_stamps(unsigned long long* host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pp::g_pp_stamps), sizeof(unsigned long long) * 256 * 4, 0, hipMemcpyDeviceToHost);
}
extern "C" int vvae_gemm_pp_timeline(unsigned long long* host_out)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pp::g_pp_tl), sizeof(unsigned long long) * 256 * 8, 0, hipMemcpyDeviceToHost);
}
#endif

// 1 if vvae_gemm_pp_bf16 takes this shape (M % 256 == 0, K % 64 == 0, K >= 128, N % 192 == 0 or N % 128 == 0, N <= 1536 resp. 2048 (the bias
// vector lives in LDS), 16-byte aligned pitches that keep a tile's panels inside 32-bit element offsets).
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
        ((uintptr_t)C % 16) || (bias && ((uintptr_t)bias % 16))) return VVAE_ERR_BAD_ARG;
    if ((epi == pp::EPI_RES || epi == pp::EPI_MUL_DSILU) && (!res || ldr % 8 || ldr < N || ((uintptr_t)res % 16))) return VVAE_ERR_BAD_ARG;
    if (epi == pp::EPI_SILU && (!C2 || ldc2 % 8 || ldc2 < N || ((uintptr_t)C2 % 16))) return VVAE_ERR_BAD_ARG;
    pp::Dims d{M, N, K, lda, ldb, ldc, ldr, ldc2, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    if (pp::pick(M, N, K) == 192) return pp::launch<pp::Pp192>(A, B, C, bias, res, C2, d, epi, s);
    return pp::launch<pp::Pp128>(A, B, C, bias, res, C2, d, epi, s);
}
