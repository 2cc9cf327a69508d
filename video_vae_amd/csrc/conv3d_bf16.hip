// bf16 MFMA fast path for NDHWC Conv3d (SAME, stride 1): forward, data-grad and weight-grad.   gfx950 / CDNA4 only.
//
// Replaces the XLA lowering of nnx.Conv at /root/reference/train/unet.py:13-21 (3x3x3) and :111-113 (3x7x7).
//
// Shape of the problem: tiny channel counts (12..128) at huge spatial extent, so the GEMM N dimension is 1..8 MFMA tiles wide and a
// naive implicit GEMM is bound by operand staging, not by the matrix cores.  Three forward / input-gradient kernels share one packed
// weight layout and one K order:
//   * conv3d_bf16_roll_kernel  -- layers whose K channels fit one chunk (16 / 32: every 256^2 / 128^2 layer, the patch mixer): a
//     workgroup owns a spatial tile and MARCHES over time with the halo planes of frames t-1, t, t+1 in a ring of four LDS slots, one
//     new plane fetched per step; weights resident in registers (or, for the mixer, in LDS);
//   * conv3d_bf16_deep_kernel  -- K = 64 / 128 channels or >= 64 output channels: the same march, the waves of a workgroup splitting
//     (K chunk, output-channel tile, row group), partial sums folded through LDS in chunk order (round 3);
//   * conv3d_bf16_kernel       -- the per-frame form both grew out of (one (n, t) x TH x 16 tile per workgroup, halo staged once per
//     tile, weights re-fetched through L1): kept as the fallback for shapes / extents the marches decline and as the reference the
//     tests compare them with bit for bit (vvae_conv3d_roll_config / vvae_conv3d_deep_config).
// Common to all:
//   * K is ordered (dy | dt, dx, ci): for a fixed kernel row dy the operand fragment read for halo row r serves the KH output rows
//     r-dy that the wave owns, so each LDS fragment read feeds up to KH*NT_W MFMAs instead of 1;
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand and the voxels as B: the accumulator then holds 4 consecutive output
//     channels of one voxel per lane -> 8-byte bf16 stores, contiguous per voxel;
//   * weights are pre-packed once per optimizer step (pack kernels below) into fragment order: a wave's fragment is one coalesced 1-KiB load;
//   * dgrad is the same kernel on weights packed with flipped taps and swapped channel roles;
//   * the marches address global memory through buffer descriptors (out-of-volume = out-of-range offset: zeros / dropped store), so
//     their loops are branch-free and the compiler's s_waitcnt are counted ones (round 3).
// The weight gradient (conv3d_wgrad_bf16_kernel) is described at its definition.
#include "common.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Buffer-descriptor access (hardware range check): a load whose offset lies outside the descriptor returns zeros, a store there is
// dropped.  The halo stagers and epilogues below therefore issue EVERY load / store unconditionally, with the offset of an out-of-volume
// voxel set to OOB: no exec-masked branches in the march loop, so the compiler can COUNT the outstanding memory operations and waits
// for the halo plane with s_waitcnt vmcnt(stores behind it) instead of vmcnt(0) -- with predicated loads it lost count, and every step
// of the march waited for the previous step's output stores to drain (and for the second plane in flight) before it parked its plane.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned OOB = 0x80000000u;                         // descriptors span < 2 GiB (checked by the launchers)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
__device__ __forceinline__ void buf_store8(__amdgpu_buffer_rsrc_t r, unsigned off, uint2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)off, 0, 0);
}

template <int CKB_, int KT_, int KH_, int KW_, int MT_W_, int NT_W_, int WM_, int WN_>
struct ConvCfg {
    static constexpr int CKB = CKB_, KT = KT_, KH = KH_, KW = KW_, MT_W = MT_W_, NT_W = NT_W_, WM = WM_, WN = WN_;
    static constexpr int TH = MT_W * WM, TW = 16;
    static constexpr int HR = TH + KH - 1, WR = TW + KW - 1;
    static constexpr int PITCH = CKB == 16 ? 32 : 96;          // bytes per halo voxel in LDS
    static constexpr int SLAB_K = KT * KW * CKB;               // K per kernel row dy and channel chunk
    static constexpr int KSTEPS = (SLAB_K + 31) / 32;
    static constexpr int NVOX = KT * HR * WR;
    static constexpr int LDS_BYTES = NVOX * PITCH;
    static constexpr int CO_BLK = 16 * NT_W * WN;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(CKB == 16 || CKB == 32, "channel chunk");
};

struct BfDims { int N, T, H, W, CK, CO, tiles_h, tiles_w; };
// 1-D launch of nblk x nsub workgroups where the nsub workgroups of a block index share operand tiles (the channel blocks of one spatial
// tile re-read the same X planes): blockIdx round-robins over the 8 XCDs, so the nsub mates get CONSECUTIVE slots of ONE XCD -- dispatched
// together, behind one private L2 -- instead of being nblk slots apart (then the shared planes came from HBM once per channel block).
__device__ __forceinline__ void block_and_sub(int nblk, int nsub, int& blk, int& sub)
{
    const int L = blockIdx.x;
    if ((nblk & 7) == 0) { const int j = L >> 3; blk = (j / nsub) * 8 + (L & 7); sub = j % nsub; }
    else { blk = L / nsub; sub = L % nsub; }
}
// bytes a (voxel, channel) tensor with row pitch ld spans from its first element (the extent of its buffer descriptor)
inline __host__ __device__ long span_bytes(long vox, int ld, int ch) { return ((vox - 1) * ld + ch) * 2; }

// Optional second tensor on either side of a single-chunk layer (the decoder's concat([up, skip]) at 16 + 16 channels, reference
// train/unet.py:79, without the joint buffer: a producer that writes a 32-byte channel half of 64-byte voxels runs at a third of the
// HBM rate, tools/convt_pitch_probe.py).  Input channels >= xsplit come from x2 (channel c at x2[c - xsplit]); produced channels
// >= ysplit go to y2.  x2 / y2 == nullptr: one tensor.  Splits are multiples of 8 (input) / 16 (output) channels.
struct Split2 { const bf16_t* x2; int ldx2, xsplit; bf16_t* y2; int ldy2, ysplit; };
constexpr Split2 NO_SPLIT{nullptr, 0, 0, nullptr, 0, 0};

// Packed weight layout (uint4 = 8 bf16 per lane): [chunk][dy][kstep][co_tile][lane]
//   lane l: co = co_tile*16 + (l & 15), k = 32*kstep + 8*(l >> 4) + e,  slot = k / CKB -> (dt, dx), ci = chunk*CKB + k % CKB
// dgrad=0: value = w[dt][dy][dx][ci][co]                (K channels = Cin,  produced = Cout)
// dgrad=1: value = w[KT-1-dt][KH-1-dy][KW-1-dx][co][ci] (K channels = Cout, produced = Cin; "co" indexes Cin here)
// CR < CKB ("real" K channels): the tensor carries CKB channels per voxel of which only the first CR are not padding; K then runs
// over (dt, dx, ci < CR) -- for the 12-channel patch mixer 3 * 7 * 12 = 252 -> 8 k-steps instead of 3 * 7 * 16 = 336 -> 11.
template <int CKB, int KT, int KH, int KW, int CR = CKB>
__device__ __forceinline__ void pack_fragment(const float* __restrict__ w, uint4* __restrict__ wp, int Cin, int Cout, int dgrad, long i)
{
    constexpr int KSTEPS = (KT * KW * CR + 31) / 32;
    const int CO = dgrad ? Cin : Cout;
    const int co_tiles = CO / 16;
    const int l = (int)(i & 63); long q = i >> 6;
    const int ct = (int)(q % co_tiles); q /= co_tiles;
    const int j = (int)(q % KSTEPS); q /= KSTEPS;
    const int dy = (int)(q % KH); const int chunk = (int)(q / KH);
    const int co = ct * 16 + (l & 15);
    uint32_t pk[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 32 * j + 8 * (l >> 4) + e;
        const int slot = k / CR, ci = chunk * CKB + k % CR;
        float v = 0.f;
        if (slot < KT * KW) {
            const int dt = slot / KW, dx = slot % KW;
            if (!dgrad) v = w[((((long)dt * KH + dy) * KW + dx) * Cin + ci) * Cout + co];
            else v = w[((((long)(KT - 1 - dt) * KH + (KH - 1 - dy)) * KW + (KW - 1 - dx)) * Cin + co) * Cout + ci];
        }
        const uint32_t b = f2bf(v);
        if (e & 1) pk[e >> 1] |= b << 16; else pk[e >> 1] = b;
    }
    wp[i] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
}

template <int CKB, int KT, int KH, int KW, int CR = CKB>
__global__ void pack_weights_kernel(const float* __restrict__ w, uint4* __restrict__ wp, int Cin, int Cout, int dgrad)
{
    constexpr int KSTEPS = (KT * KW * CR + 31) / 32;
    const int CK = dgrad ? Cout : Cin, CO = dgrad ? Cin : Cout;
    const long total = (long)(CK / CKB) * KH * KSTEPS * (CO / 16) * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        pack_fragment<CKB, KT, KH, KW, CR>(w, wp, Cin, Cout, dgrad, i);
}

// Grouped form: every conv layer of a network (forward and flipped input-gradient packings) in ONE launch -- weights change once
// per optimizer step, and 28 five-microsecond launches per step were pure launch latency.  Blocks [block_start_e,
// block_start_{e+1}) belong to entry e, 256 packed 16-byte fragments-lanes each.  variant: 0 = 3x7x7 / 16-channel chunks,
// 1 = 3x3x3 / 16, 2 = 3x3x3 / 32, 3 = 3x7x7 / 16-channel voxels with 12 real K channels.
constexpr int PACK_MAX = 64;
struct PackEntry { const float* w; uint4* wp; long total; int Cin, Cout, variant, dgrad, block_start; };
struct PackArgs { PackEntry e[PACK_MAX]; int n; };

__global__ __launch_bounds__(256) void pack_weights_grouped_kernel(PackArgs g)
{
    int ei = 0;
    for (int i = 1; i < g.n; ++i) ei = (int)blockIdx.x >= g.e[i].block_start ? i : ei;
    const PackEntry& E = g.e[ei];
    const long i = ((long)((int)blockIdx.x - E.block_start)) * 256 + threadIdx.x;
    if (i >= E.total) return;
    if (E.variant == 0) pack_fragment<16, 3, 7, 7>(E.w, E.wp, E.Cin, E.Cout, E.dgrad, i);
    else if (E.variant == 3) pack_fragment<16, 3, 7, 7, 12>(E.w, E.wp, E.Cin, E.Cout, E.dgrad, i);
    else if (E.variant == 1) pack_fragment<16, 3, 3, 3>(E.w, E.wp, E.Cin, E.Cout, E.dgrad, i);
    else pack_fragment<32, 3, 3, 3>(E.w, E.wp, E.Cin, E.Cout, E.dgrad, i);
}

// Stage a KT x HR x WR halo of 16-byte channel parts into LDS, zero-filled outside the volume.  Every thread owns one
// (column, part) of the tile and walks the (plane, row) pairs with a fully unrolled, predicated loop: the loop-invariant
// column address, bounds and LDS offset are computed once, each step is a few integer adds + one 16-byte load, and the
// compiler can issue all loads before the first LDS store.  (The div/mod-per-element form of this loop cost 2-3x the
// MFMA time of the small-channel layers.)   src already points at the first channel of the chunk.
template <int NTHREADS, int KT, int HR, int WR, int PARTS, int PITCH>
__device__ __forceinline__ void stage_halo(const bf16_t* __restrict__ src, int ld, unsigned char* __restrict__ lds, int n, int t0, int h0,
                                           int w0, int T, int H, int W, int tid)
{
    constexpr int ROW_ITEMS = WR * PARTS;
    static_assert(ROW_ITEMS <= NTHREADS, "a halo row must fit one pass");
    constexpr int RPP = NTHREADS / ROW_ITEMS;             // (plane, row) pairs per pass
    constexpr int ROWS = KT * HR;
    constexpr int ITERS = (ROWS + RPP - 1) / RPP;
    const int rip = tid / ROW_ITEMS, item = tid - rip * ROW_ITEMS;
    const int wc = item / PARTS, part = item - wc * PARTS;
    const int wi = w0 + wc;
    const bool col_ok = rip < RPP && (unsigned)wi < (unsigned)W;
    int dt = rip / HR, hr = rip - dt * HR;
    const bf16_t* col = src + (long)wi * ld + part * 8;
    unsigned char* l = lds + (rip * WR + wc) * PITCH + part * 16;
    const long plane = (long)W * ld;
    uint4 v[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int ti = t0 + dt, hi = h0 + hr;
        const bool ok = col_ok && (rip + it * RPP) < ROWS && (unsigned)ti < (unsigned)T && (unsigned)hi < (unsigned)H;
        v[it] = make_uint4(0, 0, 0, 0);
        if (ok) v[it] = *reinterpret_cast<const uint4*>(col + (((long)n * T + ti) * H + hi) * plane);
        hr += RPP;
#pragma unroll
        for (int k = 0; k < (RPP + HR - 1) / HR; ++k)
            if (hr >= HR) { hr -= HR; ++dt; }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
        if (rip < RPP && (rip + it * RPP) < ROWS) *reinterpret_cast<uint4*>(l + it * RPP * WR * PITCH) = v[it];
}

template <class C>
__global__ __launch_bounds__(256) void conv3d_bf16_kernel(const bf16_t* __restrict__ x, int ldx, const uint4* __restrict__ wp,
                                                          const float* __restrict__ bias, bf16_t* __restrict__ y, int ldy, BfDims d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CKB = C::CKB, KT = C::KT, KH = C::KH, KW = C::KW, MT_W = C::MT_W, NT_W = C::NT_W;
    constexpr int HR = C::HR, WR = C::WR, PITCH = C::PITCH, KSTEPS = C::KSTEPS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int r = lane & 15, g = lane >> 4;

    // ---- tile decode: XCD-aware remap so that time-neighbours (which share halo planes) share an L2 ----
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    int tt = bid % d.T; int q = bid / d.T;
    const int tw = q % d.tiles_w; q /= d.tiles_w;
    const int th = q % d.tiles_h; const int n = q / d.tiles_h;
    const int h0 = th * C::TH, w0 = tw * C::TW;

    const int co_tiles = d.CO / 16;
    const int ct0 = blockIdx.y * (NT_W * C::WN) + wn * NT_W;     // first output-channel tile of this wave

    f32x4 acc[MT_W][NT_W];
#pragma unroll
    for (int m = 0; m < MT_W; ++m)
#pragma unroll
        for (int i = 0; i < NT_W; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int chunks = d.CK / CKB;
    for (int chunk = 0; chunk < chunks; ++chunk) {
        if (chunk) __syncthreads();
        // ---- stage the halo tile of this channel chunk (zero fill outside the volume) ----
        stage_halo<256, KT, HR, WR, CKB / 8, PITCH>(x + chunk * CKB, ldx, smem, n, tt - KT / 2, h0 - KH / 2, w0 - KW / 2, d.T, d.H, d.W, tid);
        __syncthreads();

        const uint4* wchunk = wp + (long)chunk * KH * KSTEPS * co_tiles * 64;
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j) {
            // weight fragments for the KH kernel rows of this k-step
            bf16x8 wf[KH][NT_W];
#pragma unroll
            for (int dy = 0; dy < KH; ++dy)
#pragma unroll
                for (int i = 0; i < NT_W; ++i)
                    wf[dy][i] = __builtin_bit_cast(bf16x8, wchunk[((long)(dy * KSTEPS + j) * co_tiles + ct0 + i) * 64 + lane]);
            // per-lane halo offset of this k-step: slot -> (dt, dx), 8-channel group
            int off;
            if (CKB == 32) {
                constexpr int dummy = 0; (void)dummy;
                const int slot = j < KT * KW ? j : 0;
                off = ((slot / KW) * HR * WR + (slot % KW)) * PITCH + 16 * g;
            } else {
                int slot = 2 * j + (g >> 1);
                if (slot >= KT * KW) slot = 0;
                off = ((slot / KW) * HR * WR + (slot % KW)) * PITCH + 16 * (g & 1);
            }
            const unsigned char* base = smem + off + r * PITCH + (wm * MT_W) * WR * PITCH;
#pragma unroll
            for (int hr = 0; hr < MT_W + KH - 1; ++hr) {
                const bf16x8 xf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + hr * WR * PITCH));
#pragma unroll
                for (int dy = 0; dy < KH; ++dy) {
                    const int m = hr - dy;
                    if (m >= 0 && m < MT_W) {
#pragma unroll
                        for (int i = 0; i < NT_W; ++i)
                            acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[dy][i], xf, acc[m][i], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: D[row = co 4g+j][col = voxel r]; lane stores 4 consecutive channels (8 bytes) of one voxel ----
    const int wo = w0 + r;
#pragma unroll
    for (int m = 0; m < MT_W; ++m) {
        const int ho = h0 + wm * MT_W + m;
        if (ho >= d.H || wo >= d.W) continue;
        const long v = (((long)n * d.T + tt) * d.H + ho) * d.W + wo;
#pragma unroll
        for (int i = 0; i < NT_W; ++i) {
            const int co = (ct0 + i) * 16 + 4 * g;
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
            if (bias) { b0 = bias[co]; b1 = bias[co + 1]; b2 = bias[co + 2]; b3 = bias[co + 3]; }
            uint2 o;
            o.x = (uint32_t)f2bf(acc[m][i][0] + b0) | ((uint32_t)f2bf(acc[m][i][1] + b1) << 16);
            o.y = (uint32_t)f2bf(acc[m][i][2] + b2) | ((uint32_t)f2bf(acc[m][i][3] + b3) << 16);
            *reinterpret_cast<uint2*>(y + v * ldy + co) = o;
        }
    }
}

// ---- configuration table ---------------------------------------------------------------------------------
//                 CKB KT KH KW MT_W NT_W WM WN
typedef ConvCfg<16, 3, 3, 3, 4, 1, 4, 1> C333_k16_o16;     // 16 -> 16            TH 16, LDS 31 KB
typedef ConvCfg<32, 3, 3, 3, 2, 1, 4, 1> C333_k32_o16;     // 32,64.. -> 16       TH 8,  LDS 52 KB
typedef ConvCfg<16, 3, 3, 3, 2, 2, 4, 1> C333_k16_o32;     // 16 -> 32            TH 8,  LDS 17 KB
typedef ConvCfg<32, 3, 3, 3, 2, 2, 4, 1> C333_k32_o32;     // 32,64.. -> 32       TH 8
typedef ConvCfg<16, 3, 3, 3, 4, 2, 2, 2> C333_k16_o64;     // 16 -> 64k           TH 8
typedef ConvCfg<32, 3, 3, 3, 4, 2, 2, 2> C333_k32_o64;     // 32.. -> 64k         TH 8, 64 output channels per workgroup
typedef ConvCfg<16, 3, 7, 7, 2, 1, 4, 1> C377_k16_o16;     // 3x7x7 patch mixer   TH 8,  LDS 30 KB

template <class C>
int launch_cfg(const bf16_t* x, int ldx, const uint4* wp, const float* bias, bf16_t* y, int ldy, BfDims d, hipStream_t s)
{
    d.tiles_h = ceil_div(d.H, C::TH);
    d.tiles_w = ceil_div(d.W, C::TW);
    dim3 grid((unsigned)((long)d.N * d.T * d.tiles_h * d.tiles_w), d.CO / C::CO_BLK);
    auto k = conv3d_bf16_kernel<C>;
    static bool attr_done = false;                 // once per instantiation: keeps the launch path free of non-stream calls
    if (C::LDS_BYTES > 65536 && !attr_done) {      // (hipGraph capture of the training step replays only stream work)
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(256), C::LDS_BYTES, s, x, ldx, wp, bias, y, ldy, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

template <int CKB, int KT, int KH, int KW, int CR = CKB>
int launch_pack(const float* w, uint4* wp, int Cin, int Cout, int dgrad, hipStream_t s)
{
    constexpr int KSTEPS = (KT * KW * CR + 31) / 32;
    const int CK = dgrad ? Cout : Cin, CO = dgrad ? Cin : Cout;
    const long total = (long)(CK / CKB) * KH * KSTEPS * (CO / 16) * 64;
    long blocks = (total + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((pack_weights_kernel<CKB, KT, KH, KW, CR>), dim3((unsigned)blocks), dim3(256), 0, s, w, wp, Cin, Cout, dgrad);
    VVAE_LAUNCH_CHECK();
    return 0;
}

inline int chunk_of(int CK) { return (CK % 32 == 0) ? 32 : 16; }

inline size_t packed_bytes(int CK, int CO, int kt, int kh, int kw, int k_real = 0)
{
    const int ckb = chunk_of(CK);
    const int ksteps = (kt * kw * (k_real ? k_real : ckb) + 31) / 32;
    return (size_t)(CK / ckb) * kh * ksteps * (CO / 16) * 64 * 16;
}

}  // namespace

extern "C" size_t vvae_conv3d_wgrad_bf16_ws_bytes(int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw);

namespace {
bool roll_enabled();
int launch_roll_any(const bf16_t* x, int ldx, const uint4* wp, const float* bias, bf16_t* y, int ldy, BfDims d, int kh, hipStream_t s,
                    float* gn_part = nullptr, int gn_groups = 0, Split2 sp = NO_SPLIT, int k_real = 0);
// flags of the pack / forward entry points: bit 0 = input gradient, bits 8-15 = real K channels of a padded layer (0 = all).  Honoured
// for the 3x7x7 mixer with 12 real channels on the rolling kernel; anything else packs and multiplies the padded product.
inline int real_k(int flags, int kh) { return (kh == 7 && ((flags >> 8) & 0xff) == 12 && roll_enabled()) ? 12 : 0; }
int roll_gn_blocks_any(BfDims d, int kh, int groups);
int launch_deep_any(const bf16_t* x, int ldx, const uint4* wp, const float* bias, bf16_t* y, int ldy, BfDims d, hipStream_t s, float* gn_part = nullptr,
                    int gn_groups = 0);
int deep_gn_blocks_any(BfDims d, int kh, int groups);
}

// which: 0 fwd (K = Cin, produced = Cout), 1 dgrad (K = Cout, produced = Cin), 2 wgrad (ld_in = ldx, ld_out = lddy).
extern "C" int vvae_conv3d_bf16_supported(int Cin, int Cout, int kt, int kh, int kw, int ld_in, int ld_out, int which, int flags)
{
    (void)flags;
    if (which == 2) {
        if (Cin % 16 || Cout % 16 || ld_in % 8 || ld_out % 8) return 0;
        if (kt == 3 && kh == 3 && kw == 3) return 1;
        return (kt == 3 && kh == 7 && kw == 7 && Cin == 16 && Cout == 16) ? 1 : 0;
    }
    const int CK = which == 1 ? Cout : Cin, CO = which == 1 ? Cin : Cout;
    if (CK % 16 || CO % 16 || ld_in % 8 || ld_out % 4) return 0;
    if (kt == 3 && kh == 3 && kw == 3) return (CO == 16 || CO == 32 || CO % 64 == 0) ? 1 : 0;
    if (kt == 3 && kh == 7 && kw == 7) return (CK == 16 && CO == 16) ? 1 : 0;
    return 0;
}

extern "C" size_t vvae_conv3d_bf16_ws_bytes(int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int which)
{
    if (which == 2) return vvae_conv3d_wgrad_bf16_ws_bytes(N, T, H, W, Cin, Cout, kt, kh, kw);
    if (!vvae_conv3d_bf16_supported(Cin, Cout, kt, kh, kw, 8, 8, which, 0)) return 0;
    const int CK = which == 1 ? Cout : Cin, CO = which == 1 ? Cin : Cout;
    return packed_bytes(CK, CO, kt, kh, kw);
}

// Pack the Flax kernel w (kt,kh,kw,Cin,Cout) fp32 into the bf16 fragment order the kernel streams (see pack_weights_kernel).
// ws must hold vvae_conv3d_bf16_ws_bytes(.., which = dgrad) bytes.  Weights change once per optimizer step, so a caller
// may pack once and pass prepacked = 1 to every vvae_conv3d_fwd_bf16 call of that step.
extern "C" int vvae_conv3d_pack_bf16(const float* w, void* ws, size_t ws_bytes, int Cin, int Cout, int kt, int kh, int kw,
                                     int flags, void* stream)
{
    const int dgrad = flags & 1, kr = real_k(flags, kh);
    if (!w || !ws || ((uintptr_t)ws % 16)) return VVAE_ERR_BAD_ARG;
    if (!vvae_conv3d_bf16_supported(Cin, Cout, kt, kh, kw, 8, 8, dgrad ? 1 : 0, 0)) return VVAE_ERR_BAD_ARG;
    const int CK = dgrad ? Cout : Cin, CO = dgrad ? Cin : Cout;
    if (ws_bytes < packed_bytes(CK, CO, kt, kh, kw, kr)) return VVAE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (kh == 7 && kr == 12) return launch_pack<16, 3, 7, 7, 12>(w, (uint4*)ws, Cin, Cout, dgrad, s);
    if (kh == 7) return launch_pack<16, 3, 7, 7>(w, (uint4*)ws, Cin, Cout, dgrad, s);
    if (chunk_of(CK) == 16) return launch_pack<16, 3, 3, 3>(w, (uint4*)ws, Cin, Cout, dgrad, s);
    return launch_pack<32, 3, 3, 3>(w, (uint4*)ws, Cin, Cout, dgrad, s);
}

// n <= 64 packings in one launch: entry i packs w[i] (kt = 3, kh[i] = kw[i] in {3, 7}, Cin[i] -> Cout[i]) for the forward
// (dgrad[i] = 0) or the input-gradient (1) kernel into ws[i] (>= vvae_conv3d_bf16_ws_bytes(.., which = dgrad[i]) bytes, 16-byte
// aligned).  Host arrays of device pointers / ints.  A caller packs once per optimizer step and passes prepacked = 1 afterwards.
extern "C" int vvae_conv3d_pack_grouped_bf16(const float* const* w, void* const* ws, const size_t* ws_bytes, const int* Cin, const int* Cout,
                                             const int* kh, const int* dgrad, int n, void* stream)
{
    if (!w || !ws || !ws_bytes || !Cin || !Cout || !kh || !dgrad || n <= 0 || n > PACK_MAX) return VVAE_ERR_BAD_ARG;
    PackArgs g;
    g.n = n;
    long blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!w[i] || !ws[i] || ((uintptr_t)ws[i] % 16) || (kh[i] != 3 && kh[i] != 7)) return VVAE_ERR_BAD_ARG;
        const int dg = dgrad[i] & 1, kr = real_k(dgrad[i], kh[i]);          // dgrad[i]: the flags word of vvae_conv3d_pack_bf16
        if (!vvae_conv3d_bf16_supported(Cin[i], Cout[i], 3, kh[i], kh[i], 8, 8, dg, 0)) return VVAE_ERR_BAD_ARG;
        const int CK = dg ? Cout[i] : Cin[i], CO = dg ? Cin[i] : Cout[i];
        const size_t need = packed_bytes(CK, CO, 3, kh[i], kh[i], kr);
        if (ws_bytes[i] < need) return VVAE_ERR_WORKSPACE;
        const int variant = kh[i] == 7 ? (kr == 12 ? 3 : 0) : (chunk_of(CK) == 16 ? 1 : 2);
        g.e[i] = PackEntry{w[i], (uint4*)ws[i], (long)(need / 16), Cin[i], Cout[i], variant, dg, (int)blocks};
        blocks += ceil_div((long)(need / 16), 256L);
    }
    hipLaunchKernelGGL(pack_weights_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// flags bit 0 = 0: y = conv(x, w) + bias.   bit 0 = 1 (input gradient): "x" is dy (Cout channels), "y" is dx (Cin channels), bias
// ignored.  Bits 8-15: real K channels of a zero-padded layer (see real_k; the SAME flags must be given to the pack call).
// prepacked = 1: ws already holds the packed weights (w may be NULL).
extern "C" int vvae_conv3d_fwd_bf16(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy,
                                    int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int flags,
                                    int prepacked, void* ws, size_t ws_bytes, void* stream)
{
    const int dgrad = flags & 1, kr = real_k(flags, kh);
    if (!x || (!w && !prepacked) || !y || N <= 0 || T <= 0 || H <= 0 || W <= 0) return VVAE_ERR_BAD_ARG;
    if (!vvae_conv3d_bf16_supported(Cin, Cout, kt, kh, kw, ldx, ldy, dgrad ? 1 : 0, 0)) return VVAE_ERR_BAD_ARG;
    if (((uintptr_t)x % 16) || ((uintptr_t)y % 8)) return VVAE_ERR_BAD_ARG;
    const int CK = dgrad ? Cout : Cin, CO = dgrad ? Cin : Cout;
    if (ldx < CK || ldy < CO) return VVAE_ERR_BAD_ARG;
    const size_t need = packed_bytes(CK, CO, kt, kh, kw, kr);
    if (!ws || ws_bytes < need || ((uintptr_t)ws % 16)) return VVAE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    uint4* wp = (uint4*)ws;
    const bf16_t* xp = (const bf16_t*)x;
    bf16_t* yp = (bf16_t*)y;
    const float* bp = dgrad ? nullptr : bias;
    BfDims d{N, T, H, W, CK, CO, 0, 0};
    if (!prepacked) {
        const int rc = vvae_conv3d_pack_bf16(w, ws, ws_bytes, Cin, Cout, kt, kh, kw, flags, stream);
        if (rc) return rc;
    }
    if (roll_enabled() && CK == chunk_of(CK)) {                     // single channel chunk: rolling time-column kernel
        const int rc = launch_roll_any(xp, ldx, wp, bp, yp, ldy, d, kh, s, nullptr, 0, NO_SPLIT, kr);
        if (rc != VVAE_ERR_BAD_ARG) return rc;
    }
    if (kr) return VVAE_ERR_BAD_ARG;                                // packed for the real-channel K order: only the rolling kernel reads it
    if (kh == 3 && chunk_of(CK) == 32) {                            // deeper layers: time march with the waves splitting the K chunks
        const int rc = launch_deep_any(xp, ldx, wp, bp, yp, ldy, d, s);
        if (rc != VVAE_ERR_BAD_ARG) return rc;
    }
    if (kh == 7) return launch_cfg<C377_k16_o16>(xp, ldx, wp, bp, yp, ldy, d, s);
    if (chunk_of(CK) == 16) {
        if (CO == 16) return launch_cfg<C333_k16_o16>(xp, ldx, wp, bp, yp, ldy, d, s);
        if (CO == 32) return launch_cfg<C333_k16_o32>(xp, ldx, wp, bp, yp, ldy, d, s);
        return launch_cfg<C333_k16_o64>(xp, ldx, wp, bp, yp, ldy, d, s);
    }
    if (CO == 16) return launch_cfg<C333_k32_o16>(xp, ldx, wp, bp, yp, ldy, d, s);
    if (CO == 32) return launch_cfg<C333_k32_o32>(xp, ldx, wp, bp, yp, ldy, d, s);
    return launch_cfg<C333_k32_o64>(xp, ldx, wp, bp, yp, ldy, d, s);
}

// Workgroups per sample of the rolling forward kernel for this layer = rows per sample of the GroupNorm partial buffer
// vvae_conv3d_fwd_bf16_gn writes (part[N][blocks][groups][2] floats); 0: the layer does not take that path (use vvae_gn_stats).
extern "C" int vvae_conv3d_gn_blocks(int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int ld_in, int ld_out,
                                     int groups)
{
    if (N <= 0 || T <= 0 || H <= 0 || W <= 0 || !vvae_conv3d_bf16_supported(Cin, Cout, kt, kh, kw, ld_in, ld_out, 0, 0)) return 0;
    const long vox = (long)N * T * H * W;
    if (span_bytes(vox, ld_in, Cin) >= (1L << 31) || span_bytes(vox, ld_out, Cout) >= (1L << 31)) return 0;    // see launch_roll
    BfDims d{N, T, H, W, Cin, Cout, 0, 0};
    if (Cin == chunk_of(Cin)) {
        const int nb = roll_gn_blocks_any(d, kh, groups);
        if (nb > 0) return nb;
    }
    return chunk_of(Cin) == 32 ? deep_gn_blocks_any(d, kh, groups) : 0;   // 64 / 128-channel layers, 32 -> >= 64: the deep rolling kernel
}

// vvae_conv3d_fwd_bf16 (forward only) that also emits the per-group sums of the rounded outputs, so the GroupNorm behind the
// conv (reference train/unet.py:13-23) needs no statistics pass over the tensor: gn_part must hold
// N * vvae_conv3d_gn_blocks(...) * groups * 2 floats and is consumed by vvae_gn_finalize.
extern "C" int vvae_conv3d_fwd_bf16_gn(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy,
                                       int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                                       int prepacked, void* ws, size_t ws_bytes, float* gn_part, int groups, void* stream)
{
    if (!x || (!w && !prepacked) || !y || !gn_part) return VVAE_ERR_BAD_ARG;
    if (vvae_conv3d_gn_blocks(N, T, H, W, Cin, Cout, kt, kh, kw, ldx, ldy, groups) <= 0) return VVAE_ERR_BAD_ARG;
    if (((uintptr_t)x % 16) || ((uintptr_t)y % 8) || ldx < Cin || ldy < Cout) return VVAE_ERR_BAD_ARG;
    const size_t need = packed_bytes(Cin, Cout, kt, kh, kw);
    if (!ws || ws_bytes < need || ((uintptr_t)ws % 16)) return VVAE_ERR_WORKSPACE;
    if (!prepacked) {
        const int rc = vvae_conv3d_pack_bf16(w, ws, ws_bytes, Cin, Cout, kt, kh, kw, 0, stream);
        if (rc) return rc;
    }
    BfDims d{N, T, H, W, Cin, Cout, 0, 0};
    if (Cin == chunk_of(Cin) && roll_gn_blocks_any(d, kh, groups) > 0)
        return launch_roll_any((const bf16_t*)x, ldx, (const uint4*)ws, bias, (bf16_t*)y, ldy, d, kh, (hipStream_t)stream, gn_part, groups);
    return launch_deep_any((const bf16_t*)x, ldx, (const uint4*)ws, bias, (bf16_t*)y, ldy, d, (hipStream_t)stream, gn_part, groups);
}

// Single-chunk layers with a second tensor on one side (Split2 above): which = 0 forward over concat([x, x2], channels) -- x holds the
// first c_split input channels, x2 the other Cin - c_split -- optionally with the GroupNorm partials of vvae_conv3d_fwd_bf16_gn
// (gn_part != NULL); which = 1 input gradient (x = dY) whose Cin produced channels are split between y (first c_split) and y2.
// ws: weights already packed (vvae_conv3d_pack_bf16 / _grouped).  Only layers the rolling kernel takes (vvae_conv3d_cat2_supported:
// K channels and produced channels both 16 or 32, 3x3x3); anything else is VVAE_ERR_BAD_ARG.
extern "C" int vvae_conv3d_cat2_supported(int Cin, int Cout, int c_split, int kt, int kh, int kw)
{
    if (kt != 3 || kh != 3 || kw != 3 || !roll_enabled()) return 0;
    if (!(Cin == 16 || Cin == 32) || !(Cout == 16 || Cout == 32)) return 0;
    return (c_split > 0 && c_split < Cin && c_split % 16 == 0) ? 1 : 0;
}

extern "C" int vvae_conv3d_fwd_bf16_cat2(const void* x, int ldx, const void* x2, int ldx2, const float* bias, void* y, int ldy, void* y2,
                                         int ldy2, int c_split, int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                                         int which, const void* ws, size_t ws_bytes, float* gn_part, int groups, void* stream)
{
    if (!x || !y || !ws || N <= 0 || T <= 0 || H <= 0 || W <= 0 || (which != 0 && which != 1)) return VVAE_ERR_BAD_ARG;
    if (!vvae_conv3d_cat2_supported(Cin, Cout, c_split, kt, kh, kw)) return VVAE_ERR_BAD_ARG;
    const int CK = which ? Cout : Cin, CO = which ? Cin : Cout;
    if (ws_bytes < packed_bytes(CK, CO, kt, kh, kw) || ((uintptr_t)ws % 16)) return VVAE_ERR_WORKSPACE;
    Split2 sp = NO_SPLIT;
    if (which == 0) {
        if (!x2 || y2 || ((uintptr_t)x % 16) || ((uintptr_t)x2 % 16) || ldx % 8 || ldx2 % 8 || ldx < c_split || ldx2 < Cin - c_split ||
            ldy < Cout || ldy % 4 || ((uintptr_t)y % 8)) return VVAE_ERR_BAD_ARG;
        if (gn_part && vvae_conv3d_gn_blocks(N, T, H, W, Cin, Cout, kt, kh, kw, 8, ldy, groups) <= 0) return VVAE_ERR_BAD_ARG;
        sp.x2 = (const bf16_t*)x2; sp.ldx2 = ldx2; sp.xsplit = c_split;
    } else {
        if (!y2 || x2 || gn_part || ((uintptr_t)x % 16) || ldx % 8 || ldx < Cout || ((uintptr_t)y % 8) || ((uintptr_t)y2 % 8) || ldy % 4 ||
            ldy2 % 4 || ldy < c_split || ldy2 < Cin - c_split) return VVAE_ERR_BAD_ARG;
        sp.y2 = (bf16_t*)y2; sp.ldy2 = ldy2; sp.ysplit = c_split;
    }
    BfDims d{N, T, H, W, CK, CO, 0, 0};
    return launch_roll_any((const bf16_t*)x, ldx, (const uint4*)ws, which ? nullptr : bias, (bf16_t*)y, ldy, d, kh, (hipStream_t)stream, gn_part,
                           groups, sp);
}

namespace {
// =============================================================================================== weight gradient
// dW[dt][dy][dx][ci][co] = sum_v X[v + off(dt,dy,dx)][ci] * dY[v][co]          (and dbias[co] = sum_v dY[v][co])
//
// GEMM view per tap: M = ci, N = co, K = voxels (millions).  Both operands are K-major in NDHWC memory, so the MFMA
// fragments (8 consecutive k per lane) come from LDS through ds_read_b64_tr_b16 (hardware 4x16 transpose), with the
// k -> voxel map chosen so that each 32-lane half reads 8 consecutive voxel rows = one conflict-free 256-byte bank row.
//   * one wave owns the KH*KW (or, for the 7x7 mixer, 2*KW) taps of ONE temporal offset dt for ONE 16x16 (ci, co) tile
//     -- 9..14 accumulator tiles that stay in registers for the workgroup's whole life -- and a workgroup is the
//     3 * (CIB/16) * (COB/16) * dy-groups * row-groups waves that share a halo: 3, 12 or 12 light waves (<= 170 VGPRs),
//     so every SIMD of a CU carries the same load (3-wave workgroups holding 36 tiles each left one SIMD in four idle);
//   * workgroups are PERSISTENT: each walks a contiguous run of (n, h-tile, w-tile, t) tiles, staging the X halo
//     (KT x (TH+KH-1) x (32+KW-1) voxels) and the dY tile (TH x 32 voxels) in LDS, so the accumulators are written out
//     once per workgroup (fp32 slab) and a second tiny kernel sums the slabs -- deterministic, no float atomics;
//   * the X fragment read for halo row r and column shift dx serves the KH taps (dy, dx) of output rows r-dy, and the dY
//     fragment of a row serves all of the wave's taps: ~0.5 LDS fragment reads per MFMA instead of 4;
//   * 32-channel X voxels sit at their natural 64-byte pitch with the 32-byte half XORed by (voxel >> 2) & 1 (rows padded to
//     a multiple of 8 voxels so the bit is row-independent): conflict-free transposed reads without the 96-byte pitch that
//     cost a third of the LDS budget.
typedef short s16x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p, int row16_bytes)
{
    // two transposed 4x16 reads: k elements 0..3 from voxel rows (4g+q), 4..7 from rows 16+(4g+q)
    typedef __attribute__((address_space(3))) s16x4v* lds_ptr;
    const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + row16_bytes));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int CIB_, int COB_, int KH_, int KW_, int TH_, int AG_, int RG_, int MINW_, int BPC_, int PD_ = 1>
struct WgCfg {
    static constexpr int PD = PD_;                                           // X planes / dY tiles in flight per workgroup (1 or 2)
    static constexpr int CIB = CIB_, COB = COB_, KT = 3, KH = KH_, KW = KW_, TH = TH_, TW = 32;
    static constexpr int CIT = CIB / 16, COT = COB / 16;
    static constexpr int AG = AG_, AGS = (KH + AG - 1) / AG;                 // kernel rows dy are split into AG groups of AGS rows
    static constexpr int RG = RG_, THR = TH / RG;                            // output rows are split into RG groups of THR rows (own slab each)
    static constexpr int NW = KT * CIT * COT * AG * RG, NTHREADS = 64 * NW;  // one wave per (dt, ci tile, co tile, dy group, row group)
    static_assert(TH % RG == 0 && NW <= 16, "wave grid");
    static constexpr int MINW = MINW_;                                       // waves per SIMD the register allocation must leave room for
    static constexpr int BLOCKS = 256 * BPC_;                                // persistent grid = what the chip holds at once
    static constexpr int HR = TH + KH - 1, WR = TW + KW - 1;
    static constexpr int PX = 2 * CIB, PY = COB == 16 ? 32 : 96;             // 32-channel X voxels: 64-byte pitch, swizzled parts
    static constexpr bool SWX = CIB == 32;
    static constexpr int WRP = SWX ? ((WR + 7) & ~7) : WR;                  // rows a multiple of 8 voxels: the swizzle bit is row-independent
    static constexpr int PLANE = HR * WRP * PX, YBYTES = TH * TW * PY;      // one X halo plane, one dY tile
    static constexpr int LDS_BYTES = 4 * PLANE + 2 * YBYTES;               // ring of 4 planes + double-buffered dY
    static constexpr int SLAB_FLOATS = KT * KH * KW * CIB * COB + COB;     // + dbias partial
};

struct WgDims { int N, T, H, W, CI, CO, tiles_h, tiles_w, ncols, cols_per_block; };

// One HR x WR halo plane of 16-byte channel parts held in registers between its global fetch and its LDS store, so the
// fetch of step t+1 can be in flight while step t computes (register-staged software pipeline).
// 16-byte part index of an LDS voxel, XORed so that every ds_read_b128 lane group of 16 consecutive voxels lands on 16 distinct slots of the
// 256-byte bank row for any tap shift.  Mode 1: 64-byte voxels (32 channels); 2: 128-byte (64 channels); 3: 256-byte (128 channels) -- found by
// exhaustive search over XOR-linear maps of the voxel index against the lane groups of MI355X_MICROARCH.md's LDS table.
template <int MODE> __device__ __forceinline__ int swz_part(int part, int lin) {
    return MODE == 1 ? (part ^ ((lin >> 1) & 2)) : MODE == 2 ? (part ^ (lin & 6)) : MODE == 3 ? (part ^ ((lin & 7) << 1)) : part;
}
// SWCOL: the swizzle is a function of the voxel's COLUMN in the halo row instead of its linear index (conflict freedom only involves the 16
// consecutive voxels of one row, so either works; with the column a reader's swizzle does not depend on the halo row and the row becomes an
// immediate offset).
template <int NTHREADS, int HR, int WR, int PARTS, int PITCH, int SWZ = 0, int WRP = WR, bool SWCOL = false>   // WRP: LDS row pitch in voxels
struct PlaneStager {
    static constexpr int ROW_ITEMS = WR * PARTS;
    static_assert(ROW_ITEMS <= NTHREADS, "a halo row must fit one pass");
    static constexpr int RPP = NTHREADS / ROW_ITEMS;
    static constexpr int ITERS = (HR + RPP - 1) / RPP;
    uint4 v[ITERS];

    __device__ __forceinline__ void fetch(const bf16_t* __restrict__ src, int ld, int n, int t, int h0, int w0, int T, int H, int W, int tid) {
        const int rip = tid / ROW_ITEMS, item = tid - rip * ROW_ITEMS;
        const int wc = item / PARTS, part = item - wc * PARTS;
        const int wi = w0 + wc;
        const bool ok0 = rip < RPP && (unsigned)wi < (unsigned)W && (unsigned)t < (unsigned)T;
        const bf16_t* col = src + (((long)n * T + t) * H * (long)W + wi) * ld + part * 8;
        const long rowstride = (long)W * ld;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int hr = rip + it * RPP, hi = h0 + hr;
            v[it] = make_uint4(0, 0, 0, 0);
            if (ok0 && hr < HR && (unsigned)hi < (unsigned)H) v[it] = *reinterpret_cast<const uint4*>(col + hi * rowstride);
        }
    }
    // The same through a buffer descriptor: every load is issued, out-of-volume pieces with an OOB offset (-> zeros).
    // ``c0``: first channel of the tile within the tensor.
    __device__ __forceinline__ void fetch(__amdgpu_buffer_rsrc_t r, int ld, int c0, int n, int t, int h0, int w0, int T, int H, int W, int tid) {
        const int rip = tid / ROW_ITEMS, item = tid - rip * ROW_ITEMS;
        const int wc = item / PARTS, part = item - wc * PARTS;
        const int wi = w0 + wc;
        const bool ok0 = rip < RPP && (unsigned)wi < (unsigned)W && (unsigned)t < (unsigned)T;
        const unsigned col = (unsigned)((((long)n * T + t) * H * (long)W + wi) * ld + c0 + part * 8) * 2u;
        const unsigned rowstride = (unsigned)W * (unsigned)ld * 2u;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int hr = rip + it * RPP, hi = h0 + hr;
            const bool ok = ok0 && hr < HR && (unsigned)hi < (unsigned)H;
            v[it] = buf_load16(r, ok ? col + (unsigned)hi * rowstride : OOB);
        }
    }
    __device__ __forceinline__ void store(unsigned char* __restrict__ lds, int tid) const {
        const int rip = tid / ROW_ITEMS, item = tid - rip * ROW_ITEMS;
        const int wc = item / PARTS, part = item - wc * PARTS;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int lin = (rip + it * RPP) * WRP + wc;                   // SWZ: part ^= ((voxel >> 2) & 1) << 1 (64-byte voxels)
            const int p = swz_part<SWZ>(part, SWCOL ? wc : lin);
            if (rip < RPP && rip + it * RPP < HR) {
                if (PITCH % 16 == 0) *reinterpret_cast<uint4*>(lds + lin * PITCH + p * 16) = v[it];
                else {                                                     // 24-byte voxels (12 real of 16 channels): 8-byte pieces,
                    unsigned char* q = lds + lin * PITCH + part * 16;      // the 4 padding channels are dropped
                    *reinterpret_cast<uint2*>(q) = make_uint2(v[it].x, v[it].y);
                    if (part == 0) *reinterpret_cast<uint2*>(q + 8) = make_uint2(v[it].z, v[it].w);
                }
            }
        }
    }
};

// =============================================================================================== rolling forward / input gradient
// The same product as conv3d_bf16_kernel for layers whose K channels fit one chunk (CK = 16 or 32: every 256^2 / 128^2 layer
// and the patch mixer), restructured around TIME: a workgroup owns a spatial tile and marches over a run of frames, keeping
// the halo planes of frames t-1, t, t+1 in a ring of four LDS slots.  Each step fetches ONE new plane (t+2) into registers
// while step t multiplies, and parks it after the barrier -- so a plane is read from L2 once per workgroup instead of three
// times, the staging work per output tile drops 3x and its latency hides under the MFMAs of the previous step.  One barrier
// per step (a slot is overwritten two steps after its last reader).
//
// Where the packed weights live decides what bounds the inner loop: re-fetched through the vector L1 (64 B/clk/CU) they cost
// more than the MFMAs they feed, so a march keeps them either in registers (W_REG: <= 27 fragments per wave) or in LDS behind
// the ring (W_LDS: 256 B/clk/CU with ds_read_b128, loaded once per workgroup).  32-channel voxels are stored at their natural
// 64-byte pitch with the 16-byte part index XORed by ((voxel >> 2) & 1) << 1, which keeps every ds_read_b128 lane group
// ({r 0-3, 12-15 | part p} + {r 4-11 | part p^1}) on 16 distinct slots of the 256-byte bank row for any tap shift.
enum { W_REG = 1, W_LDS = 2 };

// CR_ < CKB_: only the first CR_ of a voxel's CKB_ channels are real (the rest is zero padding in memory): LDS voxels are packed to
// 2 CR_ bytes, K runs over (dt, dx, ci < CR_) and a lane's 8-element run is read as two 8-byte halves (a half never straddles a
// frame plane: KW * CR_ is a multiple of 4).
// PD_: halo planes in flight per workgroup (1 or 2).  2 costs one more plane of staging registers (12-20 VGPRs): the 4-wave configurations
// have them, the 8-wave ones (256-register cap, 108 of them resident weights) would spill.
// LA_ > 0 (register-resident weights only): the X fragments of a step form ONE stream of KSTEPS * (MT_W + KH - 1) LDS reads, kept LA_ reads ahead of
// the products that consume them in a rotating window of LA_ + 1 fragments.  Left to itself the compiler issued each read right in front of its
// first product and waited for it (r [lgkmcnt(0)] M M M r [lgkmcnt(0)] ...): 30 exposed LDS round trips per step and wave against 960 cycles of
// matrix work -- the matrix pipes of the 16-channel layers were 38 % busy.
template <int CKB_, int KH_, int KW_, int MT_W_, int NT_W_, int WM_, int WN_, int WMODE_, bool PF_, int CR_ = CKB_, int PD_ = 1, int LA_ = 0>
struct RollCfg {
    static constexpr int CKB = CKB_, KT = 3, KH = KH_, KW = KW_, MT_W = MT_W_, NT_W = NT_W_, WM = WM_, WN = WN_, WMODE = WMODE_, CR = CR_;
    static constexpr bool PF = PF_;                          // request the fragments of k-step j+1 before multiplying k-step j
    static constexpr int PD = PD_, LA = LA_;
    static_assert(PD == 1 || PD == 2, "planes in flight");
    static_assert(LA == 0 || (WMODE_ == W_REG && CR_ == CKB_ && !PF_), "read-ahead window: register weights, whole voxels");
    static constexpr int NTHREADS = 64 * WM * WN;
    static constexpr int TH = MT_W * WM, TW = 16, HR = TH + KH - 1, WR = TW + KW - 1;
    static constexpr int PITCH = 2 * CR;
    static constexpr bool SWZ = CKB == 32;
    static constexpr int KSTEPS = (KT * KW * CR + 31) / 32;
    static_assert(CR == CKB || (CKB == 16 && CR % 4 == 0 && (KW * CR) % 4 == 0), "real-channel form: 16-channel voxels, whole 8-byte pieces");
    static constexpr int PLANE = HR * WR * PITCH;
    static constexpr int CO_T = NT_W * WN, CO_BLK = 16 * CO_T;
    static constexpr int WFRAGS = KH * KSTEPS * CO_T;
    static constexpr int LDS_BYTES = 4 * PLANE + (WMODE == W_LDS ? WFRAGS * 1024 : 0);
    static_assert(CKB == 16 || CKB == 32, "channel chunk");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// GN: the workgroup also emits, per GroupNorm group, the sum and the sum of squares of the (bf16-rounded) outputs it wrote --
// one row of the partial buffer gn_stats would have produced by re-reading the whole tensor: part[n][blk][group][2].
// TWO: a second tensor on one side (Split2): that instantiation keeps per-thread pointers and predicated accesses (a lane's tensor is
// not wave-uniform, a buffer descriptor is); the one-tensor instantiations go through descriptors and have a branch-free march loop.
template <class C, bool GN, bool TWO>
__global__ __launch_bounds__(C::NTHREADS) void conv3d_bf16_roll_kernel(const bf16_t* __restrict__ x, int ldx, const uint4* __restrict__ wp,
                                                                       const float* __restrict__ bias, bf16_t* __restrict__ y, int ldy,
                                                                       BfDims d, int tchunk, float* __restrict__ gn_part, int gn_groups,
                                                                       Split2 sp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CKB = C::CKB, KT = C::KT, KH = C::KH, KW = C::KW, MT_W = C::MT_W, NT_W = C::NT_W;
    constexpr int HR = C::HR, WR = C::WR, PITCH = C::PITCH, KSTEPS = C::KSTEPS, PLANE = C::PLANE, CO_T = C::CO_T;
    constexpr bool WREG = C::WMODE == W_REG;
    const long vox_all = (long)d.N * d.T * d.H * d.W;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, TWO ? 0u : (unsigned)span_bytes(vox_all, ldx, d.CK));
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y, TWO ? 0u : (unsigned)span_bytes(vox_all, ldy, d.CO));
    if (TWO && sp.x2 && (int)(threadIdx.x % (WR * (CKB / 8))) % (CKB / 8) * 8 >= sp.xsplit) {     // this thread stages a 16-byte part of x2
        x = sp.x2 - sp.xsplit;
        ldx = sp.ldx2;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int r = lane & 15, g = lane >> 4;

    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);       // XCD-aware: neighbours share halo columns and planes
    const int nch = (d.T + tchunk - 1) / tchunk;
    const int tc = bid % nch; int q = bid / nch;
    const int tw = q % d.tiles_w; q /= d.tiles_w;
    const int th = q % d.tiles_h; const int n = q / d.tiles_h;
    const int h0 = th * C::TH, w0 = tw * C::TW;
    const int t_beg = tc * tchunk;
    int t_end = t_beg + tchunk;
    if (t_end > d.T) t_end = d.T;

    const int co_tiles = d.CO / 16;
    const int cb0 = blockIdx.y * CO_T;                                     // first output-channel tile of this workgroup
    const int ct0 = cb0 + wn * NT_W;                                       // ... and of this wave

    uint4* wl = reinterpret_cast<uint4*>(smem + 4 * PLANE);               // W_LDS: [dy][kstep][co tile of the block][lane]
    if (!WREG) {
        for (int i = tid; i < C::WFRAGS * 64; i += C::NTHREADS) {
            const int l = i & 63, f = i >> 6;
            const int c = f % CO_T, dyj = f / CO_T;
            wl[i] = wp[((long)dyj * co_tiles + cb0 + c) * 64 + l];
        }
    }
    bf16x8 wreg[WREG ? KSTEPS : 1][KH][NT_W];
    if (WREG) {
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j)
#pragma unroll
            for (int dy = 0; dy < KH; ++dy)
#pragma unroll
                for (int i = 0; i < NT_W; ++i)
                    wreg[j][dy][i] = __builtin_bit_cast(bf16x8, wp[((long)(dy * KSTEPS + j) * co_tiles + ct0 + i) * 64 + lane]);
    }
    float bv[NT_W][4];
#pragma unroll
    for (int i = 0; i < NT_W; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[i][e] = bias ? bias[(ct0 + i) * 16 + 4 * g + e] : 0.f;

    PlaneStager<C::NTHREADS, HR, WR, CKB / 8, PITCH, C::SWZ> sx;
    const int hx = h0 - KH / 2, wx = w0 - KW / 2;
#define FETCH(S, T_) do { if (TWO) (S).fetch(x, ldx, n, (T_), hx, wx, d.T, d.H, d.W, tid); else (S).fetch(rx, ldx, 0, n, (T_), hx, wx, d.T, d.H, d.W, tid); } while (0)
    FETCH(sx, t_beg - 1);                                                  // out-of-range frames come back as zeros
    sx.store(smem + ((t_beg - 1) & 3) * PLANE, tid);
    FETCH(sx, t_beg);
    sx.store(smem + (t_beg & 3) * PLANE, tid);
    // TWO planes in flight per workgroup: plane tt + 1 is parked at the top of step tt, plane tt + 2 is already on its way in the second
    // register set.  With one plane in flight a step could not be shorter than a memory round trip (a 16 -> 16 step is 0.45 us of matrix
    // work against 1-2 us of loaded-HBM latency: the 256^2 layers ran at half the HBM rate with the matrix pipes 38 % busy).
    PlaneStager<C::NTHREADS, HR, WR, CKB / 8, PITCH, C::SWZ> sx2;
    // The march loop below is straight-line code: every fetch and every store is issued in every step (a frame index past the chunk
    // becomes -1 -> all offsets OOB -> zeros without traffic; the stores of step tt are issued by step tt + 1, see flush()).
    FETCH(sx, t_beg + 1);
    if (C::PD == 2) FETCH(sx2, t_beg + 1 < t_end ? t_beg + 2 : -1);
    const int wo = w0 + r;
    const int lin_w = (wm * MT_W) * WR + r;                                // this lane's voxel in the wave's first halo row
    // GroupNorm partials per channel PAIR (a group is an even number of consecutive channels): v_dot2c_f32_bf16 adds the two
    // rounded outputs of a packed word (against 1.0 | 1.0), or their squares, to an fp32 accumulator in one instruction
    float gs[GN ? NT_W : 1][2], gss[GN ? NT_W : 1][2];
#pragma unroll
    for (int i = 0; i < (GN ? NT_W : 1); ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) { gs[i][e] = 0.f; gss[i][e] = 0.f; }
    uint2 held[MT_W][NT_W];
    unsigned hoff[MT_W];
#pragma unroll
    for (int m = 0; m < MT_W; ++m) {
        hoff[m] = OOB;
#pragma unroll
        for (int i = 0; i < NT_W; ++i) held[m][i] = make_uint2(0, 0);
    }
    auto frame = [&](int tt) {

        f32x4 acc[MT_W][NT_W];
#pragma unroll
        for (int m = 0; m < MT_W; ++m)
#pragma unroll
            for (int i = 0; i < NT_W; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // PF: operand fragments of k-step j+1 are requested before the MFMAs of k-step j are issued (the scheduling fences keep
        // the compiler from folding the two buffers back into a depth-2 read/wait/multiply chain).  Worth ~15 % on the mixer,
        // where weights come from LDS too; the 3x3x3 layers sit at the clock-limited MFMA rate without it and prefer the VGPRs.
        constexpr int NX = MT_W + KH - 1;
        bf16x8 xf[2][NX];
        bf16x8 wf[2][KH][NT_W];
        auto request = [&](int j, bf16x8 (&xo)[NX], bf16x8 (&wo)[KH][NT_W]) {
            if (!WREG) {
#pragma unroll
                for (int dy = 0; dy < KH; ++dy)
#pragma unroll
                    for (int i = 0; i < NT_W; ++i)
                        wo[dy][i] = __builtin_bit_cast(bf16x8, wl[((dy * KSTEPS + j) * CO_T + wn * NT_W + i) * 64 + lane]);
            }
            if (C::CR != CKB) {
                // K element k = (dt, dx, ci) flattened with ci < CR: the lane's run 32 j + 8 g .. + 7 as two 8-byte halves; the run past
                // the end of K (zero weights) re-reads the last real one (finite data)
                constexpr int KP = KW * C::CR, KALL = KT * KP;
                const unsigned char* hb[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    int k0 = 32 * j + 8 * g + 4 * hh;
                    if (k0 >= KALL) k0 = KALL - 4;
                    const int dt = (k0 >= KP) + (k0 >= 2 * KP);
                    hb[hh] = smem + ((tt + dt - 1) & 3) * PLANE + lin_w * PITCH + (k0 - dt * KP) * 2;
                }
#pragma unroll
                for (int hr = 0; hr < NX; ++hr) {
                    const uint2 lo = *reinterpret_cast<const uint2*>(hb[0] + hr * WR * PITCH);
                    const uint2 hi = *reinterpret_cast<const uint2*>(hb[1] + hr * WR * PITCH);
                    xo[hr] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                }
                return;
            }
            int slot, part;
            if (CKB == 32) { slot = j < KT * KW ? j : 0; part = g; }
            else { slot = 2 * j + (g >> 1); if (slot >= KT * KW) slot = 0; part = g & 1; }
            const int dt = slot / KW, dx = slot - dt * KW;
            const unsigned char* pbase = smem + ((tt + dt - 1) & 3) * PLANE;
            const int lin0 = lin_w + dx;
#pragma unroll
            for (int hr = 0; hr < NX; ++hr) {
                const int lin = lin0 + hr * WR;
                const int p = C::SWZ ? (part ^ ((lin >> 1) & 2)) : part;
                xo[hr] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(pbase + lin * PITCH + p * 16));
            }
        };
        if (C::LA > 0) {
            constexpr int LA = C::LA > 0 ? C::LA : 1, NR = KSTEPS * NX;
            bf16x8 win[LA + 1];
            auto rd = [&](int idx) {
                const int j = idx / NX, hr = idx - j * NX;
                int slot, part;
                if (CKB == 32) { slot = j < KT * KW ? j : 0; part = g; }
                else { slot = 2 * j + (g >> 1); if (slot >= KT * KW) slot = 0; part = g & 1; }
                const int dt = slot / KW, dx = slot - dt * KW;
                const int lin = lin_w + dx + hr * WR;
                const int p = C::SWZ ? (part ^ ((lin >> 1) & 2)) : part;
                return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(smem + ((tt + dt - 1) & 3) * PLANE + lin * PITCH + p * 16));
            };
#pragma unroll
            for (int i = 0; i < LA && i < NR; ++i) win[i % (LA + 1)] = rd(i);
#pragma unroll
            for (int idx = 0; idx < NR; ++idx) {
                if (idx + LA < NR) win[(idx + LA) % (LA + 1)] = rd(idx + LA);
                const int j = idx / NX, hr = idx - j * NX;
#pragma unroll
                for (int dy = 0; dy < KH; ++dy) {
                    const int m = hr - dy;
                    if (m >= 0 && m < MT_W) {
#pragma unroll
                        for (int i = 0; i < NT_W; ++i)
                            acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[WREG ? j : 0][dy][i], win[idx % (LA + 1)], acc[m][i], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        if (C::PF) request(0, xf[0], wf[0]);
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j) {
            if (C::PF) {
                if (j + 1 < KSTEPS) request(j + 1, xf[(j + 1) & 1], wf[(j + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                request(j, xf[j & 1], wf[j & 1]);
            }
#pragma unroll
            for (int hr = 0; hr < NX; ++hr) {
#pragma unroll
                for (int dy = 0; dy < KH; ++dy) {
                    const int m = hr - dy;
                    if (m >= 0 && m < MT_W) {
#pragma unroll
                        for (int i = 0; i < NT_W; ++i)
                            acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WREG ? wreg[WREG ? j : 0][dy][i] : wf[j & 1][dy][i],
                                                                                xf[j & 1][hr], acc[m][i], 0, 0, 0);
                    }
                }
            }
            if (C::PF) __builtin_amdgcn_sched_barrier(0);
        }
        }
        // D[row = co 4g+j][col = voxel r]; a lane holds 4 consecutive channels (8 bytes) of one voxel.  One tensor: the rounded tile is
        // PARKED (held[], hoff[]) and stored by flush() behind the NEXT step's barrier, in front of that step's plane request -- so the only
        // memory operations younger than a plane's loads when the march waits for them are none, and the stores have a whole step to drain.
#pragma unroll
        for (int m = 0; m < MT_W; ++m) {
            const int ho = h0 + wm * MT_W + m;
            const bool inside = ho < d.H && wo < d.W;                  // outside: the store is issued with an OOB offset and dropped
            const long v = (((long)n * d.T + tt) * d.H + ho) * d.W + wo;
            if (!TWO) hoff[m] = inside ? (unsigned)(v * ldy + ct0 * 16 + 4 * g) * 2u : OOB;
#pragma unroll
            for (int i = 0; i < NT_W; ++i) {
                uint2 o;
                o.x = (uint32_t)f2bf(acc[m][i][0] + bv[i][0]) | ((uint32_t)f2bf(acc[m][i][1] + bv[i][1]) << 16);
                o.y = (uint32_t)f2bf(acc[m][i][2] + bv[i][2]) | ((uint32_t)f2bf(acc[m][i][3] + bv[i][3]) << 16);
                const int c0 = (ct0 + i) * 16 + 4 * g;
                if (TWO) {                                         // produced channels >= ysplit live in the second tensor
                    if (inside) {
                        if (sp.y2 && c0 >= sp.ysplit) *reinterpret_cast<uint2*>(sp.y2 + v * sp.ldy2 + c0 - sp.ysplit) = o;
                        else *reinterpret_cast<uint2*>(y + v * ldy + c0) = o;
                    }
                } else {
                    held[m][i] = o;
                }
                if (GN) {                                          // statistics of what GroupNorm will read: the rounded values
                    const bf16x2 ones = __builtin_bit_cast(bf16x2, 0x3f803f80u);
                    const bf16x2 p0 = __builtin_bit_cast(bf16x2, inside ? o.x : 0u), p1 = __builtin_bit_cast(bf16x2, inside ? o.y : 0u);
                    gs[i][0] = __builtin_amdgcn_fdot2_f32_bf16(p0, ones, gs[i][0], false);
                    gs[i][1] = __builtin_amdgcn_fdot2_f32_bf16(p1, ones, gs[i][1], false);
                    gss[i][0] = __builtin_amdgcn_fdot2_f32_bf16(p0, p0, gss[i][0], false);
                    gss[i][1] = __builtin_amdgcn_fdot2_f32_bf16(p1, p1, gss[i][1], false);
                }
            }
        }
    };
    auto flush = [&]() {                                               // the parked tile of the previous step (OOB offsets before the first)
        if (!TWO) {
#pragma unroll
            for (int m = 0; m < MT_W; ++m)
#pragma unroll
                for (int i = 0; i < NT_W; ++i) buf_store8(ry, hoff[m] == OOB ? OOB : hoff[m] + 32u * i, held[m][i]);
        }
    };
    if (C::PD == 2) {
        for (int tt = t_beg; tt < t_end; tt += 2) {
            sx.store(smem + ((tt + 1) & 3) * PLANE, tid);                   // waits for plane tt + 1 only: plane tt + 2 stays in flight
            __syncthreads();
            flush();
            FETCH(sx, tt + 2 < t_end ? tt + 3 : -1);
            frame(tt);
            if (tt + 1 >= t_end) break;
            sx2.store(smem + ((tt + 2) & 3) * PLANE, tid);
            __syncthreads();
            flush();
            FETCH(sx2, tt + 3 < t_end ? tt + 4 : -1);
            frame(tt + 1);
        }
    } else {
        for (int tt = t_beg; tt < t_end; ++tt) {
            sx.store(smem + ((tt + 1) & 3) * PLANE, tid);
            __syncthreads();
            flush();
            FETCH(sx, tt + 1 < t_end ? tt + 2 : -1);
            frame(tt);
        }
    }
    flush();                                                            // the last step's tile
#undef FETCH
    if (GN) {
        // fold: the 16 voxel lanes of a channel quad (xor-shuffles), then the WM waves that share the channels (LDS), then the
        // channels of each group -- fixed order, no atomics
        static_assert(!GN || C::CO_T * 16 * C::WM * 2 * 4 <= 4 * C::PLANE, "fold scratch fits the ring");
        __syncthreads();                                           // every wave is done with the ring
        float* red = reinterpret_cast<float*>(smem);               // [wm][channel pair][2]
        constexpr int CP = C::CO_BLK / 2;
#pragma unroll
        for (int i = 0; i < NT_W; ++i)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float a = gs[i][e], b = gss[i][e];
                a = butterfly_sum<8, 1>(a); b = butterfly_sum<8, 1>(b);          // the 16 voxel lanes of this channel pair (DPP only)
                if (r == 0) {
                    const int cp = (wn * NT_W + i) * 8 + 2 * g + e;
                    red[(wm * CP + cp) * 2 + 0] = a;
                    red[(wm * CP + cp) * 2 + 1] = b;
                }
            }
        __syncthreads();
        if (tid < gn_groups) {
            const int ppg = CP / gn_groups;                        // pairs per group (roll_gn_blocks: channels per group is even)
            float a = 0.f, b = 0.f;
            for (int c = tid * ppg; c < (tid + 1) * ppg; ++c)
#pragma unroll
                for (int w2 = 0; w2 < C::WM; ++w2) { a += red[(w2 * CP + c) * 2]; b += red[(w2 * CP + c) * 2 + 1]; }
            const long nblk = (long)d.tiles_h * d.tiles_w * nch;
            const long blk = ((long)th * d.tiles_w + tw) * nch + tc;
            float* pp = gn_part + ((n * nblk + blk) * gn_groups + tid) * 2;
            pp[0] = a; pp[1] = b;
        }
    }
}

int g_roll = 1, g_roll_tchunk = 0;

bool roll_enabled() { return g_roll != 0; }

template <class C>
int roll_tchunk(BfDims& d)
{
    d.tiles_h = ceil_div(d.H, C::TH);
    d.tiles_w = ceil_div(d.W, C::TW);
    const long cols = (long)d.N * d.tiles_h * d.tiles_w * (d.CO / C::CO_BLK);
    const long want = C::LDS_BYTES > 80 * 1024 ? 256 : 2048;             // one workgroup per CU (big rings): ONE full round of whole clips beats two
                                                                           // rounds of half clips (32->32 @128^2: 62 -> 58.5 us); small rings: ~2.7 rounds over 3 slots per CU
    int tchunk = d.T;                                                      // whole clip per workgroup unless that starves the chip
    while (tchunk > 2 && cols * ceil_div(d.T, tchunk) < want) tchunk = (tchunk + 1) / 2;
    if (g_roll_tchunk > 0) tchunk = g_roll_tchunk;
    return tchunk;
}

// workgroups per sample = rows of the GroupNorm partial buffer per sample (0: this configuration cannot emit them)
template <class C>
int roll_gn_blocks(BfDims d, int groups)
{
    if (d.CO != C::CO_BLK || groups <= 0 || d.CO % (2 * groups) || groups > C::NTHREADS) return 0;      // whole channel pairs per group
    const int tchunk = roll_tchunk<C>(d);
    return d.tiles_h * d.tiles_w * ceil_div(d.T, tchunk);
}

template <class C, bool GN>
int launch_roll(const bf16_t* x, int ldx, const uint4* wp, const float* bias, bf16_t* y, int ldy, BfDims d, hipStream_t s,
                float* gn_part, int gn_groups, Split2 sp)
{
    const int tchunk = roll_tchunk<C>(d);
    const long vox = (long)d.N * d.T * d.H * d.W, lim = 1L << 31;              // buffer descriptors: 32-bit offsets, OOB = 2^31
    if (span_bytes(vox, ldx, d.CK) >= lim || span_bytes(vox, ldy, d.CO) >= lim || (sp.x2 && span_bytes(vox, sp.ldx2, d.CK) >= lim) ||
        (sp.y2 && span_bytes(vox, sp.ldy2, d.CO) >= lim)) return VVAE_ERR_BAD_ARG;
    dim3 grid((unsigned)((long)d.N * d.tiles_h * d.tiles_w * ceil_div(d.T, tchunk)), d.CO / C::CO_BLK);
    const bool two = sp.x2 || sp.y2;
    auto k = two ? conv3d_bf16_roll_kernel<C, GN, true> : conv3d_bf16_roll_kernel<C, GN, false>;
    static bool attr_done[2] = {false, false};
    if (C::LDS_BYTES > 65536 && !attr_done[two]) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done[two] = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(C::NTHREADS), C::LDS_BYTES, s, x, ldx, wp, bias, y, ldy, d, tchunk, gn_part, gn_groups, sp);
    VVAE_LAUNCH_CHECK();
    return 0;
}

//              CKB KH KW MT_W NT_W WM WN weights prefetch                 (per-layer sweep: tools/conv_bench.py)
typedef RollCfg<16, 7, 7, 2, 1, 8, 1, W_LDS, true> R377;         // patch mixer: TH 16, 77 KB of weights behind a 61 KB ring
typedef RollCfg<16, 7, 7, 4, 1, 8, 1, W_LDS, true, 12> R377_12;  // ... with 12 real of its 16 K channels: 8 k-steps instead of 11, and
                                                                 // TH 32: a weight fragment read from LDS feeds 4 output rows
typedef RollCfg<16, 3, 3, 4, 1, 4, 1, W_REG, false, 16, 1, 2> R16_16;   // TH 16, two planes in flight
typedef RollCfg<16, 3, 3, 4, 1, 2, 2, W_REG, false, 16, 1, 2> R16_32;   // TH 8, one output-channel tile per wave, two planes in flight
typedef RollCfg<32, 3, 3, 4, 1, 8, 1, W_REG, false, 32, 1, 3> R32_16;      // TH 32, 8 waves, 157 KB ring (TH 16 with 2 rows per wave read 4 X
                                                                 // fragments per 6 products: 153 -> 129 us at 256^2; taller tiles lose elsewhere)
typedef RollCfg<32, 3, 3, 4, 1, 4, 2, W_REG, false, 32, 1, 3> R32_32;      // TH 16, 8 waves, one output-channel tile per wave

#define ROLL(C) do { if (gn_part) return launch_roll<C, true>(x, ldx, wp, bias, y, ldy, d, s, gn_part, gn_groups, sp); \
                     return launch_roll<C, false>(x, ldx, wp, bias, y, ldy, d, s, nullptr, 0, sp); } while (0)
int launch_roll_any(const bf16_t* x, int ldx, const uint4* wp, const float* bias, bf16_t* y, int ldy, BfDims d, int kh, hipStream_t s,
                    float* gn_part, int gn_groups, Split2 sp, int k_real)
{
    if (kh == 7) {
        if (gn_part || sp.x2 || sp.y2) return VVAE_ERR_BAD_ARG;
        if (k_real == 12) return launch_roll<R377_12, false>(x, ldx, wp, bias, y, ldy, d, s, nullptr, 0, NO_SPLIT);
        return launch_roll<R377, false>(x, ldx, wp, bias, y, ldy, d, s, nullptr, 0, NO_SPLIT);
    }
    if (d.CK == 16 && d.CO == 16) ROLL(R16_16);
    if (d.CK == 16 && d.CO == 32) ROLL(R16_32);
    if (d.CK == 32 && d.CO == 16) ROLL(R32_16);
    if (d.CK == 32 && d.CO == 32) ROLL(R32_32);
    return VVAE_ERR_BAD_ARG;                                               // >= 64 output channels: the per-frame kernel is as fast
}
#undef ROLL

int roll_gn_blocks_any(BfDims d, int kh, int groups)
{
    if (kh != 3 || !roll_enabled()) return 0;
    if (d.CK == 16 && d.CO == 16) return roll_gn_blocks<R16_16>(d, groups);
    if (d.CK == 16 && d.CO == 32) return roll_gn_blocks<R16_32>(d, groups);
    if (d.CK == 32 && d.CO == 16) return roll_gn_blocks<R32_16>(d, groups);
    if (d.CK == 32 && d.CO == 32) return roll_gn_blocks<R32_32>(d, groups);
    return 0;
}


// =============================================================================================== deep rolling forward / input gradient
// The layers whose K channels do not fit one chunk (CK = 64, 128) or that produce >= 64 channels from 32 ran on the per-frame kernel
// (conv3d_bf16_kernel): stage -> barrier -> multiply with nothing in flight, packed weights re-fetched through the vector L1 for every tile
// -- 0.6-0.95 PF/s with the matrix pipes 34 % busy.  Same march over time as conv3d_bf16_roll_kernel (ring of four halo planes, one new plane
// per step fetched while the step multiplies, parked output tile stored behind the next barrier), with the WAVES of a workgroup splitting
// (K chunk, output-channel tile, row group) so that each wave's weights -- 27 fragments: one 32-channel chunk x one 16-channel tile x
// 3 x 3 x 3 taps -- stay in registers for the whole march.  The NCH chunk-waves of an output tile each hold a partial sum; at the end
// of a step every wave parks the rows it does not own in an LDS scratch, one barrier, and the owner of a row adds the NCH - 1 partials in
// chunk order (fixed order: deterministic), rounds and stores it.
template <int NCH_, int NCT_, int NRG_, int MT_, int LA_>
struct DeepCfg {
    static constexpr int NCH = NCH_, NCT = NCT_, NRG = NRG_, MT = MT_, LA = LA_;
    static constexpr int CK = 32 * NCH, CO_BLK = 16 * NCT, NW = NCH * NCT * NRG, NTHREADS = 64 * NW;
    static constexpr int KT = 3, KH = 3, KW = 3, KSTEPS = KT * KW;            // one k-step = the 32 channels of a chunk at one (dt, dx)
    static constexpr int TH = MT * NRG, TW = 16, HR = TH + KH - 1, WR = TW + KW - 1, NX = MT + KH - 1;
    static constexpr int PITCH = 2 * CK, SWZ = NCH == 1 ? 1 : NCH == 2 ? 2 : 3;
    static constexpr int PLANE = HR * WR * PITCH;
    static constexpr int OWN = MT / NCH;                                     // rows of a wave's MT that it finishes itself
    static constexpr int SCRATCH = NCH > 1 ? NW * MT * 1024 : 0;             // [wave][row][lane] f32x4
    static constexpr int LDS_BYTES = 4 * PLANE + SCRATCH;
    static_assert(NW == 8 && MT % NCH == 0 && (NCH == 1 || NCH == 2 || NCH == 4), "wave grid");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// GN: as conv3d_bf16_roll_kernel<.., GN>: the workgroup also emits, per GroupNorm group it covers, the sum and the sum of squares of the
// rounded outputs it wrote -- part[n][blk][group][2], blk = (h tile, w tile, time chunk); a workgroup of CO_BLK < CO channels writes the
// groups of its channel block only (whole groups per block: deep_gn_blocks checks it).
template <class C, bool GN>
__global__ __launch_bounds__(C::NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3d_bf16_deep_kernel(const bf16_t* __restrict__ x, int ldx, const uint4* __restrict__ wp,
                                                                       const float* __restrict__ bias, bf16_t* __restrict__ y, int ldy,
                                                                       BfDims d, int tchunk, float* __restrict__ gn_part, int gn_groups)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NCH = C::NCH, NCT = C::NCT, MT = C::MT, KH = C::KH, KW = C::KW, KSTEPS = C::KSTEPS, NX = C::NX;
    constexpr int HR = C::HR, WR = C::WR, PITCH = C::PITCH, PLANE = C::PLANE, OWN = C::OWN;
    const long vox_all = (long)d.N * d.T * d.H * d.W;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (unsigned)span_bytes(vox_all, ldx, d.CK));
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y, (unsigned)span_bytes(vox_all, ldy, d.CO));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = wave % NCH, wi = (wave / NCH) % NCT, rg = wave / (NCH * NCT);      // K chunk, output-channel tile, row group of this wave
    const int r = lane & 15, g = lane >> 4;

    const int nsub = d.CO / C::CO_BLK;
    const int nblk = gridDim.x / nsub;
    int bid, yb;
    block_and_sub(nblk, nsub, bid, yb);                                    // the channel blocks of a tile: neighbours on one XCD
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);       // XCD-aware: neighbours share halo columns and planes
    const int nch = (d.T + tchunk - 1) / tchunk;
    const int tc = bid % nch; int q = bid / nch;
    const int tw = q % d.tiles_w; q /= d.tiles_w;
    const int th = q % d.tiles_h; const int n = q / d.tiles_h;
    const int h0 = th * C::TH, w0 = tw * C::TW;
    const int t_beg = tc * tchunk;
    int t_end = t_beg + tchunk;
    if (t_end > d.T) t_end = d.T;

    const int co_tiles = d.CO / 16;
    const int ct = yb * NCT + wi;                                          // this wave's output-channel tile
    bf16x8 wreg[KSTEPS][KH];
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j)
#pragma unroll
        for (int dy = 0; dy < KH; ++dy)
            wreg[j][dy] = __builtin_bit_cast(bf16x8, wp[((long)((c * KH + dy) * KSTEPS + j) * co_tiles + ct) * 64 + lane]);
    float bv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = bias ? bias[ct * 16 + 4 * g + e] : 0.f;

    PlaneStager<C::NTHREADS, HR, WR, C::CK / 8, PITCH, C::SWZ, WR, true> sx;
    const int hx = h0 - KH / 2, wx = w0 - KW / 2;
#define FETCH(T_) sx.fetch(rx, ldx, 0, n, (T_), hx, wx, d.T, d.H, d.W, tid)
    FETCH(t_beg - 1);                                                      // out-of-range frames come back as zeros
    sx.store(smem + ((t_beg - 1) & 3) * PLANE, tid);
    FETCH(t_beg);
    sx.store(smem + (t_beg & 3) * PLANE, tid);
    FETCH(t_beg + 1);
    const int wo = w0 + r;
    const int lin_w = (rg * MT) * WR + r;                                  // this lane's voxel in the wave's first halo row
    f32x4* scr = reinterpret_cast<f32x4*>(smem + 4 * PLANE);               // [wave][row][lane]
    uint2 held[OWN];
    unsigned hoff[OWN];
#pragma unroll
    for (int m = 0; m < OWN; ++m) { held[m] = make_uint2(0, 0); hoff[m] = OOB; }
    float gs[2] = {0.f, 0.f}, gss[2] = {0.f, 0.f};                         // GN: this lane's two channel pairs over its rows and frames

    for (int tt = t_beg; tt < t_end; ++tt) {
        sx.store(smem + ((tt + 1) & 3) * PLANE, tid);
        __syncthreads();                                                   // also: every wave has read last step's scratch
#pragma unroll
        for (int m = 0; m < OWN; ++m) buf_store8(ry, hoff[m], held[m]);    // the previous step's rows (OOB before the first)
        FETCH(tt + 1 < t_end ? tt + 2 : -1);

        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int LA = C::LA, NR = KSTEPS * NX;
        bf16x8 win[LA + 1];
        auto rd = [&](int idx) {
            const int j = idx / NX, hr = idx - j * NX;
            const int dt = j / KW, dx = j - dt * KW;
            const int p = swz_part<C::SWZ>(c * 4 + g, r + dx);             // column swizzle: independent of the halo row
            return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(smem + ((tt + dt - 1) & 3) * PLANE + (lin_w + dx) * PITCH + p * 16
                                                                                + hr * (WR * PITCH)));
        };
#pragma unroll
        for (int i = 0; i < LA; ++i) win[i % (LA + 1)] = rd(i);
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j) {
#pragma unroll
            for (int hr = 0; hr < NX; ++hr) {
                const int idx = j * NX + hr;
                if (idx + LA < NR) win[(idx + LA) % (LA + 1)] = rd(idx + LA);
#pragma unroll
                for (int dy = 0; dy < KH; ++dy) {
                    const int m = hr - dy;
                    if (m >= 0 && m < MT) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[j][dy], win[idx % (LA + 1)], acc[m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- fold the NCH partial sums of every row: wave c owns rows [c * OWN, (c + 1) * OWN) of its (tile, row group)
        if (NCH > 1) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (m / OWN != c) scr[(wave * MT + m) * 64 + lane] = acc[m];
            __syncthreads();
        }
#pragma unroll
        for (int o = 0; o < OWN; ++o) {
            float sum[4] = {0.f, 0.f, 0.f, 0.f};                           // scalar adds on purpose: no packed fp32 VALU forms (Makefile)
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc) {                             // chunk order, own partial in its place
                f32x4 part = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mm = 0; mm < MT; ++mm)                            // (static register index: the owned row is c * OWN + o)
                    if (mm == c * OWN + o) part = acc[mm];
                if (cc != c) part = scr[((wave - c + cc) * MT + c * OWN + o) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] = cc == 0 ? part[e] : sum[e] + part[e];
            }
            const int ho = h0 + rg * MT + c * OWN + o;
            const bool inside = ho < d.H && wo < d.W;
            const long v = (((long)n * d.T + tt) * d.H + ho) * d.W + wo;
            hoff[o] = inside ? (unsigned)(v * ldy + ct * 16 + 4 * g) * 2u : OOB;
            held[o].x = (uint32_t)f2bf(sum[0] + bv[0]) | ((uint32_t)f2bf(sum[1] + bv[1]) << 16);
            held[o].y = (uint32_t)f2bf(sum[2] + bv[2]) | ((uint32_t)f2bf(sum[3] + bv[3]) << 16);
            if (GN) {                                                      // statistics of what GroupNorm will read: the rounded values
                const bf16x2 ones = __builtin_bit_cast(bf16x2, 0x3f803f80u);
                const bf16x2 p0 = __builtin_bit_cast(bf16x2, inside ? held[o].x : 0u), p1 = __builtin_bit_cast(bf16x2, inside ? held[o].y : 0u);
                gs[0] = __builtin_amdgcn_fdot2_f32_bf16(p0, ones, gs[0], false);
                gs[1] = __builtin_amdgcn_fdot2_f32_bf16(p1, ones, gs[1], false);
                gss[0] = __builtin_amdgcn_fdot2_f32_bf16(p0, p0, gss[0], false);
                gss[1] = __builtin_amdgcn_fdot2_f32_bf16(p1, p1, gss[1], false);
            }
        }
    }
#undef FETCH
#pragma unroll
    for (int m = 0; m < OWN; ++m) buf_store8(ry, hoff[m], held[m]);        // the last step's rows
    if (GN) {
        // fold: the 16 voxel lanes of a channel quad (DPP), then the waves that share the output-channel tile (LDS, wave order), then the
        // channel pairs of each group -- fixed order, no atomics
        __syncthreads();                                                   // every wave is done with the ring and the scratch
        float* red = reinterpret_cast<float*>(smem);                       // [wave][pair 0..7][2]
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float a = butterfly_sum<8, 1>(gs[e]), b = butterfly_sum<8, 1>(gss[e]);
            if (r == 0) { red[(wave * 8 + 2 * g + e) * 2] = a; red[(wave * 8 + 2 * g + e) * 2 + 1] = b; }
        }
        __syncthreads();
        const int gpb = gn_groups * C::CO_BLK / d.CO;                      // groups this workgroup's channel block covers
        if (tid < gpb) {
            constexpr int CP = C::CO_BLK / 2;                              // channel pairs of the block: pair cp lives in tile cp / 8
            const int ppg = CP / gpb;
            float a = 0.f, b = 0.f;
            for (int cp = tid * ppg; cp < (tid + 1) * ppg; ++cp) {
                const int t8 = cp >> 3, q8 = cp & 7;
                for (int w2 = 0; w2 < C::NW; ++w2)
                    if ((w2 / NCH) % NCT == t8) { a += red[(w2 * 8 + q8) * 2]; b += red[(w2 * 8 + q8) * 2 + 1]; }
            }
            const long nblk_s = (long)d.tiles_h * d.tiles_w * nch;
            const long blk = ((long)th * d.tiles_w + tw) * nch + tc;
            float* pp = gn_part + ((n * nblk_s + blk) * gn_groups + yb * gpb + tid) * 2;
            pp[0] = a; pp[1] = b;
        }
    }
}

int g_deep = 1;

//              NCH NCT NRG MT LA
typedef DeepCfg<1, 4, 2, 8, 3> D32_64;       // K 32 -> 64 per workgroup: TH 16
typedef DeepCfg<2, 2, 2, 4, 3> D64_32;       // K 64 -> 32: TH 8
typedef DeepCfg<2, 4, 1, 8, 3> D64_64;       // K 64 -> 64 per workgroup: TH 8
typedef DeepCfg<4, 2, 1, 4, 3> D128_32;      // K 128 -> 32 per workgroup: TH 4

template <class C>
int deep_tchunk(BfDims& d)
{
    d.tiles_h = ceil_div(d.H, C::TH);
    d.tiles_w = ceil_div(d.W, C::TW);
    const long cols = (long)d.N * d.tiles_h * d.tiles_w * (d.CO / C::CO_BLK);
    int tchunk = d.T;                                                      // whole clip per workgroup unless that starves the chip
    while (tchunk > 2 && cols * ceil_div(d.T, tchunk) < 256) tchunk = (tchunk + 1) / 2;
    return tchunk;
}

// workgroups per sample and channel block = rows per sample of the GroupNorm partial buffer (0: this configuration cannot emit them)
template <class C>
int deep_gn_blocks(BfDims d, int groups)
{
    if (groups <= 0 || d.CO % (2 * groups) || (groups * C::CO_BLK) % d.CO || groups * C::CO_BLK / d.CO > C::NTHREADS) return 0;
    const int tchunk = deep_tchunk<C>(d);
    return d.tiles_h * d.tiles_w * ceil_div(d.T, tchunk);
}

template <class C>
int launch_deep(const bf16_t* x, int ldx, const uint4* wp, const float* bias, bf16_t* y, int ldy, BfDims d, hipStream_t s, float* gn_part, int gn_groups)
{
    const long vox = (long)d.N * d.T * d.H * d.W, lim = 1L << 31;
    if (span_bytes(vox, ldx, d.CK) >= lim || span_bytes(vox, ldy, d.CO) >= lim) return VVAE_ERR_BAD_ARG;
    if (gn_part && deep_gn_blocks<C>(d, gn_groups) <= 0) return VVAE_ERR_BAD_ARG;
    const int tchunk = deep_tchunk<C>(d);
    dim3 grid((unsigned)((long)d.N * d.tiles_h * d.tiles_w * ceil_div(d.T, tchunk) * (d.CO / C::CO_BLK)));      // 1-D: block_and_sub
    const bool gn = gn_part != nullptr;
    auto k = gn ? conv3d_bf16_deep_kernel<C, true> : conv3d_bf16_deep_kernel<C, false>;
    static bool attr_done[2] = {false, false};
    if (C::LDS_BYTES > 65536 && !attr_done[gn]) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done[gn] = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(C::NTHREADS), C::LDS_BYTES, s, x, ldx, wp, bias, y, ldy, d, tchunk, gn_part, gn_groups);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// 3 x 3 x 3 layers the deep kernel takes; anything else: VVAE_ERR_BAD_ARG (the caller falls back to the per-frame kernel)
int launch_deep_any(const bf16_t* x, int ldx, const uint4* wp, const float* bias, bf16_t* y, int ldy, BfDims d, hipStream_t s, float* gn_part,
                    int gn_groups)
{
    if (!g_deep) return VVAE_ERR_BAD_ARG;
    if (d.CK == 32 && d.CO % 64 == 0) return launch_deep<D32_64>(x, ldx, wp, bias, y, ldy, d, s, gn_part, gn_groups);
    if (d.CK == 64 && d.CO == 32) return launch_deep<D64_32>(x, ldx, wp, bias, y, ldy, d, s, gn_part, gn_groups);
    if (d.CK == 64 && d.CO % 64 == 0) return launch_deep<D64_64>(x, ldx, wp, bias, y, ldy, d, s, gn_part, gn_groups);
    if (d.CK == 128 && d.CO % 32 == 0) return launch_deep<D128_32>(x, ldx, wp, bias, y, ldy, d, s, gn_part, gn_groups);
    return VVAE_ERR_BAD_ARG;
}

int deep_gn_blocks_any(BfDims d, int kh, int groups)
{
    if (kh != 3 || !g_deep) return 0;
    if (d.CK == 32 && d.CO % 64 == 0) return deep_gn_blocks<D32_64>(d, groups);
    if (d.CK == 64 && d.CO == 32) return deep_gn_blocks<D64_32>(d, groups);
    if (d.CK == 64 && d.CO % 64 == 0) return deep_gn_blocks<D64_64>(d, groups);
    if (d.CK == 128 && d.CO % 32 == 0) return deep_gn_blocks<D128_32>(d, groups);
    return 0;
}

#ifndef WG_ABL            // timing-only builds (tools/r04_wg_abl.sh): bit 0 no MFMAs, bit 1 no fragment reads, bit 2 no staging behind the first planes
#define WG_ABL 0
#endif

template <class C, bool TWO>
__global__ __launch_bounds__(C::NTHREADS, C::MINW) void conv3d_wgrad_bf16_kernel(const bf16_t* __restrict__ x, int ldx,
                                                                                 const bf16_t* __restrict__ dy, int lddy,
                                                                                 float* __restrict__ slab, WgDims d, Split2 sp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CIB = C::CIB, COB = C::COB, KH = C::KH, KW = C::KW, TH = C::TH, TW = C::TW, AGS = C::AGS;
    constexpr int CIT = C::CIT, COT = C::COT, HR = C::HR, WR = C::WR, PX = C::PX, PY = C::PY;
    unsigned char* ring = smem;                                   // 4 X halo planes: plane t lives in slot t & 3
    unsigned char* ybuf = smem + 4 * C::PLANE;                    // dY tile of step t in buffer t & 1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // wave -> (dt, ci tile wi, co tile wj, dy group): every SIMD gets the same number of equally heavy waves
    const int rgp = wave % C::RG, wv = wave / C::RG;
    const int dt = wv / (CIT * COT * C::AG);
    const int wr = wv % (CIT * COT * C::AG);
    const int wi = wr / (COT * C::AG), wj = (wr / C::AG) % COT, a0 = (wr % C::AG) * AGS;
    const int r0 = rgp * C::THR;                                             // first output row of this wave
    const bool bias_wave = dt == 0 && wi == 0 && a0 == 0;
    const int co_subs = d.CO / COB, nsub = (d.CI / CIB) * co_subs;
    const int nbx = gridDim.x / nsub;
    int bx, by;
    block_and_sub(nbx, nsub, bx, by);                                        // the (ci, co) sub-blocks of a column run: neighbours on one XCD
    const int ci0 = (by / co_subs) * CIB, co0 = (by % co_subs) * COB;
    // transposed-read lane offsets: lane (g = l>>4, q = (l>>2)&3, p = l&3) addresses voxel row 4g+q, channels 4p..4p+3
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int lvox = 4 * g + qq, loffy = (4 * g + qq) * PY + 8 * pp + wj * 32;

    f32x4 acc[AGS][KW];
#pragma unroll
    for (int a = 0; a < AGS; ++a)
#pragma unroll
        for (int b = 0; b < KW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};
    const s16x8 ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};   // bf16 1.0
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);

    [[maybe_unused]] bf16x8 fk = ones;
    if (WG_ABL & 2) { typedef unsigned u32x4b __attribute__((ext_vector_type(4))); u32x4b u = __builtin_bit_cast(u32x4b, ones); asm volatile("" : "+v"(u)); fk = __builtin_bit_cast(bf16x8, u); }
    PlaneStager<C::NTHREADS, HR, WR, CIB / 8, PX, C::SWX, C::WRP> sx;
    PlaneStager<C::NTHREADS, TH, TW, COB / 8, PY> sy;
    PlaneStager<C::NTHREADS, HR, WR, CIB / 8, PX, C::SWX, C::WRP> sx2;      // PD == 2 only
    PlaneStager<C::NTHREADS, TH, TW, COB / 8, PY> sy2;
    // one input tensor: buffer descriptors (unconditional loads, see make_rsrc); TWO: per-thread pointers as a lane's tensor is not uniform
    const long vox_all = (long)d.N * d.T * d.H * d.W;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, TWO ? 0u : (unsigned)span_bytes(vox_all, ldx, d.CI));
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dy, (unsigned)span_bytes(vox_all, lddy, d.CO));
    const bf16_t* xsrc = x + ci0;
    if (TWO && sp.x2 && ci0 + (int)(threadIdx.x % (WR * (CIB / 8))) % (CIB / 8) * 8 >= sp.xsplit) {   // this thread stages a 16-byte part of x2
        xsrc = sp.x2 + ci0 - sp.xsplit;
        ldx = sp.ldx2;
    }
#define FX(S, T_) do { if ((WG_ABL & 4) && (T_) > 1) break; if (TWO) (S).fetch(xsrc, ldx, n, (T_), hx, wx, d.T, d.H, d.W, tid); else (S).fetch(rx, ldx, ci0, n, (T_), hx, wx, d.T, d.H, d.W, tid); } while (0)
#define FY(S, T_) do { if ((WG_ABL & 4) && (T_) > 0) break; (S).fetch(rdy, lddy, co0, n, (T_), h0, w0, d.T, d.H, d.W, tid); } while (0)
#define SX(S, P_, T_) do { if (!((WG_ABL & 4) && (T_) > 1)) (S).store((P_), tid); } while (0)
#define SY(S, P_, T_) do { if (!((WG_ABL & 4) && (T_) > 0)) (S).store((P_), tid); } while (0)

    // A workgroup walks whole time-columns: for a fixed (n, h-tile, w-tile) it marches t = 0..T-1, so every X plane is
    // fetched from memory once (not KT times) and lives in the LDS ring for the three steps that use it.
    const int col_beg = bx * d.cols_per_block;
    int col_end = col_beg + d.cols_per_block;
    if (col_end > d.ncols) col_end = d.ncols;
    for (int colidx = col_beg; colidx < col_end; ++colidx) {
        const int tw = colidx % d.tiles_w; const int q = colidx / d.tiles_w;
        const int th = q % d.tiles_h; const int n = q / d.tiles_h;
        const int h0 = th * TH, w0 = tw * TW;
        const int hx = h0 - KH / 2, wx = w0 - KW / 2;
        __syncthreads();                                          // previous column is done with the ring
        FX(sx, -1);   // plane -1 = zeros
        SX(sx, ring + 3 * C::PLANE, -1);
        FX(sx, 0);
        SX(sx, ring, 0);
        FX(sx, 1);
        FY(sy, 0);
        if (C::PD == 2 && d.T > 1) {                              // second register set: plane 2 and dY tile 1 are on their way too
            FX(sx2, 2);
            FY(sy2, 1);
        }
        auto step = [&](int tt) {
            const unsigned char* ys = ybuf + (tt & 1) * C::YBYTES + loffy + r0 * (TW * PY);
            // halo rows r0 + a0 + rr, rr = 0 .. THR+AGS-2, meet kernel rows a0 + aa at output rows r0 + h, h = rr - aa
            const unsigned char* xplane = ring + ((tt + dt - 1) & 3) * C::PLANE + (r0 + a0) * (C::WRP * PX) + 8 * pp;
            bf16x8 bfr[AGS];                                      // rolling window of dY fragments: row h lives in slot h % AGS
            typedef unsigned u32x4a __attribute__((ext_vector_type(4)));
            auto fake = [&](int) { return fk; };                  // timing builds without fragment reads: one register quad made up at kernel entry
#pragma unroll
            for (int rr = 0; rr < C::THR + AGS - 1; ++rr) {
                if (rr < C::THR) {
                    bfr[rr % AGS] = (WG_ABL & 2) ? fake(rr) : tr_frag(ys + rr * TW * PY, 16 * PY);
                    if (bias_wave && !(WG_ABL & 1)) accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bfr[rr % AGS], accb, 0, 0, 0);
                }
                if (KH % AGS != 0 && a0 + rr >= C::THR + KH - 1) continue;   // last dy group is short: its tail rows are not needed
#pragma unroll
                for (int b = 0; b < KW; ++b) {
                    const int lin = rr * C::WRP + b + lvox;       // SWX: 32-byte half ^= (voxel >> 2) & 1, same for voxel + 16
                    const int sw = C::SWX ? ((lin >> 2) & 1) << 5 : 0;
                    const bf16x8 afr = (WG_ABL & 2) ? fake(rr * 8 + b) : tr_frag(xplane + lin * PX + ((wi * 32) ^ sw), 16 * PX);
#pragma unroll
                    for (int aa = 0; aa < AGS; ++aa) {
                        const int h = rr - aa;
                        if (h >= 0 && h < C::THR && (KH % AGS == 0 || a0 + aa < KH)) {
                            if (!(WG_ABL & 1)) acc[aa][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[h % AGS], acc[aa][b], 0, 0, 0);
                            else { const u32x4a ua = __builtin_bit_cast(u32x4a, afr), ub = __builtin_bit_cast(u32x4a, bfr[h % AGS]); asm volatile("" :: "v"(ua), "v"(ub)); }
                        }
                    }
                }
            }
        };
        if (C::PD == 2) {
            // two steps' operands in flight: a step of the 16-channel layers is ~0.5 us of matrix work against 1-2 us of loaded-HBM latency
            for (int tt = 0; tt < d.T; tt += 2) {
                SX(sx, ring + ((tt + 1) & 3) * C::PLANE, tt + 1);
                SY(sy, ybuf + (tt & 1) * C::YBYTES, tt);
                __syncthreads();
                if (tt + 2 < d.T) {
                    FX(sx, tt + 3);
                    FY(sy, tt + 2);
                }
                step(tt);
                if (tt + 1 >= d.T) break;
                SX(sx2, ring + ((tt + 2) & 3) * C::PLANE, tt + 2);
                SY(sy2, ybuf + ((tt + 1) & 1) * C::YBYTES, tt + 1);
                __syncthreads();
                if (tt + 3 < d.T) {
                    FX(sx2, tt + 4);
                    FY(sy2, tt + 3);
                }
                step(tt + 1);
            }
        } else {
            for (int tt = 0; tt < d.T; ++tt) {
                SX(sx, ring + ((tt + 1) & 3) * C::PLANE, tt + 1);     // plane tt+1 (zeros past the end)
                SY(sy, ybuf + (tt & 1) * C::YBYTES, tt);
                __syncthreads();
                if (tt + 1 < d.T) {                                   // next step's operands fly while this step computes
                    FX(sx, tt + 2);
                    FY(sy, tt + 1);
                }
                step(tt);
            }
        }
    }
#undef FX
#undef FY
#undef SX
#undef SY
    // ---- write this workgroup's partial sums: slab[block][dt][dy][dx][ci_local][co_local] (+ COB dbias partials) ----
    float* out = slab + (((long)bx * C::RG + rgp) * nsub + by) * C::SLAB_FLOATS;
    const int col = lane & 15, rg = lane >> 4;
#pragma unroll
    for (int aa = 0; aa < AGS; ++aa) {
        if (KH % AGS != 0 && a0 + aa >= KH) continue;
#pragma unroll
        for (int b = 0; b < KW; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = wi * 16 + rg * 4 + e, co = wj * 16 + col;
                out[(((dt * KH + a0 + aa) * KW + b) * CIB + ci) * COB + co] = acc[aa][b][e];
            }
    }
    if (bias_wave && rg == 0) out[C::KT * KH * KW * CIB * COB + wj * 16 + col] = accb[0];
}

// dw[tap][ci][co] = sum_b slab[b][sub(ci,co)][tap][ci%CIB][co%COB];  dbias[co] = sum_b (ci-sub 0) slab dbias part
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, float* __restrict__ dbias,
                                                           int nblk, int taps, int CI, int CO, int CIB, int COB)
{
    // 256 threads = 32 output elements x 8 slab lanes; fixed summation order (lane-strided, then a fixed tree) => reproducible
    __shared__ float red[8][32];
    const int co_subs = CO / COB, nsub = (CI / CIB) * co_subs;
    const long slab_floats = (long)taps * CIB * COB + COB;
    const long total = (long)taps * CI * CO;
    const int el = threadIdx.x & 31, bl = threadIdx.x >> 5;
    const long i = (long)blockIdx.x * 32 + el;
    long base = -1;
    if (i < total) {
        const int co = (int)(i % CO); const long r = i / CO; const int ci = (int)(r % CI); const int tap = (int)(r / CI);
        base = (long)((ci / CIB) * co_subs + co / COB) * slab_floats + ((long)tap * CIB + ci % CIB) * COB + co % COB;
    } else if (dbias && i < total + CO) {
        const int co = (int)(i - total);
        base = (long)(co / COB) * slab_floats + (long)taps * CIB * COB + co % COB;          // ci-sub 0
    }
    float s = 0.f;
    if (base >= 0) {
        // the slabs sit in L2: this loop pays latency, so 8 loads are in flight per thread (same summation order)
        const long stride = (long)nsub * slab_floats;
        const float* p = slab + base;
        int b = bl;
        for (; b + 56 < nblk; b += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long)(b + 8 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nblk; b += 8) s += p[(long)b * stride];
    }
    red[bl][el] = s;
    __syncthreads();
    if (bl == 0 && base >= 0) {
        const float t = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
        if (i < total) dw[i] = t; else dbias[i - total] = t;
    }
}

//            CIB COB KH KW TH AG RG MINW workgroups/CU
typedef WgCfg<16, 16, 3, 3, 8, 1, 2, 3, 2, 1> W333_16_16;    //  6 waves: two row groups of an 8-row tile, 2 workgroups per CU
typedef WgCfg<32, 16, 3, 3, 8, 1, 2, 3, 1> W333_32_16;       // 12 waves: two row groups of an 8-row tile
typedef WgCfg<32, 32, 3, 3, 8, 1, 1, 3, 1> W333_32_32;       // 12 waves, 8-row tiles: twice the products per barrier and staged plane (64->64 @64^2 70.8 -> 63.1 us,
                                                             // 64->32 @128^2 127 -> 116) -- where they still fill the chip (wg32_tall)
typedef WgCfg<32, 32, 3, 3, 4, 1, 1, 3, 1> W333_32_32_T4;    // 12 waves, 4-row tiles (32->64 @64^2: 8-row tiles leave 128 workgroups for 256 CUs, 41 -> 53 us)
typedef WgCfg<16, 16, 7, 7, 8, 4, 1, 3, 1, 2> W377_16_16;    // 12 waves: 3 temporal taps x 4 groups of kernel rows

int g_wg_cob16 = 0;                        // tuning: 1 = 16 output channels per workgroup even where Cin, Cout % 32 == 0
int g_wg_blocks = 0;                       // tuning: > 0 overrides the persistent grid size

template <class C>
inline int wg_blocks_x(long ncols, int nsub)
{
    long nb = (g_wg_blocks > 0 ? g_wg_blocks : C::BLOCKS) / nsub;
    if (nb < 16) nb = 16;
    if (nb > ncols) nb = ncols;
    return (int)nb;
}

template <class C>
size_t wg_ws_bytes(int N, int T, int H, int W, int CI, int CO)
{
    (void)T;
    const long ncols = (long)N * ceil_div(H, C::TH) * ceil_div(W, C::TW);
    const int nsub = (CI / C::CIB) * (CO / C::COB);
    return (size_t)wg_blocks_x<C>(ncols, nsub) * C::RG * nsub * C::SLAB_FLOATS * sizeof(float);
}

template <class C>
int launch_wgrad_cfg(const bf16_t* x, int ldx, const bf16_t* dy, int lddy, float* dw, float* dbias, int N, int T, int H, int W,
                     int CI, int CO, void* ws, size_t ws_bytes, hipStream_t s, Split2 sp = NO_SPLIT)
{
    WgDims d{N, T, H, W, CI, CO, ceil_div(H, C::TH), ceil_div(W, C::TW), 0, 0};
    const long vox = (long)N * T * H * W, lim = 1L << 31;                     // buffer descriptors: 32-bit offsets, OOB = 2^31
    if (span_bytes(vox, ldx, CI) >= lim || span_bytes(vox, lddy, CO) >= lim || (sp.x2 && span_bytes(vox, sp.ldx2, CI) >= lim)) return VVAE_ERR_BAD_ARG;
    const long ncols = (long)N * d.tiles_h * d.tiles_w;
    const int nsub = (CI / C::CIB) * (CO / C::COB);
    const int nbx = wg_blocks_x<C>(ncols, nsub);
    d.ncols = (int)ncols;
    d.cols_per_block = ceil_div(ncols, nbx);
    const int nblk = ceil_div(ncols, d.cols_per_block);        // blocks that own at least one time-column
    if (!ws || ws_bytes < (size_t)nblk * C::RG * nsub * C::SLAB_FLOATS * sizeof(float)) return VVAE_ERR_WORKSPACE;
    const bool two = sp.x2 != nullptr;
    auto k = two ? conv3d_wgrad_bf16_kernel<C, true> : conv3d_wgrad_bf16_kernel<C, false>;
    static bool attr_done[2] = {false, false};     // once per instantiation: keeps the launch path free of non-stream calls
    if (C::LDS_BYTES > 65536 && !attr_done[two]) { // (hipGraph capture of the training step replays only stream work)
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_done[two] = true;
    }
    hipLaunchKernelGGL(k, dim3(nblk * nsub), dim3(C::NTHREADS), C::LDS_BYTES, s, x, ldx, dy, lddy, (float*)ws, d, sp);   // 1-D: block_and_sub
    VVAE_LAUNCH_CHECK();
    const int taps = C::KT * C::KH * C::KW;
    const long total = (long)taps * CI * CO + CO;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(total, 32)), dim3(256), 0, s, (const float*)ws, dw, dbias, nblk * C::RG, taps, CI,
                       CO, C::CIB, C::COB);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// 8-row tiles for the 32 x 32 configuration where the time columns of 8 x 32 tiles x (ci, co) sub-blocks still give every CU a workgroup
inline bool wg32_tall(int N, int H, int W, int CI, int CO)
{
    const long cols = (long)N * ceil_div(H, W333_32_32::TH) * ceil_div(W, W333_32_32::TW);
    return cols * (CI / 32) * (CO / 32) >= 256;
}

inline bool wgrad_shape_ok(int Cin, int Cout, int kt, int kh, int kw)
{
    if (Cin % 16 || Cout % 16) return false;
    if (kt == 3 && kh == 3 && kw == 3) return true;
    return kt == 3 && kh == 7 && kw == 7 && Cin == 16 && Cout == 16;
}

}  // namespace

extern "C" size_t vvae_conv3d_wgrad_bf16_ws_bytes(int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw)
{
    if (!wgrad_shape_ok(Cin, Cout, kt, kh, kw)) return 0;
    if (kh == 7) return wg_ws_bytes<W377_16_16>(N, T, H, W, Cin, Cout);
    const bool i32 = Cin % 32 == 0, o32 = i32 && Cout % 32 == 0 && !g_wg_cob16;
    if (i32 && o32) return wg32_tall(N, H, W, Cin, Cout) ? wg_ws_bytes<W333_32_32>(N, T, H, W, Cin, Cout) : wg_ws_bytes<W333_32_32_T4>(N, T, H, W, Cin, Cout);
    if (i32) return wg_ws_bytes<W333_32_16>(N, T, H, W, Cin, Cout);
    return wg_ws_bytes<W333_16_16>(N, T, H, W, Cin, Cout);
}

// dw (kt,kh,kw,Cin,Cout) fp32 and dbias (Cout, may be NULL) are overwritten.  ws: vvae_conv3d_wgrad_bf16_ws_bytes(...) bytes.
extern "C" int vvae_conv3d_wgrad_bf16(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias,
                                      int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                                      void* ws, size_t ws_bytes, void* stream)
{
    if (!x || !dy || !dw || N <= 0 || T <= 0 || H <= 0 || W <= 0 || !wgrad_shape_ok(Cin, Cout, kt, kh, kw)) return VVAE_ERR_BAD_ARG;
    if (ldx < Cin || lddy < Cout || ldx % 8 || lddy % 8 || ((uintptr_t)x % 16) || ((uintptr_t)dy % 16)) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bf16_t* xp = (const bf16_t*)x;
    const bf16_t* dyp = (const bf16_t*)dy;
    if (kh == 7) return launch_wgrad_cfg<W377_16_16>(xp, ldx, dyp, lddy, dw, dbias, N, T, H, W, Cin, Cout, ws, ws_bytes, s);
    const bool i32 = Cin % 32 == 0, o32 = i32 && Cout % 32 == 0 && !g_wg_cob16;
    if (i32 && o32 && wg32_tall(N, H, W, Cin, Cout)) return launch_wgrad_cfg<W333_32_32>(xp, ldx, dyp, lddy, dw, dbias, N, T, H, W, Cin, Cout, ws, ws_bytes, s);
    if (i32 && o32) return launch_wgrad_cfg<W333_32_32_T4>(xp, ldx, dyp, lddy, dw, dbias, N, T, H, W, Cin, Cout, ws, ws_bytes, s);
    if (i32) return launch_wgrad_cfg<W333_32_16>(xp, ldx, dyp, lddy, dw, dbias, N, T, H, W, Cin, Cout, ws, ws_bytes, s);
    return launch_wgrad_cfg<W333_16_16>(xp, ldx, dyp, lddy, dw, dbias, N, T, H, W, Cin, Cout, ws, ws_bytes, s);
}

// vvae_conv3d_wgrad_bf16 for a layer whose input is concat([x, x2], channels) held as two tensors (x: the first c_split channels):
// the X halo parts are staged from whichever tensor holds them.  3x3x3, c_split a multiple of 8; same workspace query.
extern "C" int vvae_conv3d_wgrad_bf16_cat2(const void* x, int ldx, const void* x2, int ldx2, int c_split, const void* dy, int lddy, float* dw,
                                           float* dbias, int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, void* ws,
                                           size_t ws_bytes, void* stream)
{
    if (!x || !x2 || !dy || !dw || N <= 0 || T <= 0 || H <= 0 || W <= 0 || kh != 3 || !wgrad_shape_ok(Cin, Cout, kt, kh, kw)) return VVAE_ERR_BAD_ARG;
    if (c_split <= 0 || c_split >= Cin || c_split % 8 || ldx < c_split || ldx2 < Cin - c_split || lddy < Cout || ldx % 8 || ldx2 % 8 || lddy % 8 ||
        ((uintptr_t)x % 16) || ((uintptr_t)x2 % 16) || ((uintptr_t)dy % 16)) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bf16_t* xp = (const bf16_t*)x;
    const bf16_t* dyp = (const bf16_t*)dy;
    const Split2 sp{(const bf16_t*)x2, ldx2, c_split, nullptr, 0, 0};
    const bool i32 = Cin % 32 == 0, o32 = i32 && Cout % 32 == 0 && !g_wg_cob16;
    if (i32 && o32 && wg32_tall(N, H, W, Cin, Cout)) return launch_wgrad_cfg<W333_32_32>(xp, ldx, dyp, lddy, dw, dbias, N, T, H, W, Cin, Cout, ws, ws_bytes, s, sp);
    if (i32 && o32) return launch_wgrad_cfg<W333_32_32_T4>(xp, ldx, dyp, lddy, dw, dbias, N, T, H, W, Cin, Cout, ws, ws_bytes, s, sp);
    if (i32) return launch_wgrad_cfg<W333_32_16>(xp, ldx, dyp, lddy, dw, dbias, N, T, H, W, Cin, Cout, ws, ws_bytes, s, sp);
    return launch_wgrad_cfg<W333_16_16>(xp, ldx, dyp, lddy, dw, dbias, N, T, H, W, Cin, Cout, ws, ws_bytes, s, sp);
}

// Tuning hook for the weight-gradient kernel: output channels per workgroup (16 / 32 where Cout allows) and persistent grid size.
extern "C" int vvae_conv3d_wgrad_config(int cob16, int blocks)
{
    g_wg_cob16 = cob16; g_wg_blocks = blocks;
    return 0;
}

// Test / tuning hook: on = 0 routes the multi-chunk / wide layers through the per-frame kernel again (vvae_conv3d_deep_config).
extern "C" int vvae_conv3d_deep_config(int on)
{
    g_deep = on;
    return 0;
}

// Test / tuning hook: on = 0 routes single-chunk layers through the per-frame kernel again; tchunk > 0 forces the frames per workgroup.
extern "C" int vvae_conv3d_roll_config(int on, int tchunk)
{
    g_roll = on; g_roll_tchunk = tchunk;
    return 0;
}
