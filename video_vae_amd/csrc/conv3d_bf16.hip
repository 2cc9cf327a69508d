// bf16 MFMA fast path for NDHWC Conv3d (SAME, stride 1): forward and data-grad.   gfx950 / CDNA4 only.
//
// Replaces the XLA lowering of nnx.Conv at /root/reference/train/unet.py:13-21 (3x3x3) and :111-113 (3x7x7).
//
// Shape of the problem: tiny channel counts (16..128) at huge spatial extent, so the GEMM N dimension is 1..8
// MFMA tiles wide and a naive implicit GEMM is bound by LDS operand reads, not by the matrix cores.  Design:
//   * one workgroup (4 waves) = one (n, t) x TH x 16 output tile for a block of output channels; the input halo
//     tile (KT x (TH+KH-1) x (16+KW-1) voxels x CKB channels) is staged ONCE in LDS, zero-filled at the borders;
//     the voxel pitch is padded to 32 / 96 bytes (= 2 / 6 sixteen-byte slots, both = 2 mod 4) which makes every
//     ds_read_b128 operand read conflict-free for any tap shift;
//   * K is ordered (dy | dt, dx, ci): for a fixed kernel row dy the operand fragment read for halo row r serves
//     the KH output rows r-dy that the wave owns, so each LDS fragment read feeds up to KH*NT_W MFMAs instead of 1;
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand and the voxels as B: the accumulator then holds
//     4 consecutive output channels of one voxel per lane -> 8-byte bf16 stores, contiguous per voxel;
//   * weights are pre-packed (pack kernel below, ~KB..MB, once per call) into fragment order so a wave's B^T
//     fragment is one coalesced 1-KiB load that stays L1/L2 resident across workgroups;
//   * dgrad is the same kernel on weights packed with flipped taps and swapped channel roles.
#include "common.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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

// Packed weight layout (uint4 = 8 bf16 per lane): [chunk][dy][kstep][co_tile][lane]
//   lane l: co = co_tile*16 + (l & 15), k = 32*kstep + 8*(l >> 4) + e,  slot = k / CKB -> (dt, dx), ci = chunk*CKB + k % CKB
// dgrad=0: value = w[dt][dy][dx][ci][co]                (K channels = Cin,  produced = Cout)
// dgrad=1: value = w[KT-1-dt][KH-1-dy][KW-1-dx][co][ci] (K channels = Cout, produced = Cin; "co" indexes Cin here)
template <int CKB, int KT, int KH, int KW>
__global__ void pack_weights_kernel(const float* __restrict__ w, uint4* __restrict__ wp, int Cin, int Cout, int dgrad)
{
    constexpr int KSTEPS = (KT * KW * CKB + 31) / 32;
    const int CK = dgrad ? Cout : Cin, CO = dgrad ? Cin : Cout;
    const int co_tiles = CO / 16, chunks = CK / CKB;
    const long total = (long)chunks * KH * KSTEPS * co_tiles * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int l = (int)(i & 63); long q = i >> 6;
        const int ct = (int)(q % co_tiles); q /= co_tiles;
        const int j = (int)(q % KSTEPS); q /= KSTEPS;
        const int dy = (int)(q % KH); const int chunk = (int)(q / KH);
        const int co = ct * 16 + (l & 15);
        uint32_t pk[4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 32 * j + 8 * (l >> 4) + e;
            const int slot = k / CKB, ci = chunk * CKB + k % CKB;
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
        // ---- stage the halo tile of this channel chunk: 16-byte parts, zero fill outside the volume ----
        constexpr int PARTS = CKB / 8;
        for (int i = tid; i < C::NVOX * PARTS; i += 256) {
            const int part = i % PARTS, vox = i / PARTS;
            const int wc = vox % WR; const int qq = vox / WR; const int hr = qq % HR, dt = qq / HR;
            const int ti = tt + dt - KT / 2, hi = h0 + hr - KH / 2, wi = w0 + wc - KW / 2;
            uint4 val = make_uint4(0, 0, 0, 0);
            if ((unsigned)ti < (unsigned)d.T && (unsigned)hi < (unsigned)d.H && (unsigned)wi < (unsigned)d.W) {
                const long v = (((long)n * d.T + ti) * d.H + hi) * d.W + wi;
                val = *reinterpret_cast<const uint4*>(x + v * ldx + chunk * CKB + part * 8);
            }
            *reinterpret_cast<uint4*>(smem + vox * PITCH + part * 16) = val;
        }
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
    if (C::LDS_BYTES > 65536) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k, grid, dim3(256), C::LDS_BYTES, s, x, ldx, wp, bias, y, ldy, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

template <int CKB, int KT, int KH, int KW>
int launch_pack(const float* w, uint4* wp, int Cin, int Cout, int dgrad, hipStream_t s)
{
    constexpr int KSTEPS = (KT * KW * CKB + 31) / 32;
    const int CK = dgrad ? Cout : Cin, CO = dgrad ? Cin : Cout;
    const long total = (long)(CK / CKB) * KH * KSTEPS * (CO / 16) * 64;
    long blocks = (total + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((pack_weights_kernel<CKB, KT, KH, KW>), dim3((unsigned)blocks), dim3(256), 0, s, w, wp, Cin, Cout, dgrad);
    VVAE_LAUNCH_CHECK();
    return 0;
}

inline int chunk_of(int CK) { return (CK % 32 == 0) ? 32 : 16; }

inline size_t packed_bytes(int CK, int CO, int kt, int kh, int kw)
{
    const int ckb = chunk_of(CK);
    const int ksteps = (kt * kw * ckb + 31) / 32;
    return (size_t)(CK / ckb) * kh * ksteps * (CO / 16) * 64 * 16;
}

}  // namespace

// which: 0 fwd (K = Cin, produced = Cout), 1 dgrad (K = Cout, produced = Cin), 2 wgrad (not on this path yet).
extern "C" int vvae_conv3d_bf16_supported(int Cin, int Cout, int kt, int kh, int kw, int ld_in, int ld_out, int which, int flags)
{
    (void)flags;
    if (which == 2) return 0;
    const int CK = which == 1 ? Cout : Cin, CO = which == 1 ? Cin : Cout;
    if (CK % 16 || CO % 16 || ld_in % 8 || ld_out % 4) return 0;
    if (kt == 3 && kh == 3 && kw == 3) return (CO == 16 || CO == 32 || CO % 64 == 0) ? 1 : 0;
    if (kt == 3 && kh == 7 && kw == 7) return (CK == 16 && CO == 16) ? 1 : 0;
    return 0;
}

extern "C" size_t vvae_conv3d_bf16_ws_bytes(int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int which)
{
    (void)N; (void)T; (void)H; (void)W;
    if (which == 2) return 0;
    if (!vvae_conv3d_bf16_supported(Cin, Cout, kt, kh, kw, 8, 8, which, 0)) return 0;
    const int CK = which == 1 ? Cout : Cin, CO = which == 1 ? Cin : Cout;
    return packed_bytes(CK, CO, kt, kh, kw);
}

// dgrad = 0: y = conv(x, w) + bias.   dgrad = 1: "x" is dy (Cout channels), "y" is dx (Cin channels), bias ignored.
extern "C" int vvae_conv3d_fwd_bf16(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy,
                                    int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int dgrad,
                                    void* ws, size_t ws_bytes, void* stream)
{
    if (!x || !w || !y || N <= 0 || T <= 0 || H <= 0 || W <= 0) return VVAE_ERR_BAD_ARG;
    if (!vvae_conv3d_bf16_supported(Cin, Cout, kt, kh, kw, ldx, ldy, dgrad ? 1 : 0, 0)) return VVAE_ERR_BAD_ARG;
    if (((uintptr_t)x % 16) || ((uintptr_t)y % 8)) return VVAE_ERR_BAD_ARG;
    const int CK = dgrad ? Cout : Cin, CO = dgrad ? Cin : Cout;
    if (ldx < CK || ldy < CO) return VVAE_ERR_BAD_ARG;
    const size_t need = packed_bytes(CK, CO, kt, kh, kw);
    if (!ws || ws_bytes < need || ((uintptr_t)ws % 16)) return VVAE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    uint4* wp = (uint4*)ws;
    const bf16_t* xp = (const bf16_t*)x;
    bf16_t* yp = (bf16_t*)y;
    const float* bp = dgrad ? nullptr : bias;
    BfDims d{N, T, H, W, CK, CO, 0, 0};
    int rc;
    if (kh == 7) {
        if ((rc = launch_pack<16, 3, 7, 7>(w, wp, Cin, Cout, dgrad, s))) return rc;
        return launch_cfg<C377_k16_o16>(xp, ldx, wp, bp, yp, ldy, d, s);
    }
    if (chunk_of(CK) == 16) {
        if ((rc = launch_pack<16, 3, 3, 3>(w, wp, Cin, Cout, dgrad, s))) return rc;
        if (CO == 16) return launch_cfg<C333_k16_o16>(xp, ldx, wp, bp, yp, ldy, d, s);
        if (CO == 32) return launch_cfg<C333_k16_o32>(xp, ldx, wp, bp, yp, ldy, d, s);
        return launch_cfg<C333_k16_o64>(xp, ldx, wp, bp, yp, ldy, d, s);
    }
    if ((rc = launch_pack<32, 3, 3, 3>(w, wp, Cin, Cout, dgrad, s))) return rc;
    if (CO == 16) return launch_cfg<C333_k32_o16>(xp, ldx, wp, bp, yp, ldy, d, s);
    if (CO == 32) return launch_cfg<C333_k32_o32>(xp, ldx, wp, bp, yp, ldy, d, s);
    return launch_cfg<C333_k32_o64>(xp, ldx, wp, bp, yp, ldy, d, s);
}

extern "C" int vvae_conv3d_wgrad_bf16(const void*, int, const void*, int, float*, float*, int, int, int, int, int, int, int, int,
                                      int, void*, size_t, void*)
{
    return VVAE_ERR_BAD_ARG;
}
