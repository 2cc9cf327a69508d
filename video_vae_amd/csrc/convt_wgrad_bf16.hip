// Weight and bias gradient of ConvTranspose (1,2,2)/(1,2,2) on the bf16 matrix cores (reference train/unet.py:61-69 under autodiff):
//   dK[1-a][1-b][ci][co] = sum_v x[v][ci] * dy[up(v,a,b)][co],    db[co] = sum over all output voxels of dy
// for the three up-blocks of the UNet (128->64 @32^2, 64->32 @64^2, 32->16 @128^2 at the production shape).  Four K-major
// products with a tiny output (Cin x Cout) and K = millions of voxels: the recipe of gemm_tn.hip with one twist --
//   * wave w of a workgroup IS tap (a, b) = (w >> 1, w & 1): the four waves share the staged x tile (32 voxels x Cin) and each
//     transposes its own gathered dy tile, so x is read from HBM once for the four taps and the whole Cin x Cout accumulator of
//     a tap (<= 128 VGPRs) stays in one wave's registers for the workgroup's life;
//   * fragments through ds_read_b64_tr_b16 from voxel-major LDS tiles whose row pitch (in 32-byte slots) is odd: the 8 rows
//     a half-wave transposes fall in 8 different bank slots;
//   * register-staged double buffering, one barrier per 32-voxel step;
//   * every workgroup owns a contiguous voxel range and writes one fp32 slab; a second kernel adds the slabs in fixed order
//     (the generic path used float atomics: this one is bitwise reproducible) and folds the four taps' bias sums (all-ones
//     row operand) into db.
#include "common.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4v __attribute__((ext_vector_type(4)));

struct CwDims { int NT, H, W, ldx, lddy; long V; int vpw; };          // vpw = voxels per workgroup (multiple of 32)

__device__ __forceinline__ long up_voxel_cw(long v, int a, int b, int H, int W)
{
    const int w = (int)(v % W); const long q = v / W; const int h = (int)(q % H); const long p = q / H;
    return (p * (2 * H) + 2 * h + a) * (2L * W) + 2 * w + b;
}

template <int CIN, int COUT>
struct CwCfg {
    static constexpr int KS = 32;
    static constexpr int PX = CIN * 2 + (((CIN / 16) & 1) ? 64 : 32);         // pitch / 32 odd
    static constexpr int PY = COUT * 2 + (((COUT / 16) & 1) ? 64 : 32);
    static constexpr int X_BYTES = KS * PX, Y_BYTES = KS * PY, STAGE = X_BYTES + 4 * Y_BYTES;
    static constexpr int XC = KS * CIN / 8, YC = 4 * KS * COUT / 8;            // 16-byte chunks per step
    static constexpr int X_IT = (XC + 255) / 256, Y_IT = (YC + 255) / 256;
    static constexpr int MT = CIN / 16, NT = COUT / 16;
    static constexpr int SLAB = 4 * CIN * COUT + 4 * COUT;                      // floats per workgroup: [tap][ci][co] then [tap][co]
    static_assert(CIN % 16 == 0 && COUT % 16 == 0 && (PX / 32) % 2 == 1 && (PY / 32) % 2 == 1, "tile geometry");
};

template <int PITCH>
__device__ __forceinline__ bf16x8 tr_frag_cw(const unsigned char* p)
{
    typedef __attribute__((address_space(3))) s16x4v* lds_ptr;
    const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 16 * PITCH));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <typename C, int CIN, int COUT>
__global__ __launch_bounds__(256) void convt_wgrad_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ slab,
                                                               CwDims d)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ta = wave >> 1, tb = wave & 1;                                 // this wave's tap
    const long vbeg = (long)blockIdx.x * d.vpw;
    long vend = vbeg + d.vpw;
    if (vend > d.V) vend = d.V;

    // staging registers: x chunks then dy chunks (tap-major)
    uint4 rx[C::X_IT], ry[C::Y_IT];
    auto fetch = [&](long v0) {
#pragma unroll
        for (int it = 0; it < C::X_IT; ++it) {
            const int q = it * 256 + tid;
            rx[it] = make_uint4(0, 0, 0, 0);
            if (q < C::XC) {
                const int row = q / (CIN / 8), part = q % (CIN / 8);
                const long v = v0 + row;
                if (v < vend) rx[it] = *reinterpret_cast<const uint4*>(x + v * d.ldx + part * 8);
            }
        }
#pragma unroll
        for (int it = 0; it < C::Y_IT; ++it) {
            const int q = it * 256 + tid;
            ry[it] = make_uint4(0, 0, 0, 0);
            if (q < C::YC) {
                const int tap = q / (C::KS * COUT / 8), rem = q % (C::KS * COUT / 8);
                const int row = rem / (COUT / 8), part = rem % (COUT / 8);
                const long v = v0 + row;
                if (v < vend) ry[it] = *reinterpret_cast<const uint4*>(dy + up_voxel_cw(v, tap >> 1, tap & 1, d.H, d.W) * d.lddy + part * 8);
            }
        }
    };
    auto park = [&](unsigned char* st) {
#pragma unroll
        for (int it = 0; it < C::X_IT; ++it) {
            const int q = it * 256 + tid;
            if (q < C::XC) *reinterpret_cast<uint4*>(st + (q / (CIN / 8)) * C::PX + (q % (CIN / 8)) * 16) = rx[it];
        }
#pragma unroll
        for (int it = 0; it < C::Y_IT; ++it) {
            const int q = it * 256 + tid;
            if (q < C::YC) {
                const int tap = q / (C::KS * COUT / 8), rem = q % (C::KS * COUT / 8);
                *reinterpret_cast<uint4*>(st + C::X_BYTES + tap * C::Y_BYTES + (rem / (COUT / 8)) * C::PY + (rem % (COUT / 8)) * 16) = ry[it];
            }
        }
    };

    f32x4 acc[C::MT][C::NT], accb[C::NT];
#pragma unroll
    for (int i = 0; i < C::MT; ++i)
#pragma unroll
        for (int j = 0; j < C::NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < C::NT; ++j) accb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const s16x8 ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_s);
    // transposed-read lane offset: lane (g = l>>4, q = (l>>2)&3, p = l&3) -> voxel row 4g+q, channels 4p..4p+3
    const int rowl = 4 * (lane >> 4) + ((lane >> 2) & 3), chl = 8 * (lane & 3);

    fetch(vbeg);
    int buf = 0;
    for (long v0 = vbeg; v0 < vend; v0 += C::KS) {
        unsigned char* st = smem + buf * C::STAGE;
        park(st);
        __syncthreads();
        if (v0 + C::KS < vend) fetch(v0 + C::KS);
        const unsigned char* xs = st + rowl * C::PX + chl;
        const unsigned char* ys = st + C::X_BYTES + wave * C::Y_BYTES + rowl * C::PY + chl;
        bf16x8 bfr[C::NT];
#pragma unroll
        for (int j = 0; j < C::NT; ++j) bfr[j] = tr_frag_cw<C::PY>(ys + j * 32);
#pragma unroll
        for (int i = 0; i < C::MT; ++i) {
            const bf16x8 afr = tr_frag_cw<C::PX>(xs + i * 32);
#pragma unroll
            for (int j = 0; j < C::NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < C::NT; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bfr[j], accb[j], 0, 0, 0);
        buf ^= 1;
    }
    // D[row = ci (4g+e)][col = co (lane & 15)]; slab layout [tap][ci][co] with tap = (1-a)*2 + (1-b)
    const int tap = (1 - ta) * 2 + (1 - tb);
    float* out = slab + (long)blockIdx.x * C::SLAB + (long)tap * CIN * COUT;
    const int col = lane & 15, rg = lane >> 4;
#pragma unroll
    for (int i = 0; i < C::MT; ++i)
#pragma unroll
        for (int j = 0; j < C::NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[(i * 16 + rg * 4 + e) * COUT + j * 16 + col] = acc[i][j][e];
    if (rg == 0) {
        float* ob = slab + (long)blockIdx.x * C::SLAB + 4 * CIN * COUT + tap * COUT;
#pragma unroll
        for (int j = 0; j < C::NT; ++j) ob[j * 16 + col] = accb[j][0];
    }
}

// Fold the per-workgroup slabs in fixed order: column c < nw -> dw[c]; the 4 x cout bias columns behind them (one run per tap) -> db[co] as
// ((t0 + t1) + (t2 + t3)) of the four folded taps -- one launch (round 3; the taps were folded by a second 5 us launch).
// Block = 32 columns x 8 row lanes, 8 loads in flight per thread (the slabs are L2-resident: latency is what the fold pays for).  Blocks
// [0, ceil(nw / 32)) take weight columns; the blocks behind them take 8 output channels x 4 taps each, so that a tap fold stays inside a block
// and no thread walks more rows than any other.
__global__ __launch_bounds__(256) void cw_reduce_kernel(const float* __restrict__ part, int rows, long stride, int nw, int cout,
                                                        float* __restrict__ dw, float* __restrict__ db)
{
    __shared__ float red[8][32];
    __shared__ float taps[32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int wblocks = (nw + 31) / 32;
    const bool bias_blk = (int)blockIdx.x >= wblocks;
    long col = -1;
    int co = -1;
    if (!bias_blk) { const int c = blockIdx.x * 32 + cl; if (c < nw) col = c; }
    else {
        co = ((int)blockIdx.x - wblocks) * 8 + (cl & 7);
        if (co < cout) col = nw + (long)(cl >> 3) * cout + co;      // tap cl >> 3
    }
    float s = 0.f;
    if (col >= 0) {
        for (int r0 = rl; r0 < rows; r0 += 8 * 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + 8 * i;
                v[i] = r < rows ? part[(long)r * stride + col] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
        }
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0) {
        const float t = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]));
        if (!bias_blk) { if (col >= 0) dw[col] = t; }
        else taps[cl] = t;
    }
    __syncthreads();
    if (bias_blk && threadIdx.x < 8 && co >= 0 && co < cout && db)
        db[co] = (taps[threadIdx.x] + taps[8 + threadIdx.x]) + (taps[16 + threadIdx.x] + taps[24 + threadIdx.x]);
}

inline bool cw_shape(int Cin, int Cout) { return (Cin == 128 && Cout == 64) || (Cin == 64 && Cout == 32) || (Cin == 32 && Cout == 16); }

// workgroups: enough to fill the chip, but at most ~16 MB of slabs
inline int cw_blocks(long V, int Cin, int Cout)
{
    const long slab_bytes = (4L * Cin * Cout + 4 * Cout) * 4;
    long nb = (16L << 20) / slab_bytes;
    if (nb > 1024) nb = 1024;
    const long by_work = (V + 63) / 64;                                       // at least two 32-voxel steps per workgroup
    if (nb > by_work) nb = by_work;
    return (int)(nb < 1 ? 1 : nb);
}

template <int CIN, int COUT>
int launch_cw(const void* x, const void* dy, float* dw, float* db, float* slab, CwDims d, int nblk, hipStream_t s)
{
    typedef CwCfg<CIN, COUT> C;
    auto k = convt_wgrad_bf16_kernel<C, CIN, COUT>;
    constexpr int lds = 2 * C::STAGE;
    static bool attr_done = false;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(nblk), dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)dy, slab, d);
    VVAE_LAUNCH_CHECK();
    // fold the slabs in fixed order: weight columns straight into dw, the 4 x COUT bias columns (one run per tap) into db
    const int nw = 4 * CIN * COUT;
    hipLaunchKernelGGL(cw_reduce_kernel, dim3(ceil_div(nw, 32) + (db ? ceil_div(COUT, 8) : 0)), dim3(256), 0, s, slab, nblk, (long)C::SLAB, nw, COUT, dw, db);
    VVAE_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// 1 if vvae_convt_1x2x2_wgrad_bf16 takes this layer (bf16; 128->64, 64->32 or 32->16; 16-byte aligned pitches).
extern "C" int vvae_convt_wgrad_bf16_supported(int Cin, int Cout, int ldx, int lddy)
{
    return (cw_shape(Cin, Cout) && ldx >= Cin && lddy >= Cout && ldx % 8 == 0 && lddy % 8 == 0) ? 1 : 0;
}

extern "C" size_t vvae_convt_wgrad_bf16_ws_bytes(int NT, int H, int W, int Cin, int Cout)
{
    if (!cw_shape(Cin, Cout)) return 0;
    return ((size_t)cw_blocks((long)NT * H * W, Cin, Cout) + 1) * (4 * (size_t)Cin * Cout + 4 * Cout) * sizeof(float);
}

// x (NT,H,W,Cin) row pitch ldx; dy (NT,2H,2W,Cout) row pitch lddy; dw (1,2,2,Cin,Cout) fp32 and dbias (Cout) fp32 (or NULL) overwritten.
extern "C" int vvae_convt_1x2x2_wgrad_bf16(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias, int NT, int H, int W,
                                           int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    if (!x || !dy || !dw || NT <= 0 || H <= 0 || W <= 0 || !vvae_convt_wgrad_bf16_supported(Cin, Cout, ldx, lddy) || ((uintptr_t)x % 16) ||
        ((uintptr_t)dy % 16)) return VVAE_ERR_BAD_ARG;
    if (!ws || ws_bytes < vvae_convt_wgrad_bf16_ws_bytes(NT, H, W, Cin, Cout) || ((uintptr_t)ws % 16)) return VVAE_ERR_WORKSPACE;
    const long V = (long)NT * H * W;
    const int nblk = cw_blocks(V, Cin, Cout);
    long vpw = (V + nblk - 1) / nblk;
    vpw = (vpw + 31) / 32 * 32;
    const int used = (int)((V + vpw - 1) / vpw);
    CwDims d{NT, H, W, ldx, lddy, V, (int)vpw};
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 128) return launch_cw<128, 64>(x, dy, dw, dbias, (float*)ws, d, used, s);
    if (Cin == 64) return launch_cw<64, 32>(x, dy, dw, dbias, (float*)ws, d, used, s);
    return launch_cw<32, 16>(x, dy, dw, dbias, (float*)ws, d, used, s);
}
