// Spatial max-pool (1,2,2) and ConvTranspose (1,2,2)/(1,2,2), channels-last, forward and backward.
//
// max-pool  : nnx.max_pool(x, (1,2,2), strides (1,2,2)) of DownBlock3D, /root/reference/train/unet.py:50.
// convT     : nnx.ConvTranspose(kernel (1,2,2), strides (1,2,2), SAME) of UpBlock3D, unet.py:61-69,78.
//             lax.conv_transpose does NOT flip the kernel, so out[2i+d] = x[i] * K[1-d] per spatial axis
//             (SURVEY.md Appendix A.4): every input voxel owns a disjoint 2x2 output block.
//
// Pool is a pure HBM stream (16-byte vectors).  ConvTranspose is a pointwise GEMM + pixel-shuffle store on
// the fp32 matrix cores (v_mfma_f32_16x16x4_f32), writing straight into a channel slice (row pitch ldy)
// of the concat buffer so jnp.concatenate (unet.py:80) never materialises.
#include "common.hpp"

namespace {

struct PoolDims { int NT; int H, W, C; };   // NT = n*t planes; H, W = input (full) resolution

template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, PoolDims d)
{
    const int cvecs = d.C / VEC, Ho = d.H / 2, Wo = d.W / 2;
    const long items = (long)d.NT * Ho * Wo * cvecs;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c0 = (int)(it % cvecs) * VEC; long q = it / cvecs;
        const int wo = (int)(q % Wo); q /= Wo;
        const int ho = (int)(q % Ho); const long p = q / Ho;
        const long vi = (p * d.H + 2 * ho) * d.W + 2 * wo;
        float a[VEC], b[VEC], c[VEC], e[VEC];
        VecIO<T, VEC>::load(x + vi * ldx + c0, a);
        VecIO<T, VEC>::load(x + (vi + 1) * ldx + c0, b);
        VecIO<T, VEC>::load(x + (vi + d.W) * ldx + c0, c);
        VecIO<T, VEC>::load(x + (vi + d.W + 1) * ldx + c0, e);
#pragma unroll
        for (int i = 0; i < VEC; ++i) a[i] = fmaxf(fmaxf(a[i], b[i]), fmaxf(c[i], e[i]));
        VecIO<T, VEC>::store(y + ((p * Ho + ho) * Wo + wo) * ldy + c0, a);
    }
}

// dx[v] = (dskip ? dskip[v] : 0) + (v is the first arg-max of its window ? dpool[window] : 0)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dpool, int lddp,
                                                          const T* __restrict__ dskip, int ldds, T* __restrict__ dx, int lddx,
                                                          PoolDims d)
{
    const int cvecs = d.C / VEC, Ho = d.H / 2, Wo = d.W / 2;
    const long items = (long)d.NT * Ho * Wo * cvecs;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int c0 = (int)(it % cvecs) * VEC; long q = it / cvecs;
        const int wo = (int)(q % Wo); q /= Wo;
        const int ho = (int)(q % Ho); const long p = q / Ho;
        const long vi = (p * d.H + 2 * ho) * d.W + 2 * wo;
        const long vs[4] = {vi, vi + 1, vi + d.W, vi + d.W + 1};
        float in[4][VEC], g[VEC], out[4][VEC];
#pragma unroll
        for (int k = 0; k < 4; ++k) VecIO<T, VEC>::load(x + vs[k] * ldx + c0, in[k]);
        VecIO<T, VEC>::load(dpool + ((p * Ho + ho) * Wo + wo) * lddp + c0, g);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (dskip) VecIO<T, VEC>::load(dskip + vs[k] * ldds + c0, out[k]);
            else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) out[k][i] = 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            int best = 0; float m = in[0][i];
#pragma unroll
            for (int k = 1; k < 4; ++k) if (in[k][i] > m) { m = in[k][i]; best = k; }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k == best) out[k][i] += g[i];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) VecIO<T, VEC>::store(dx + vs[k] * lddx + c0, out[k]);
    }
}

// ------------------------------------------------------------------------------------------- ConvTranspose 1x2x2
struct CtDims { int NT; int H, W, Cin, Cout; };   // H, W = input (low) resolution; output is 2H x 2W

__device__ __forceinline__ long up_voxel(long v, int a, int b, int H, int W) {
    const int w = (int)(v % W); const long q = v / W; const int h = (int)(q % H); const long p = q / H;
    return (p * (2 * H) + 2 * h + a) * (2L * W) + 2 * w + b;
}

// MODE 0 (fwd)  : y[up(v,a,b)][co] = bias[co] + sum_ci x[v][ci] * K[1-a][1-b][ci][co]   (blockIdx.z = a*2+b)
// MODE 1 (dgrad): dx[v][ci] = sum_{a,b,co} dy[up(v,a,b)][co] * K[1-a][1-b][ci][co]
template <typename T, int MODE, int NT>
__global__ __launch_bounds__(256) void convt_f32mfma_kernel(const T* __restrict__ in, int ldin, const float* __restrict__ w,
                                                            const float* __restrict__ bias, T* __restrict__ out, int ldout, CtDims d)
{
    const int CK = MODE == 0 ? d.Cin : d.Cout, CO = MODE == 0 ? d.Cout : d.Cin;
    const long V = (long)d.NT * d.H * d.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4;
    const long tile0 = ((long)blockIdx.x * 4 + wave) * 16;
    const int o_base = blockIdx.y * NT * 16;
    long v = tile0 + r;
    const bool vvalid = v < V;
    if (!vvalid) v = V - 1;
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ab_lo = MODE == 0 ? blockIdx.z : 0, ab_hi = MODE == 0 ? blockIdx.z + 1 : 4;
    for (int ab = ab_lo; ab < ab_hi; ++ab) {
        const int a = ab >> 1, b = ab & 1;
        const float* wt = w + (long)((1 - a) * 2 + (1 - b)) * d.Cin * d.Cout;
        const T* row = MODE == 0 ? in + v * (long)ldin : in + up_voxel(v, a, b, d.H, d.W) * (long)ldin;
        for (int k0 = 0; k0 < CK; k0 += 4) {
            const int kk = k0 + kq; const bool kin = kk < CK;
            const float av = (vvalid && kin) ? ldf(row + kk) : 0.f;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int o = o_base + i * 16 + r;
                float bv = 0.f;
                if (kin && o < CO) bv = MODE == 0 ? wt[(long)kk * d.Cout + o] : wt[(long)o * d.Cout + kk];
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int o = o_base + i * 16 + r;
        if (o >= CO) continue;
        const float bb = (MODE == 0 && bias) ? bias[o] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long vo = tile0 + kq * 4 + j;
            if (vo >= V) continue;
            const long dst = MODE == 0 ? up_voxel(vo, blockIdx.z >> 1, blockIdx.z & 1, d.H, d.W) : vo;
            stf(out + dst * (long)ldout + o, acc[i][j] + bb);
        }
    }
}

// part[chunk][1-a][1-b][ci][co] = sum over the chunk's voxels of x[v][ci] * dy[up(v,a,b)][co]; every element written by one workgroup, the chunks
// folded in index order afterwards (no atomics).   grid: x = voxel chunk, y = ab, z = ci tile
template <typename T, int NT>
__global__ __launch_bounds__(256) void convt_wgrad_f32mfma_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                                  float* __restrict__ part, CtDims d, int co_tile_base, int voxels_per_block)
{
    __shared__ float red[4][NT][64][4];
    const long V = (long)d.NT * d.H * d.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4;
    const int a = blockIdx.y >> 1, b = blockIdx.y & 1;
    const int tap = (1 - a) * 2 + (1 - b);
    const int ci = blockIdx.z * 16 + r;
    const long vbeg = (long)blockIdx.x * voxels_per_block;
    long vend = vbeg + voxels_per_block;
    if (vend > V) vend = V;
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long v0 = vbeg + wave * 4; v0 < vend; v0 += 16) {
        const long v = v0 + kq; const bool vok = v < vend; const long vc = vok ? v : vbeg;
        const float av = (vok && ci < d.Cin) ? ldf(x + vc * (long)ldx + ci) : 0.f;
        const T* dyrow = dy + up_voxel(vc, a, b, d.H, d.W) * (long)lddy;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int co = (co_tile_base + i) * 16 + r;
            const float bv = (vok && co < d.Cout) ? ldf(dyrow + co) : 0.f;
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave][i][lane][j] = acc[i][j];
    __syncthreads();
    for (int e = threadIdx.x; e < NT * 256; e += 256) {
        const int i = e >> 8, l = (e >> 2) & 63, j = e & 3;
        const float s = red[0][i][l][j] + red[1][i][l][j] + red[2][i][l][j] + red[3][i][l][j];
        const int cii = blockIdx.z * 16 + (l >> 4) * 4 + j, co = (co_tile_base + i) * 16 + (l & 15);
        if (cii < d.Cin && co < d.Cout) part[(((long)blockIdx.x * 4 + tap) * d.Cin + cii) * d.Cout + co] = s;
    }
}

template <typename T> bool vec_ok2(const void* p, int ld, int C) {
    constexpr int V = VecWidth<T>::value;
    return C % V == 0 && ld % V == 0 && ((uintptr_t)p % 16) == 0;
}

template <typename T, int MODE>
int launch_convt(const void* in, int ldin, const float* w, const float* bias, void* out, int ldout, CtDims d, hipStream_t s)
{
    const long V = (long)d.NT * d.H * d.W;
    const int CO = MODE == 0 ? d.Cout : d.Cin;
    const int tiles = ceil_div(CO, 16);
    const int nt = tiles >= 8 ? 8 : tiles >= 4 ? 4 : tiles >= 2 ? 2 : 1;
    dim3 grid(ceil_div(V, 64), ceil_div(tiles, nt), MODE == 0 ? 4 : 1);
#define GO(NTV) hipLaunchKernelGGL((convt_f32mfma_kernel<T, MODE, NTV>), grid, dim3(256), 0, s, (const T*)in, ldin, w, bias, (T*)out, ldout, d)
    switch (nt) { case 8: GO(8); break; case 4: GO(4); break; case 2: GO(2); break; default: GO(1); break; }
#undef GO
    VVAE_LAUNCH_CHECK();
    return 0;
}

inline long convt_wgrad_chunks(const CtDims& d, long* vpb_out)
{
    const long V = (long)d.NT * d.H * d.W;
    long want = 2048 / (4L * ceil_div(d.Cin, 16)); if (want < 1) want = 1;
    long vpb = (V + want - 1) / want; vpb = ((vpb + 15) / 16) * 16; if (vpb < 256) vpb = 256;
    if (vpb_out) *vpb_out = vpb;
    return ceil_div(V, vpb);
}

template <typename T>
int launch_convt_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, CtDims d, float* ws, hipStream_t s)
{
    const int ci_tiles = ceil_div(d.Cin, 16), co_tiles = ceil_div(d.Cout, 16);
    long vpb = 0;
    const long chunks = convt_wgrad_chunks(d, &vpb);
    dim3 grid((unsigned)chunks, 4, ci_tiles);
    for (int base = 0; base < co_tiles;) {
        const int rem = co_tiles - base;
        const int nt = rem >= 8 ? 8 : rem >= 4 ? 4 : rem >= 2 ? 2 : 1;
#define GO(NTV) hipLaunchKernelGGL((convt_wgrad_f32mfma_kernel<T, NTV>), grid, dim3(256), 0, s, (const T*)x, ldx, (const T*)dy, lddy, ws, d, base, (int)vpb)
        switch (nt) { case 8: GO(8); break; case 4: GO(4); break; case 2: GO(2); break; default: GO(1); break; }
#undef GO
        VVAE_LAUNCH_CHECK();
        base += nt;
    }
    const long ncols = 4L * d.Cin * d.Cout;
    hipLaunchKernelGGL(vvae_reduce_rows_kernel, dim3((unsigned)ceil_div(ncols, 32L)), dim3(256), 0, s, (const float*)ws, (int)chunks, ncols, (int)ncols,
                       dw, (int)ncols, (float*)nullptr);
    VVAE_LAUNCH_CHECK();
    return 0;
}

}  // namespace

#define POOL_DISPATCH(KERNEL, VOK, ...)                                                                       \
    do {                                                                                                      \
        if (dtype == VVAE_DT_F32) { typedef float T;                                                          \
            if (VOK) hipLaunchKernelGGL((KERNEL<T, 4>), grid, dim3(256), 0, s, __VA_ARGS__);                  \
            else hipLaunchKernelGGL((KERNEL<T, 1>), grid, dim3(256), 0, s, __VA_ARGS__);                      \
        } else { typedef bf16_t T;                                                                            \
            if (VOK) hipLaunchKernelGGL((KERNEL<T, 8>), grid, dim3(256), 0, s, __VA_ARGS__);                  \
            else hipLaunchKernelGGL((KERNEL<T, 1>), grid, dim3(256), 0, s, __VA_ARGS__);                      \
        }                                                                                                     \
    } while (0)

extern "C" int vvae_maxpool_1x2x2_fwd(const void* x, int ldx, void* y, int ldy, int NT, int H, int W, int C, int dtype, void* stream)
{
    if (!x || !y || NT <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || ldx < C || ldy < C ||
        (dtype != VVAE_DT_F32 && dtype != VVAE_DT_BF16)) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    PoolDims d{NT, H, W, C};
    const bool vok = dtype == VVAE_DT_F32 ? (vec_ok2<float>(x, ldx, C) && vec_ok2<float>(y, ldy, C))
                                          : (vec_ok2<bf16_t>(x, ldx, C) && vec_ok2<bf16_t>(y, ldy, C));
    const int vec = vok ? (dtype == VVAE_DT_F32 ? 4 : 8) : 1;
    const long items = (long)NT * (H / 2) * (W / 2) * (C / vec);
    dim3 grid((unsigned)(items / 256 + 1 > 8192 ? 8192 : items / 256 + 1));
    POOL_DISPATCH(maxpool_fwd_kernel, vok, (const T*)x, ldx, (T*)y, ldy, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

extern "C" int vvae_maxpool_1x2x2_bwd(const void* x, int ldx, const void* dpool, int lddp, const void* dskip, int ldds,
                                      void* dx, int lddx, int NT, int H, int W, int C, int dtype, void* stream)
{
    if (!x || !dpool || !dx || NT <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || ldx < C || lddp < C ||
        lddx < C || (dskip && ldds < C) || (dtype != VVAE_DT_F32 && dtype != VVAE_DT_BF16)) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    PoolDims d{NT, H, W, C};
    bool vok;
    if (dtype == VVAE_DT_F32)
        vok = vec_ok2<float>(x, ldx, C) && vec_ok2<float>(dpool, lddp, C) && vec_ok2<float>(dx, lddx, C) && (!dskip || vec_ok2<float>(dskip, ldds, C));
    else
        vok = vec_ok2<bf16_t>(x, ldx, C) && vec_ok2<bf16_t>(dpool, lddp, C) && vec_ok2<bf16_t>(dx, lddx, C) && (!dskip || vec_ok2<bf16_t>(dskip, ldds, C));
    const int vec = vok ? (dtype == VVAE_DT_F32 ? 4 : 8) : 1;
    const long items = (long)NT * (H / 2) * (W / 2) * (C / vec);
    dim3 grid((unsigned)(items / 256 + 1 > 8192 ? 8192 : items / 256 + 1));
    POOL_DISPATCH(maxpool_bwd_kernel, vok, (const T*)x, ldx, (const T*)dpool, lddp, (const T*)dskip, ldds, (T*)dx, lddx, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// w: Flax ConvTranspose kernel (1,2,2,Cin,Cout) fp32.  y has spatial size 2H x 2W and row pitch ldy.
extern "C" int vvae_convt_1x2x2_fwd(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy,
                                    int NT, int H, int W, int Cin, int Cout, int dtype, void* stream)
{
    if (!x || !w || !y || NT <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ldx < Cin || ldy < Cout) return VVAE_ERR_BAD_ARG;
    CtDims d{NT, H, W, Cin, Cout};
    if (dtype == VVAE_DT_F32) return launch_convt<float, 0>(x, ldx, w, bias, y, ldy, d, (hipStream_t)stream);
    if (dtype == VVAE_DT_BF16) return launch_convt<bf16_t, 0>(x, ldx, w, bias, y, ldy, d, (hipStream_t)stream);
    return VVAE_ERR_BAD_ARG;
}

extern "C" int vvae_convt_1x2x2_dgrad(const void* dy, int lddy, const float* w, void* dx, int lddx,
                                      int NT, int H, int W, int Cin, int Cout, int dtype, void* stream)
{
    if (!dy || !w || !dx || NT <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || lddy < Cout || lddx < Cin) return VVAE_ERR_BAD_ARG;
    CtDims d{NT, H, W, Cin, Cout};
    if (dtype == VVAE_DT_F32) return launch_convt<float, 1>(dy, lddy, w, nullptr, dx, lddx, d, (hipStream_t)stream);
    if (dtype == VVAE_DT_BF16) return launch_convt<bf16_t, 1>(dy, lddy, w, nullptr, dx, lddx, d, (hipStream_t)stream);
    return VVAE_ERR_BAD_ARG;
}

// dw (1,2,2,Cin,Cout) fp32 overwritten.  (dbias = vvae_colsum(dy).)
// Scratch bytes of vvae_convt_1x2x2_wgrad: one fp32 partial dK per voxel chunk.
extern "C" size_t vvae_convt_1x2x2_wgrad_ws_bytes(int NT, int H, int W, int Cin, int Cout)
{
    if (NT <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    CtDims d{NT, H, W, Cin, Cout};
    return (size_t)convt_wgrad_chunks(d, nullptr) * 4 * Cin * Cout * sizeof(float);
}

// dw (1,2,2,Cin,Cout) fp32 overwritten; ws: vvae_convt_1x2x2_wgrad_ws_bytes(...) bytes.  No atomics: per-chunk partials folded in index order.
extern "C" int vvae_convt_1x2x2_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw,
                                      int NT, int H, int W, int Cin, int Cout, int dtype, void* ws, size_t ws_bytes, void* stream)
{
    if (!x || !dy || !dw || NT <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || ldx < Cin || lddy < Cout) return VVAE_ERR_BAD_ARG;
    if (!ws || ws_bytes < vvae_convt_1x2x2_wgrad_ws_bytes(NT, H, W, Cin, Cout) || ((uintptr_t)ws % 16)) return VVAE_ERR_WORKSPACE;
    CtDims d{NT, H, W, Cin, Cout};
    if (dtype == VVAE_DT_F32) return launch_convt_wgrad<float>(x, ldx, dy, lddy, dw, d, (float*)ws, (hipStream_t)stream);
    if (dtype == VVAE_DT_BF16) return launch_convt_wgrad<bf16_t>(x, ldx, dy, lddy, dw, d, (float*)ws, (hipStream_t)stream);
    return VVAE_ERR_BAD_ARG;
}

// =========================================================================================== ConvTranspose, bf16 MFMA path
// Pointwise GEMM + pixel shuffle on v_mfma_f32_16x16x32_bf16, no LDS at all: the voxel operand fragment (8 consecutive
// channels of one voxel per lane) is a 16-byte global load straight from the NDHWC row, the weights (<= 64 KB) are packed
// once into fragment order and held in REGISTERS for the life of the wave, and with the weights as the A operand a lane
// ends up with 4 consecutive output channels of one voxel (8-byte stores into the pitched destination slice).
//   MODE 0 (fwd)  : y[up(v,a,b)][co] = bias[co] + sum_ci x[v][ci] K[1-a][1-b][ci][co]      K dim = Cin,     tiles = 4*Cout/16
//   MODE 1 (dgrad): dx[v][ci]        = sum_{a,b,co} dy[up(v,a,b)][co] K[1-a][1-b][ci][co]   K dim = 4*Cout,  tiles = Cin/16
// WSPLIT waves share a voxel tile and split the output tiles (keeps <= 16 weight fragments = 64 VGPRs per wave).
namespace {

typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));

template <int MODE, int CIN, int COUT> struct CtCfg {
    static constexpr int KSTEPS = MODE == 0 ? CIN / 32 : (4 * COUT) / 32;
    static constexpr int OT = MODE == 0 ? (4 * COUT) / 16 : CIN / 16;
    static constexpr int WSPLIT = (OT * KSTEPS > 16) ? 4 : 1;
    static constexpr int NTL = OT / WSPLIT;
    static_assert(NTL * KSTEPS <= 16 && OT % WSPLIT == 0, "weight fragments must fit in registers");
};

// packed[(ot*KSTEPS + ks)*64 + lane] = 8 bf16: row = lane&15 (output channel in tile), k = 32ks + 8(lane>>4) + e
template <int MODE, int CIN, int COUT>
__device__ __forceinline__ void convt_pack_item(const float* __restrict__ w, uint4* __restrict__ wp, int i)
{
    typedef CtCfg<MODE, CIN, COUT> C;
    const int l = i & 63, ks = (i >> 6) % C::KSTEPS, ot = (i >> 6) / C::KSTEPS;
    const int row = l & 15, g = l >> 4;
    uint32_t pk[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int a, b, ci, co;
        if (MODE == 0) {
            const int cot = ot % (COUT / 16), ab = ot / (COUT / 16);
            a = ab >> 1; b = ab & 1; co = cot * 16 + row; ci = 32 * ks + 8 * g + e;
        } else {
            ci = ot * 16 + row;
            int ab;
            if (COUT >= 32) { ab = ks / (COUT / 32); co = (ks % (COUT / 32)) * 32 + 8 * g + e; }
            else { ab = 2 * ks + (g >> 1); co = 8 * (g & 1) + e; }
            a = ab >> 1; b = ab & 1;
        }
        const float v = w[((long)((1 - a) * 2 + (1 - b)) * CIN + ci) * COUT + co];
        const uint32_t h = f2bf(v);
        if (e & 1) pk[e >> 1] |= h << 16; else pk[e >> 1] = h;
    }
    wp[i] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
}

template <int MODE, int CIN, int COUT>
__global__ void convt_pack_kernel(const float* __restrict__ w, uint4* __restrict__ wp)
{
    typedef CtCfg<MODE, CIN, COUT> C;
    const int total = C::OT * C::KSTEPS * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) convt_pack_item<MODE, CIN, COUT>(w, wp, i);
}

// Grouped form: the forward and input-gradient packings of every ConvTranspose layer of a network in ONE launch, once per optimizer
// step (they were a 5 us launch in front of each of the six ConvTranspose launches of a step).  Every packing is 4 Cin Cout / 8 fragment
// lanes whatever the mode; blocks [block_start_e, block_start_{e+1}) of 256 lanes belong to entry e.
constexpr int CT_PACK_MAX = 16;
struct CtPackEntry { const float* w; uint4* wp; int mode, cin, total, block_start; };
struct CtPackArgs { CtPackEntry e[CT_PACK_MAX]; int n; };

__global__ __launch_bounds__(256) void convt_pack_grouped_kernel(CtPackArgs g)
{
    int ei = 0;
    for (int i = 1; i < g.n; ++i) ei = (int)blockIdx.x >= g.e[i].block_start ? i : ei;
    const CtPackEntry& E = g.e[ei];
    const int i = ((int)blockIdx.x - E.block_start) * 256 + threadIdx.x;
    if (i >= E.total) return;
    if (E.mode == 0) {
        if (E.cin == 128) convt_pack_item<0, 128, 64>(E.w, E.wp, i);
        else if (E.cin == 64) convt_pack_item<0, 64, 32>(E.w, E.wp, i);
        else convt_pack_item<0, 32, 16>(E.w, E.wp, i);
    } else {
        if (E.cin == 128) convt_pack_item<1, 128, 64>(E.w, E.wp, i);
        else if (E.cin == 64) convt_pack_item<1, 64, 32>(E.w, E.wp, i);
        else convt_pack_item<1, 32, 16>(E.w, E.wp, i);
    }
}

template <int MODE, int CIN, int COUT>
__global__ __launch_bounds__(256) void convt_bf16_kernel(const bf16_t* __restrict__ in, int ldin, const uint4* __restrict__ wp,
                                                         const float* __restrict__ bias, bf16_t* __restrict__ out, int ldout, int NT, int H, int W)
{
    typedef CtCfg<MODE, CIN, COUT> C;
    constexpr int KSTEPS = C::KSTEPS, NTL = C::NTL, WSPLIT = C::WSPLIT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int wsub = wave % WSPLIT, vsub = wave / WSPLIT;          // which output tiles / which voxel tile of the group
    constexpr int VT_PER_BLOCK = 4 / WSPLIT;
    const long V = (long)NT * H * W;
    const long nvt = (V + 15) / 16;

    bf16x8c wfr[NTL][KSTEPS];
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
            wfr[i][ks] = __builtin_bit_cast(bf16x8c, wp[((wsub + WSPLIT * i) * KSTEPS + ks) * 64 + lane]);

    for (long vt = (long)blockIdx.x * VT_PER_BLOCK + vsub; vt < nvt; vt += (long)gridDim.x * VT_PER_BLOCK) {
        const long v = vt * 16 + r;
        const bool ok = v < V;
        const long vc = ok ? v : 0;
        const int wq = (int)(vc % W); const long q2 = vc / W; const int hq = (int)(q2 % H); const long pq = q2 / H;
        const long up00 = (pq * (2 * H) + 2 * hq) * (2L * W) + 2 * wq;        // up(v, 0, 0); up(v,a,b) = up00 + a*2W + b
        bf16x8c xf[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const bf16_t* p;
            if (MODE == 0) p = in + vc * ldin + 32 * ks + 8 * g;
            else {
                int ab, ch;
                if (COUT >= 32) { ab = ks / (COUT / 32); ch = (ks % (COUT / 32)) * 32 + 8 * g; }
                else { ab = 2 * ks + (g >> 1); ch = 8 * (g & 1); }
                p = in + (up00 + (ab >> 1) * 2L * W + (ab & 1)) * ldin + ch;
            }
            uint4 t = make_uint4(0, 0, 0, 0);
            if (ok) t = *reinterpret_cast<const uint4*>(p);
            xf[ks] = __builtin_bit_cast(bf16x8c, t);
        }
#pragma unroll
        for (int i = 0; i < NTL; ++i) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[i][ks], xf[ks], acc, 0, 0, 0);
            const int ot = wsub + WSPLIT * i;
            long dst; int ch;
            if (MODE == 0) {
                const int cot = ot % (COUT / 16), ab = ot / (COUT / 16);
                dst = up00 + (ab >> 1) * 2L * W + (ab & 1); ch = cot * 16 + 4 * g;
            } else { dst = vc; ch = ot * 16 + 4 * g; }
            if (ok) {
                float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
                if (MODE == 0 && bias) { b0 = bias[ch]; b1 = bias[ch + 1]; b2 = bias[ch + 2]; b3 = bias[ch + 3]; }
                uint2 o;
                o.x = (uint32_t)f2bf(acc[0] + b0) | ((uint32_t)f2bf(acc[1] + b1) << 16);
                o.y = (uint32_t)f2bf(acc[2] + b2) | ((uint32_t)f2bf(acc[3] + b3) << 16);
                *reinterpret_cast<uint2*>(out + dst * ldout + ch) = o;
            }
        }
    }
}

template <int MODE, int CIN, int COUT>
int launch_convt_bf16(const void* in, int ldin, const float* w, const float* bias, void* out, int ldout, int NT, int H, int W, void* ws,
                      hipStream_t s, bool prepacked)
{
    typedef CtCfg<MODE, CIN, COUT> C;
    uint4* wp = (uint4*)ws;
    if (!prepacked) {
        hipLaunchKernelGGL((convt_pack_kernel<MODE, CIN, COUT>), dim3(ceil_div(C::OT * C::KSTEPS * 64, 256)), dim3(256), 0, s, w, wp);
        VVAE_LAUNCH_CHECK();
    }
    const long nvt = ((long)NT * H * W + 15) / 16;
    long blocks = (nvt + (4 / C::WSPLIT) - 1) / (4 / C::WSPLIT);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((convt_bf16_kernel<MODE, CIN, COUT>), dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)in, ldin, wp, bias,
                       (bf16_t*)out, ldout, NT, H, W);
    VVAE_LAUNCH_CHECK();
    return 0;
}

inline bool convt_bf16_shape(int Cin, int Cout) { return (Cin == 128 && Cout == 64) || (Cin == 64 && Cout == 32) || (Cin == 32 && Cout == 16); }

}  // namespace

// 1 if the bf16 MFMA ConvTranspose path takes this shape (the three UNet decoder levels), pitches in elements.
extern "C" int vvae_convt_bf16_supported(int Cin, int Cout, int ld_in, int ld_out)
{
    return (convt_bf16_shape(Cin, Cout) && ld_in % 8 == 0 && ld_out % 4 == 0) ? 1 : 0;
}

extern "C" size_t vvae_convt_bf16_ws_bytes(int Cin, int Cout) { return convt_bf16_shape(Cin, Cout) ? (size_t)4 * Cin * Cout * 2 : 0; }

// Pack n <= 16 ConvTranspose kernels w[i] (1,2,2,Cin[i],Cout[i]) fp32 for the forward (dgrad[i] = 0) or input-gradient (1) kernel into
// ws[i] (>= vvae_convt_bf16_ws_bytes(Cin[i], Cout[i]) bytes, 16-byte aligned) in one launch; pass dgrad | 0x100 to vvae_convt_1x2x2_bf16
// afterwards.  Host arrays of device pointers / ints.
extern "C" int vvae_convt_pack_grouped_bf16(const float* const* w, void* const* ws, const int* Cin, const int* Cout, const int* dgrad, int n,
                                            void* stream)
{
    if (!w || !ws || !Cin || !Cout || !dgrad || n <= 0 || n > CT_PACK_MAX) return VVAE_ERR_BAD_ARG;
    CtPackArgs g;
    g.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!w[i] || !ws[i] || ((uintptr_t)ws[i] % 16) || !convt_bf16_shape(Cin[i], Cout[i])) return VVAE_ERR_BAD_ARG;
        const int total = 4 * Cin[i] * Cout[i] / 8;
        g.e[i] = CtPackEntry{w[i], (uint4*)ws[i], dgrad[i] & 1, Cin[i], total, blocks};
        blocks += ceil_div(total, 256);
    }
    hipLaunchKernelGGL(convt_pack_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// dgrad bit 0 = 0: x (NT,H,W,Cin) -> y (NT,2H,2W,Cout) + bias.  bit 0 = 1: "x" is dy (NT,2H,2W,Cout), "y" is dx (NT,H,W,Cin).
// bit 8 (0x100): ws already holds this direction's packed weights (vvae_convt_pack_grouped_bf16); w may then be NULL.
extern "C" int vvae_convt_1x2x2_bf16(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy,
                                     int NT, int H, int W, int Cin, int Cout, int dgrad, void* ws, size_t ws_bytes, void* stream)
{
    const bool prepacked = (dgrad & 0x100) != 0;
    dgrad &= 1;
    if (!x || (!w && !prepacked) || !y || NT <= 0 || H <= 0 || W <= 0 || !convt_bf16_shape(Cin, Cout)) return VVAE_ERR_BAD_ARG;
    const int cin_side = dgrad ? Cout : Cin, cout_side = dgrad ? Cin : Cout;
    if (ldx < cin_side || ldy < cout_side || ldx % 8 || ldy % 4 || ((uintptr_t)x % 16) || ((uintptr_t)y % 8)) return VVAE_ERR_BAD_ARG;
    if (!ws || ws_bytes < vvae_convt_bf16_ws_bytes(Cin, Cout) || ((uintptr_t)ws % 16)) return VVAE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
#define CT_GO(CI, CO)                                                                                                   \
    if (Cin == CI && Cout == CO)                                                                                        \
        return dgrad ? launch_convt_bf16<1, CI, CO>(x, ldx, w, nullptr, y, ldy, NT, H, W, ws, s, prepacked)             \
                     : launch_convt_bf16<0, CI, CO>(x, ldx, w, bias, y, ldy, NT, H, W, ws, s, prepacked);
    CT_GO(128, 64) CT_GO(64, 32) CT_GO(32, 16)
#undef CT_GO
    return VVAE_ERR_BAD_ARG;
}
