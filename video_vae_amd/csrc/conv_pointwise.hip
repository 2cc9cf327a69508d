// 1x1x1 convolutions onto a few channels: UNet.final_conv (16 -> 3, reference train/unet.py:144-153,188) and the
// PatchUnEmbedding down-projection (12 -> 3, train/layers.py:60-79 as a per-voxel Linear) -- 4.2 million voxels each at the
// production shape.  The matrix-core kernels waste 13/16 of every output tile on them; these are plain HBM streams:
// one thread per voxel, the (Cin x Cout) weights in registers, 8/16-byte vector loads of the voxel's channels.
//   fwd  : y[v][co] = bias[co] + sum_ci x[v][ci] w[ci][co]
//   dgrad: dx[v][ci] = sum_co dy[v][co] w[ci][co]
//   wgrad: dw[ci][co] = sum_v x[v][ci] dy[v][co], db[co] = sum_v dy[v][co]: per-thread fp32 accumulators over a strided run of
//          voxels, folded wave -> workgroup -> one partial row per workgroup; a second kernel sums the rows in fixed order
//          (deterministic; no atomics).
#include "common.hpp"

namespace {

struct PwDims { long V; int ldx, ldy; };

constexpr int PW_BLOCKS = 1024;

template <typename T_, int CIN>
__device__ __forceinline__ void load_vox(const T_* __restrict__ p, float (&x)[CIN])
{
#pragma unroll
    for (int c = 0; c < CIN / 4; ++c) {
        float t[4];
        VecIO<T_, 4>::load(p + 4 * c, t);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[4 * c + e] = t[e];
    }
}
template <typename T_, int CIN>
__device__ __forceinline__ void store_vox(T_* __restrict__ p, const float (&x)[CIN])
{
#pragma unroll
    for (int c = 0; c < CIN / 4; ++c) {
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = x[4 * c + e];
        VecIO<T_, 4>::store(p + 4 * c, t);
    }
}

// addend (row pitch ldadd) != nullptr: y = addend + conv(x) -- the decoder's `coarse + UNet(features)` (reference train/model.py:97) inside
// the product that ends the UNet, rounded once, instead of a 100 MB add launch behind it.
template <typename T_, int CIN, int COUT>
__global__ __launch_bounds__(256) void pw_fwd_kernel(const T_* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                     T_* __restrict__ y, PwDims d, const T_* __restrict__ addend = nullptr, int ldadd = 0)
{
    float wr[CIN][COUT], br[COUT];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int co = 0; co < COUT; ++co) wr[ci][co] = w[ci * COUT + co];
#pragma unroll
    for (int co = 0; co < COUT; ++co) br[co] = bias ? bias[co] : 0.f;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < d.V; v += (long)gridDim.x * 256) {
        float xv[CIN];
        load_vox<T_, CIN>(x + v * d.ldx, xv);
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float a = br[co];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) a += xv[ci] * wr[ci][co];
            if (addend) a += ldf(addend + v * ldadd + co);
            stf(y + v * d.ldy + co, a);
        }
    }
}

// addend (row pitch ldadd) != nullptr: dx = addend + dy W^T -- the gradient arriving at the same tensor from its other consumer (the features feed
// the UNet AND the coarse projection, reference train/layers.py:60-79 / train/model.py:95-97) joins here instead of in a 134 MB add launch.
template <typename T_, int CIN, int COUT>
__global__ __launch_bounds__(256) void pw_dgrad_kernel(const T_* __restrict__ dy, const float* __restrict__ w, T_* __restrict__ dx, PwDims d,
                                                       const T_* __restrict__ addend = nullptr, int ldadd = 0)
{
    float wr[CIN][COUT];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int co = 0; co < COUT; ++co) wr[ci][co] = w[ci * COUT + co];
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < d.V; v += (long)gridDim.x * 256) {
        float g[COUT], o[CIN];
#pragma unroll
        for (int co = 0; co < COUT; ++co) g[co] = ldf(dy + v * d.ldy + co);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            float a = 0.f;
#pragma unroll
            for (int co = 0; co < COUT; ++co) a += g[co] * wr[ci][co];
            o[ci] = a;
        }
        if (addend) {
            float ad[CIN];
            load_vox<T_, CIN>(addend + v * ldadd, ad);
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) o[ci] = round_to<T_>(o[ci]) + ad[ci];          // as the separate add saw the rounded input gradient
        }
        store_vox<T_, CIN>(dx + v * d.ldx, o);
    }
}

// part: (gridDim.x, CIN*COUT + COUT) fp32
template <typename T_, int CIN, int COUT>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(const T_* __restrict__ x, const T_* __restrict__ dy, float* __restrict__ part, PwDims d)
{
    constexpr int NA = CIN * COUT + COUT;
    __shared__ float red[4][NA];
    float acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0.f;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < d.V; v += (long)gridDim.x * 256) {
        float xv[CIN], g[COUT];
        load_vox<T_, CIN>(x + v * d.ldx, xv);
#pragma unroll
        for (int co = 0; co < COUT; ++co) g[co] = ldf(dy + v * d.ldy + co);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[ci * COUT + co] += xv[ci] * g[co];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[CIN * COUT + co] += g[co];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const float t = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = t;
    }
    __syncthreads();
    if (threadIdx.x < NA) part[(long)blockIdx.x * NA + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

inline int pw_blocks(long V)
{
    const long b = (V + 255) / 256;
    return (int)(b < PW_BLOCKS ? (b < 1 ? 1 : b) : PW_BLOCKS);
}

inline bool pw_shape(int Cin, int Cout, int kt, int kh, int kw) { return kt == 1 && kh == 1 && kw == 1 && Cout == 3 && (Cin == 12 || Cin == 16); }

template <typename T_> bool pw_aligned(const void* a, int lda)
{
    const int bytes = 4 * (int)sizeof(T_);                 // one 4-channel vector
    return ((uintptr_t)a % bytes) == 0 && (lda * (int)sizeof(T_)) % bytes == 0;
}

}  // namespace

// 1 if the pointwise kernels take this convolution (1x1x1, Cout = 3, Cin in {12, 16}; pointers / pitches 4-channel aligned).
extern "C" int vvae_conv_pointwise_supported(int Cin, int Cout, int kt, int kh, int kw, int ldx, int dtype, const void* x)
{
    if (!pw_shape(Cin, Cout, kt, kh, kw) || ldx < Cin) return 0;
    return (dtype == VVAE_DT_F32 ? pw_aligned<float>(x, ldx) : dtype == VVAE_DT_BF16 ? pw_aligned<bf16_t>(x, ldx) : false) ? 1 : 0;
}

// scratch for vvae_conv_pointwise_wgrad
extern "C" size_t vvae_conv_pointwise_ws_bytes(long V, int Cin, int Cout) { return (size_t)pw_blocks(V) * (Cin * Cout + Cout) * sizeof(float); }

// addend (V rows of Cout channels, row pitch ldadd, same dtype) may be NULL; otherwise y = addend + conv(x) + bias.
extern "C" int vvae_conv_pointwise_fwd_add(const void* x, int ldx, const float* w, const float* bias, const void* addend, int ldadd, void* y, int ldy,
                                           long V, int Cin, int Cout, int dtype, void* stream)
{
    if (!x || !w || !y || V <= 0 || ldy < Cout || (addend && ldadd < Cout) || !vvae_conv_pointwise_supported(Cin, Cout, 1, 1, 1, ldx, dtype, x))
        return VVAE_ERR_BAD_ARG;
    PwDims d{V, ldx, ldy};
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(pw_blocks(V) * 4 > (V + 255) / 256 ? (unsigned)((V + 255) / 256) : (unsigned)(pw_blocks(V) * 4));
    if (dtype == VVAE_DT_F32) {
        if (Cin == 16) hipLaunchKernelGGL((pw_fwd_kernel<float, 16, 3>), grid, dim3(256), 0, s, (const float*)x, w, bias, (float*)y, d, (const float*)addend, ldadd);
        else hipLaunchKernelGGL((pw_fwd_kernel<float, 12, 3>), grid, dim3(256), 0, s, (const float*)x, w, bias, (float*)y, d, (const float*)addend, ldadd);
    } else {
        if (Cin == 16) hipLaunchKernelGGL((pw_fwd_kernel<bf16_t, 16, 3>), grid, dim3(256), 0, s, (const bf16_t*)x, w, bias, (bf16_t*)y, d, (const bf16_t*)addend, ldadd);
        else hipLaunchKernelGGL((pw_fwd_kernel<bf16_t, 12, 3>), grid, dim3(256), 0, s, (const bf16_t*)x, w, bias, (bf16_t*)y, d, (const bf16_t*)addend, ldadd);
    }
    VVAE_LAUNCH_CHECK();
    return 0;
}

extern "C" int vvae_conv_pointwise_fwd(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, long V, int Cin, int Cout,
                                       int dtype, void* stream)
{
    return vvae_conv_pointwise_fwd_add(x, ldx, w, bias, nullptr, 0, y, ldy, V, Cin, Cout, dtype, stream);
}

// addend (V rows of Cin channels, row pitch ldadd, same dtype) may be NULL; otherwise dx = addend + dy W^T.
extern "C" int vvae_conv_pointwise_dgrad_add(const void* dy, int lddy, const float* w, const void* addend, int ldadd, void* dx, int lddx, long V, int Cin,
                                             int Cout, int dtype, void* stream)
{
    if (!dy || !w || !dx || V <= 0 || lddy < Cout || (addend && (ldadd < Cin || ldadd % 4 || ((uintptr_t)addend % 8))) ||
        !vvae_conv_pointwise_supported(Cin, Cout, 1, 1, 1, lddx, dtype, dx)) return VVAE_ERR_BAD_ARG;
    PwDims d{V, lddx, lddy};
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(pw_blocks(V) * 4 > (V + 255) / 256 ? (unsigned)((V + 255) / 256) : (unsigned)(pw_blocks(V) * 4));
    if (dtype == VVAE_DT_F32) {
        if (Cin == 16) hipLaunchKernelGGL((pw_dgrad_kernel<float, 16, 3>), grid, dim3(256), 0, s, (const float*)dy, w, (float*)dx, d, (const float*)addend, ldadd);
        else hipLaunchKernelGGL((pw_dgrad_kernel<float, 12, 3>), grid, dim3(256), 0, s, (const float*)dy, w, (float*)dx, d, (const float*)addend, ldadd);
    } else {
        if (Cin == 16) hipLaunchKernelGGL((pw_dgrad_kernel<bf16_t, 16, 3>), grid, dim3(256), 0, s, (const bf16_t*)dy, w, (bf16_t*)dx, d, (const bf16_t*)addend, ldadd);
        else hipLaunchKernelGGL((pw_dgrad_kernel<bf16_t, 12, 3>), grid, dim3(256), 0, s, (const bf16_t*)dy, w, (bf16_t*)dx, d, (const bf16_t*)addend, ldadd);
    }
    VVAE_LAUNCH_CHECK();
    return 0;
}

extern "C" int vvae_conv_pointwise_dgrad(const void* dy, int lddy, const float* w, void* dx, int lddx, long V, int Cin, int Cout, int dtype,
                                         void* stream)
{
    return vvae_conv_pointwise_dgrad_add(dy, lddy, w, nullptr, 0, dx, lddx, V, Cin, Cout, dtype, stream);
}

// dw: (Cin, Cout) fp32 overwritten; dbias (Cout) fp32 or NULL.  ws: vvae_conv_pointwise_ws_bytes(V, Cin, Cout) bytes.
extern "C" int vvae_conv_pointwise_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias, long V, int Cin, int Cout,
                                         int dtype, void* ws, size_t ws_bytes, void* stream)
{
    if (!x || !dy || !dw || V <= 0 || lddy < Cout || !vvae_conv_pointwise_supported(Cin, Cout, 1, 1, 1, ldx, dtype, x)) return VVAE_ERR_BAD_ARG;
    if (!ws || ws_bytes < vvae_conv_pointwise_ws_bytes(V, Cin, Cout)) return VVAE_ERR_WORKSPACE;
    PwDims d{V, ldx, lddy};
    hipStream_t s = (hipStream_t)stream;
    const int nblk = pw_blocks(V);
    float* part = (float*)ws;
    if (dtype == VVAE_DT_F32) {
        if (Cin == 16) hipLaunchKernelGGL((pw_wgrad_kernel<float, 16, 3>), dim3(nblk), dim3(256), 0, s, (const float*)x, (const float*)dy, part, d);
        else hipLaunchKernelGGL((pw_wgrad_kernel<float, 12, 3>), dim3(nblk), dim3(256), 0, s, (const float*)x, (const float*)dy, part, d);
    } else {
        if (Cin == 16) hipLaunchKernelGGL((pw_wgrad_kernel<bf16_t, 16, 3>), dim3(nblk), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, part, d);
        else hipLaunchKernelGGL((pw_wgrad_kernel<bf16_t, 12, 3>), dim3(nblk), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, part, d);
    }
    VVAE_LAUNCH_CHECK();
    const int nw = Cin * Cout;
    hipLaunchKernelGGL(vvae_reduce_rows_kernel, dim3(ceil_div(nw + Cout, 32)), dim3(256), 0, s, part, nblk, (long)(nw + Cout), nw + Cout, dw, nw,
                       dbias);
    VVAE_LAUNCH_CHECK();
    return 0;
}
