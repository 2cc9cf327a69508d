// PatchUnEmbedding's rearrange "b t (h w) (p1 p2 cu) -> b t (h p1) (w p2) cu" (reference train/layers.py:48; cu = c * u channels per
// pixel) fused with the zero padding of the channel axis to the multiple of 16 the UNet's matrix-core kernels take, and its transpose.
// HBM stream: one thread per 8-byte piece of the padded side, so whole padded voxels are written once (the framework route was a
// rearrange copy + a strided copy into 24 of every 32 bytes + a fill of the other 8, then the same again backwards).
#include "common.hpp"

namespace {

struct UpDims { long pieces; int h, w, p, cu4, c4; };     // cu4 / c4: 8-byte pieces per pixel before / after padding (2-byte elements)

// patch-major element offset of pixel (frame f, row y, column x), in 8-byte pieces
__device__ __forceinline__ long src_piece(const UpDims& d, long f, int y, int x)
{
    const int hi = y / d.p, p1 = y - hi * d.p, wi = x / d.p, p2 = x - wi * d.p;
    return ((f * d.h + hi) * d.w + wi) * ((long)d.p * d.p * d.cu4) + ((long)p1 * d.p + p2) * d.cu4;
}

__global__ __launch_bounds__(256) void unpatch_pad_fwd_kernel(const uint2* __restrict__ x, uint2* __restrict__ y, UpDims d)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= d.pieces) return;
    const int q = (int)(i % d.c4);
    long v = i / d.c4;
    const int W = d.w * d.p, H = d.h * d.p;
    const int xx = (int)(v % W); v /= W;
    const int yy = (int)(v % H); const long f = v / H;
    uint2 o = make_uint2(0u, 0u);
    if (q < d.cu4) o = x[src_piece(d, f, yy, xx) + q];
    y[i] = o;
}

__global__ __launch_bounds__(256) void unpatch_pad_bwd_kernel(const uint2* __restrict__ g, uint2* __restrict__ gx, UpDims d)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;               // one thread per piece of the UNPADDED pixel-major index space
    const long total = d.pieces / d.c4 * d.cu4;
    if (i >= total) return;
    const int q = (int)(i % d.cu4);
    long v = i / d.cu4;
    const int W = d.w * d.p, H = d.h * d.p;
    const int xx = (int)(v % W); v /= W;
    const int yy = (int)(v % H); const long f = v / H;
    gx[src_piece(d, f, yy, xx) + q] = g[(i / d.cu4) * d.c4 + q];
}

int up_check(const void* a, const void* b, long frames, int h, int w, int p, int cu, int c, int dtype)
{
    if (!a || !b || frames <= 0 || h <= 0 || w <= 0 || p <= 0 || cu <= 0 || c < cu || dtype != VVAE_DT_BF16) return VVAE_ERR_BAD_ARG;
    if (cu % 4 || c % 4 || ((uintptr_t)a % 8) || ((uintptr_t)b % 8)) return VVAE_ERR_BAD_ARG;
    return 0;
}

}  // namespace

// x: (frames, h*w, p*p*cu) bf16 contiguous -> y: (frames, h*p, w*p, c) bf16 contiguous, channels [cu, c) zero.  cu, c multiples of 4.
extern "C" int vvae_unpatch_pad_fwd(const void* x, void* y, long frames, int h, int w, int p, int cu, int c, int dtype, void* stream)
{
    const int rc = up_check(x, y, frames, h, w, p, cu, c, dtype);
    if (rc) return rc;
    UpDims d{frames * h * p * w * p * (c / 4), h, w, p, cu / 4, c / 4};
    hipLaunchKernelGGL(unpatch_pad_fwd_kernel, dim3((unsigned)((d.pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint2*)x,
                       (uint2*)y, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// g: gradient of y (frames, h*p, w*p, c) -> gx: gradient of x (frames, h*w, p*p*cu); the pad channels of g are ignored.
extern "C" int vvae_unpatch_pad_bwd(const void* g, void* gx, long frames, int h, int w, int p, int cu, int c, int dtype, void* stream)
{
    const int rc = up_check(g, gx, frames, h, w, p, cu, c, dtype);
    if (rc) return rc;
    UpDims d{frames * h * p * w * p * (c / 4), h, w, p, cu / 4, c / 4};
    const long total = d.pieces / d.c4 * d.cu4;
    hipLaunchKernelGGL(unpatch_pad_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint2*)g,
                       (uint2*)gx, d);
    VVAE_LAUNCH_CHECK();
    return 0;
}
