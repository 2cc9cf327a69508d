// Zero-pad / un-pad the last two dims of a few small fp32 tensors in ONE launch.
//
// The UNet's 12-channel patch mixer (3,7,7,12,12), its bias (12) and the first encoder conv (3,3,3,12,16) run on the 16-channel
// matrix-core kernels over zero-padded weights (/root/reference/train/unet.py:98-104,166-170 are the layers).  As framework ops the three
// pads were six launches forward (fill + copy each) and four backward (slice copies) in every step; they are one launch each way.
//   pad  : dst (R, d0, d1) = src (R, s0, s1) in the leading corner, zeros elsewhere
//   unpad: dst (R, s0, s1) = src (R, d0, d1)[:, :s0, :s1]
#include "common.hpp"

namespace {

constexpr int PADG_MAX = 8;
struct PadEntry { const float* src; float* dst; long total; int s0, s1, d0, d1, block_start; };
struct PadArgs { PadEntry e[PADG_MAX]; int n, unpad; };

__global__ __launch_bounds__(256) void pad_last2_grouped_kernel(PadArgs g)
{
    int ei = 0;
    for (int i = 1; i < g.n; ++i) ei = (int)blockIdx.x >= g.e[i].block_start ? i : ei;
    const PadEntry& E = g.e[ei];
    const long i = ((long)((int)blockIdx.x - E.block_start)) * 256 + threadIdx.x;      // index into the destination
    if (i >= E.total) return;
    if (!g.unpad) {
        const int j = (int)(i % E.d1); const long q = i / E.d1; const int a = (int)(q % E.d0); const long r = q / E.d0;
        E.dst[i] = (a < E.s0 && j < E.s1) ? E.src[(r * E.s0 + a) * E.s1 + j] : 0.f;
    } else {
        const int j = (int)(i % E.s1); const long q = i / E.s1; const int a = (int)(q % E.s0); const long r = q / E.s0;
        E.dst[i] = E.src[(r * E.d0 + a) * E.d1 + j];
    }
}

}  // namespace

// n <= 8 contiguous fp32 tensors; entry i is rows[i] matrices.  unpad = 0: src[i] (rows, s0, s1) -> dst[i] (rows, d0, d1), zero-filled
// outside the (s0, s1) corner.  unpad = 1: src[i] (rows, d0, d1) -> dst[i] (rows, s0, s1), the corner.  s0 <= d0, s1 <= d1.
// Host arrays of device pointers / ints.
extern "C" int vvae_pad_last2_grouped(const float* const* src, float* const* dst, const long* rows, const int* s0, const int* s1, const int* d0,
                                      const int* d1, int n, int unpad, void* stream)
{
    if (!src || !dst || !rows || !s0 || !s1 || !d0 || !d1 || n <= 0 || n > PADG_MAX) return VVAE_ERR_BAD_ARG;
    PadArgs g;
    g.n = n; g.unpad = unpad ? 1 : 0;
    long blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!src[i] || !dst[i] || rows[i] <= 0 || s0[i] <= 0 || s1[i] <= 0 || s0[i] > d0[i] || s1[i] > d1[i]) return VVAE_ERR_BAD_ARG;
        const long total = rows[i] * (unpad ? (long)s0[i] * s1[i] : (long)d0[i] * d1[i]);
        g.e[i] = PadEntry{src[i], dst[i], total, s0[i], s1[i], d0[i], d1[i], (int)blocks};
        blocks += (total + 255) / 256;
    }
    hipLaunchKernelGGL(pad_last2_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    VVAE_LAUNCH_CHECK();
    return 0;
}
