// Fused GroupNorm(eps=1e-6, fast variance) + SiLU, forward and backward, channels-last.
//
// Replaces nnx.GroupNorm + nnx.silu of ConvBlock3D (/root/reference/train/unet.py:22-23,28-29):
// statistics per (sample, group) over (t, h, w, C/G) in fp32 (accumulated across workgroups in
// fp64 so the E[x^2]-E[x]^2 form does not cancel), y = silu((x-mean)*rstd*gamma + beta).
//
// All kernels are pure HBM streams: 16-byte vector loads per lane (VEC = 4 fp32 / 8 bf16 channels),
// wave-level partial sums, one fp64 atomic per (block, group|channel).
//   stats : read x                      -> sums[N][G][2]  (sum, sum of squares), fp64
//   fwd   : read x, write y
//   bwd 1 : read x, dy                  -> csum[N][C][2]  (sum dz*xhat, sum dz), fp64
//   bwd 2 : read x, dy, write dx        (+ tiny kernel for dgamma, dbeta)
#include "common.hpp"

namespace {

constexpr int kMaxC = 1024;

struct GnDims {
    int N; long S; int C, G; float eps;      // S = voxels per sample
};

// Sum (a, b) over the threads that share a channel column (tid % cvecs); threads tid < cvecs end with the block totals.
// Power-of-two cvecs <= 64: xor-shuffles across the wave, then one LDS hop across the 4 waves.  Otherwise (odd channel
// counts on the unvectorised path) a plain LDS gather by the column owners.  red: [256][2] floats.  Ends synchronised.
__device__ __forceinline__ void column_reduce(float& a, float& b, int cvecs, int rows, float (*red)[2])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (cvecs <= 64 && (cvecs & (cvecs - 1)) == 0) {
        // the lanes of a wave that hold the same channel vector (lane % cvecs): VALU-only exchanges (common.hpp), cvecs is
        // uniform over the grid, so the ladder below is one scalar branch chain
        a += xor_lane<32>(a); b += xor_lane<32>(b);
        if (cvecs <= 16) { a += xor_lane<16>(a); b += xor_lane<16>(b); }
        if (cvecs <= 8) { a += xor_lane<8>(a); b += xor_lane<8>(b); }
        if (cvecs <= 4) { a += xor_lane<4>(a); b += xor_lane<4>(b); }
        if (cvecs <= 2) { a += xor_lane<2>(a); b += xor_lane<2>(b); }
        if (cvecs <= 1) { a += xor_lane<1>(a); b += xor_lane<1>(b); }
        __syncthreads();                      // previous use of red is finished
        if (lane < cvecs) { red[wave * 64 + lane][0] = a; red[wave * 64 + lane][1] = b; }
        __syncthreads();
        if (tid < cvecs) {
            a = red[tid][0] + red[64 + tid][0] + red[128 + tid][0] + red[192 + tid][0];
            b = red[tid][1] + red[64 + tid][1] + red[128 + tid][1] + red[192 + tid][1];
        }
    } else {
        __syncthreads();
        red[tid][0] = a; red[tid][1] = b;
        __syncthreads();
        if (tid < cvecs) {
            a = 0.f; b = 0.f;
            for (int r2 = 0; r2 < rows; ++r2) { a += red[r2 * cvecs + tid][0]; b += red[r2 * cvecs + tid][1]; }
        }
    }
}

// Thread layout shared by all kernels: cvecs = C/VEC lanes across channels, rows = 256/cvecs voxels per pass.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, int ldx, GnDims d,
                                                       float* __restrict__ part, int voxels_per_block)
{
    __shared__ float red[256][2];
    __shared__ float chan[kMaxC][2];
    const int cvecs = d.C / VEC, rows = 256 / cvecs;
    const int cl = threadIdx.x % cvecs, rl = threadIdx.x / cvecs;
    const int n = blockIdx.y;
    const long vbeg = (long)blockIdx.x * voxels_per_block;
    long vend = vbeg + voxels_per_block;
    if (vend > d.S) vend = d.S;
    const T* xs = x + (long)n * d.S * ldx;
    float s[VEC], ss[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = ss[i] = 0.f;
    if (rl < rows) {
        long v = vbeg + rl;
        for (; v + rows < vend; v += 2 * rows) {                  // two rows per trip: 2 independent 16-byte loads in flight
            float t[VEC], u[VEC];
            VecIO<T, VEC>::load(xs + v * ldx + cl * VEC, t);
            VecIO<T, VEC>::load(xs + (v + rows) * ldx + cl * VEC, u);
#pragma unroll
            for (int i = 0; i < VEC; ++i) { s[i] += t[i] + u[i]; ss[i] += t[i] * t[i] + u[i] * u[i]; }
        }
        if (v < vend) {
            float t[VEC];
            VecIO<T, VEC>::load(xs + v * ldx + cl * VEC, t);
#pragma unroll
            for (int i = 0; i < VEC; ++i) { s[i] += t[i]; ss[i] += t[i] * t[i]; }
        }
    }
    // reduce over rows for each channel, one channel-of-the-vector at a time
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        float a = s[i], b = ss[i];
        column_reduce(a, b, cvecs, rows, red);
        if (threadIdx.x < cvecs) { chan[cl * VEC + i][0] = a; chan[cl * VEC + i][1] = b; }
    }
    __syncthreads();
    const int cpg = d.C / d.G;
    if (threadIdx.x < d.G) {                     // one partial row per workgroup: part[n][block][g][2] (no atomics: fp64 atomics
        float a = 0.f, b = 0.f;                  // on a handful of addresses cost 30-70 us per launch)
        for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) { a += chan[c][0]; b += chan[c][1]; }
        float* p = part + (((long)n * gridDim.x + blockIdx.x) * d.G + threadIdx.x) * 2;
        p[0] = a; p[1] = b;
    }
}

// out[n][j][2] (fp64) = sum over nblk workgroup partials part[n][blk][j][2] (fp32), fixed order.  J = G or C.
// grid (ceil(2J/32), N); 256 threads = 32 outputs x 8 partial lanes (lane-strided sums, then a fixed tree).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, double* __restrict__ out, int nblk, int J)
{
    __shared__ double red[8][32];
    const int n = blockIdx.y;
    const int el = threadIdx.x & 31, bl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + el;
    double s = 0.0;
    if (i < 2 * J) {
        // the partials sit in L2: what this loop pays is latency, so 8 loads are in flight per thread (same summation order)
        const float* p = part + (long)n * nblk * 2 * J + i;
        int b = bl;
        for (; b + 56 < nblk; b += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long)(b + 8 * u) * 2 * J];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; b < nblk; b += 8) s += (double)p[(long)b * 2 * J];
    }
    red[bl][el] = s;
    __syncthreads();
    if (bl == 0 && i < 2 * J)
        out[(long)n * 2 * J + i] = ((red[0][el] + red[1][el]) + (red[2][el] + red[3][el])) + ((red[4][el] + red[5][el]) + (red[6][el] + red[7][el]));
}

// per-block prologue: a_c = rstd_g*gamma_c, b_c = beta_c - mean_g*a_c for sample n
__device__ __forceinline__ void load_affine(const double* sums, const float* gamma, const float* beta, const GnDims& d,
                                            int n, float (*ab)[2], float* mean_g, float* rstd_g)
{
    const int cpg = d.C / d.G;
    const double M = (double)d.S * cpg;
    if (threadIdx.x < d.G) {
        const double m = sums[((long)n * d.G + threadIdx.x) * 2] / M;
        double var = sums[((long)n * d.G + threadIdx.x) * 2 + 1] / M - m * m;
        if (var < 0.0) var = 0.0;
        mean_g[threadIdx.x] = (float)m;
        rstd_g[threadIdx.x] = rsqrtf((float)var + d.eps);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d.C; c += 256) {
        const int g = c / cpg;
        const float a = rstd_g[g] * gamma[c];
        ab[c][0] = a;
        ab[c][1] = beta[c] - mean_g[g] * a;
    }
    __syncthreads();
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void gn_silu_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                          const double* __restrict__ sums, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, GnDims d, int voxels_per_block)
{
    __shared__ float ab[kMaxC][2];
    __shared__ float mean_g[64], rstd_g[64];
    const int n = blockIdx.y;
    load_affine(sums, gamma, beta, d, n, ab, mean_g, rstd_g);
    const int cvecs = d.C / VEC;
    const long vbeg = (long)blockIdx.x * voxels_per_block;
    long vend = vbeg + voxels_per_block;
    if (vend > d.S) vend = d.S;
    // thread = fixed channel vector cl, rows rl + k*rows: no per-item division, per-channel affine in registers
    const int rows = 256 / cvecs, cl = threadIdx.x % cvecs, rl = threadIdx.x / cvecs, c0 = cl * VEC;
    const T* xs = x + (long)n * d.S * ldx + c0;
    T* ys = y + (long)n * d.S * ldy + c0;
    float aa[VEC], bb[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { aa[i] = ab[c0 + i][0]; bb[i] = ab[c0 + i][1]; }
    if (rl < rows)
        for (long v = vbeg + rl; v < vend; v += rows) {
            float t[VEC];
            VecIO<T, VEC>::load(xs + v * ldx, t);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float z = t[i] * aa[i] + bb[i];
                t[i] = z * sigmoidf_(z);
            }
            VecIO<T, VEC>::store(ys + v * ldy, t);
        }
}

// y = silu(GroupNorm(x)) AND pool = max over the (1,2,2) windows of y in one pass: conv2 -> GroupNorm -> SiLU -> max_pool of an encoder level
// (/root/reference/train/unet.py:44-51).  The separate pool re-read the skip tensor it had just written (3 launches, 66 us per step).
// A thread owns VEC channels of one 2 x 2 window: four loads of x, four stores of y (the rounded values, as the pool of the stored tensor
// saw them), one store of their maximum.  grid (blocks, N); items = T * (H/2) * (W/2) * (C/VEC) per sample; 256 % (C/VEC) == 0.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gn_silu_pool_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, T* __restrict__ pool,
                                                               int ldp, const double* __restrict__ sums, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, GnDims d, int H, int W, long items_per_block)
{
    __shared__ float ab[kMaxC][2];
    __shared__ float mean_g[64], rstd_g[64];
    const int n = blockIdx.y;
    load_affine(sums, gamma, beta, d, n, ab, mean_g, rstd_g);
    const int cvecs = d.C / VEC, Ho = H / 2, Wo = W / 2;
    const long items = d.S / 4 * cvecs;
    const long ibeg = (long)blockIdx.x * items_per_block;
    long iend = ibeg + items_per_block;
    if (iend > items) iend = items;
    const int c0 = (threadIdx.x % cvecs) * VEC;              // 256 % cvecs == 0: a thread keeps its channel vector
    float aa[VEC], bb[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { aa[i] = ab[c0 + i][0]; bb[i] = ab[c0 + i][1]; }
    const T* xs = x + (long)n * d.S * ldx + c0;
    T* ys = y + (long)n * d.S * ldy + c0;
    T* ps = pool + (long)n * (d.S / 4) * ldp + c0;
    for (long it = ibeg + threadIdx.x; it < iend; it += 256) {
        long q = it / cvecs;
        const int wo = (int)(q % Wo); q /= Wo;
        const int ho = (int)(q % Ho); const long t = q / Ho;
        const long vi = (t * H + 2 * ho) * W + 2 * wo;
        const long vs[4] = {vi, vi + 1, vi + W, vi + W + 1};
        float v[4][VEC], m[VEC];
#pragma unroll
        for (int k = 0; k < 4; ++k) VecIO<T, VEC>::load(xs + vs[k] * ldx, v[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float z = v[k][i] * aa[i] + bb[i];
                v[k][i] = round_to<T>(z * sigmoidf_(z));
            }
#pragma unroll
        for (int i = 0; i < VEC; ++i) m[i] = fmaxf(fmaxf(v[0][i], v[1][i]), fmaxf(v[2][i], v[3][i]));
#pragma unroll
        for (int k = 0; k < 4; ++k) VecIO<T, VEC>::store(ys + vs[k] * ldy, v[k]);
        VecIO<T, VEC>::store(ps + ((t * Ho + ho) * Wo + wo) * ldp, m);
    }
}

// dz = dy * silu'(z), silu'(z) = s*(1 + z*(1-s)), s = sigmoid(z)
__device__ __forceinline__ float dsilu(float z) {
    const float s = sigmoidf_(z);
    return s * (1.f + z * (1.f - s));
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void gn_silu_bwd_reduce_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                                 const double* __restrict__ sums, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, GnDims d, float* __restrict__ part,
                                                                 int voxels_per_block)
{
    __shared__ float ab[kMaxC][2];
    __shared__ float mean_g[64], rstd_g[64];
    __shared__ float red[256][2];
    const int n = blockIdx.y;
    load_affine(sums, gamma, beta, d, n, ab, mean_g, rstd_g);
    const int cvecs = d.C / VEC, rows = 256 / cvecs, cpg = d.C / d.G;
    const int cl = threadIdx.x % cvecs, rl = threadIdx.x / cvecs;
    const long vbeg = (long)blockIdx.x * voxels_per_block;
    long vend = vbeg + voxels_per_block;
    if (vend > d.S) vend = d.S;
    const T* xs = x + (long)n * d.S * ldx;
    const T* dys = dy + (long)n * d.S * lddy;
    float s1[VEC], s2[VEC], mu[VEC], rs[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        s1[i] = s2[i] = 0.f;
        const int g = (cl * VEC + i) / cpg;
        mu[i] = mean_g[g]; rs[i] = rstd_g[g];
    }
    float aa[VEC], bb[VEC];                                       // this thread's channels never change: affine in registers
#pragma unroll
    for (int i = 0; i < VEC; ++i) { aa[i] = ab[cl * VEC + i][0]; bb[i] = ab[cl * VEC + i][1]; }
    if (rl < rows) {
        long v = vbeg + rl;
        for (; v + rows < vend; v += 2 * rows) {                  // two rows per trip: 4 independent 16-byte loads in flight
            float t[VEC], g[VEC], t2[VEC], g2[VEC];
            VecIO<T, VEC>::load(xs + v * ldx + cl * VEC, t);
            VecIO<T, VEC>::load(dys + v * lddy + cl * VEC, g);
            VecIO<T, VEC>::load(xs + (v + rows) * ldx + cl * VEC, t2);
            VecIO<T, VEC>::load(dys + (v + rows) * lddy + cl * VEC, g2);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float dz = g[i] * dsilu(t[i] * aa[i] + bb[i]);
                const float dz2 = g2[i] * dsilu(t2[i] * aa[i] + bb[i]);
                s1[i] += (dz * (t[i] - mu[i]) + dz2 * (t2[i] - mu[i])) * rs[i];
                s2[i] += dz + dz2;
            }
        }
        if (v < vend) {
            float t[VEC], g[VEC];
            VecIO<T, VEC>::load(xs + v * ldx + cl * VEC, t);
            VecIO<T, VEC>::load(dys + v * lddy + cl * VEC, g);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float dz = g[i] * dsilu(t[i] * aa[i] + bb[i]);
                s1[i] += dz * (t[i] - mu[i]) * rs[i];
                s2[i] += dz;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        float a = s1[i], b = s2[i];
        column_reduce(a, b, cvecs, rows, red);
        if (threadIdx.x < cvecs) {
            const int c = cl * VEC + i;
            float* p = part + (((long)n * gridDim.x + blockIdx.x) * d.C + c) * 2;       // part[n][block][c][2]
            p[0] = a; p[1] = b;
        }
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void gn_silu_bwd_apply_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                                T* __restrict__ dx, int lddx, const double* __restrict__ sums,
                                                                const double* __restrict__ csum, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, GnDims d, int voxels_per_block)
{
    __shared__ float ab[kMaxC][2];
    __shared__ float mean_g[64], rstd_g[64], m1_g[64], m2_g[64];
    const int n = blockIdx.y;
    if (blockIdx.x == 0 && n == 0)                        // the affine-parameter gradients ride along in one workgroup (was a launch)
        for (int c = threadIdx.x; c < d.C; c += 256) {
            double a = 0.0, b = 0.0;
            for (int m = 0; m < d.N; ++m) { a += csum[((long)m * d.C + c) * 2]; b += csum[((long)m * d.C + c) * 2 + 1]; }
            dgamma[c] = (float)a;
            dbeta[c] = (float)b;
        }
    load_affine(sums, gamma, beta, d, n, ab, mean_g, rstd_g);
    const int cvecs = d.C / VEC, cpg = d.C / d.G;
    if (threadIdx.x < d.G) {
        const double M = (double)d.S * cpg;
        double a = 0.0, b = 0.0;
        for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) {
            a += (double)gamma[c] * csum[((long)n * d.C + c) * 2 + 1];   // sum gamma*dz
            b += (double)gamma[c] * csum[((long)n * d.C + c) * 2];       // sum gamma*dz*xhat
        }
        m1_g[threadIdx.x] = (float)(a / M);
        m2_g[threadIdx.x] = (float)(b / M);
    }
    __syncthreads();
    const long vbeg = (long)blockIdx.x * voxels_per_block;
    long vend = vbeg + voxels_per_block;
    if (vend > d.S) vend = d.S;
    const int rows = 256 / cvecs, cl = threadIdx.x % cvecs, rl = threadIdx.x / cvecs, c0 = cl * VEC;
    const T* xs = x + (long)n * d.S * ldx + c0;
    const T* dys = dy + (long)n * d.S * lddy + c0;
    T* dxs = dx + (long)n * d.S * lddx + c0;
    // per-channel constants in registers: z = x*aa+bb;  dx = k1*dz - k2 - x*k3  with xhat folded in
    float aa[VEC], bb[VEC], k1[VEC], k2[VEC], k3[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int c = c0 + i, gi = c / cpg;
        aa[i] = ab[c][0]; bb[i] = ab[c][1];
        const float rs = rstd_g[gi], mu = mean_g[gi];
        k1[i] = rs * gamma[c];
        k3[i] = rs * rs * m2_g[gi];                       // rstd * (xhat * m2) = rs*rs*m2 * (x - mu)
        k2[i] = rs * m1_g[gi] - k3[i] * mu;
    }
    if (rl < rows)
        for (long v = vbeg + rl; v < vend; v += rows) {
            float t[VEC], g[VEC];
            VecIO<T, VEC>::load(xs + v * ldx, t);
            VecIO<T, VEC>::load(dys + v * lddy, g);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float dz = g[i] * dsilu(t[i] * aa[i] + bb[i]);
                t[i] = k1[i] * dz - k2[i] - t[i] * k3[i];
            }
            VecIO<T, VEC>::store(dxs + v * lddx, t);
        }
}

int g_gn_blocks = 4096;      // workgroups of the streaming passes (forward, backward apply); tuning hook vvae_gn_config

inline int pick_vpb(long S, int N, int total_blocks = 0) {
    if (total_blocks == 0) total_blocks = g_gn_blocks;
    // ~total_blocks workgroups over the whole tensor, at least 128 voxels each.  The reducing kernels pay a fixed
    // epilogue per workgroup (cross-wave folds + fp64 atomics), so they take fewer, longer workgroups (~4 per CU).
    long want = total_blocks / (N > 0 ? N : 1);
    if (want < 1) want = 1;
    long vpb = (S + want - 1) / want;
    if (vpb < 128) vpb = 128;
    return (int)vpb;
}

template <typename T> bool vec_ok(const void* p, int ld, int C) {
    constexpr int V = VecWidth<T>::value;
    return C % V == 0 && ld % V == 0 && ((uintptr_t)p % 16) == 0 && (256 % (C / V) == 0 || C / V <= 256);
}

bool gn_ok(const GnDims& d) {
    return d.N > 0 && d.S > 0 && d.C > 0 && d.G > 0 && d.C % d.G == 0 && d.C <= kMaxC && d.G <= 64 && d.C <= 256 * 8;
}

}  // namespace

#define GN_DISPATCH(KERNEL, VOK, ...)                                                                         \
    do {                                                                                                      \
        if (dtype == VVAE_DT_F32) {                                                                           \
            typedef float T;                                                                                  \
            if (VOK) hipLaunchKernelGGL((KERNEL<T, 4>), grid, dim3(256), 0, s, __VA_ARGS__);                  \
            else hipLaunchKernelGGL((KERNEL<T, 1>), grid, dim3(256), 0, s, __VA_ARGS__);                      \
        } else {                                                                                              \
            typedef bf16_t T;                                                                                 \
            if (VOK) hipLaunchKernelGGL((KERNEL<T, 8>), grid, dim3(256), 0, s, __VA_ARGS__);                  \
            else hipLaunchKernelGGL((KERNEL<T, 1>), grid, dim3(256), 0, s, __VA_ARGS__);                      \
        }                                                                                                     \
    } while (0)

// fp32 scratch floats vvae_gn_stats / vvae_gn_silu_bwd need in `part` (per-workgroup partial sums).
extern "C" size_t vvae_gn_part_floats(int N, long S, int C)
{
    if (N <= 0 || S <= 0 || C <= 0) return 0;
    return (size_t)N * ceil_div(S, pick_vpb(S, N, 1024)) * C * 2;
}

// sums[n][g][2] (fp64) = fixed-order sum of the nblk partial rows part[n][blk][g][2] (fp32) -- the second half of vvae_gn_stats,
// for partials that the producing convolution wrote itself (vvae_conv3d_fwd_bf16_gn).
extern "C" int vvae_gn_finalize(const float* part, int N, int nblk, int G, double* sums, void* stream)
{
    if (!part || !sums || N <= 0 || nblk <= 0 || G <= 0 || G > 64) return VVAE_ERR_BAD_ARG;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(ceil_div(2 * G, 32), N), dim3(256), 0, (hipStream_t)stream, part, sums, nblk, G);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// sums: fp64 [N][G][2], overwritten.  part: fp32 scratch, >= vvae_gn_part_floats(N, S, C) floats.
extern "C" int vvae_gn_stats(const void* x, int ldx, int N, long S, int C, int G, double* sums, float* part, int dtype, void* stream)
{
    GnDims d{N, S, C, G, 0.f};
    if (!x || !sums || !part || !gn_ok(d) || ldx < C || (dtype != VVAE_DT_F32 && dtype != VVAE_DT_BF16)) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int vpb = pick_vpb(S, N, 1024);
    dim3 grid(ceil_div(S, vpb), N);
    const bool vok = dtype == VVAE_DT_F32 ? vec_ok<float>(x, ldx, C) : vec_ok<bf16_t>(x, ldx, C);
    if (!vok && C > 256) return VVAE_ERR_BAD_ARG;
    GN_DISPATCH(gn_stats_kernel, vok, (const T*)x, ldx, d, part, vpb);
    VVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(ceil_div(2 * G, 32), N), dim3(256), 0, s, part, sums, (int)grid.x, G);
    VVAE_LAUNCH_CHECK();
    return 0;
}

extern "C" int vvae_gn_silu_fwd(const void* x, int ldx, void* y, int ldy, const double* sums, const float* gamma,
                                const float* beta, int N, long S, int C, int G, float eps, int dtype, void* stream)
{
    GnDims d{N, S, C, G, eps};
    if (!x || !y || !sums || !gamma || !beta || !gn_ok(d) || ldx < C || ldy < C ||
        (dtype != VVAE_DT_F32 && dtype != VVAE_DT_BF16)) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int vpb = pick_vpb(S, N);
    dim3 grid(ceil_div(S, vpb), N);
    const bool vok = dtype == VVAE_DT_F32 ? (vec_ok<float>(x, ldx, C) && vec_ok<float>(y, ldy, C))
                                          : (vec_ok<bf16_t>(x, ldx, C) && vec_ok<bf16_t>(y, ldy, C));
    if (!vok && C > 256) return VVAE_ERR_BAD_ARG;
    GN_DISPATCH(gn_silu_fwd_kernel, vok, (const T*)x, ldx, (T*)y, ldy, sums, gamma, beta, d, vpb);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// 1 if vvae_gn_silu_pool_fwd takes the layer: 16-byte channel vectors that tile a 256-thread workgroup, even H and W.
extern "C" int vvae_gn_silu_pool_supported(int H, int W, int C, int G, int ldx, int ldy, int ldp, int dtype)
{
    const int vec = dtype == VVAE_DT_F32 ? 4 : dtype == VVAE_DT_BF16 ? 8 : 0;
    if (!vec || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || C % vec || G <= 0 || G > 64 || C % G || C > kMaxC) return 0;
    const int cvecs = C / vec;
    return (256 % cvecs == 0 && ldx % vec == 0 && ldy % vec == 0 && ldp % vec == 0 && ldx >= C && ldy >= C && ldp >= C) ? 1 : 0;
}

// y (N, T, H, W, C) = silu(GroupNorm(x)) with the statistics `sums` (vvae_gn_stats / vvae_gn_finalize) and pool (N, T, H/2, W/2, C) = the
// (1,2,2) max-pool of y, in one launch.  Row pitches in elements.
extern "C" int vvae_gn_silu_pool_fwd(const void* x, int ldx, void* y, int ldy, void* pool, int ldp, const double* sums, const float* gamma,
                                     const float* beta, int N, int T, int H, int W, int C, int G, float eps, int dtype, void* stream)
{
    if (!x || !y || !pool || !sums || !gamma || !beta || N <= 0 || T <= 0 || !vvae_gn_silu_pool_supported(H, W, C, G, ldx, ldy, ldp, dtype) ||
        ((uintptr_t)x % 16) || ((uintptr_t)y % 16) || ((uintptr_t)pool % 16)) return VVAE_ERR_BAD_ARG;
    const long S = (long)T * H * W;
    GnDims d{N, S, C, G, eps};
    const int vec = dtype == VVAE_DT_F32 ? 4 : 8;
    const long items = S / 4 * (C / vec);
    long want = 4096 / N; if (want < 1) want = 1;
    long ipb = (items + want - 1) / want;
    ipb = (ipb + 255) / 256 * 256;                               // whole passes of the workgroup: a thread's channel vector stays put
    dim3 grid((unsigned)((items + ipb - 1) / ipb), N);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VVAE_DT_F32)
        hipLaunchKernelGGL((gn_silu_pool_fwd_kernel<float, 4>), grid, dim3(256), 0, s, (const float*)x, ldx, (float*)y, ldy, (float*)pool, ldp, sums, gamma,
                           beta, d, H, W, ipb);
    else
        hipLaunchKernelGGL((gn_silu_pool_fwd_kernel<bf16_t, 8>), grid, dim3(256), 0, s, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, (bf16_t*)pool, ldp, sums,
                           gamma, beta, d, H, W, ipb);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// csum: fp64 workspace [N][C][2] (overwritten).  dgamma/dbeta fp32 [C] (overwritten).
extern "C" int vvae_gn_silu_bwd(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, const double* sums,
                                const float* gamma, const float* beta, double* csum, float* part, float* dgamma, float* dbeta,
                                int N, long S, int C, int G, float eps, int dtype, void* stream)
{
    GnDims d{N, S, C, G, eps};
    if (!x || !dy || !dx || !sums || !gamma || !beta || !csum || !part || !dgamma || !dbeta || !gn_ok(d) || ldx < C || lddy < C ||
        lddx < C || (dtype != VVAE_DT_F32 && dtype != VVAE_DT_BF16)) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int vpb = pick_vpb(S, N);
    dim3 grid(ceil_div(S, vpb), N);
    const bool vok = dtype == VVAE_DT_F32
        ? (vec_ok<float>(x, ldx, C) && vec_ok<float>(dy, lddy, C) && vec_ok<float>(dx, lddx, C))
        : (vec_ok<bf16_t>(x, ldx, C) && vec_ok<bf16_t>(dy, lddy, C) && vec_ok<bf16_t>(dx, lddx, C));
    if (!vok && C > 256) return VVAE_ERR_BAD_ARG;
    {
        const int vpb_r = pick_vpb(S, N, 1024);
        dim3 grid(ceil_div(S, vpb_r), N);
        GN_DISPATCH(gn_silu_bwd_reduce_kernel, vok, (const T*)x, ldx, (const T*)dy, lddy, sums, gamma, beta, d, part, vpb_r);
        VVAE_LAUNCH_CHECK();
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(ceil_div(2 * C, 32), N), dim3(256), 0, s, part, csum, (int)grid.x, C);
        VVAE_LAUNCH_CHECK();
    }
    GN_DISPATCH(gn_silu_bwd_apply_kernel, vok, (const T*)x, ldx, (const T*)dy, lddy, (T*)dx, lddx, sums, csum, gamma, beta, dgamma, dbeta, d,
                vpb);
    VVAE_LAUNCH_CHECK();
    return 0;
}

// Tuning hook: workgroups over the whole tensor for the GroupNorm forward / backward-apply passes (default 4096; the reducing passes keep 1024).
extern "C" int vvae_gn_config(int stream_blocks)
{
    g_gn_blocks = stream_blocks > 0 ? stream_blocks : 4096;
    return 0;
}
