// Generic NDHWC Conv3d (SAME, stride 1, odd kernel) forward / data-grad / weight-grad.
//
// Replaces the XLA lowering of nnx.Conv at /root/reference/train/unet.py:13-21 (3x3x3),
// :111-113 (3x7x7 patch_mixer) and :144-153 (1x1x1 final_conv), and their autodiff.
//
// This is the *any-shape, any-dtype* path: operands are converted to fp32 on load and
// multiplied on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains,
// 157 TFLOP/s peak), so results match an fp32 reference to rounding.  Its fragment
// shapes need no transposes in NDHWC: the MFMA k index lives on lane>>4, so
//   fwd/dgrad: A[voxel][k=channel] and B[k=channel][out-channel] are plain 4-byte loads,
//   wgrad:     A[cin][k=voxel]     and B[k=voxel][cout]          likewise.
// The bf16 fast path (conv3d_bf16.hip) takes over for 16-multiple channel counts.
#include "common.hpp"

namespace {

struct ConvDims {
    int N, T, H, W, Cin, Cout, kt, kh, kw;
};

// NT = number of 16-wide output-channel tiles per wave.
// DGRAD=false: y[v][co] = bias[co] + sum_{tap,ci} x[v+off(tap)][ci] * w[tap][ci][co]
// DGRAD=true : dx[v][ci] =          sum_{tap,co} dy[v-off(tap)][co] * w[tap][ci][co]
template <typename T, bool DGRAD, int NT>
__global__ __launch_bounds__(256) void conv3d_f32mfma_kernel(
    const T* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bias,
    T* __restrict__ y, int ldy, ConvDims d)
{
    const int CK = DGRAD ? d.Cout : d.Cin;     // contracted channels
    const int CO = DGRAD ? d.Cin : d.Cout;     // produced channels
    const long V = (long)d.N * d.T * d.H * d.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const long tile0 = ((long)blockIdx.x * 4 + wave) * 16;
    const int o_base = blockIdx.y * NT * 16;

    long v = tile0 + r;
    const bool vvalid = v < V;
    if (!vvalid) v = V - 1;
    int ww = (int)(v % d.W); long q = v / d.W;
    int hh = (int)(q % d.H); q /= d.H;
    int tt = (int)(q % d.T); const int n = (int)(q / d.T);
    const int pt = d.kt / 2, ph = d.kh / 2, pw = d.kw / 2;

    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int a = 0; a < d.kt; ++a) {
        const int ti = DGRAD ? tt - a + pt : tt + a - pt;
        for (int b = 0; b < d.kh; ++b) {
            const int hi = DGRAD ? hh - b + ph : hh + b - ph;
            for (int c = 0; c < d.kw; ++c) {
                const int wi = DGRAD ? ww - c + pw : ww + c - pw;
                const bool inb = vvalid && (unsigned)ti < (unsigned)d.T && (unsigned)hi < (unsigned)d.H &&
                                 (unsigned)wi < (unsigned)d.W;
                const long vin = (((long)n * d.T + ti) * d.H + hi) * d.W + wi;
                const T* xrow = x + (inb ? vin : 0) * (long)ldx;
                const int tap = (a * d.kh + b) * d.kw + c;
                const float* wtap = w + (long)tap * d.Cin * d.Cout;
                for (int k0 = 0; k0 < CK; k0 += 4) {
                    const int kk = k0 + kq;
                    const bool kin = kk < CK;
                    const float av = (inb && kin) ? ldf(xrow + kk) : 0.f;
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        const int o = o_base + i * 16 + r;
                        float bv = 0.f;
                        if (kin && o < CO)
                            bv = DGRAD ? wtap[(long)o * d.Cout + kk] : wtap[(long)kk * d.Cout + o];
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    // C/D layout: col = lane&15 (channel), row = (lane>>4)*4 + j (voxel in the tile).
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int o = o_base + i * 16 + r;
        if (o >= CO) continue;
        const float bb = (!DGRAD && bias) ? bias[o] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long vo = tile0 + kq * 4 + j;
            if (vo < V) stf(y + vo * (long)ldy + o, acc[i][j] + bb);
        }
    }
}

// part[chunk][tap][ci][co] = sum over the chunk's voxels v of x[v+off(tap)][ci] * dy[v][co]: every (chunk, tap, ci tile, co tile) is written by
// exactly one workgroup (no atomics, no zero fill); vvae_reduce_rows_kernel folds the chunks in index order -> bitwise reproducible.
// grid: x = voxel chunk, y = tap, z = ci tile.  Each wave strides over 4-voxel k-steps.
template <typename T, int NT>
__global__ __launch_bounds__(256) void conv3d_wgrad_f32mfma_kernel(
    const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy, float* __restrict__ part,
    ConvDims d, int co_tile_base, int voxels_per_block)
{
    __shared__ float red[4][NT][64][4];
    const long V = (long)d.N * d.T * d.H * d.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int tap = blockIdx.y;
    const int c = tap % d.kw, b = (tap / d.kw) % d.kh, a = tap / (d.kw * d.kh);
    const int ot = a - d.kt / 2, oh = b - d.kh / 2, ow = c - d.kw / 2;
    const int ci = blockIdx.z * 16 + r;
    const bool ci_ok = ci < d.Cin;
    const long vbeg = (long)blockIdx.x * voxels_per_block;
    long vend = vbeg + voxels_per_block;
    if (vend > V) vend = V;

    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (long v0 = vbeg + wave * 4; v0 < vend; v0 += 16) {
        const long v = v0 + kq;
        const bool vok = v < vend;
        const long vc = vok ? v : vbeg;
        int ww = (int)(vc % d.W); long q = vc / d.W;
        int hh = (int)(q % d.H); q /= d.H;
        int tt = (int)(q % d.T); const int n = (int)(q / d.T);
        const int ti = tt + ot, hi = hh + oh, wi = ww + ow;
        const bool inb = vok && ci_ok && (unsigned)ti < (unsigned)d.T && (unsigned)hi < (unsigned)d.H &&
                         (unsigned)wi < (unsigned)d.W;
        const long vin = (((long)n * d.T + ti) * d.H + hi) * d.W + wi;
        const float av = inb ? ldf(x + vin * (long)ldx + ci) : 0.f;
        const T* dyrow = dy + vc * (long)lddy;
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
    // 256 threads reduce NT*64*4 values over the 4 waves.
    for (int e = threadIdx.x; e < NT * 256; e += 256) {
        const int i = e >> 8, l = (e >> 2) & 63, j = e & 3;
        const float s = red[0][i][l][j] + red[1][i][l][j] + red[2][i][l][j] + red[3][i][l][j];
        const int row = (l >> 4) * 4 + j, col = l & 15;           // row = ci in tile, col = co in tile
        const int cii = blockIdx.z * 16 + row, co = (co_tile_base + i) * 16 + col;
        if (cii < d.Cin && co < d.Cout) part[(((long)blockIdx.x * gridDim.y + tap) * d.Cin + cii) * d.Cout + co] = s;
    }
}

// stage 1 of vvae_colsum: part[blockIdx.x][c] = sum of this workgroup's rows (lane-strided, then a fixed tree through LDS)
template <typename T>
__global__ __launch_bounds__(256) void colsum_part_kernel(const T* __restrict__ x, int ld, long V, int C, float* __restrict__ part,
                                                          int rows_per_block)
{
    __shared__ float red[256];
    int Cp = 1;
    while (Cp < C && Cp < 256) Cp <<= 1;
    const int rows = 256 / Cp;
    const int cl = threadIdx.x % Cp, rl = threadIdx.x / Cp;
    const long vbeg = (long)blockIdx.x * rows_per_block;
    long vend = vbeg + rows_per_block;
    if (vend > V) vend = V;
    for (int c0 = 0; c0 < C; c0 += Cp) {
        const int c = c0 + cl;
        float s = 0.f;
        if (c < C)
            for (long v = vbeg + rl; v < vend; v += rows) s += ldf(x + v * (long)ld + c);
        red[threadIdx.x] = s;
        __syncthreads();
        if (rl == 0 && c < C) {
            float t = 0.f;
            for (int i = 0; i < rows; ++i) t += red[i * Cp + cl];
            part[(long)blockIdx.x * C + c] = t;
        }
        __syncthreads();
    }
}
// stage 1, vector form (C % VEC == 0, rows 16-byte aligned): a thread owns VEC consecutive columns and walks rows with 16-byte loads;
// 256 / (C / VEC) row lanes per pass (column groups beyond 256 threads are walked in turn), folded through LDS in lane order.
// The scalar form above reads 2 bytes per load: 198 us for the (16 384, 768) bias gradient of the decoder's first Linear, this one 10.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void colsum_part_vec_kernel(const T* __restrict__ x, int ld, long V, int C, float* __restrict__ part,
                                                              int rows_per_block)
{
    __shared__ float red[256][VEC];
    const int cvecs = C / VEC;
    const int cpp = cvecs < 256 ? cvecs : 256;                       // column vectors per pass
    const int rows = 256 / cpp;
    const int cl = threadIdx.x % cpp, rl = threadIdx.x / cpp;
    const long vbeg = (long)blockIdx.x * rows_per_block;
    long vend = vbeg + rows_per_block;
    if (vend > V) vend = V;
    for (int c0 = 0; c0 < cvecs; c0 += cpp) {
        const int cv = c0 + cl;
        float a[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] = 0.f;
        if (rl < rows && cv < cvecs)
            for (long v = vbeg + rl; v < vend; v += rows) {
                float t[VEC];
                VecIO<T, VEC>::load(x + v * (long)ld + cv * VEC, t);
#pragma unroll
                for (int e = 0; e < VEC; ++e) a[e] += t[e];
            }
#pragma unroll
        for (int e = 0; e < VEC; ++e) red[threadIdx.x][e] = a[e];
        __syncthreads();
        if (rl == 0 && cv < cvecs) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float t = 0.f;
                for (int i = 0; i < rows; ++i) t += red[i * cpp + cl][e];
                part[(long)blockIdx.x * C + cv * VEC + e] = t;
            }
        }
        __syncthreads();
    }
}
// stage 2: out[c] = sum_b part[b][c], b in index order
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ out)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[(long)b * C + c];
    out[c] = s;
}

template <typename T, bool DGRAD>
int launch_fwd(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, ConvDims d,
               hipStream_t s)
{
    const long V = (long)d.N * d.T * d.H * d.W;
    const int CO = DGRAD ? d.Cin : d.Cout;
    const int tiles = ceil_div(CO, 16);
    const int nt = tiles >= 8 ? 8 : tiles >= 4 ? 4 : tiles >= 2 ? 2 : 1;
    dim3 grid(ceil_div(V, 64), ceil_div(tiles, nt));
    const T* xp = (const T*)x;
    T* yp = (T*)y;
#define GO(NTV) hipLaunchKernelGGL((conv3d_f32mfma_kernel<T, DGRAD, NTV>), grid, dim3(256), 0, s, xp, ldx, w, bias, yp, ldy, d)
    switch (nt) {
        case 8: GO(8); break;
        case 4: GO(4); break;
        case 2: GO(2); break;
        default: GO(1); break;
    }
#undef GO
    VVAE_LAUNCH_CHECK();
    return 0;
}

// voxel chunks of the generic weight gradient (= rows of its partial buffer) and the voxels per chunk
inline long wgrad_chunks(const ConvDims& d, long* vpb_out)
{
    const long V = (long)d.N * d.T * d.H * d.W;
    const int taps = d.kt * d.kh * d.kw;
    const long blocks_xy = (long)taps * ceil_div(d.Cin, 16);
    long want = 4096 / (blocks_xy > 0 ? blocks_xy : 1);          // aim for >= ~2048 blocks in total, chunk a multiple of 16 voxels
    if (want < 1) want = 1;
    long vpb = (V + want - 1) / want;
    vpb = ((vpb + 15) / 16) * 16;
    if (vpb < 256) vpb = 256;
    if (vpb_out) *vpb_out = vpb;
    return ceil_div(V, vpb);
}

inline size_t wgrad_ws_bytes(const ConvDims& d)
{
    const long V = (long)d.N * d.T * d.H * d.W;
    const size_t slab = (size_t)wgrad_chunks(d, nullptr) * d.kt * d.kh * d.kw * d.Cin * d.Cout;
    const size_t bias = (size_t)((V + 255) / 256 < 1024 ? (V + 255) / 256 : 1024) * d.Cout;
    return (slab + bias) * sizeof(float);
}

template <typename T>
int launch_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias, ConvDims d, float* ws, hipStream_t s)
{
    const long V = (long)d.N * d.T * d.H * d.W;
    const int taps = d.kt * d.kh * d.kw;
    const int ci_tiles = ceil_div(d.Cin, 16), co_tiles = ceil_div(d.Cout, 16);
    long vpb = 0;
    const long chunks = wgrad_chunks(d, &vpb);
    dim3 grid((unsigned)chunks, taps, ci_tiles);
    const T* xp = (const T*)x;
    const T* dyp = (const T*)dy;
    for (int base = 0; base < co_tiles;) {
        const int rem = co_tiles - base;
        const int nt = rem >= 8 ? 8 : rem >= 4 ? 4 : rem >= 2 ? 2 : 1;
#define GO(NTV) hipLaunchKernelGGL((conv3d_wgrad_f32mfma_kernel<T, NTV>), grid, dim3(256), 0, s, xp, ldx, dyp, lddy, ws, d, base, (int)vpb)
        switch (nt) {
            case 8: GO(8); break;
            case 4: GO(4); break;
            case 2: GO(2); break;
            default: GO(1); break;
        }
#undef GO
        VVAE_LAUNCH_CHECK();
        base += nt;
    }
    const long ncols = (long)taps * d.Cin * d.Cout;
    hipLaunchKernelGGL(vvae_reduce_rows_kernel, dim3((unsigned)ceil_div(ncols, 32L)), dim3(256), 0, s, (const float*)ws, (int)chunks, ncols, (int)ncols,
                       dw, (int)ncols, (float*)nullptr);
    VVAE_LAUNCH_CHECK();
    if (dbias) {                                                  // two-stage column sum (vvae_colsum's kernels), partial rows behind the slab
        float* part = ws + chunks * ncols;
        const long nb = (V + 255) / 256 < 1024 ? (V + 255) / 256 : 1024;
        const int vb = (int)((V + nb - 1) / nb);
        hipLaunchKernelGGL((colsum_part_kernel<T>), dim3((unsigned)nb), dim3(256), 0, s, dyp, lddy, V, d.Cout, part, vb);
        VVAE_LAUNCH_CHECK();
        hipLaunchKernelGGL(colsum_fold_kernel, dim3(ceil_div(d.Cout, 256)), dim3(256), 0, s, (const float*)part, (int)nb, d.Cout, dbias);
        VVAE_LAUNCH_CHECK();
    }
    return 0;
}

bool dims_ok(const ConvDims& d, int ldx, int ldy, bool dgrad)
{
    if (d.N <= 0 || d.T <= 0 || d.H <= 0 || d.W <= 0 || d.Cin <= 0 || d.Cout <= 0) return false;
    if (!(d.kt & 1) || !(d.kh & 1) || !(d.kw & 1)) return false;
    const int cx = dgrad ? d.Cout : d.Cin, cy = dgrad ? d.Cin : d.Cout;
    return ldx >= cx && ldy >= cy;
}

}  // namespace

// ---------------------------------------------------------------------------- C ABI (generic path)
extern "C" int vvae_conv3d_fwd_generic(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy,
                                       int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                                       int dtype, void* stream)
{
    ConvDims d{N, T, H, W, Cin, Cout, kt, kh, kw};
    if (!x || !w || !y || !dims_ok(d, ldx, ldy, false)) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VVAE_DT_F32) return launch_fwd<float, false>(x, ldx, w, bias, y, ldy, d, s);
    if (dtype == VVAE_DT_BF16) return launch_fwd<bf16_t, false>(x, ldx, w, bias, y, ldy, d, s);
    return VVAE_ERR_BAD_ARG;
}

extern "C" int vvae_conv3d_dgrad_generic(const void* dy, int lddy, const float* w, void* dx, int lddx,
                                         int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                                         int dtype, void* stream)
{
    ConvDims d{N, T, H, W, Cin, Cout, kt, kh, kw};
    if (!dy || !w || !dx || !dims_ok(d, lddy, lddx, true)) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VVAE_DT_F32) return launch_fwd<float, true>(dy, lddy, w, nullptr, dx, lddx, d, s);
    if (dtype == VVAE_DT_BF16) return launch_fwd<bf16_t, true>(dy, lddy, w, nullptr, dx, lddx, d, s);
    return VVAE_ERR_BAD_ARG;
}

// Scratch bytes of vvae_conv3d_wgrad_generic: one fp32 partial dW per voxel chunk + the partial rows of the bias column sum.
extern "C" size_t vvae_conv3d_wgrad_generic_ws_bytes(int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw)
{
    ConvDims d{N, T, H, W, Cin, Cout, kt, kh, kw};
    if (N <= 0 || T <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    return wgrad_ws_bytes(d);
}

// dw / dbias overwritten; ws: vvae_conv3d_wgrad_generic_ws_bytes(...) bytes, 16-byte aligned.  No atomics: per-chunk partials folded in index order.
extern "C" int vvae_conv3d_wgrad_generic(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias,
                                         int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                                         int dtype, void* ws, size_t ws_bytes, void* stream)
{
    ConvDims d{N, T, H, W, Cin, Cout, kt, kh, kw};
    if (!x || !dy || !dw || !dims_ok(d, ldx, lddy, false)) return VVAE_ERR_BAD_ARG;
    if (!ws || ws_bytes < wgrad_ws_bytes(d) || ((uintptr_t)ws % 16)) return VVAE_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VVAE_DT_F32) return launch_wgrad<float>(x, ldx, dy, lddy, dw, dbias, d, (float*)ws, s);
    if (dtype == VVAE_DT_BF16) return launch_wgrad<bf16_t>(x, ldx, dy, lddy, dw, dbias, d, (float*)ws, s);
    return VVAE_ERR_BAD_ARG;
}

// Rows of the fp32 partial buffer vvae_colsum needs: one per workgroup.
extern "C" int vvae_colsum_blocks(long V)
{
    if (V <= 0) return 0;
    const long b = (V + 31) / 32;                                    // >= 32 rows per workgroup, at most 1024 workgroups
    return (int)(b < 1024 ? b : 1024);
}

// out[c] (fp32, overwritten) = sum over V rows of x[v][c].  part: fp32 scratch of vvae_colsum_blocks(V) x C floats.  Two stages, no
// atomics: every workgroup writes its own partial row, one workgroup folds the rows in index order -- bitwise reproducible.  (The bias
// gradients of the Linear layers the GEMM kernels do not take -- 96-wide latent heads -- used the framework's multi-block reduction, whose
// result inside a replayed hipGraph depended on what had run before: tools/step_determinism.py.)
extern "C" int vvae_colsum(const void* x, int ld, long V, int C, float* out, float* part, int dtype, void* stream)
{
    if (!x || !out || !part || V <= 0 || C <= 0 || ld < C || (dtype != VVAE_DT_F32 && dtype != VVAE_DT_BF16)) return VVAE_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = vvae_colsum_blocks(V);
    const int vb = (int)((V + nblk - 1) / nblk);
    const bool al = ((uintptr_t)x % 16) == 0;
    if (dtype == VVAE_DT_F32) {
        if (al && C % 4 == 0 && ld % 4 == 0) hipLaunchKernelGGL((colsum_part_vec_kernel<float, 4>), dim3(nblk), dim3(256), 0, s, (const float*)x, ld, V, C, part, vb);
        else hipLaunchKernelGGL((colsum_part_kernel<float>), dim3(nblk), dim3(256), 0, s, (const float*)x, ld, V, C, part, vb);
    } else {
        if (al && C % 8 == 0 && ld % 8 == 0) hipLaunchKernelGGL((colsum_part_vec_kernel<bf16_t, 8>), dim3(nblk), dim3(256), 0, s, (const bf16_t*)x, ld, V, C, part, vb);
        else hipLaunchKernelGGL((colsum_part_kernel<bf16_t>), dim3(nblk), dim3(256), 0, s, (const bf16_t*)x, ld, V, C, part, vb);
    }
    VVAE_LAUNCH_CHECK();
    // the partial rows sit in L2: 8 row lanes x 8 loads in flight per thread (common.hpp), same index order every time
    hipLaunchKernelGGL(vvae_reduce_rows_kernel, dim3(ceil_div(C, 32)), dim3(256), 0, s, (const float*)part, nblk, (long)C, C, out, C, (float*)nullptr);
    VVAE_LAUNCH_CHECK();
    return 0;
}
