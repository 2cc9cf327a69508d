// Public Conv3d entry points: pick the bf16 MFMA fast path when the shape qualifies, else the generic fp32-MFMA path.
#include "common.hpp"

extern "C" {
int vvae_conv3d_fwd_generic(const void*, int, const float*, const float*, void*, int, int, int, int, int, int, int, int, int, int, int, void*);
int vvae_conv3d_dgrad_generic(const void*, int, const float*, void*, int, int, int, int, int, int, int, int, int, int, int, void*);
int vvae_conv3d_wgrad_generic(const void*, int, const void*, int, float*, float*, int, int, int, int, int, int, int, int, int, int, void*, size_t, void*);
size_t vvae_conv3d_wgrad_generic_ws_bytes(int, int, int, int, int, int, int, int, int);
int vvae_conv3d_fwd_bf16(const void*, int, const float*, const float*, void*, int, int, int, int, int, int, int, int, int, int, int, int, void*, size_t, void*);
int vvae_conv3d_wgrad_bf16(const void*, int, const void*, int, float*, float*, int, int, int, int, int, int, int, int, int, void*, size_t, void*);
size_t vvae_conv3d_bf16_ws_bytes(int, int, int, int, int, int, int, int, int, int);
int vvae_conv3d_bf16_supported(int, int, int, int, int, int, int, int, int);
int vvae_conv_pointwise_supported(int, int, int, int, int, int, int, const void*);
size_t vvae_conv_pointwise_ws_bytes(long, int, int);
int vvae_conv_pointwise_fwd(const void*, int, const float*, const float*, void*, int, long, int, int, int, void*);
int vvae_conv_pointwise_dgrad(const void*, int, const float*, void*, int, long, int, int, int, void*);
int vvae_conv_pointwise_wgrad(const void*, int, const void*, int, float*, float*, long, int, int, int, void*, size_t, void*);
}

// 1x1x1 convolutions onto 3 channels (final_conv, the un-embedding down-projection) are HBM streams: conv_pointwise.hip
static inline bool pointwise_shape(int Cin, int Cout, int kt, int kh, int kw)
{
    return kt == 1 && kh == 1 && kw == 1 && Cout == 3 && (Cin == 12 || Cin == 16);
}

static int g_force_generic = 0;

// The bf16 kernels address through buffer descriptors (32-bit byte offsets): a tensor spanning 2 GB or more is the generic path's
// (conv3d_bf16.hip launch_roll returns VVAE_ERR_BAD_ARG for it; ops._bf16_fast says once that this happened).
static inline bool fits32(int N, int T, int H, int W, int ld_a, int Ca, int ld_b, int Cb)
{
    const long vox = (long)N * T * H * W;
    return ((vox - 1) * ld_a + Ca) * 2 < (1L << 31) && ((vox - 1) * ld_b + Cb) * 2 < (1L << 31);
}

// Test hook: 1 = always take the generic path (used to cross-check the fast path on the GPU).
extern "C" void vvae_conv3d_force_generic(int on) { g_force_generic = on; }

// Scratch bytes the caller must provide to fwd / dgrad / wgrad for this shape (0 = none needed).
extern "C" size_t vvae_conv3d_workspace_bytes(int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int dtype, int which)
{
    const size_t generic = which == 2 ? vvae_conv3d_wgrad_generic_ws_bytes(N, T, H, W, Cin, Cout, kt, kh, kw) : 0;
    if (g_force_generic) return generic;
    if (pointwise_shape(Cin, Cout, kt, kh, kw)) {                     // (vvae_conv3d_wgrad falls back to generic when the pointwise kernels decline)
        const size_t pw = which == 2 ? vvae_conv_pointwise_ws_bytes((long)N * T * H * W, Cin, Cout) : 0;
        return pw > generic ? pw : generic;
    }
    if (dtype != VVAE_DT_BF16) return generic;
    const size_t fast = vvae_conv3d_bf16_ws_bytes(N, T, H, W, Cin, Cout, kt, kh, kw, which);
    return fast > generic ? fast : generic;                           // (the fast path may decline a pitch: the generic path then needs its own)
}

// y[n,t,h,w,co] = bias[co] + sum_{a,b,c,ci} x[n,t+a-pt,h+b-ph,w+c-pw,ci] * w[a,b,c,ci,co]   (zero padding)
// x: (N,T,H,W,Cin) row pitch ldx; y: (N,T,H,W,Cout) row pitch ldy; w: Flax kernel (kt,kh,kw,Cin,Cout) fp32; bias fp32 or NULL.
extern "C" int vvae_conv3d_fwd(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy,
                               int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int dtype,
                               void* ws, size_t ws_bytes, void* stream)
{
    if (!g_force_generic && vvae_conv_pointwise_supported(Cin, Cout, kt, kh, kw, ldx, dtype, x))
        return vvae_conv_pointwise_fwd(x, ldx, w, bias, y, ldy, (long)N * T * H * W, Cin, Cout, dtype, stream);
    if (dtype == VVAE_DT_BF16 && !g_force_generic && fits32(N, T, H, W, ldx, Cin, ldy, Cout) && vvae_conv3d_bf16_supported(Cin, Cout, kt, kh, kw, ldx, ldy, 0, 0))
        return vvae_conv3d_fwd_bf16(x, ldx, w, bias, y, ldy, N, T, H, W, Cin, Cout, kt, kh, kw, 0, 0, ws, ws_bytes, stream);
    return vvae_conv3d_fwd_generic(x, ldx, w, bias, y, ldy, N, T, H, W, Cin, Cout, kt, kh, kw, dtype, stream);
}

// dx[n,t,h,w,ci] = sum_{a,b,c,co} dy[n,t-a+pt,h-b+ph,w-c+pw,co] * w[a,b,c,ci,co]
extern "C" int vvae_conv3d_dgrad(const void* dy, int lddy, const float* w, void* dx, int lddx,
                                 int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int dtype,
                                 void* ws, size_t ws_bytes, void* stream)
{
    if (!g_force_generic && vvae_conv_pointwise_supported(Cin, Cout, kt, kh, kw, lddx, dtype, dx))
        return vvae_conv_pointwise_dgrad(dy, lddy, w, dx, lddx, (long)N * T * H * W, Cin, Cout, dtype, stream);
    if (dtype == VVAE_DT_BF16 && !g_force_generic && fits32(N, T, H, W, lddx, Cin, lddy, Cout) && vvae_conv3d_bf16_supported(Cin, Cout, kt, kh, kw, lddx, lddy, 1, 0))
        return vvae_conv3d_fwd_bf16(dy, lddy, w, nullptr, dx, lddx, N, T, H, W, Cin, Cout, kt, kh, kw, 1, 0, ws, ws_bytes, stream);
    return vvae_conv3d_dgrad_generic(dy, lddy, w, dx, lddx, N, T, H, W, Cin, Cout, kt, kh, kw, dtype, stream);
}

// dw[a,b,c,ci,co] = sum_v x[v+off(a,b,c)][ci] * dy[v][co] (fp32, overwritten); dbias[co] = sum_v dy[v][co] (or NULL).
extern "C" int vvae_conv3d_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias,
                                 int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int dtype,
                                 void* ws, size_t ws_bytes, void* stream)
{
    if (!g_force_generic && vvae_conv_pointwise_supported(Cin, Cout, kt, kh, kw, ldx, dtype, x))
        return vvae_conv_pointwise_wgrad(x, ldx, dy, lddy, dw, dbias, (long)N * T * H * W, Cin, Cout, dtype, ws, ws_bytes, stream);
    if (dtype == VVAE_DT_BF16 && !g_force_generic && fits32(N, T, H, W, ldx, Cin, lddy, Cout) && vvae_conv3d_bf16_supported(Cin, Cout, kt, kh, kw, ldx, lddy, 2, 0))
        return vvae_conv3d_wgrad_bf16(x, ldx, dy, lddy, dw, dbias, N, T, H, W, Cin, Cout, kt, kh, kw, ws, ws_bytes, stream);
    return vvae_conv3d_wgrad_generic(x, ldx, dy, lddy, dw, dbias, N, T, H, W, Cin, Cout, kt, kh, kw, dtype, ws, ws_bytes, stream);
}
