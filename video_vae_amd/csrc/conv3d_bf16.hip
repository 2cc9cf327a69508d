// bf16 MFMA fast path for Conv3d (placeholder: reports "unsupported" until the tiled kernels land).
#include "common.hpp"

extern "C" int vvae_conv3d_bf16_supported(int Cin, int Cout, int kt, int kh, int kw, int ld_in, int ld_out, int which, int flags)
{
    (void)Cin; (void)Cout; (void)kt; (void)kh; (void)kw; (void)ld_in; (void)ld_out; (void)which; (void)flags;
    return 0;
}

extern "C" size_t vvae_conv3d_bf16_ws_bytes(int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int which)
{
    (void)N; (void)T; (void)H; (void)W; (void)Cin; (void)Cout; (void)kt; (void)kh; (void)kw; (void)which;
    return 0;
}

extern "C" int vvae_conv3d_fwd_bf16(const void*, int, const float*, const float*, void*, int, int, int, int, int, int, int, int, int,
                                    int, int, void*, size_t, void*)
{
    return VVAE_ERR_BAD_ARG;
}

extern "C" int vvae_conv3d_wgrad_bf16(const void*, int, const void*, int, float*, float*, int, int, int, int, int, int, int, int,
                                      int, void*, size_t, void*)
{
    return VVAE_ERR_BAD_ARG;
}
