"""ConvTranspose / GroupNorm+SiLU writing a dense tensor vs the channel half of a twice-as-wide buffer (the decoder's joint buffer):
is the strided half-row write what makes convt_bf16_kernel<0, 32, 16> run at a third of the HBM rate?"""
import sys
import torch
sys.path.insert(0, ".")
from video_vae_amd import ops

dev = "cuda"


def tmg(f, n=10):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): f()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): f()
        g.replay(); st.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): g.replay()
        e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


for cin, s in ((32, 128), (64, 64), (128, 32)):
    cout = cin // 2
    x = torch.randn(4, 16, s, s, cin, device=dev, dtype=torch.bfloat16)
    w = torch.randn(1, 2, 2, cin, cout, device=dev) * 0.1
    b = torch.zeros(cout, device=dev)
    dense = torch.empty(4, 16, 2 * s, 2 * s, cout, device=dev, dtype=torch.bfloat16)
    joint = torch.empty(4, 16, 2 * s, 2 * s, 2 * cout, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        t0 = tmg(lambda: ops.conv_transpose_1x2x2(x, w, b, dense))
        t1 = tmg(lambda: ops.conv_transpose_1x2x2(x, w, b, joint[..., :cout]))
        t2 = tmg(lambda: ops.conv_transpose_1x2x2(x, w, b, joint[..., cout:]))
    mb = (x.numel() + dense.numel()) * 2 / 1e6
    print(f"convT {cin}->{cout} @{s}->{2*s}: dense {t0:6.1f} us  lower half {t1:6.1f} us  upper half {t2:6.1f} us   ({mb:.0f} MB algorithmic, "
          f"{mb / t0 / 1e3:.2f} / {mb / t1 / 1e3:.2f} TB/s)", flush=True)

# ---- the other users of a 16-channel half at 256^2: GroupNorm+SiLU forward writing the skip half, and the backward readers
c, s = 16, 256
x = torch.randn(4, 16, s, s, c, device=dev, dtype=torch.bfloat16)
sc = torch.ones(c, device=dev); bi = torch.zeros(c, device=dev)
dense = torch.empty_like(x)
joint = torch.empty(4, 16, s, s, 2 * c, device=dev, dtype=torch.bfloat16)
with torch.no_grad():
    t0 = tmg(lambda: ops.group_norm_silu(x, sc, bi, 8, 1e-6, dense))
    t1 = tmg(lambda: ops.group_norm_silu(x, sc, bi, 8, 1e-6, joint[..., c:]))
print(f"gn_silu fwd 16ch @256: dense {t0:6.1f} us  half of joint {t1:6.1f} us", flush=True)
