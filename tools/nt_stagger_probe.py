"""The start-time stagger of the own NT GEMM (vvae_gemm_nt_stagger: every other workgroup of an XCD sleeps n x 2048 cycles before its
first load): the four epilogue forms on a production shape, graph replays of 20 launches.   python tools/nt_stagger_probe.py [M N K]"""
import ctypes, glob, os
import torch
dev = "cuda"
import sys
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 1536, 768)
a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
b = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K ** -0.5
bias = torch.randn(N, device=dev)
res = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
c2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def tmg(f, n=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): f(st.cuda_stream)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): f(torch.cuda.current_stream().cuda_stream)
        g.replay(); st.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): g.replay()
        e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


sys.path.insert(0, ".")
from video_vae_amd._lib import lib as _lib
lib = _lib()
for units in (0, 1, 2, 3, 4, 6):
    assert lib.vvae_gemm_nt_stagger(units) == 0
    def run(epi, stream):
        return lib.vvae_gemm_nt_bf16(P(a), K, P(b), K, P(c), N, P(bias), P(res) if epi in (1, 3) else None, N, P(c2) if epi == 2 else None, N,
                                     epi, M, N, K, ctypes.c_void_p(stream))
    assert run(0, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    print(f"{M}x{N} K{K} stagger {units}:", " ".join(f"epi{e} {tmg(lambda s, e=e: run(e, s)):6.1f} us" for e in (0, 1, 2, 3)), flush=True)
lib.vvae_gemm_nt_stagger(0)
