import torch


def tmg(f, n=20):
    """GPU microseconds per call from a replayed hipGraph of n back-to-back calls."""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            f()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                f()
        g.replay(); st.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            g.replay()
        e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
