import torch


def tmg(f, n=10):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2):
            f()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                f()
        g.replay()
        st.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(3):
            g.replay()
        e1.record(st)
        st.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3
