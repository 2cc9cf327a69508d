"""What a plain elementwise stream reaches on this chip with the LayerNorm kernels' access mix, from HBM (rotating over buffer sets larger than the
Infinity Cache): torch's add (2 reads + 1 write), copy (1 + 1), a 3-read + 1-write fused expression -- the yardstick for layernorm_{fwd,bwd}.
    python tools/stream_ceiling.py"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from pp_bench_util import tmg
n, c, NSET = 16384, 768, 8
torch.manual_seed(0)
sets = [tuple(torch.randn(n, c, device="cuda", dtype=torch.bfloat16) for _ in range(5)) for _ in range(NSET)]
k = [0]
def nxt():
    s = sets[k[0] % NSET]; k[0] += 1
    return s
def copy():
    a, b, c_, d, e = nxt(); d.copy_(a)
def add():
    a, b, c_, d, e = nxt(); torch.add(a, b, out=d)
def add3():
    a, b, c_, d, e = nxt(); torch.add(a, b, out=d); d.add_(c_)
comp = torch.compile(lambda a, b, c_: a + b * c_) if False else None
for name, f, streams in (("copy 1r+1w", copy, 2), ("add 2r+1w", add, 3)):
    k[0] = 0
    t = tmg(f, n=24)
    print(f"{name}: {t:5.1f} us  {streams * n * c * 2 / t / 1e6:4.2f} TB/s", flush=True)
# larger streams (the rate is not a small-tensor effect)
big = [tuple(torch.randn(4 * n, c, device="cuda", dtype=torch.bfloat16) for _ in range(3)) for _ in range(4)]
kb = [0]
def addb():
    a, b, d = big[kb[0] % 4]; kb[0] += 1
    torch.add(a, b, out=d)
t = tmg(addb, n=12)
print(f"add 2r+1w, 100 MB tensors: {t:5.1f} us  {3 * 4 * n * c * 2 / t / 1e6:4.2f} TB/s", flush=True)
