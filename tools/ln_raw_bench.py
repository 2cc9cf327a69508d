"""LayerNorm forward / backward (+ skip gradient) through the C ABI on the trunk's shape, rotating over NSET buffer sets so the inputs come from
HBM, not from the 256 MB Infinity Cache (one set = 75-100 MB); GPU us per launch from a replayed graph.  VVAE_AB_LIB=path for an A/B.
    python tools/ln_raw_bench.py [label]"""
import os
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
import video_vae_amd._lib as _L
if os.environ.get("VVAE_AB_LIB"):
    _L.LIB_PATH = os.path.abspath(os.environ["VVAE_AB_LIB"])
from video_vae_amd import ops
from video_vae_amd._lib import lib
from pp_bench_util import tmg

label = sys.argv[1] if len(sys.argv) > 1 else "default"
if len(sys.argv) > 2:
    lib().vvae_layernorm_fwd_config(int(sys.argv[2]))
if len(sys.argv) > 3:
    lib().vvae_layernorm_config(int(sys.argv[3]))
NSET = 6
n, c = 16384, 768
p = lambda t: None if t is None else t.data_ptr()
torch.manual_seed(0)
sets = []
for i in range(NSET):
    x = torch.randn(n, c, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(n, c, device="cuda", dtype=torch.bfloat16)
    sk = torch.randn(n, c, device="cuda", dtype=torch.bfloat16)
    sets.append((x, dy, sk, torch.empty_like(x), torch.empty_like(x)))
g = torch.randn(c, device="cuda"); b = torch.randn(c, device="cuda")
mean = torch.empty(n, device="cuda"); rstd = torch.empty(n, device="cuda")
dt = ops._dt(sets[0][0])
st = lambda: torch.cuda.current_stream().cuda_stream
nblk = lib().vvae_layernorm_bwd_blocks(n, c, dt)
part = torch.empty((nblk, 2, c), device="cuda")
lib().vvae_layernorm_fwd(p(sets[0][0]), p(sets[0][3]), p(g), p(b), p(mean), p(rstd), None, None, n, c, n, 0, c, 1e-6, dt, st())
torch.cuda.synchronize()
ref_dx = None
k = [0]
def fwd(add):
    x, dy, sk, o1, o2 = sets[k[0] % NSET]; k[0] += 1
    lib().vvae_layernorm_fwd(p(x), p(o1), p(g), p(b), p(mean), p(rstd), p(sk) if add else None, p(o2) if add else None, n, c, n, 0, c, 1e-6, dt, st())
def bwd(skip):
    x, dy, sk, o1, o2 = sets[k[0] % NSET]; k[0] += 1
    lib().vvae_layernorm_bwd(p(x), p(dy), p(g), p(mean), p(rstd), p(sk) if skip else None, p(o1), p(part), n, c, n, 0, c, dt, st())
# checksum of one backward (+skip) for bitwise comparison between builds
k[0] = 0
bwd(True); torch.cuda.synchronize()
chk = sets[0][3].float().double().sum().item(); chk2 = part.double().sum().item()
res = []
for name, f, nbytes in (("fwd", lambda: fwd(False), 2), ("fwd+add", lambda: fwd(True), 4), ("bwd", lambda: bwd(False), 3), ("bwd+skip", lambda: bwd(True), 4)):
    k[0] = 0
    t = tmg(f, n=24)
    res.append(f"{name} {t:5.1f} us ({nbytes * n * c * 2 / t / 1e6:4.2f} TB/s)")
print(f"{label:10s} | " + " | ".join(res) + f" | dx sum {chk:.6f} partial sum {chk2:.4f}", flush=True)
