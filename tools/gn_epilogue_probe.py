import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from video_vae_amd import ops
from conv_bench_util import tmg
dev="cuda"
for cin,cout,H in ((16,16,256),(32,16,256),(32,32,128),(16,32,128)):
    x=torch.randn(4,16,H,H,cin,device=dev,dtype=torch.bfloat16); w=torch.randn(3,3,3,cin,cout,device=dev)*0.05; b=torch.randn(cout,device=dev)
    pk=ops.conv3d_prepack([w])[0]
    nblk=ops.conv3d_gn_blocks(x,w,8)
    t0=tmg(lambda: ops.conv3d_fwd_raw(x,w,b,packed=pk.fwd))
    t1=tmg(lambda: ops.conv3d_fwd_gn_raw(x,w,b,8,nblk,pk.fwd))
    print(f"{cin}->{cout} @{H}: plain {t0:.1f} us  with GroupNorm partials {t1:.1f} us", flush=True)
