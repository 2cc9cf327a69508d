"""bench.py against another build of the C-ABI library (A/B of kernel variants on one box):   VVAE_AB_LIB=path python tools/bench_with_lib.py <bench flags>"""
import os
import sys
sys.path.insert(0, ".")
import video_vae_amd._lib as _L
if os.environ.get("VVAE_AB_LIB"):
    _L.LIB_PATH = os.path.abspath(os.environ["VVAE_AB_LIB"])
import bench
bench.main()
