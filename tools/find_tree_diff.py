"""Hunt for a ray on which the quantised 4-wide and 8-wide trees of a scene in HBM disagree (they must not: culling is
conservative, hits are decided by the triangle tests): renders config 5's geometry at 16 spp under several seeds with either
tree, bisects a counter difference down to one pixel and asks the oracle which of the two is right."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import mitransient_amd.mi as mi
from mitransient_amd.scenes import staircase
from oracle import oracle
mi.set_variant("llvm_ad_rgb")
W = 512; SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 16
KEYS = ("rays_closest", "rays_shadow", "splats_issued", "bounces")

def make(no8):
    if no8: os.environ["MTR_NO_WIDE8Q"] = "1"
    else: os.environ.pop("MTR_NO_WIDE8Q", None)
    sc = staircase(width=W, height=W, temporal_bins=64, spp=SPP, max_depth=65)
    f = sc.sensors()[0].film(); f.start_opl, f.bin_width_opl = 0.0, 40.0 / 64
    sc.integrator().collect_stats = True
    sc.integrator().render(sc, seed=0, spp=1)          # creates the device scene under this environment
    return sc

def counters(sc, seed, prange=None):
    integ = sc.integrator()
    integ.render(sc, seed=seed, spp=SPP, pixel_range=prange)
    torch.cuda.synchronize()
    return tuple(integ.total_counters[k] for k in KEYS)

A, B = make(False), make(True)
found = 0
for seed in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    ca, cb = counters(A, seed), counters(B, seed)
    if ca == cb:
        continue
    lo, hi = 0, W * W
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if counters(A, seed, (lo, mid)) != counters(B, seed, (lo, mid)): hi = mid
        else: lo = mid
    ca, cb = counters(A, seed, (lo, hi)), counters(B, seed, (lo, hi))
    sd = A.data(); integ = A.integrator(); film = A.sensors()[0].film()
    p = integ.render_params(film, seed, SPP, 0, SPP, lo, hi)
    _, _, co = oracle.render(sd, p, use_bvh=True)
    _, _, cbf = oracle.render(sd, p, use_bvh=False)
    print("seed", seed, "pixel", lo, "8-wide", ca, "4-wide", cb, "oracle (own BVH)", tuple(co[k] for k in KEYS), "oracle (brute force)", tuple(cbf[k] for k in KEYS), flush=True)
    found += 1
    if found >= 3: break
print("differences found:", found)
