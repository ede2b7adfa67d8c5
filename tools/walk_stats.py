"""tools/walk_stats.py [px] [spp] — node steps and primitive passes per ray of the quantised 8-wide walk over the staircase (config 5's
tree), counted on the HOST build of the walker (tests/host_harness.cpp with -DHH_WALK_STATS, one thread), and the tree's leaves by size.
The trace kernel is bound by the divergent loads it issues (4 per node step + 1 per child reference, 5 per triangle pair): this prices them."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
px = int(sys.argv[1]) if len(sys.argv) > 1 else 48
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
out = os.path.join(ROOT, "tests", "_build", "libhost_harness_stats.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = [os.path.join(ROOT, "tests", "host_harness.cpp")] + [os.path.join(ROOT, "mitransient_amd", "csrc", f) for f in ("mtr_scene_host.cpp", "mtr_bvh.cpp")]
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-DMTR_EXPERIMENTS", "-DHH_WALK_STATS", "-o", out] + srcs)
lib = C.CDLL(out)
from mitransient_amd.scenes import staircase
from conftest import hh_render
sc = staircase(width=px, height=px, temporal_bins=64, max_depth=65, materials="smooth", spp=spp)
sd = sc.data()
sizes = (C.c_uint64 * 7)()
d = sd.desc()
assert lib.hh_leaf_sizes(C.byref(d), sizes) == 0
s = list(sizes)
print(f"quantised 8-wide tree: {s[5]} nodes, {s[6] / s[5]:.2f} children per node; leaves: {s[0]} rectangles, " + ", ".join(f"{s[k]} x {k} tri" for k in range(1, 5)))
lib.hh_set_wide(3)
params = sc.integrator().render_params(sc.sensors()[0].film(), 0, spp)
t4, s4, cnt = hh_render(lib, sd, params)
st = (C.c_uint64 * 2)()
lib.hh_walk_steps(st, 1)
rays = cnt["rays_closest"] + cnt["rays_shadow"]
print(f"{rays} rays ({cnt['rays_closest']} closest, {cnt['rays_shadow']} shadow): {st[0] / rays:.2f} node steps, {st[1] / rays:.2f} primitive passes per ray")
