"""tools/plan_edges.py — k_fused's plan at its edges: NLOS (grey and coloured) and Cornell (f32 and deterministic rows) films whose rows just fit, just
do not fit, or leave LDS altogether, fused against wavefront on the same samples (rel-L2 of the transient film, counters)."""
import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import make_nlos, make_cornell, rel_l2

def both(make, spp):
    out = {}
    for mode in ("fused", "wavefront"):
        scene = make(mode)
        integ = scene.integrator(); integ.collect_stats = True
        s, t = integ.render(scene, spp=spp)
        torch.cuda.synchronize()
        out[mode] = (np.asarray(t.cpu() if hasattr(t, "cpu") else t), dict(integ.last_counters))
    a, b = out["fused"], out["wavefront"]
    same = all(a[1][k] == b[1][k] for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"))
    return rel_l2(a[0], b[0]), same, int(np.count_nonzero(b[0]))

bad = 0
for bins in (1, 3, 1365, 4095, 4096, 4097, 12288, 12289, 13000, 40000):
    for colour in (None, {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.9, 0.5, 0.2]}}):
        mk = lambda mode: make_nlos(sx=6, sy=5, capture="confocal", hidden="quad", bins=bins, bin_width=3.0 / bins, start=1.8, hidden_bsdf=colour, amd_mode=mode)
        e, same, nz = both(mk, 64)
        ok = e <= 1e-5 and same
        bad += not ok
        print("nlos %-8s bins %6d: rel-L2 %.2e counters %s nonzero %d %s" % ("grey" if colour is None else "coloured", bins, e, same, nz, "" if ok else "<-- FAIL"), flush=True)
for det in (False, True):
    for bins in (2, 1024, 2731, 3072, 3073, 4000, 4500, 13000, 14000):
        mk = lambda mode: make_cornell(width=8, height=6, bins=bins, amd_mode=mode, amd_deterministic=det)
        e, same, nz = both(mk, 32)
        ok = e <= 1e-5 and same
        bad += not ok
        print("cornell %-13s bins %6d: rel-L2 %.2e counters %s nonzero %d %s" % ("deterministic" if det else "f32 rows", bins, e, same, nz, "" if ok else "<-- FAIL"), flush=True)
print("FAILURES", bad)
