"""tools/nlos_bins.py — config 4's share with fewer time bins over the SAME window: how much the NLOS kernel gains from more row slots in its LDS ring
(an rgb row of 4096 bins is 48 KB: one slot per workgroup, no overlap between pixels; 2048 / 1365 / 1024 bins: 2 / 3 / 4 slots)."""
import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from conftest import make_nlos
for bins in (4096, 2048, 1365, 1024):
    scene = make_nlos(sx=256, sy=256, capture="confocal", bins=bins, bin_width=2.0 / bins, start=1.85, hidden="z", max_depth=-1, rr_depth=5)
    integ = scene.integrator(); integ.collect_stats = True
    ms = []
    for _ in range(4):
        integ.render(scene, spp=512); torch.cuda.synchronize()
        ms.append(integ.total_times['total_ms'])
    print('bins %4d: %s ms' % (bins, ' '.join('%.2f' % m for m in ms[1:])), flush=True)
