import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mitransient_amd.mi as mi
from mitransient_amd.scenes import staircase_like
mi.set_variant('llvm_ad_rgb')
tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 200
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
res = int(sys.argv[3]) if len(sys.argv) > 3 else 256
for mode in ('fused', 'wavefront'):
    t0 = time.time()
    d = staircase_like(n_steps=16, balusters=3, tiles=tiles, width=res, height=res, temporal_bins=2048, spp=spp)
    d['integrator']['amd_mode'] = mode
    scene = mi.load_dict(d); sd = scene.data()
    t1 = time.time()
    integ = scene.integrator(); integ.collect_stats = True
    for _ in range(2):
        s, t = integ.render(scene, spp=spp)
    torch.cuda.synchronize()
    c, tm = integ.last_counters, integ.last_times
    rays = c['rays_closest'] + c['rays_shadow']
    print(mode, 'tris', sd.tri_verts.shape[0], 'build %.1fs' % (t1 - t0), 'render %.1f ms' % tm['total_ms'],
          '%.0f Mray/s' % (rays / tm['total_ms'] / 1e3), 'bounces/path %.2f' % (c['bounces'] / c['paths']), tm)
