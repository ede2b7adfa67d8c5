import sys, time; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import make_nlos
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
scene = make_nlos(sx=res, sy=res, capture="confocal", bins=4096, bin_width=2.0**-11, start=1.85, hidden="z", max_depth=-1, rr_depth=5)
integ = scene.integrator(); integ.collect_stats = True
for _ in range(2):
    s, t = integ.render(scene, spp=spp)
torch.cuda.synchronize()
c, tm = integ.total_counters, integ.total_times
rays = c['rays_closest'] + c['rays_shadow']
print('config-4 share: %dx%d, T=4096, %d spp: %.1f ms, %.0f Mray/s, %.3g bins/s, bounces/path %.2f' % (res, res, spp, tm['total_ms'], rays/tm['total_ms']/1e3, c['splats_issued']/tm['total_ms']*1e3, c['bounces']/c['paths']), c)
from oracle import oracle
sd = scene.data(); film = scene.sensors()[0].film()
p = integ.render_params(film, 0, spp, 0, 2)
bufs = oracle.alloc_film(sd.film, prefault=True)
oracle.render(sd, p, use_bvh=False, out=bufs)
t0 = time.time(); _,_,cc = oracle.render(sd, integ.render_params(film, 0, spp, 2, 10), use_bvh=False, out=bufs); dt = time.time()-t0
print('oracle: %.2f Mray/s on %d threads' % ((cc['rays_closest']+cc['rays_shadow'])/dt/1e6, oracle.num_threads()))
