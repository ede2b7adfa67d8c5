import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# -DMTR_PROFILE_OCC build: how full are k_fused's persistent waves?  (wave iterations with a live lane, live lanes, lanes
# holding a sample whose row slot is not free yet)
import bench, torch
scene = bench.build_scene(512,512,1024)
integ = scene.integrator(); integ.collect_stats=True
s,t = integ.render(scene, spp=1024)
c = integ.last_counters
it, alive, wait = c['splats_overflow'], c['reserved'][0], c['reserved'][1]
print('wave iterations %d  alive lanes/iteration %.1f  waiting lanes/iteration %.1f  (bounces %d, ideal iterations %d)' % (it, alive/it, wait/it, c['bounces'], c['bounces']//64))
