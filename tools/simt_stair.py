import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
bench.SCENE = "staircase"
scene = bench.build_scene(360, 640, 400, mode="fused")
integ = scene.integrator(); integ.collect_stats=True
s,t = integ.render(scene, spp=8)
c = integ.last_counters
print(c, integ.last_times)
for name, x in (('node step', c['splats_overflow']), ('tri test', c['reserved'][0])):
    lane, wave = x >> 32, x & 0xffffffff
    print('%-10s lane-steps %d wave-steps %d  SIMT efficiency %.1f%%  per ray %.2f' % (name, lane, wave, 100.0*lane/(64.0*wave), lane/float(c['rays_closest']+c['rays_shadow'])))
