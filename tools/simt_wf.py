import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
which = sys.argv[1] if len(sys.argv) > 1 else "cornell"
bench.SCENE = which
scene = bench.build_scene(512,512,1024, mode="wavefront") if which == "cornell" else bench.build_scene(360, 640, 400, mode="wavefront")
integ = scene.integrator(); integ.collect_stats=True
s,t = integ.render(scene, spp=64 if which == "cornell" else 8)
c = integ.last_counters
print(c, integ.last_times)
for name, x in (('node', c['reserved'][0]), ('leaf', c['reserved'][1])):
    it, lanes64 = x >> 32, x & 0xffffffff
    print('%s iterations %d, mean active lanes %.1f / 64' % (name, it, 64.0 * lanes64 / max(it, 1)))
print('closest rays', c['rays_closest'], ' node iterations per 64 rays: %.1f' % ((c['reserved'][0] >> 32) / (c['rays_closest'] / 64.0)))
