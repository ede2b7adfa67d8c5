import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# -DMTR_PROFILE_TAIL build: first workgroup start, first and last workgroup end of k_fused (100 MHz wall clock)
import bench, torch
scene = bench.build_scene(512,512,1024)
integ = scene.integrator(); integ.collect_stats=True
for _ in range(2):
    s,t = integ.render(scene, spp=1024)
c = integ.last_counters
M = (1 << 64) - 1
print('shortest workgroup %.2f ms, longest %.2f ms, mean %.2f ms (1024 workgroups)' % ((M - c['reserved'][0]) / 1e5, c['reserved'][1] / 1e5, c['splats_overflow'] / 1e5 / 1024), integ.last_times)
