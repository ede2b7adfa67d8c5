"""wave-clock sections of k_fused's flat walk on config 2 (-DMTR_PROFILE_CYCLES=2 build: MITRANSIENT_AMD_LIB=ab/libs/lib_prof_CYC2.so)"""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
scene = bench.build_scene(512, 512, 1024)
integ = scene.integrator(); integ.collect_stats = True
for _ in range(2):
    s, t = integ.render(scene, spp=1024)
c = integ.last_counters; tm = integ.last_times
print(tm)
v = [c['splats_overflow'], c['reserved'][0], c['reserved'][1]]
sec = []
for x in v: sec += [x >> 32, x & 0xffffffff]
tot = sum(sec)
names = ['box selection (both walks)', 'shading (A + B)', 'rectangle slab tests', 'everything else (path start, bookkeeping, flush, idle)', 'rectangle tests', 'box face tests']
for n, x in zip(names, sec): print('%-56s %5.1f%%' % (n, 100.0 * x / tot))
