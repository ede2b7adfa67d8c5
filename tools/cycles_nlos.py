"""wave-clock sections of k_fused<NLOS> on config 4's share (-DMTR_PROFILE_CYCLES=1 build: MITRANSIENT_AMD_LIB=ab/libs/lib_prof_CYC1.so).
nlos_bounce carries no marks of its own: the whole bounce is counted with the end-of-path bookkeeping."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
bench.SCENE = "nlos"
scene = bench.build_scene(256, 256, 4096)
integ = scene.integrator(); integ.collect_stats = True
for _ in range(2):
    s, t = integ.render(scene, spp=512)
c = integ.last_counters; tm = integ.last_times
print(tm)
v = [c['splats_overflow'], c['reserved'][0], c['reserved'][1]]
sec = []
for x in v: sec += [x >> 32, x & 0xffffffff]
tot = sum(sec)
names = ['(traversal marks: none)', '(shading marks: none)', 'path start (ticket, owner, nlos_begin)', 'the bounce (walks + shading) + end-of-path bookkeeping', 'row flush + loop top', 'waiting for a row slot + kernel start / end']
for n, x in zip(names, sec): print('%-60s %5.1f%%' % (n, 100.0 * x / tot))
