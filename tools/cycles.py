import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
scene = bench.build_scene(512,512,1024)
integ = scene.integrator(); integ.collect_stats=True
for _ in range(2):
    s,t = integ.render(scene, spp=1024)
c = integ.last_counters; tm = integ.last_times
print(tm)
v = [c['splats_overflow'], c['reserved'][0], c['reserved'][1]]
sec = []
for x in v: sec += [x >> 32, x & 0xffffffff]
tot = sum(sec)
names = ['traversal up to the second node step (root, rectangles)','shading (A + B)','path start (ticket, owner, path_begin)','end-of-path bookkeeping (steady, done, next)','row flush + loop top','traversal after it (object nodes, their leaves) + idle']
for n,x in zip(names, sec): print('%-28s %5.1f%%' % (n, 100.0*x/tot))
