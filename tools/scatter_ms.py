import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
scene = bench.build_scene(512,512,1024, mode='wavefront')
integ = scene.integrator(); integ.collect_stats=True
for _ in range(2):
    s,t = integ.render(scene, spp=1024)
print(integ.last_times)
