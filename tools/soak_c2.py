import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# config 2 rendered repeatedly: counters must repeat exactly, films up to the summation order; fused vs wavefront likewise
import bench, torch, numpy as np
def run(mode):
    scene = bench.build_scene(512,512,1024, mode=mode)
    integ = scene.integrator(); integ.collect_stats = True
    s, t = integ.render(scene, spp=1024, seed=0)
    torch.cuda.synchronize()
    c = dict(integ.last_counters); c.pop('reserved', None); c.pop('splats_overflow', None)
    return t.torch_tensor().clone() if hasattr(t, 'torch_tensor') else torch.as_tensor(np.array(t)), c
ref_t, ref_c = run(None)
for i in range(4):
    t, c = run(None)
    rel = float((t.double() - ref_t.double()).norm() / ref_t.double().norm())
    print('fused run', i, 'counters equal', c == ref_c, 'rel-L2 vs first %.2e' % rel)
t, c = run('wavefront')
rel = float((t.double() - ref_t.double()).norm() / ref_t.double().norm())
print('wavefront counters equal', c == ref_c, 'rel-L2 vs fused %.2e' % rel)
