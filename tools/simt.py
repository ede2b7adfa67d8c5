import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
scene = bench.build_scene(512,512,1024)
integ = scene.integrator(); integ.collect_stats=True
s,t = integ.render(scene, spp=int(sys.argv[1]) if len(sys.argv) > 1 else 64)
c = integ.last_counters
# u64 sums of packed (lane<<32 | wave) overflow the low half into the high half; spp=64 keeps them small enough
for name, x in (('node step', c['splats_overflow']), ('tri test', c['reserved'][0])):
    lane, wave = x >> 32, x & 0xffffffff
    print('%-10s lane-steps %d wave-steps %d  SIMT efficiency %.1f%%  per ray %.2f' % (name, lane, wave, 100.0*lane/(64.0*wave), lane/float(c['rays_closest']+c['rays_shadow'])))
calls, wmax = max(1, c['reserved'][1] >> 32), c['reserved'][1] & 0xffffffff
wn, wp = c['splats_overflow'] & 0xffffffff, c['reserved'][0] & 0xffffffff
print('wave traversals %d: node steps %.2f, primitive steps %.2f per traversal; wave-max of the lanes\' node steps %.2f per traversal (floor = %.1f%% of the node wave-steps)' % (
    calls, wn / calls, wp / calls, wmax / calls, 100.0 * wmax / wn))
