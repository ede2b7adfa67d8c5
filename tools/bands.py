import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# config 2 rendered as ONE launch vs as 8 row-band launches (what the multi-GPU pipeline does per rank): cost of the per-launch tails
import bench, torch
scene = bench.build_scene(512,512,1024)
integ = scene.integrator(); integ.collect_stats = (len(sys.argv) > 1 and sys.argv[1] == "stats")
sens = scene.sensors()[0]; film = sens.film()
LANES = (torch.cuda.Stream(), torch.cuda.Stream())
def run2(nb):
    # the same, consecutive bands on two alternating streams (what DistributedRenderer does)
    passes = integ.prepare(scene, sens, 0, 1024, integ.aov_names())
    total = sum(s for _, s in passes)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rows = 512 // nb
    for b in range(nb):
        with torch.cuda.stream(LANES[b & 1]):
            integ.accumulate(scene, sens, passes, total, pixel_range=(b * rows * 512, (b + 1) * rows * 512), rows_are_zero=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
def run(nb):
    passes = integ.prepare(scene, sens, 0, 1024, integ.aov_names())
    total = sum(s for _, s in passes)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rows = 512 // nb
    for b in range(nb):
        integ.accumulate(scene, sens, passes, total, pixel_range=(b * rows * 512, (b + 1) * rows * 512), rows_are_zero=ZERO)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for ZERO in (False, True):
    for nb in (1, 8, 1, 8, 16):
        print('rows_are_zero %s bands %2d: %.2f ms' % (ZERO, nb, run(nb)))
for nb in (8, 8, 16):
    print('two alternating streams, bands %2d: %.2f ms' % (nb, run2(nb)))
