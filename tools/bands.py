import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# config 2 rendered as ONE launch, as 8 row-band launches on two alternating streams (what the multi-GPU pipeline does per rank by
# default) and as ONE launch that publishes a completion word per band (mtr_render_params.n_bands; DistributedRenderer(single_launch=True)):
# the cost of the per-launch tails against the cost of a release fence per flushed pixel.  (raw 4-channel film, rows vouched zero)
import bench, torch
scene = bench.build_scene(512, 512, 1024)
integ = scene.integrator(); integ.direct_develop = False
sens = scene.sensors()[0]; film = sens.film()
LANES = (torch.cuda.Stream(), torch.cuda.Stream())
WORDS = torch.zeros(16, dtype=torch.int32, device="cuda")
EPOCH = [0]

def run(kind, nb):
    passes = integ.prepare(scene, sens, 0, 1024, integ.aov_names())
    total = sum(s for _, s in passes)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if kind == "one":
        integ.accumulate(scene, sens, passes, total, rows_are_zero=True)
    elif kind == "words":
        EPOCH[0] += 1
        integ.accumulate(scene, sens, passes, total, rows_are_zero=True, bands=(nb, EPOCH[0], WORDS.data_ptr()))
    else:
        rows = 512 // nb
        for b in range(nb):
            with torch.cuda.stream(LANES[b & 1]):
                integ.accumulate(scene, sens, passes, total, pixel_range=(b * rows * 512, (b + 1) * rows * 512), rows_are_zero=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3

for rep in range(3):
    for kind, nb in (("one", 1), ("launches", 8), ("words", 8), ("words", 16)):
        ms = min(run(kind, nb) for _ in range(3))
        print('%-9s bands %2d: %.2f ms' % ({"one": "1 launch", "launches": "N launches", "words": "1 launch + band words"}[kind], nb, ms))
