import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# config 2 with and without the two boxes: how much of the render is the object-node part of the traversal?
import torch, mitransient_amd as mitr, mitransient_amd.mi as mi
mi.set_variant("llvm_ad_rgb")
for boxes in (True, False):
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=512, height=512, temporal_bins=1024, start_opl=3.5, bin_width_opl=6.0 / 1024)
    d["integrator"]["max_depth"] = 8
    if not boxes:
        del d["small-box"], d["large-box"]
    scene = mi.load_dict(d)
    integ = scene.integrator(); integ.collect_stats = True
    for _ in range(3):
        integ.render(scene, spp=1024, seed=0)
    c, t = integ.last_counters, integ.last_times
    rays = c["rays_closest"] + c["rays_shadow"]
    print("boxes" if boxes else "no boxes", "kernel %.2f ms" % t["trace_ms"], "rays %.3e" % rays, "bounces %.3e" % c["bounces"], "ns per wave-bounce %.1f" % (t["trace_ms"] * 1e6 / (c["bounces"] / 64)))
