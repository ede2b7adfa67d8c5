"""tools/mirror_box.py — config 2's film over the Cornell box with its small box a MIRROR (conductor) and a second light: a scene with a flat top
level that runs the GENERAL shading code (k_fused<..., kTrFlatTop | kTrLeafPair>); with MTR_NO_FLAT=1 (experiments build) the tree walk."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mitransient_amd as mitr, mitransient_amd.mi as mi
from mitransient_amd.transform import ScalarTransform4f as T
mi.set_variant("llvm_ad_rgb")
d = mitr.cornell_box()
d["sensor"]["film"].update(width=512, height=512, temporal_bins=1024, start_opl=3.5, bin_width_opl=6.0 / 1024)
d["integrator"]["max_depth"] = 8
d["mirror"] = {"type": "conductor", "material": "Al"} if False else {"type": "conductor", "eta": {"type": "rgb", "value": [1.65, 0.88, 0.52]}, "k": {"type": "rgb", "value": [9.2, 6.3, 4.8]}}
d["small-box"]["bsdf"] = {"type": "ref", "id": "mirror"}
scene = mi.load_dict(d)
integ = scene.integrator(); integ.collect_stats = True
for _ in range(3):
    integ.render(scene, spp=1024); torch.cuda.synchronize()
print("traits", scene.gpu_traits(), "ms", integ.last_times["total_ms"], {k: integ.last_counters[k] for k in ("rays_closest", "rays_shadow")})
