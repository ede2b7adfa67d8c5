#!/usr/bin/env python
"""Pins parity against the REAL reference, on a machine that has it (SURVEY §8c pin 8).

Needs `mitsuba>=3.6,<3.9` and `mitransient` (the reference) importable — neither exists in the authoring
container nor on the GPU box, which is why DESIGN.md §2 says PARITY UNPINNED.  Run once wherever they are:

    python tools/gen_golden_with_mitsuba.py            # writes tests/golden/mitsuba_c1.npz, mitsuba_rough.npz (< 1 MB each)

and commit the file: tests/test_reference_golden.py then holds the oracle (CPU) and the HIP path (GPU) to it and decides
the open questions of SURVEY App. A without a code change (which PCG32 seeding the sampler uses; whether per-sample
arithmetic lines up to 1e-5 or only statistically).

Contents, per file (mitsuba_c1.npz: BASELINE config 1 — cornell_box() at 64 x 64, 64 bins over OPL 3.5 .. 9.5, llvm_ad_rgb;
mitsuba_rough.npz: the same box with GGX roughconductor / roughplastic lobes on four shapes, `rough_cornell`) —
  * `lo_*`  16 spp, seed 0: marginals in f32/f64 + 20 000 exact cells (the sample-for-sample comparison);
  * `hi_*`  1024 spp, seed 1: per-bin / per-pixel marginals (the k-sigma statistical comparison).
The packing is shared with the test (`pack_render`), which also uses it to build synthetic files for its self-test.
"""
import os
import sys

import numpy as np


def pack_render(prefix, steady, transient, n_cells=20000):
    """the summary of one render that goes into the .npz (a full (H,W,T,3) f32 tensor would be 3 MB per render)"""
    steady, transient = np.asarray(steady, dtype=np.float32), np.asarray(transient, dtype=np.float32)
    rng = np.random.default_rng(0)
    idx = rng.choice(transient.size // 3, size=min(n_cells, transient.size // 3), replace=False)
    return {f"{prefix}_per_bin": transient.sum(axis=(0, 1)).astype(np.float64),
            f"{prefix}_per_pixel": transient.sum(axis=2), f"{prefix}_steady": steady,
            f"{prefix}_sample_index": idx.astype(np.int64), f"{prefix}_sample_value": transient.reshape(-1, 3)[idx],
            f"{prefix}_norm": np.float64(np.linalg.norm(transient.astype(np.float64)))}


def rough_cornell(d):
    """the second pinned scene: cornell_box() with GGX lobes (the restated roughconductor / roughplastic) on four shapes —
    the same dictionary works in mitsuba and in mitransient_amd (tests/test_reference_golden.py builds it with this function)"""
    d["sensor"]["film"].update(width=64, height=64, temporal_bins=64, start_opl=3.5, bin_width_opl=6.0 / 64)
    d["integrator"].update(max_depth=8, rr_depth=5, camera_unwarp=False)
    d["floor"]["bsdf"] = {"type": "roughplastic", "distribution": "ggx", "alpha": 0.1, "int_ior": 1.5, "ext_ior": 1.0, "nonlinear": True,
                          "diffuse_reflectance": {"type": "rgb", "value": [0.58, 0.42, 0.3]}}
    d["large-box"]["bsdf"] = {"type": "roughconductor", "distribution": "ggx", "alpha": 0.15, "eta": {"type": "rgb", "value": [1.657, 0.880, 0.521]},
                              "k": {"type": "rgb", "value": [9.224, 6.270, 4.837]}}
    d["back"]["bsdf"] = {"type": "twosided", "bsdf": {"type": "roughplastic", "distribution": "ggx", "alpha": 0.3, "int_ior": 1.49, "ext_ior": 1.000277,
                                                      "diffuse_reflectance": {"type": "rgb", "value": [0.2, 0.5, 0.7]}}}
    d["small-box"]["bsdf"] = {"type": "twosided", "bsdf": {"type": "roughconductor", "distribution": "ggx", "alpha": 0.05,
                                                           "eta": {"type": "rgb", "value": [0.2, 0.2, 0.2]}, "k": {"type": "rgb", "value": [3.9, 3.9, 3.9]}}}
    return d


def main():
    import mitsuba as mi
    mi.set_variant("llvm_ad_rgb")
    import mitransient as mitr
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ("c1", "rough"):
        d = mitr.cornell_box()
        if name == "c1":
            d["sensor"]["film"].update(width=64, height=64, temporal_bins=64, start_opl=3.5, bin_width_opl=6.0 / 64)
            d["integrator"].update(max_depth=8, rr_depth=5, camera_unwarp=False)
        else:
            d = rough_cornell(d)
        scene = mi.load_dict(d)
        out = {}
        for prefix, spp, seed in (("lo", 16, 0), ("hi", 1024, 1)):
            steady, transient = mi.render(scene, spp=spp, seed=seed)
            steady, transient = np.array(steady, dtype=np.float32), np.array(transient, dtype=np.float32)
            assert transient.shape == (64, 64, 64, 3) and steady.shape == (64, 64, 3), (transient.shape, steady.shape)
            out.update(pack_render(prefix, steady, transient))
            out[f"{prefix}_spp_seed"] = np.asarray([spp, seed])
        path = os.path.join(os.path.dirname(here), "tests", "golden", f"mitsuba_{name}.npz")
        np.savez_compressed(path, versions=np.asarray([f"mitsuba {mi.__version__}", f"mitransient {mitr.__version__}"]), **out)
        print("wrote", path, ";", sys.version.split()[0])


if __name__ == "__main__":
    main()
