#!/usr/bin/env python
"""Pins parity against the REAL reference, on a machine that has it (SURVEY §8c pin 8).

Needs `mitsuba>=3.6,<3.9` and `mitransient` (the reference) importable — neither exists in the authoring
container nor on the GPU box, which is why DESIGN.md §2 says PARITY UNPINNED.  Run once wherever they are:

    python tools/gen_golden_with_mitsuba.py            # writes tests/golden/mitsuba_c1.npz (~200 KB)

and commit the file: tests/test_reference_golden.py then compares the oracle (CPU) and the HIP path (GPU) with
it.  BASELINE config 1: cornell_box() at 64 x 64, 64 bins over OPL 3.5 .. 9.5, 16 spp, seed 0, llvm_ad_rgb.
"""
import os
import sys

import numpy as np


def main():
    import mitsuba as mi
    mi.set_variant("llvm_ad_rgb")
    import mitransient as mitr
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=64, height=64, temporal_bins=64, start_opl=3.5, bin_width_opl=6.0 / 64)
    d["integrator"].update(max_depth=8, rr_depth=5, camera_unwarp=False)
    scene = mi.load_dict(d)
    steady, transient = mi.render(scene, spp=16, seed=0)
    steady, transient = np.array(steady, dtype=np.float32), np.array(transient, dtype=np.float32)
    assert transient.shape == (64, 64, 64, 3) and steady.shape == (64, 64, 3), (transient.shape, steady.shape)
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(here), "tests", "golden", "mitsuba_c1.npz")
    # the full (H,W,T,3) tensor in f16 would lose the 1e-5 bar: keep f32 marginals + a sparse exact sample
    rng = np.random.default_rng(0)
    idx = rng.choice(transient.size // 3, size=20000, replace=False)
    np.savez_compressed(out, per_bin=transient.sum(axis=(0, 1)).astype(np.float64),
                        per_pixel=transient.sum(axis=2), steady=steady,
                        sample_index=idx.astype(np.int64), sample_value=transient.reshape(-1, 3)[idx],
                        norm=np.float64(np.linalg.norm(transient.astype(np.float64))),
                        versions=np.asarray([f"mitsuba {mi.__version__}", f"mitransient {mitr.__version__}"]))
    print("wrote", out, "with", transient.size, "cells summarised;", sys.version.split()[0])


if __name__ == "__main__":
    main()
