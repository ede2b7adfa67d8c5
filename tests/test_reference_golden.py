"""Parity against the REAL reference.

`tests/golden/mitsuba_c1.npz` is written by tools/gen_golden_with_mitsuba.py on a machine that has mitsuba + mitransient;
it is absent so far (PARITY UNPINNED), so the two reference tests skip.  The decision procedure itself is exercised on
synthetic files made from the oracle (test_pin_procedure_on_synthetic_goldens), so that the day a real file is dropped in
it answers, without a code change:
  1. which PCG32 seeding Mitsuba's independent sampler uses (TEA only / TEA + lane offset on initseq — SURVEY A.9 — / the
     64-bit sample_tea_64 words, MTR_FLAG_PCG_TEA64): the variant whose 16-spp render reproduces the file's exact cells;
  2. whether per-sample arithmetic lines up (rel-L2 <= 1e-5 = the north star's bar) or only the estimators agree:
     per-time-bin and per-pixel totals of a 1024-spp render within k sigma of the file's, sigma from batch means.
"""
import os
import sys

import numpy as np
import pytest

from conftest import make_cornell, rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = os.path.join(ROOT, "tests", "golden", "mitsuba_c1.npz")
needs_gold = pytest.mark.skipif(not os.path.exists(GOLD), reason="no reference render available (parity unpinned)")
GOLD_ROUGH = os.path.join(ROOT, "tests", "golden", "mitsuba_rough.npz")      # the GGX lobes (restated from memory of mitsuba 3)
needs_gold_rough = pytest.mark.skipif(not os.path.exists(GOLD_ROUGH), reason="no reference render of the rough scene available")


def make_rough_cornell():
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    from gen_golden_with_mitsuba import rough_cornell
    mi.set_variant("llvm_ad_rgb")
    return mi.load_dict(rough_cornell(mitr.cornell_box()))

K_SIGMA = 5.0
SEEDINGS = {"tea": "tea", "tea+lane": "tea+lane", "tea64": "tea64"}      # name -> what `render` is asked for


def set_seeding(integ, which):
    integ.pcg_initseq_plus_lane = which == "tea+lane"
    integ.pcg_tea64 = which == "tea64"


def classify(g, render):
    """`render(spp, seed, seq_plus_lane, spp_range)` -> (steady (H,W,3), transient (H,W,T,3)).  Returns a verdict dict."""
    out = {"exact_seeding": None, "exact_rel": {}, "statistical": None}
    spp, seed = (int(x) for x in g["lo_spp_seed"])
    for name, flag in SEEDINGS.items():
        s3, t3 = render(spp, seed, flag, None)
        cells = t3.reshape(-1, 3)[g["lo_sample_index"]]
        rel = max(rel_l2(cells, g["lo_sample_value"]), rel_l2(t3.sum(axis=2), g["lo_per_pixel"]), rel_l2(s3, g["lo_steady"]))
        out["exact_rel"][name] = rel
        if rel <= 1e-5 and out["exact_seeding"] is None:
            out["exact_seeding"] = name
    # statistical comparison: totals per time bin / per pixel of the high-spp render vs the file, sigma of OUR estimate from
    # batch means (the file's own noise is the same size: the two-sample difference has variance 2 sigma^2)
    spp, seed = (int(x) for x in g["hi_spp_seed"])
    nb = 16
    step = spp // nb
    bins, pix = [], []
    flag = SEEDINGS[out["exact_seeding"]] if out["exact_seeding"] else "tea"
    for b in range(nb):
        s3, t3 = render(spp, seed, flag, (b * step, (b + 1) * step))          # batch b alone, scaled by 1/spp
        bins.append(t3.sum(axis=(0, 1)).astype(np.float64).sum(axis=-1) * nb)
        pix.append(t3.sum(axis=(2, 3)).astype(np.float64) * nb)
    bins, pix = np.array(bins), np.array(pix)

    def zmax(batches, ref):
        mean, sig = batches.mean(axis=0), batches.std(axis=0, ddof=1) / np.sqrt(nb)
        ok = sig > 0
        return float(np.max(np.abs(mean - ref)[ok] / (np.sqrt(2.0) * sig[ok])))
    # pixels are binned 8 x 8 so that every cell holds enough samples for a normal approximation
    pix_c = pix.reshape(nb, 8, 8, 8, 8).sum(axis=(2, 4))
    ref_c = g["hi_per_pixel"].astype(np.float64).sum(axis=-1).reshape(8, 8, 8, 8).sum(axis=(1, 3))
    out["z_bins"] = zmax(bins, g["hi_per_bin"].sum(axis=-1))
    out["z_pixels"] = zmax(pix_c, ref_c)
    out["statistical"] = bool(out["z_bins"] <= K_SIGMA and out["z_pixels"] <= K_SIGMA)
    return out


def _oracle_render(oracle, scene):
    sd = scene.data()
    integ, film = scene.integrator(), scene.sensors()[0].film()

    def render(spp, seed, seq_plus_lane, spp_range):
        set_seeding(integ, seq_plus_lane)
        s0, s1 = (0, spp) if spp_range is None else spp_range
        t4, s4, _ = oracle.render(sd, integ.render_params(film, seed, spp, s0, s1), use_bvh=True)
        t3, s3 = oracle.develop(sd.film, t4, s4)
        return s3, t3
    return render


def test_pin_procedure_on_synthetic_goldens(oracle, tmp_path):
    """the procedure recognises (a) a file rendered with either seeding as an exact match of THAT seeding, (b) a file from
    an unrelated sample stream as 'statistical agreement only', (c) a biased file (radiance scaled by 1.1) as a failure"""
    from gen_golden_with_mitsuba import pack_render
    scene = make_cornell()
    render = _oracle_render(oracle, scene)

    def synth(flag, seed_shift=0, scale=1.0, spp_hi=256):
        g = {}
        for prefix, spp, seed in (("lo", 16, 0), ("hi", spp_hi, 1)):
            s3, t3 = render(spp, seed + seed_shift, flag, None)
            g.update(pack_render(prefix, s3 * scale, t3 * scale))
            g[f"{prefix}_spp_seed"] = np.asarray([spp, seed])
        return g
    for name, flag in SEEDINGS.items():
        v = classify(synth(flag), render)
        assert v["exact_seeding"] == name and v["statistical"], v
        for other in SEEDINGS:
            if other != name:
                assert v["exact_rel"][other] > 1e-2, v              # the seedings really are different streams
    v = classify(synth("tea", seed_shift=77), render)               # same estimator, unrelated samples
    assert v["exact_seeding"] is None and v["statistical"], v
    v = classify(synth("tea", seed_shift=77, scale=1.1), render)    # a 10 % bias must not pass
    assert v["exact_seeding"] is None and not v["statistical"], v


@needs_gold
def test_oracle_against_reference_render(oracle):
    g = np.load(GOLD)
    v = classify(g, _oracle_render(oracle, make_cornell()))
    print("oracle vs", list(g["versions"]), "->", v)
    assert v["exact_seeding"] is not None or v["statistical"], v


def test_rough_pin_scene_loads_and_self_classifies(oracle):
    """the second pinned scene (tools/gen_golden_with_mitsuba.py: rough_cornell) is a dictionary both sides accept, and
    the procedure recognises its own render"""
    from gen_golden_with_mitsuba import pack_render
    scene = make_rough_cornell()
    sd = scene.data()
    kinds = sorted(sd.materials[i].type for i in range(sd.n_materials))
    assert kinds.count(4) == 2 and kinds.count(5) == 2
    render = _oracle_render(oracle, scene)
    g = {}
    for prefix, spp, seed in (("lo", 16, 0), ("hi", 128, 1)):
        s3, t3 = render(spp, seed, "tea", None)
        g.update(pack_render(prefix, s3, t3))
        g[f"{prefix}_spp_seed"] = np.asarray([spp, seed])
    v = classify(g, render)
    assert v["exact_seeding"] == "tea" and v["statistical"], v


@needs_gold_rough
def test_oracle_rough_lobes_against_reference_render(oracle):
    g = np.load(GOLD_ROUGH)
    v = classify(g, _oracle_render(oracle, make_rough_cornell()))
    print("oracle (GGX lobes) vs", list(g["versions"]), "->", v)
    assert v["exact_seeding"] is not None or v["statistical"], v


@needs_gold
@pytest.mark.gpu
def test_hip_path_against_reference_render():
    g = np.load(GOLD)
    scene = make_cornell()
    integ, sens = scene.integrator(), scene.sensors()[0]

    def render(spp, seed, seq_plus_lane, spp_range):
        set_seeding(integ, seq_plus_lane)
        passes = integ.prepare(scene, sens, seed, spp, [])
        integ.accumulate(scene, sens, passes, spp, spp_range=spp_range)
        s, t = sens.film().develop()
        return np.array(s), np.array(t)
    v = classify(g, render)
    print("HIP path vs", list(g["versions"]), "->", v)
    assert v["exact_seeding"] is not None or v["statistical"], v
