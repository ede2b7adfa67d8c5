"""k_fused's plan at its edges (fused_plan: row slots for residency first, one plane per row for grey NLOS scenes): films whose rows just
fit LDS, just do not, or leave it altogether — the fused kernel against the wavefront organisation on the same samples."""
import numpy as np
import pytest

from conftest import make_cornell, make_nlos, rel_l2

pytestmark = pytest.mark.gpu
COLOURED = {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.9, 0.5, 0.2]}}


def _both(make, spp):
    out = {}
    for mode in ("fused", "wavefront"):
        scene = make(mode)
        integ = scene.integrator()
        integ.collect_stats = True
        s, t = integ.render(scene, spp=spp)
        out[mode] = (np.asarray(t.cpu() if hasattr(t, "cpu") else t), dict(integ.last_counters))
    (ta, ca), (tb, cb) = out["fused"], out["wavefront"]
    assert np.count_nonzero(tb) > 50
    assert rel_l2(ta, tb) <= 1e-5
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert ca[k] == cb[k], k


@pytest.mark.parametrize("colour", [None, COLOURED], ids=["grey", "coloured"])
@pytest.mark.parametrize("bins", [1, 4096, 4097, 12288, 12289, 40000])
def test_nlos_rows_at_the_edges_of_lds(bins, colour):
    """grey rows are 4 B per bin (one plane, up to 12288 bins), coloured ones 12 B; beyond LDS: f32 atomics on the film"""
    _both(lambda mode: make_nlos(sx=6, sy=5, capture="confocal", hidden="quad", bins=bins, bin_width=3.0 / bins, start=1.8,
                                 hidden_bsdf=colour, amd_mode=mode), 64)


@pytest.mark.parametrize("det", [False, True], ids=["f32", "deterministic"])
@pytest.mark.parametrize("bins", [2, 2731, 3072, 3073, 4500, 14000])
def test_cornell_rows_at_the_edges_of_lds(bins, det):
    """12 B per bin (f32) or 36 B (fixed point + overflow ring): one slot at 3 - 4 workgroups per CU, one slot at one, no LDS rows at all"""
    _both(lambda mode: make_cornell(width=8, height=6, bins=bins, amd_mode=mode, amd_deterministic=det), 32)
