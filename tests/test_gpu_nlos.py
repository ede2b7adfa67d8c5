"""GPU parity of the NLOS tier: k_fused<NLOS> + k_nlos_prepare through the C-ABI against the CPU oracle."""
import numpy as np
import pytest

from conftest import make_nlos, rel_l2
from test_nlos import CONFIGS

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _gpu(scene, spp, seed=0):
    import torch
    integ = scene.integrator()
    integ.collect_stats = True
    s, t = integ.render(scene, seed=seed, spp=spp)
    torch.cuda.synchronize()
    return np.array(s), np.array(t)


def _oracle(oracle, scene, spp, seed=0):
    sd = scene.data()
    p = scene.integrator().render_params(scene.sensors()[0].film(), seed, spp)
    t4, s4, c = oracle.render(sd, p, use_bvh=True)
    t3, s3 = oracle.develop(sd.film, t4, s4)
    return s3, t3, c


@pytest.mark.parametrize("capture,integ", CONFIGS)
@pytest.mark.parametrize("hidden", ["quad", "z"])
def test_nlos_matches_oracle(oracle, capture, integ, hidden):
    scene = make_nlos(sx=8, sy=6, capture=capture, hidden=hidden, **integ)
    s_gpu, t_gpu = _gpu(scene, 64)
    s_ref, t_ref, cnt = _oracle(oracle, scene, 64)
    assert t_gpu.shape == (6, 8, 64, 3)
    if np.linalg.norm(t_ref) == 0:
        assert not t_gpu.any()
    else:
        assert rel_l2(t_gpu, t_ref) <= TOL
    assert np.linalg.norm(s_ref) == 0 or rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


def test_refocusing_the_laser_between_renders(oracle):
    """mitransient.nlos.focus_emitter_at_relay_wall_pixel moves the laser: the next render must see it
    (mtr_scene_set_nlos re-derives the tables and the scanned points)."""
    import mitransient_amd as mitr
    scene = make_nlos(sx=8, sy=8, capture="single")
    s0, t0 = _gpu(scene, 32)
    sensor = scene.sensors()[0]
    relay = [s for s in scene.shapes() if s.sensor() is sensor][0]
    mitr.nlos.focus_emitter_at_relay_wall_pixel((1, 6), relay, scene.emitters()[0])
    s1, t1 = _gpu(scene, 32)
    s_ref, t_ref, _ = _oracle(oracle, scene, 32)
    assert rel_l2(t1, t_ref) <= TOL
    assert rel_l2(t1, t0) > 0.1


def test_nlos_config4_shape_properties():
    """BASELINE config 4 geometry at reduced size (confocal, T = 4096, bin width 2^-11): energy identity with a window
    that covers everything, and no contribution before 2 x 1.0 of OPL."""
    scene = make_nlos(sx=16, sy=16, capture="confocal", bins=4096, bin_width=2.0 ** -11, start=1.85, hidden="z", max_depth=4)
    s, t = _gpu(scene, 256)
    assert t.shape == (16, 16, 4096, 3)
    first_bin = int(np.nonzero(t.sum(axis=(0, 1, 3)))[0][0])
    assert 1.85 + first_bin * 2.0 ** -11 >= 1.99
    # wavefront mode is refused for this tier (fused kernel only)
    scene.integrator().mode = 2
    with pytest.raises(Exception):
        _gpu(scene, 4)
