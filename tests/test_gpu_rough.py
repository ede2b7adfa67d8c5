"""GPU parity of the microfacet lobes (MTR_BSDF_ROUGHCONDUCTOR / MTR_BSDF_ROUGHPLASTIC, GGX and Beckmann) against the CPU oracle, through the C-ABI:
both kernel organisations, a scene staged in LDS and a scene walked in HBM (material-sorted lists: the rough lobes share
the list of the smooth BSDFs, whose vertices sample the emitter)."""
import numpy as np
import pytest

from conftest import rel_l2
from test_gpu_parity import gpu_render, oracle_render, TOL, MODES
from test_rough_bsdf import _rough_cornell

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("distribution", ["ggx", None, "aniso", "glass", "plastic"],
                         ids=["ggx", "beckmann-by-default", "anisotropic", "roughdielectric", "plastic-thindielectric"])
@pytest.mark.parametrize("mode", MODES)
def test_rough_cornell_matches_oracle(oracle, mode, distribution):
    import mitransient_amd.mi as mi
    d = _rough_cornell(distribution, width=48, height=40)
    d["integrator"].update(max_depth=8, rr_depth=3, amd_mode=mode)
    scene = mi.load_dict(d)
    s_gpu, t_gpu, s_raw, t_raw = gpu_render(scene, 24, seed=5, raw=True)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 24, seed=5)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    assert np.array_equal(t_raw[..., :3] != 0, t4[..., :3] != 0)
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k
    assert np.isfinite(t_gpu).all() and np.count_nonzero(t_gpu) > 10000


@pytest.mark.parametrize("mode", MODES)
def test_rough_materials_scene_in_hbm(oracle, mode):
    """the staircase stand-in with the reference scene's material kinds: wood = roughplastic (nonlinear), steel / brass =
    (two-sided) roughconductor, glass = dielectric; max_depth 65, camera_unwarp"""
    import mitransient_amd.mi as mi
    from mitransient_amd.scenes import staircase_like
    d = staircase_like(n_steps=12, balusters=2, tiles=6, width=40, height=40, temporal_bins=64, spp=8)
    d["wood"] = {"type": "roughplastic", "distribution": "ggx", "alpha": 0.1, "int_ior": 1.5, "ext_ior": 1.0, "nonlinear": True,
                 "diffuse_reflectance": {"type": "rgb", "value": [0.42, 0.26, 0.13]}}
    d["steel"] = {"type": "roughconductor", "distribution": "ggx", "alpha": 0.1, "eta": [2.76, 2.54, 2.27], "k": [3.83, 3.43, 3.04]}
    d["brass"] = {"type": "twosided", "bsdf": {"type": "roughconductor", "distribution": "ggx", "alpha": 0.2,
                                               "eta": [0.44, 0.53, 1.03], "k": [3.7, 2.77, 1.97]}}
    d["integrator"]["amd_mode"] = mode
    scene = mi.load_dict(d)
    s_gpu, t_gpu = gpu_render(scene, 8)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 8)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


def test_rough_materials_unsupported_combinations_fail_loudly():
    """fused kernel + deterministic rows (or a phasor film) with rough materials: refused, not silently something else;
    AUTO picks the wavefront pipeline for them"""
    import mitransient_amd.mi as mi
    from mitransient_amd._cabi import MitransientAMDError
    d = _rough_cornell("ggx", width=16, height=16)
    d["integrator"].update(amd_mode="fused", amd_deterministic=True)
    scene = mi.load_dict(d)
    with pytest.raises(MitransientAMDError, match="wavefront"):
        gpu_render(scene, 4)
    d["integrator"].update(amd_mode="auto")
    scene = mi.load_dict(d)
    a = gpu_render(scene, 4)
    b = gpu_render(mi.load_dict(d), 4)
    assert np.array_equal(a[1], b[1]) and np.count_nonzero(a[1]) > 100


def test_staircase_config5_geometry_with_its_rough_materials(oracle):
    """BASELINE config 5 geometry with the scene file's own GGX lobes (6 roughplastic woods, 2 roughconductor metals;
    textures -> mean colour, bump map ignored), reduced film: wavefront pipeline in HBM against the oracle"""
    from mitransient_amd.scenes import staircase
    scene = staircase(width=45, height=80, spp=4, materials="rough")
    sd = scene.data()
    kinds = sorted(sd.materials[i].type for i in range(sd.n_materials))
    assert kinds.count(5) == 6 and kinds.count(4) == 2
    s_gpu, t_gpu = gpu_render(scene, 4)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 4)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


@pytest.mark.parametrize("mode", MODES)
def test_smooth_shaded_sphere_matches_oracle(oracle, tmp_path, mode):
    """interpolated vertex normals (mtr_scene_desc.tri_normals) on a mesh with a non-uniform to_world: the shading frame is
    rebuilt at the hit, the geometric normal keeps the ray offsets; diffuse, a GGX lobe and glass"""
    from test_smooth_normals import sphere_scene
    for bsdf in (None, {"type": "roughconductor", "distribution": "ggx", "alpha": 0.2, "eta": 0.2, "k": 3.9},
                 {"type": "dielectric", "int_ior": 1.5, "ext_ior": 1.0}):
        scene = sphere_scene(tmp_path, bsdf=bsdf, width=40, height=32)
        scene.integrator().mode = {"fused": 1, "wavefront": 2}[mode]
        s_gpu, t_gpu = gpu_render(scene, 16, seed=2)
        s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 16, seed=2)
        assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
        got = scene.integrator().last_counters
        for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
            assert got[k] == cnt[k], k


def test_staircase_config5_faithful_shading(oracle):
    """config 5 geometry with the scene file's GGX lobes AND its vertex normals (91.5 % of the triangles smooth-shaded)"""
    from mitransient_amd.scenes import staircase
    scene = staircase(width=45, height=80, spp=4, materials="rough", vertex_normals=True)
    sd = scene.data()
    assert sd.tri_normals is not None and 0.9 < np.any(sd.tri_normals != 0, axis=1).mean() < 0.93
    s_gpu, t_gpu = gpu_render(scene, 4)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 4)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("bsdf_type", ["diffuse", "roughplastic"])
def test_textured_scene_matches_oracle(oracle, tmp_path, mode, bsdf_type):
    """bitmap textures on a (diffuse) reflectance: bilinear, repeat; an OBJ panel (flipped v) and a cube (its own coordinates)"""
    from test_textures import textured_scene
    scene = textured_scene(tmp_path, bsdf_type, width=40, height=40)
    scene.integrator().mode = {"fused": 1, "wavefront": 2}[mode]
    s_gpu, t_gpu = gpu_render(scene, 16, seed=4)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 16, seed=4)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    # bitmaps without lobes: the extended kernel built without the lobe code (scene trait kTrNoLobes) — the same numbers
    from mitransient_amd import _cabi
    assert bool(scene.gpu_traits() & _cabi.MTR_TRAIT_NO_LOBES) == (bsdf_type == "diffuse")
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


def test_staircase_config5_as_its_file_describes_it(oracle):
    """config 5 geometry with GGX lobes, vertex normals AND the nine bitmap textures (256-px fixtures)"""
    from mitransient_amd.scenes import staircase
    scene = staircase(width=45, height=80, spp=4, materials="rough", vertex_normals=True, textures=True)
    sd = scene.data()
    assert len(sd.textures) == 9 and sum(1 for i in range(sd.n_materials) if sd.materials[i].albedo_texture) == 9
    s_gpu, t_gpu = gpu_render(scene, 4)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 4)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k


@pytest.mark.parametrize("mode", MODES)
def test_mesh_emitter_with_vertex_normals_gpu(oracle, tmp_path, mode):
    """VERDICT r2 task 8: Mesh::sample_position's interpolated normal on a mesh emitter with vertex normals"""
    from test_smooth_normals import _glowing_ball
    scene = _glowing_ball(tmp_path, False)
    scene.integrator().amd_mode = mode
    s_gpu, t_gpu = gpu_render(scene, 32, seed=3)
    s_ref, t_ref, s4, t4, cnt = oracle_render(oracle, scene, 32, seed=3)
    assert rel_l2(t_gpu, t_ref) <= TOL and rel_l2(s_gpu, s_ref) <= TOL
    got = scene.integrator().last_counters
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert got[k] == cnt[k], k
    assert np.count_nonzero(t_gpu) > 2000
