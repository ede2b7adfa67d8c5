"""Pins for the CPU oracle.  PARITY UNPINNED against real Mitsuba: the reference
holds no golden vectors for transient_path, so the oracle is anchored on published KATs, on the
reference's own stated identities and on an analytic quadrature."""
import json
import os

import numpy as np
import pytest

from conftest import make_cornell, rel_l2, hh_render

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_pcg32_known_answer(oracle):
    kat = json.load(open(os.path.join(GOLD, "pcg32_kat.json")))
    u, f = oracle.pcg32_stream(kat["initstate"], kat["initseq"], len(kat["u32"]))
    assert [hex(int(x)) for x in u] == kat["u32"]
    # next_float32 = bitcast((u >> 9) | 0x3f800000) - 1
    exp = ((u >> 9) | 0x3F800000).astype(np.uint32).view(np.float32) - np.float32(1)
    assert np.array_equal(f, exp) and np.all((f >= 0) & (f < 1))


def test_tea_reference_implementation(oracle):
    def tea(v0, v1, rounds=4):
        s = 0
        M = 0xFFFFFFFF
        for _ in range(rounds):
            s = (s + 0x9E3779B9) & M
            v0 = (v0 + ((((v1 << 4) & M) + 0xA341316C) & M ^ ((v1 + s) & M) ^ (((v1 >> 5) + 0xC8013EA4) & M))) & M
            v1 = (v1 + ((((v0 << 4) & M) + 0xAD90777D) & M ^ ((v0 + s) & M) ^ (((v0 >> 5) + 0x7E95761E) & M))) & M
        return v0, v1
    for a, b in [(0, 0), (0, 1), (7, 123456), (0xFFFFFFFF, 0xFFFFFFFF), (1, 2 ** 31)]:
        assert oracle.tea32(a, b) == tea(a, b)
    # sampler streams of different lanes are distinct and stay in [0,1)
    s0, s1 = oracle.sampler_stream(0, 0, 64), oracle.sampler_stream(0, 1, 64)
    assert not np.array_equal(s0, s1) and np.all((s0 >= 0) & (s0 < 1))


def test_bin_mapping_kat(oracle):
    rows = json.load(open(os.path.join(GOLD, "bin_mapping_kat.json")))
    assert len(rows) > 50
    for r in rows:
        d = np.uint32(r["d_bits"]).view(np.float32)
        assert oracle.bin_index(d, r["start"], r["width"], r["T"]) == r["bin"], r
    # the f32 edge cases called out in SURVEY §8c
    assert oracle.bin_index(3.52, 3.5, 0.02, 300) == 0        # (f32(3.52)-3.5)/f32(0.02) < 1
    assert oracle.bin_index(3.54, 3.5, 0.02, 300) == 1
    assert oracle.bin_index(9.5, 3.5, 0.02, 300) == -1
    assert oracle.bin_index(3.5 + 6.0 / 1024, 3.5, 6.0 / 1024, 1024) == 1   # exact widths: exact edges


def test_sincos_polynomials(oracle):
    import ctypes as C
    xs = np.linspace(-np.pi / 4, np.pi / 4, 2001).astype(np.float32)
    s, c = C.c_float(), C.c_float()
    err = 0.0
    for x in xs:
        oracle.lib().orc_sincos_q(C.c_float(float(x)), C.byref(s), C.byref(c))
        err = max(err, abs(s.value - np.sin(np.float64(x))), abs(c.value - np.cos(np.float64(x))))
    assert err < 2e-7


def test_cosine_hemisphere_warp(oracle):
    import ctypes as C
    rng = np.random.default_rng(0)
    out = (C.c_float * 3)()
    zs = []
    for u1, u2 in rng.random((4000, 2), dtype=np.float32):
        oracle.lib().orc_square_to_cos_hemi(C.c_float(float(u1)), C.c_float(float(u2)), out)
        v = np.array(out[:], np.float64)
        assert abs(np.linalg.norm(v) - 1) < 1e-6 and v[2] >= 0
        zs.append(v[2])
    assert abs(np.mean(zs) - 2.0 / 3.0) < 0.02          # E[cos] = 2/3 for a cosine-weighted hemisphere


def test_brute_force_equals_own_bvh(oracle, cornell_c1):
    scene = cornell_c1
    sd = scene.data()
    p = scene.integrator().render_params(scene.sensors()[0].film(), 3, 8)
    a = oracle.render(sd, p, n_threads=1, use_bvh=False)
    b = oracle.render(sd, p, n_threads=1, use_bvh=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


def test_golden_snapshot_c1(oracle, cornell_c1):
    """Regression snapshot of BASELINE config 1 (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(GOLD, "cornell_c1_oracle.npz"))
    scene = cornell_c1
    sd = scene.data()
    p = scene.integrator().render_params(scene.sensors()[0].film(), 0, 16)
    t4, s4, cnt = oracle.render(sd, p, use_bvh=True)
    t3, s3 = oracle.develop(sd.film, t4, s4)
    assert [cnt[k] for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces")] == list(g["counters"])
    assert int(np.count_nonzero(t3)) == int(g["nonzero_cells"])
    assert rel_l2(t3.sum(axis=(0, 1)), g["per_bin"]) < 1e-6
    assert rel_l2(t3.sum(axis=2), g["per_pixel"]) < 1e-6
    assert rel_l2(s3, g["steady"]) < 1e-6
    assert np.all(t4[..., 3] == 0)                       # channel "W" never receives anything


def test_energy_identity(oracle):
    """data_steady == data_transient.sum(axis=2) when the window covers all OPLs
    (examples/transient-nlos/1-simple-nlos-scenes.ipynb md cell 8; transientpath.py:180,217,230)."""
    scene = make_cornell(width=32, height=32, bins=128, start=0.0, window=64.0)
    sd = scene.data()
    p = scene.integrator().render_params(scene.sensors()[0].film(), 0, 32)
    t4, s4, _ = oracle.render(sd, p)
    t3, s3 = oracle.develop(sd.film, t4, s4)
    assert rel_l2(t3.sum(axis=2), s3) < 1e-5


def test_sample_slices_partition_the_render(oracle, cornell_c1):
    scene = cornell_c1
    sd = scene.data()
    integ, film = scene.integrator(), scene.sensors()[0].film()
    full = oracle.render(sd, integ.render_params(film, 0, 8), n_threads=1)
    acc_t = np.zeros_like(full[0], dtype=np.float64)
    for s0, s1 in [(0, 3), (3, 8)]:
        part = oracle.render(sd, integ.render_params(film, 0, 8, s0, s1), n_threads=1)
        acc_t += part[0]
    assert rel_l2(acc_t, full[0]) < 1e-6


# ---------------------------------------------------------------- analytic direct illumination (KAT 4)
def _direct_scene(unwarp=False):
    import mitransient_amd.mi as mi
    from mitransient_amd.transform import ScalarTransform4f as T
    mi.set_variant("llvm_ad_rgb")
    rho = [0.7, 0.5, 0.3]
    Le = [10.0, 8.0, 6.0]
    d = {
        "type": "scene",
        "integrator": {"type": "transient_path", "max_depth": 2, "camera_unwarp": unwarp},
        "sensor": {"type": "perspective", "fov": 30.0, "near_clip": 0.01, "far_clip": 100.0,
                   "to_world": T().look_at(origin=[0, 1.5, 3.0], target=[0, 0, 0], up=[0, 1, 0]),
                   "sampler": {"type": "independent", "sample_count": 4},
                   "film": {"type": "transient_hdr_film", "width": 16, "height": 16, "rfilter": {"type": "box"},
                            "temporal_bins": 120, "start_opl": 0.0 if unwarp else 2.0, "bin_width_opl": 0.1}},
        "floor": {"type": "rectangle", "to_world": T().rotate([1, 0, 0], -90).scale(4.0),
                  "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": rho}}},
        "light": {"type": "rectangle", "to_world": T().translate([0.2, 1.0, -0.3]).rotate([1, 0, 0], 90).scale(0.25),
                  "bsdf": {"type": "diffuse", "reflectance": 0.0},
                  "emitter": {"type": "area", "radiance": {"type": "rgb", "value": Le}}},
    }
    return mi.load_dict(d), np.array(rho), np.array(Le)


@pytest.mark.parametrize("unwarp", [False, True])
def test_direct_illumination_matches_quadrature(oracle, unwarp):
    """max_depth=2, one diffuse quad under one quad light: per-time-bin energy equals
    integral over pixel footprint x light of rho/pi * Le * G, binned by |cam->x| (unless unwarp) + |x->y|."""
    scene, rho, Le = _direct_scene(unwarp)
    sd = scene.data()
    film = scene.sensors()[0].film()
    spp = 512
    p = scene.integrator().render_params(film, 0, spp)
    t4, s4, _ = oracle.render(sd, p)
    t3, s3 = oracle.develop(sd.film, t4, s4)
    got = t3.sum(axis=(0, 1)).astype(np.float64)                # (T,3)

    # float64 quadrature: sub-pixel grid x light grid
    W = H = 16
    T, start, width = film.temporal_bins, film.start_opl, film.bin_width_opl
    sub, nl = 6, 24
    exp = np.zeros((T, 3))
    lu = (np.arange(nl) + 0.5) / nl * 2 - 1
    la, lb = np.meshgrid(lu, lu, indexing="ij")
    # light: center (0.2,1,-0.3), du=(0.25,0,0), dv = rotX(90)*(0,0.25,0) = (0,0,0.25), normal -y
    ly = np.stack([0.2 + 0.25 * la, np.full_like(la, 1.0), -0.3 + 0.25 * lb], -1).reshape(-1, 3)
    dA = (0.5 * 0.5) / (nl * nl)
    for py in range(H):
        for px in range(W):
            for sy in range(sub):
                for sx in range(sub):
                    o, dvec, _ = oracle.camera_ray(sd, px, py, (sx + 0.5) / sub, (sy + 0.5) / sub)
                    o, dvec = o.astype(np.float64), dvec.astype(np.float64)
                    if dvec[1] >= 0:
                        continue
                    tt = -o[1] / dvec[1]
                    x = o + tt * dvec
                    if abs(x[0]) > 4 or abs(x[2]) > 4:
                        continue
                    v = ly - x
                    r2 = (v * v).sum(-1)
                    r = np.sqrt(r2)
                    cos_x = v[:, 1] / r                      # floor normal +y
                    cos_y = v[:, 1] / r                      # light normal -y: cos = -(-v).(-y)... = v_y / r
                    G = np.clip(cos_x, 0, None) * np.clip(cos_y, 0, None) / r2
                    opl = (0.0 if unwarp else tt) + r
                    b = np.floor((opl - start) / width).astype(int)
                    ok = (b >= 0) & (b < T)
                    contrib = (G * dA / (sub * sub))[:, None] * (rho / np.pi * Le)[None, :]
                    np.add.at(exp, b[ok], contrib[ok])
    # Monte-Carlo vs quadrature: totals within 1 %, per-bin shape within 3 % of the peak
    assert abs(got.sum() - exp.sum()) / exp.sum() < 0.01
    assert np.abs(got - exp).max() / exp.max() < 0.03
    # and the steady image is the time integral
    assert rel_l2(t3.sum(axis=2), s3) < 1e-5


def test_ior_scales_optical_path(oracle):
    """A dielectric slab of thickness h adds eta*h of OPL (transientpath.py:154,232)."""
    import mitransient_amd.mi as mi
    from mitransient_amd.transform import ScalarTransform4f as T
    mi.set_variant("llvm_ad_rgb")

    def scene(with_slab):
        d = {"type": "scene",
             "integrator": {"type": "transient_path", "max_depth": 6},
             "sensor": {"type": "perspective", "fov": 1.0, "near_clip": 0.01, "far_clip": 100.0,
                        "to_world": T().look_at(origin=[0, 0, 5], target=[0, 0, 0], up=[0, 1, 0]),
                        "sampler": {"type": "independent", "sample_count": 4},
                        "film": {"type": "transient_hdr_film", "width": 1, "height": 1, "rfilter": {"type": "box"},
                                 "temporal_bins": 2000, "start_opl": 0.0, "bin_width_opl": 0.005}},
             # an emitter facing the camera at z = 0
             "light": {"type": "rectangle", "to_world": T().scale(0.5),
                       "bsdf": {"type": "diffuse", "reflectance": 0.0},
                       "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [1.0, 1.0, 1.0]}}}}
        if with_slab:
            d["slab"] = {"type": "cube", "to_world": T().translate([0, 0, 2.0]).scale([1.0, 1.0, 0.5]),
                         "bsdf": {"type": "dielectric", "int_ior": 1.5, "ext_ior": 1.0}}
        return mi.load_dict(d)

    def first_arrival(sc):
        sd = sc.data()
        p = sc.integrator().render_params(sc.sensors()[0].film(), 0, 64)
        t4, s4, _ = oracle.render(sd, p)
        prof = t4[0, 0, :, 0]
        return int(np.argmax(prof)), prof

    b0, _ = first_arrival(scene(False))
    b1, prof = first_arrival(scene(True))
    # geometric distance 4.99 (near clip 0.01); slab thickness 1.0 at eta 1.5 adds 0.5 of OPL
    assert abs(b0 * 0.005 - 4.99) < 0.011
    assert abs((b1 - b0) * 0.005 - 0.5) < 0.011


@pytest.fixture(params=[(0, 0), (1, 0), (1, 1)], ids=["plane-selects", "plane-offsets", "wide-tree"])
def node_pairs(request, host_harness):
    """the forms of the traversal: BVH2 with the node step's entry / exit planes fetched by selects (HBM scenes) or by
    sign-dependent offsets (LDS), and the 8-wide tree the fused kernel walks when the scene is staged in LDS"""
    pairs, wide = request.param
    host_harness.hh_set_node_pairs(pairs)
    host_harness.hh_set_wide(wide)
    yield request.param
    host_harness.hh_set_node_pairs(0)
    host_harness.hh_set_wide(0)


def test_host_harness_matches_oracle_bit_for_bit(oracle, host_harness, cornell_c1, node_pairs):
    """The product's per-path arithmetic (mtr_core.h + BVH2 builder, compiled for the host by a
    test-only harness) reproduces the oracle exactly: film, steady image and every counter."""
    scene = cornell_c1
    sd = scene.data()
    p = scene.integrator().render_params(scene.sensors()[0].film(), 0, 16)
    t4, s4, cnt = oracle.render(sd, p, n_threads=1)
    ht, hs, hc = hh_render(host_harness, sd, p)
    assert np.array_equal(t4, ht) and np.array_equal(s4, hs)
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert hc[k] == cnt[k]


def test_host_harness_specular_and_flags(oracle, host_harness, node_pairs):
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=24, height=24, temporal_bins=96, start_opl=0.0, bin_width_opl=0.125,
                               crop_width=20, crop_height=11, crop_offset_x=2, crop_offset_y=7)
    d["mirror"] = {"type": "conductor", "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14]}
    d["glass"] = {"type": "dielectric", "int_ior": 1.5, "ext_ior": 1.0}
    d["two"] = {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.2, 0.5, 0.7]}}}
    d["small-box"]["bsdf"] = {"type": "ref", "id": "glass"}
    d["large-box"]["bsdf"] = {"type": "ref", "id": "mirror"}
    d["back"]["bsdf"] = {"type": "ref", "id": "two"}
    d["integrator"].update(max_depth=-1, rr_depth=3, camera_unwarp=True)
    scene = mi.load_dict(d)
    sd = scene.data()
    p = scene.integrator().render_params(scene.sensors()[0].film(), 7, 12)
    t4, s4, cnt = oracle.render(sd, p, n_threads=1)
    ht, hs, hc = hh_render(host_harness, sd, p)
    assert np.array_equal(t4, ht) and np.array_equal(s4, hs) and hc["rays_shadow"] == cnt["rays_shadow"]
    assert np.count_nonzero(t4) > 1000


@pytest.mark.parametrize("wide", [0, 1, 2, 3], ids=["bvh2", "wide-8", "wide-4", "wide-8q"])
def test_host_harness_staircase_like(oracle, host_harness, wide):
    """BASELINE config-5 stand-in (procedural stair flight: conductor / dielectric / twosided mix, 852 triangles,
    BVH depth 13, max_depth 65, camera_unwarp): product arithmetic == oracle, bit for bit — through the BVH2, the
    8-wide tree (scenes staged in LDS) and the 4-wide tree (scenes walked in HBM)."""
    host_harness.hh_set_wide(wide)
    try:
        _staircase_like_bit_for_bit(oracle, host_harness)
    finally:
        host_harness.hh_set_wide(0)


def _staircase_like_bit_for_bit(oracle, host_harness):
    import mitransient_amd.mi as mi
    from mitransient_amd.scenes import staircase_like
    mi.set_variant("llvm_ad_rgb")
    scene = mi.load_dict(staircase_like(n_steps=12, balusters=2, tiles=6, width=40, height=40, temporal_bins=64, spp=4))
    sd = scene.data()
    assert sd.tri_verts.shape[0] == 852 and sd.n_materials == 7
    p = scene.integrator().render_params(scene.sensors()[0].film(), 0, 4)
    t4, s4, cnt = oracle.render(sd, p, n_threads=1, use_bvh=True)
    ht, hs, hc = hh_render(host_harness, sd, p)
    assert np.array_equal(t4, ht) and np.array_equal(s4, hs)
    assert hc["bounces"] == cnt["bounces"] and cnt["bounces"] > 5 * cnt["paths"]      # long specular chains


def mesh_light_cornell(width=24, height=24, bins=96, with_rect=True):
    """Cornell box whose light is a triangle MESH with an area emitter (as in the reference's
    examples/transient/cornell-box/cbox_diffuse.xml, where the light is an .obj) — a thin emitting cube
    below the ceiling, optionally next to the rectangle light."""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    from mitransient_amd.transform import ScalarTransform4f as T
    mi.set_variant("llvm_ad_rgb")
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=width, height=height, temporal_bins=bins, start_opl=3.0, bin_width_opl=8.0 / bins)
    if not with_rect:
        d.pop("light")
    d["mesh-light"] = {"type": "cube", "to_world": T().translate([-0.3, 0.7, 0.2]).scale([0.2, 0.03, 0.12]),
                       "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.0, 0.0, 0.0]}},
                       "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [9.0, 12.0, 15.0]}}}
    return mi.load_dict(d)


@pytest.mark.parametrize("with_rect", [False, True])
def test_mesh_area_emitter(oracle, host_harness, with_rect):
    """Area emitter on a triangle mesh [mitsuba3: Mesh::sample_position / pdf_position]: face picked by area
    with sample reuse, uniform triangle warp, pdf = 1 / mesh area.  Product arithmetic == oracle bit for bit,
    and the NEE/MIS estimate agrees with the BSDF-sampling-only picture through the usual energy check."""
    scene = mesh_light_cornell(with_rect=with_rect)
    sd = scene.data()
    em = [e for e in sd.emitters if e.is_mesh]
    assert len(em) == 1 and em[0].n_tris == 12
    p = scene.integrator().render_params(scene.sensors()[0].film(), 3, 16)
    t4, s4, cnt = oracle.render(sd, p, n_threads=1)
    ht, hs, hc = hh_render(host_harness, sd, p)
    assert np.array_equal(t4, ht) and np.array_equal(s4, hs)
    for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces"):
        assert hc[k] == cnt[k]
    assert np.count_nonzero(t4) > 2000 and np.isfinite(t4).all()
    tb, sb, _ = oracle.render(sd, p, use_bvh=True)
    assert np.array_equal(t4, tb)


def test_mesh_emitter_equals_rectangle_emitter_in_expectation(oracle, tmp_path):
    """The same quad light once as a `rectangle` and once as a two-triangle `obj` mesh: different sampling code
    (analytic rectangle vs face-pmf + triangle warp), same estimator in expectation."""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
    obj = tmp_path / "quad.obj"
    obj.write_text("v -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nf 1 2 3\nf 1 3 4\n")
    imgs = []
    for as_mesh in (False, True):
        d = mitr.cornell_box()
        d["sensor"]["film"].update(width=12, height=12, temporal_bins=32, start_opl=3.0, bin_width_opl=0.25)
        if as_mesh:
            lt = d["light"]
            d["light"] = {"type": "obj", "filename": str(obj), "to_world": lt["to_world"], "bsdf": lt["bsdf"],
                          "emitter": lt["emitter"]}
        scene = mi.load_dict(d)
        sd = scene.data()
        assert bool(sd.emitters[0].is_mesh) == as_mesh
        p = scene.integrator().render_params(scene.sensors()[0].film(), 1, 512)
        t4, s4, _ = oracle.render(sd, p)
        imgs.append((t4.sum(axis=(0, 1))[:, :3], s4[..., :3].sum(axis=(0, 1))))
    (ta, sa), (tb, sb) = imgs
    assert np.allclose(sa, sb, rtol=0.02)
    assert rel_l2(ta, tb) < 0.03


def _flipped_cornell(light=None, floor_flipped=False, **film):
    """cornell_box() with `flip_normals` on rectangles.  light="flip+turn": the light turned to face the ceiling AND
    flipped (emits downwards again, with a mirrored parameterisation); "turn": turned only (emits into the ceiling)."""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    from mitransient_amd.transform import ScalarTransform4f as T
    mi.set_variant("llvm_ad_rgb")
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=24, height=24, temporal_bins=64, start_opl=3.5, bin_width_opl=6.0 / 64)
    d["sensor"]["film"].update(film)
    if light in ("flip+turn", "turn"):
        d["light"]["to_world"] = T().translate([0, 0.99, 0.01]).rotate([1, 0, 0], -90).scale([0.23, 0.19, 0.19])
        if light == "flip+turn":
            d["light"]["flip_normals"] = True
    if floor_flipped:
        d["floor"]["flip_normals"] = True
    return mi.load_dict(d)


def test_flip_normals_on_rectangles(oracle, host_harness):
    """`flip_normals` on an analytic rectangle negates the frame normal [mitsuba3: Rectangle] — emitter side,
    one-sided BSDF side — and leaves the parameterisation alone.  Product arithmetic == oracle bit for bit; a light turned
    away and flipped back lights the box like the original; turned away only, it lights (almost) nothing; a flipped
    (one-sided, diffuse) floor reflects nothing."""
    def run(scene, spp=32):
        sd = scene.data()
        p = scene.integrator().render_params(scene.sensors()[0].film(), 0, spp)
        t4, s4, cnt = oracle.render(sd, p, n_threads=0)
        ht, hs, hc = hh_render(host_harness, sd, p)
        assert np.array_equal(t4, ht) and np.array_equal(s4, hs) and hc == {**hc, **{k: cnt[k] for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces")}}
        return s4[..., :3] / np.maximum(s4[..., 3:], 1)
    base = run(_flipped_cornell())
    back = run(_flipped_cornell("flip+turn"))
    away = run(_flipped_cornell("turn"))
    e0, e1, e2 = float(base.sum()), float(back.sum()), float(away.sum())
    assert abs(e1 - e0) <= 0.05 * e0, (e0, e1)
    assert e2 <= 0.3 * e0, (e0, e2)      # (light leaks out of the 1 cm gap below the ceiling by inter-reflection)
    sd = _flipped_cornell("flip+turn").data()
    assert sd.emitters[0].flip_normals == 1 and sd.shapes[0].is_rectangle == 3
    # flipped floor: the floor's rows of the image (below the boxes) go black — it only blocks
    dark = run(_flipped_cornell(floor_flipped=True))
    assert float(base[21:24, 8:16].sum()) > 1.0 and float(dark[21:24, 8:16].sum()) <= 0.01 * float(base[21:24, 8:16].sum())
    assert float(dark.sum()) < 0.9 * e0
