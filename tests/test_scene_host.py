"""Host logic: transforms, scene-dict flattening, plugin surface, BVH builder (via the test host harness)."""
import ctypes as C

import numpy as np
import pytest

from conftest import make_cornell


def test_transform_chaining_matches_matrix_product():
    from mitransient_amd.transform import ScalarTransform4f as T
    t = T().translate([1, 2, 3]).rotate([0, 1, 0], 90).scale([2, 1, 1])
    p = t.transform_affine(np.array([1.0, 0.0, 0.0]))            # scale -> (2,0,0); rotY(90) -> (0,0,-2); + t
    assert np.allclose(p, [1, 2, 1])
    cam = T().look_at(origin=[0, 0, 3.9], target=[0, 0, 0], up=[0, 1, 0])
    assert np.allclose(cam.transform_vector([0, 0, 1]), [0, 0, -1])
    assert np.allclose(cam.translation(), [0, 0, 3.9])
    assert np.allclose((t @ t.inverse()).matrix, np.eye(4), atol=1e-12)


def test_cornell_box_values_and_flattening():
    """utils.py:78-220: 36 triangles, 3 diffuse albedos, one quad light, the film/integrator defaults."""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
    d = mitr.cornell_box()
    assert d["integrator"]["type"] == "transient_path" and d["integrator"]["max_depth"] == 8
    f = d["sensor"]["film"]
    assert (f["type"], f["width"], f["height"], f["temporal_bins"], f["start_opl"], f["bin_width_opl"]) == \
        ("transient_hdr_film", 256, 256, 300, 3.5, 0.02)
    scene = mi.load_dict(d)
    sd = scene.data()
    assert sd.tri_verts.shape == (36, 9) and sd.n_materials == 3 and sd.n_emitters == 1
    assert np.allclose(list(sd.materials[0].a), [0.885809, 0.698859, 0.666422])
    assert np.allclose(list(sd.emitters[0].radiance), [18.387, 13.9873, 6.75357])
    assert list(sd.tri_emitter[:2]) == [0, 0] and np.all(sd.tri_emitter[2:] == -1)
    # light faces down (-y); geometry inside [-1,1]^3 except the open front
    e = sd.emitters[0]
    n = np.cross(list(e.du), list(e.dv))
    assert n[1] < 0 and abs(n[0]) < 1e-9 and abs(n[2]) < 1e-9
    assert np.abs(sd.tri_verts).max() <= 1.0101          # the tall box dips 0.01 below the floor (utils.py:214)
    # all box faces point outwards: normal . (centroid - box centre) > 0
    for name, (a, b) in zip(sd.shape_names, sd.shape_ranges):
        if "box" not in name:
            continue
        tris = sd.tri_verts[a:b].reshape(-1, 3, 3).astype(np.float64)
        ctr = tris.reshape(-1, 3).mean(0)
        nn = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
        assert np.all(((tris.mean(1) - ctr) * nn).sum(1) > 0)
    integ = scene.integrator()
    assert (integ.max_depth, integ.rr_depth, integ.camera_unwarp, integ.discard_direct_light) == (8, 5, False, False)
    film = scene.sensors()[0].film()
    assert film.end_opl() == pytest.approx(9.5)
    assert scene.sensors()[0].sampler().sample_count() == 256


def test_plugin_defaults_and_errors():
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    from mitransient_amd.scene import Properties
    mi.set_variant("cuda_ad_rgb")
    with pytest.raises(ValueError):
        mi.set_variant("scalar_rgb")
    f = mitr.TransientHDRFilm(Properties("transient_hdr_film", {"rfilter": {"type": "box"}}))
    assert (f.temporal_bins, f.bin_width_opl, f.start_opl, f.size()) == (2048, 0.003, 0.0, (768, 576))
    i = mitr.TransientPath(Properties("transient_path", {}))
    assert (i.max_depth, i.rr_depth, i.camera_unwarp) == (6, 5, False)
    assert mitr.TransientPath(Properties("transient_path", {"max_depth": -1})).max_depth == 0xFFFFFFFF
    with pytest.raises(Exception):
        mitr.TransientPath(Properties("transient_path", {"rr_depth": 0}))
    d = mitr.cornell_box()
    d["integrator"]["type"] = "transient_prbvolpath"          # the differentiable tier is out of scope
    with pytest.raises(ValueError, match="unknown plugin"):
        mi.load_dict(d)
    d = mitr.cornell_box()
    d["floor"]["bsdf"] = {"type": "hair"}
    with pytest.raises(ValueError, match="unknown plugin"):
        mi.load_dict(d).data()
    d["floor"]["bsdf"] = {"type": "roughplastic", "distribution": "phong"}      # beckmann (mitsuba's default) and ggx are built
    with pytest.raises(ValueError, match="beckmann"):
        mi.load_dict(d).data()
    from mitransient_amd import _cabi
    d["floor"]["bsdf"] = {"type": "roughplastic"}
    assert any(m.flags & _cabi.MTR_MAT_BECKMANN for m in mi.load_dict(d).data().materials)
    d = mitr.cornell_box()
    d["sensor"]["film"]["crop_width"] = 9999
    with pytest.raises(ValueError):
        mi.load_dict(d)
    p = mi.traverse(mi.load_dict(mitr.cornell_box()))
    assert p["sensor.film.temporal_bins"] == 300
    p["sensor.film.start_opl"] = 1.0
    p.update()


def test_traverse_updates_film():
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    scene = mi.load_dict(mitr.cornell_box())
    p = mi.traverse(scene)
    p["sensor.film.bin_width_opl"] = 0.5
    p["sensor.film.temporal_bins"] = 12
    p.update()
    film = scene.sensors()[0].film()
    assert film.bin_width_opl == 0.5 and film.temporal_bins == 12
    assert scene.data().film.temporal_bins == 12


def test_perspective_camera_rays(oracle):
    """centre pixel looks down -z from (0,0,3.9) offset by near_clip; corners span the fov."""
    scene = make_cornell(width=64, height=64)
    sd = scene.data()
    o, d, maxt = oracle.camera_ray(sd, 32, 32, 0.0, 0.0)
    assert np.allclose(d, [0, 0, -1], atol=1e-6) and np.allclose(o, [0, 0, 3.9 - 0.001], atol=1e-6)
    assert maxt == pytest.approx(100.0 - 0.001, rel=1e-5)
    o, d, _ = oracle.camera_ray(sd, 0, 0, 0.0, 0.0)           # top-left pixel: -x (red wall side), +y
    half = np.tan(np.radians(39.3077 / 2))
    assert d[0] < 0 and d[1] > 0
    assert d[0] / -d[2] == pytest.approx(-half, rel=1e-4) and d[1] / -d[2] == pytest.approx(half, rel=1e-4)


def _hh_intersect(lib, sd, o, d, maxt=None):
    n = o.shape[0]
    t = np.empty(n, np.float32)
    prim = np.empty(n, np.int32)
    occ = np.empty(n, np.uint8)
    fp = C.POINTER(C.c_float)
    desc = sd.desc()
    mt = np.ascontiguousarray(maxt, np.float32) if maxt is not None else None
    rc = lib.hh_intersect(C.byref(desc), n, o.ctypes.data_as(fp), d.ctypes.data_as(fp),
                          mt.ctypes.data_as(fp) if mt is not None else None, t.ctypes.data_as(fp),
                          prim.ctypes.data_as(C.POINTER(C.c_int32)), occ.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert rc == 0
    return t, prim, occ


@pytest.mark.parametrize("pairs,wide", [(0, 0), (1, 0), (1, 1), (0, 2), (0, 3)], ids=["plane-selects", "plane-offsets", "wide-tree", "quantised-4", "quantised-8"])
def test_product_bvh_equals_brute_force(oracle, host_harness, pairs, wide):
    host_harness.hh_set_node_pairs(pairs)
    host_harness.hh_set_wide(wide)
    try:
        _bvh_equals_brute_force(oracle, host_harness)
    finally:
        host_harness.hh_set_node_pairs(0)
        host_harness.hh_set_wide(0)


def _bvh_equals_brute_force(oracle, host_harness):
    """The product's SAH BVH2 + node-packet traversal returns exactly the oracle's brute-force closest
    hit (same t bits, same primitive) and the same occlusion answer, for random rays."""
    scene = make_cornell()
    sd = scene.data()
    rng = np.random.default_rng(5)
    n = 20000
    o = rng.uniform(-0.95, 0.95, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[:100, 0] = 0.0                                           # axis-parallel components (inf reciprocals)
    d[100:200, 1] = 0.0
    maxt = np.where(rng.random(n) < 0.5, np.inf, rng.uniform(0.1, 2.0, n)).astype(np.float32)
    t0, p0, occ0 = oracle.intersect(sd, o, d, maxt, use_bvh=False)
    t1, p1, occ1 = _hh_intersect(host_harness, sd, o, d, maxt)
    assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    assert np.array_equal(p0, p1) and np.array_equal(occ0, occ1)
    assert (p0 >= 0).mean() > 0.5


def test_object_nodes_equal_brute_force(oracle, host_harness, monkeypatch):
    """MTR_NO_BOX_NODES: the cubes as plain object nodes (slab tests against the six faces' object-space boxes)."""
    monkeypatch.setenv("MTR_NO_BOX_NODES", "1")
    assert host_harness.hh_count_box_nodes(C.byref(make_cornell().data().desc())) == 0
    host_harness.hh_set_node_pairs(1)
    host_harness.hh_set_wide(1)
    try:
        _bvh_equals_brute_force(oracle, host_harness)
    finally:
        host_harness.hh_set_node_pairs(0)
        host_harness.hh_set_wide(0)


@pytest.mark.parametrize("wide", [0, 1], ids=["bvh2", "wide-tree"])
@pytest.mark.parametrize("to_world", ["rotated", "sheared-far", "aligned"])
def test_box_node_equals_brute_force(oracle, host_harness, wide, to_world):
    """A `cube` is a BOX node of the 8-wide tree (mtr_core.h box_select): the slab distances in object space select the
    faces whose leaves are visited.  Bit for bit the brute-force answer over all twelve triangles — for rays aimed at the
    edges and corners, grazing the faces, starting inside the box and starting on its surface."""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
    T = mi.ScalarTransform4f
    if to_world == "rotated":
        tw = T().translate([0.3, -0.2, 0.1]).rotate([1, 2, 3], 37.0).scale([0.3, 0.6, 0.2])
        M = np.asarray(tw.matrix, np.float64)
    elif to_world == "aligned":
        tw = T().translate([0.3, -0.2, 0.1]).scale([0.3, 0.6, 0.2])
        M = np.asarray(tw.matrix, np.float64)
    else:                                  # not even orthogonal, and far from the origin (cornell-box units)
        M = np.array([[80.0, 25.0, 0.0, 350.0], [-10.0, 160.0, 30.0, 165.0], [5.0, -20.0, 90.0, 270.0], [0, 0, 0, 1.0]])
        tw = T(M)
    d = mitr.cornell_box()
    for k in list(d):
        if isinstance(d[k], dict) and d[k].get("type") in ("cube", "rectangle", "obj") and k != "light":
            del d[k]
    d["box"] = {"type": "cube", "to_world": tw, "bsdf": {"type": "diffuse"}}
    scene = mi.load_dict(d)
    sd = scene.data()
    assert host_harness.hh_count_box_nodes(C.byref(sd.desc())) == 1
    rng = np.random.default_rng(11)
    n = 150000
    A, b = M[:3, :3], M[:3, 3]
    def to_w(p): return p @ A.T + b
    # targets in object space: faces, edges (two coordinates at +-1), corners (three), each with a jitter of a few ulps .. 1e-3
    tgt = rng.uniform(-1, 1, (n, 3))
    kind = rng.integers(0, 4, n)
    for i in range(3):
        snap = (kind >= 1) & (rng.random(n) < 0.75) | (kind == 3)
        tgt[snap, i] = np.sign(tgt[snap, i])
    tgt += rng.choice([0.0, 1e-7, 1e-5, 1e-3], (n, 1)) * rng.normal(size=(n, 3))
    org = rng.uniform(-3, 3, (n, 3))
    inside = rng.random(n) < 0.2
    org[inside] = rng.uniform(-0.999, 0.999, (inside.sum(), 3))
    on_face = rng.random(n) < 0.1                                   # origins on (just off) a face, as after a bounce
    ax = rng.integers(0, 3, n)
    org[on_face, ax[on_face]] = np.sign(org[on_face, ax[on_face]]) * (1.0 + rng.choice([0.0, 1e-6, 1e-4], on_face.sum()))
    # (not IN the plane of that face: Moller-Trumbore on an in-plane ray divides noise by noise and reports hits outside the
    # triangle's own bounding box, which brute force keeps and any BVH culls)
    tgt[on_face, ax[on_face]] = rng.uniform(-0.9, 0.9, on_face.sum())
    o = to_w(org).astype(np.float32)
    dirs = to_w(tgt) - to_w(org)
    dirs[on_face & (rng.random(n) < 0.5)] *= -1.0                   # ... leaving the surface
    dirs = (dirs / np.maximum(np.linalg.norm(dirs, axis=1, keepdims=True), 1e-30)).astype(np.float32)
    dirs[:50, 0] = 0.0
    maxt = np.where(rng.random(n) < 0.7, np.inf, rng.uniform(0.1, 4.0, n) * np.abs(A).max()).astype(np.float32)
    host_harness.hh_set_node_pairs(1 if wide else 0)
    host_harness.hh_set_wide(wide)
    try:
        t0, p0, occ0 = oracle.intersect(sd, o, dirs, maxt, use_bvh=False)
        t1, p1, occ1 = _hh_intersect(host_harness, sd, o, dirs, maxt)
    finally:
        host_harness.hh_set_node_pairs(0)
        host_harness.hh_set_wide(0)
    bad = np.flatnonzero((t0.view(np.uint32) != t1.view(np.uint32)) | (p0 != p1) | (occ0 != occ1))
    assert bad.size == 0, (bad[:5], t0[bad[:5]], t1[bad[:5]], p0[bad[:5]], p1[bad[:5]], org[bad[:5]], tgt[bad[:5]], maxt[bad[:5]])
    assert 0.2 < (p0 >= 0).mean() < 0.98


def test_bvh_structure(host_harness):
    scene = make_cornell()
    desc = scene.data().desc()
    nn, dd, ll = C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert host_harness.hh_bvh_info(C.byref(desc), C.byref(nn), C.byref(dd), C.byref(ll)) == 0
    assert ll.value == nn.value + 1 and 9 <= ll.value <= 36 and dd.value <= 8     # fits the smallest LDS stack
    assert host_harness.hh_count_box_nodes(C.byref(desc)) == 2


def test_bvh_degenerate_inputs(oracle, host_harness):
    """empty scene, a single triangle, many coincident triangles."""
    import mitransient_amd.mi as mi
    from mitransient_amd.scene import SceneData
    from mitransient_amd import _cabi
    base = make_cornell().data()

    def with_tris(v):
        sd = SceneData()
        sd.tri_verts = np.ascontiguousarray(v, np.float32).reshape(-1, 9)
        n = sd.tri_verts.shape[0]
        sd.tri_material = np.zeros(n, np.uint32)
        sd.tri_emitter = np.full(n, -1, np.int32)
        sd.materials, sd.n_materials = base.materials, 1
        sd.camera, sd.film = base.camera, base.film
        return sd

    o = np.array([[0.2, 0.2, 2.0]] * 3, np.float32)
    d = np.array([[0, 0, -1], [0, 0, 1], [0.1, 0, -1]], np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tri = [[0, 0, 0, 1, 0, 0, 0, 1, 0]]
    for verts in ([], tri, tri * 9):
        sd = with_tris(np.array(verts, np.float32))
        t0, p0, occ0 = oracle.intersect(sd, o, d)
        t1, p1, occ1 = _hh_intersect(host_harness, sd, o, d)
        assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32)) and np.array_equal(p0, p1)
        assert np.array_equal(occ0, occ1)
        if len(verts):
            assert p0[0] == 0                                   # coincident triangles: lowest index wins


def test_shared_edges_are_closed(oracle, host_harness):
    """Rays aimed AT the shared diagonal of a two-triangle quad never slip between the triangles (mtr_core.h: kEdgeEps,
    oracle: MTR_EDGE_EPS).  Moller-Trumbore evaluates the two sides of an edge with different operation orders; with the
    plain 0 <= u, v, u + v <= 1 test about one ray in nine aimed at the centre of examples/transient-nlos/nlos-z-simple.xml's
    relay wall — where its laser points — missed both triangles: every NLOS frame came out 0.886 of the reference's figure
    (DESIGN.md section 2).  Oracle and product agree bit for bit on every ray, hit or not."""
    from mitransient_amd.scene import SceneData
    base = make_cornell().data()
    # Plane.ply of the reference's scene: (-1,-1) (1,-1) (1,1) | (-1,-1) (1,1) (-1,1), and a rotated, shifted, scaled copy
    quad = np.array([[-1, -1, 0, 1, -1, 0, 1, 1, 0], [-1, -1, 0, 1, 1, 0, -1, 1, 0]], np.float64).reshape(2, 3, 3)
    c, s_ = np.cos(0.7), np.sin(0.7)
    rot = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]]) @ np.array([[1, 0, 0], [0, np.cos(0.3), -np.sin(0.3)], [0, np.sin(0.3), np.cos(0.3)]])
    rng = np.random.default_rng(11)
    for verts, origin in ((quad, np.zeros(3)), (quad @ rot.T * 0.37 + np.array([3.1, -2.2, 5.3]), np.array([3.1, -2.2, 5.3]))):
        sd = SceneData()
        sd.tri_verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 9)
        sd.tri_material = np.zeros(2, np.uint32)
        sd.tri_emitter = np.full(2, -1, np.int32)
        sd.materials, sd.n_materials = base.materials, 1
        sd.camera, sd.film = base.camera, base.film
        v = sd.tri_verts.reshape(2, 3, 3).astype(np.float64)
        n = 200000
        # targets: the quad's centre (the laser spot) for half of the rays, random points of the diagonal for the rest
        lam = np.where(np.arange(n) < n // 2, 0.5, rng.uniform(0.02, 0.98, n))[:, None]
        target = v[0, 0] * (1 - lam) + v[0, 2] * lam
        nrm = np.cross(v[0, 1] - v[0, 0], v[0, 2] - v[0, 0]); nrm /= np.linalg.norm(nrm)
        dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        dirs = np.where((dirs @ nrm)[:, None] < 0, -dirs, dirs)            # origins on the normal's side ...
        dirs = dirs[np.abs(dirs @ nrm) > 0.05]                             # ... and not grazing
        o = (target[:len(dirs)] + dirs * rng.uniform(0.5, 4.0, (len(dirs), 1))).astype(np.float32)
        d = (target[:len(dirs)] - o.astype(np.float64))
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        t0, p0, occ0 = oracle.intersect(sd, o, d)
        t1, p1, occ1 = _hh_intersect(host_harness, sd, o, d)
        assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32)) and np.array_equal(p0, p1) and np.array_equal(occ0, occ1)
        # the plain 0 <= u, v test loses 8 - 12 % of these rays.  The reference's wall: none may miss.  The small, rotated quad far
        # from the origin, seen from up to 11 of its side lengths away at down to 3 degrees: the rounding of u, v grows with
        # distance / (size * cos) and passes the tolerance for a few grazing rays in 10^4 — a ray that misses there is within
        # 10^-6 of the edge of a triangle it sees under a few degrees
        miss = p0 < 0
        assert miss.mean() <= (0.0 if origin[0] == 0.0 else 5e-4), (int(miss.sum()), len(p0))
        assert not np.any(miss & (np.abs(d.astype(np.float64) @ nrm) > 0.3))
    # ... and a ray that passes OUTSIDE the quad by more than the tolerance still misses
    sd.tri_verts = np.ascontiguousarray(quad, np.float32).reshape(-1, 9)
    o = np.array([[1.0 + 1e-4, 0.3, 2.0], [0.2, -1.0 - 1e-4, 2.0]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, -1]], np.float32)
    assert np.all(oracle.intersect(sd, o, d)[1] < 0) and np.all(_hh_intersect(host_harness, sd, o, d)[1] < 0)


@pytest.mark.parametrize("wide", [0, 3], ids=["bvh2", "quantised-8"])
def test_spatial_splits_equal_brute_force(oracle, host_harness, tmp_path, monkeypatch, wide):
    """mtr_bvh.cpp's spatial splits (scenes of >= 1024 triangles): 1500 long thin triangles at random angles — the case in which
    a triangle's box is mostly empty and the builder prefers a plane that cuts references in two.  The tree must return the
    brute-force closest hit (t bits, primitive) and occlusion answer for random rays, duplicated references or not, and the
    splits must actually have happened (more leaves than the build without them)."""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
    rng = np.random.default_rng(11)
    fn = tmp_path / "sticks.obj"
    with open(fn, "w") as fh:
        for i in range(1500):
            c = rng.uniform(-0.9, 0.9, 3)
            axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
            side = np.cross(axis, rng.normal(size=3)); side /= np.linalg.norm(side)
            half = rng.uniform(0.05, 0.9) if i % 3 else rng.uniform(0.01, 0.05)
            for p in (c - half * axis, c + half * axis, c + 0.01 * side):
                fh.write("v %.9g %.9g %.9g\n" % tuple(np.clip(p, -0.99, 0.99)))
            fh.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=8, height=8, temporal_bins=8)
    for name in ("small-box", "large-box"):
        d.pop(name)
    d["sticks"] = {"type": "obj", "filename": str(fn), "face_normals": True, "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.5, 0.5, 0.5]}}}
    sd = mi.load_dict(d).data()
    assert sd.tri_verts.shape[0] >= 1500
    n = 6000
    o = rng.uniform(-0.95, 0.95, (n, 3)).astype(np.float32)
    dirs = rng.normal(size=(n, 3))
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    maxt = np.where(rng.random(n) < 0.5, np.inf, rng.uniform(0.1, 2.0, n)).astype(np.float32)
    t0, p0, occ0 = oracle.intersect(sd, o, dirs, maxt, use_bvh=False)
    leaves = {}
    host_harness.hh_set_wide(wide)
    try:
        for off in (False, True):
            if off:
                monkeypatch.setenv("MTR_BVH_NO_SBVH", "1")
            t1, p1, occ1 = _hh_intersect(host_harness, sd, o, dirs, maxt)
            assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
            assert np.array_equal(p0, p1) and np.array_equal(occ0, occ1)
            nn, dd, ll = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
            desc = sd.desc()
            assert host_harness.hh_bvh_info(C.byref(desc), C.byref(nn), C.byref(dd), C.byref(ll)) == 0
            leaves[off] = ll.value
    finally:
        host_harness.hh_set_wide(0)
    assert leaves[False] > leaves[True] * 1.1, leaves


def test_obj_loader(tmp_path):
    from mitransient_amd.scene import load_obj
    p = tmp_path / "q.obj"
    p.write_text("# quad + stray line\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nl 1 2\nf 1/1/1 2/1/1 3/1/1 4/1/1\nf -4 -3 -2\n")
    t = load_obj(str(p))
    assert t.shape == (3, 3, 3)
    assert np.allclose(t[1], [[0, 0, 0], [1, 1, 0], [0, 1, 0]])


@pytest.mark.parametrize("which", ["cornell", "staircase_like"])
def test_collapsed_trees_cover_the_bvh2(host_harness, which):
    """The 8-wide tree (LDS scenes) and the quantised 4-wide tree (HBM scenes) are collapses of the BVH2: every leaf
    referenced exactly once, every node reachable exactly once, every triangle inside its quantised leaf box, child boxes
    inside their parent's."""
    import mitransient_amd.mi as mi
    if which == "cornell":
        scene = make_cornell()
    else:
        from mitransient_amd.scenes import staircase_like
        mi.set_variant("llvm_ad_rgb")
        scene = mi.load_dict(staircase_like(n_steps=12, balusters=2, tiles=6, width=16, height=16, temporal_bins=16, spp=1))
    sd = scene.data()
    desc = sd.desc()
    n8, n4 = C.c_uint32(0), C.c_uint32(0)
    assert host_harness.hh_check_wide(C.byref(desc), C.byref(n8), C.byref(n4)) == 0
    assert n8.value >= 1 and n4.value >= 1
    if which == "cornell":
        assert n8.value == 3            # walls + light under the root, one node per box


def test_multi_pass_prepare_host_logic(oracle, host_harness, monkeypatch):
    """TransientADIntegrator.prepare above the single-pass limit (common.py:56-85): pass sizes, the remainder pass, seeds =
    UInt32(seeder.next_1d() * 2^32) with the seeder seeded by (seed, num_passes); product arithmetic == oracle on one pass with
    sample_scale = 1 / total_spp (mtr_render_params.spp_scale).  film.prepare needs a GPU, so it is stubbed here."""
    from conftest import make_cornell, hh_render
    from mitransient_amd.films.transient_hdr_film import TransientHDRFilm
    monkeypatch.setattr(TransientHDRFilm, "prepare", lambda self, aovs=(): 4)
    scene = make_cornell(width=12, height=10, bins=32)
    integ = scene.integrator()
    sens = scene.sensors()[0]
    assert len(integ.prepare(scene, sens, 0, 16, [])) == 1                       # 1920 lanes: one pass, seed = base + seed
    integ.max_wavefront_size, integ.pass_wavefront_size = 1000, 12 * 10 * 4 + 5
    passes = integ.prepare(scene, sens, 3, 11, [])
    assert [s for _, s in passes] == [4, 4, 3]
    seeder = [oracle.sampler_stream(3, j, 1)[0] for j in range(3)]
    want = [int(np.uint32(np.float32(f) * np.float32(2 ** 32))) for f in seeder]
    assert [p.seed_value() for p, _ in passes] == want
    assert [s for _, s in integ.prepare(scene, sens, 3, 12, [])] == [4, 4, 4]    # no remainder pass
    integ.pass_wavefront_size = 100
    with pytest.raises(Exception, match="film is too big"):
        integ.prepare(scene, sens, 3, 11, [])
    sd = scene.data()
    film = sens.film()
    p = integ.render_params(film, passes[2][0].seed_value(), 3, spp_scale=11)
    t4, s4, c = oracle.render(sd, p, n_threads=1)
    ht, hs, hc = hh_render(host_harness, sd, p)
    assert np.array_equal(t4, ht) and np.array_equal(s4, hs) and c["paths"] == 12 * 10 * 3
    p1 = integ.render_params(film, passes[2][0].seed_value(), 3)                 # the same lanes scaled by 1/3 instead of 1/11
    t1, _, _ = oracle.render(sd, p1, n_threads=1)
    assert np.allclose(t4 * np.float32(11.0 / 3.0), t1, rtol=2e-6, atol=0)


def test_c_abi_rejects_bad_materials(host_harness):
    """derive_scene (the C-ABI's own validation, not only the Python loader's): a non-positive second roughness of an
    anisotropic lobe and MTR_MAT_TWOSIDED on a transmissive BSDF are refused (ggx_eval / beck_eval divide by alpha_v; the
    two-sided adapter is defined for materials without a transmission component)"""
    import copy
    from mitransient_amd import _cabi
    base = make_cornell().data()
    nn, dd, ll = C.c_uint32(), C.c_uint32(), C.c_uint32()

    def accepted(edit):
        sd = copy.copy(base)
        mats = (_cabi.mtr_material * base.n_materials)()
        for i in range(base.n_materials):
            C.memmove(C.byref(mats[i]), C.byref(base.materials[i]), C.sizeof(_cabi.mtr_material))
        edit(mats[0])
        sd.materials = mats
        desc = sd.desc()
        return host_harness.hh_bvh_info(C.byref(desc), C.byref(nn), C.byref(dd), C.byref(ll)) == 0

    def aniso(av):
        def edit(m):
            m.type = 4; m.flags = _cabi.MTR_MAT_ANISOTROPIC; m.alpha = 0.2; m.c2[0] = av      # roughconductor: alpha_v in c2[0]
        return edit

    def twosided(t):
        def edit(m):
            m.type = t; m.flags = _cabi.MTR_MAT_TWOSIDED; m.alpha = 0.2; m.int_ior = 1.5; m.ext_ior = 1.0
        return edit
    assert accepted(lambda m: None)
    assert accepted(aniso(0.3)) and not accepted(aniso(0.0)) and not accepted(aniso(-0.1))
    assert accepted(twosided(0)) and accepted(twosided(4))
    for t in (2, 6, 7):                      # dielectric, roughdielectric, thindielectric
        assert not accepted(twosided(t)), t


def test_parallel_build_makes_the_sequential_tree(host_harness, tmp_path, monkeypatch):
    """mtr_bvh.cpp builds the subtrees of a large scene on worker threads (Builder::build_parallel_finish).  Every split is a function
    of its node's references alone and the pieces are stitched in the order they were deferred, so the tree — BVH2 packets, both
    quantised collapses, slot order, pair records — is the one a single thread builds, byte for byte: 17 000 random triangles with
    spatial splits on, 1 / 3 / 8 threads."""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
    rng = np.random.default_rng(5)
    fn = tmp_path / "soup.obj"
    n = 17000
    with open(fn, "w") as fh:
        for i in range(n):
            c = rng.uniform(-0.9, 0.9, 3)
            axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
            side = np.cross(axis, rng.normal(size=3)); side /= np.linalg.norm(side)
            half = rng.uniform(0.02, 0.5) if i % 4 == 0 else rng.uniform(0.005, 0.03)
            for p in (c - half * axis, c + half * axis, c + 0.01 * side):
                fh.write("v %.9g %.9g %.9g\n" % tuple(np.clip(p, -0.99, 0.99)))
            fh.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=8, height=8, temporal_bins=8)
    d["soup"] = {"type": "obj", "filename": str(fn), "face_normals": True, "bsdf": {"type": "ref", "id": "white"}}
    sd = mi.load_dict(d).data()
    desc = sd.desc()
    host_harness.hh_tree_hash.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    hashes = {}
    for threads in (1, 3, 8):
        monkeypatch.setenv("MTR_BVH_THREADS", str(threads))
        h = C.c_uint64(0)
        assert host_harness.hh_tree_hash(C.byref(desc), C.byref(h)) == 0
        hashes[threads] = h.value
    assert len(set(hashes.values())) == 1, hashes
    nn, dep, lv = C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert host_harness.hh_bvh_info(C.byref(desc), C.byref(nn), C.byref(dep), C.byref(lv)) == 0
    assert lv.value > n // 2, lv.value           # (spatial splits happened: more leaves than triangle pairs)


def test_depth_budget_keeps_the_tree_walkable(oracle, host_harness, tmp_path, monkeypatch):
    """the walkers' stacks end at 64 levels, and a scene whose tree is deeper has no kernel at all.  Near the limit the builder
    splits at the median instead of where the SAH says (mtr_bvh.cpp, DEPTH BUDGET): depth + ceil(log2 references) stays below
    the budget.  No geometry that fits f32 drives a binned SAH 64 levels deep, so the test lowers the budget to 14 on a scene
    whose tree is 17 levels deep without it — 400 triangles falling off geometrically — and checks depth and brute-force hits."""
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
    fn = tmp_path / "chain.obj"
    n = 400
    with open(fn, "w") as fh:
        for i in range(n):
            s = 0.9 * 0.93 ** i
            x = -0.95 + 1.9 * (1.0 - 0.93 ** i)
            for p in ((x, -s, -s), (x, s, -s), (x, 0.0, s)):
                fh.write("v %.9g %.9g %.9g\n" % p)
            fh.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=8, height=8, temporal_bins=8)
    for name in ("small-box", "large-box"):
        d.pop(name)
    d["chain"] = {"type": "obj", "filename": str(fn), "face_normals": True, "bsdf": {"type": "ref", "id": "white"}}
    sd = mi.load_dict(d).data()
    desc = sd.desc()
    rng = np.random.default_rng(3)
    m = 4000
    o = rng.uniform(-0.95, 0.95, (m, 3)).astype(np.float32)
    dirs = rng.normal(size=(m, 3)); dirs[:, 0] *= 3.0
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    maxt = np.full(m, np.inf, np.float32)
    t0, p0, occ0 = oracle.intersect(sd, o, dirs, maxt, use_bvh=False)
    depths = {}
    for budget in (None, 14):
        if budget:
            monkeypatch.setenv("MTR_BVH_DEPTH_BUDGET", str(budget))
        nn, dep, lv = C.c_uint32(), C.c_uint32(), C.c_uint32()
        assert host_harness.hh_bvh_info(C.byref(desc), C.byref(nn), C.byref(dep), C.byref(lv)) == 0
        depths[budget] = dep.value
        for wide in (0, 3):
            host_harness.hh_set_wide(wide)
            try:
                t1, p1, occ1 = _hh_intersect(host_harness, sd, o, dirs, maxt)
            finally:
                host_harness.hh_set_wide(0)
            assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32)) and np.array_equal(p0, p1) and np.array_equal(occ0, occ1)
    assert depths[None] > 15 and depths[14] <= 15, depths
