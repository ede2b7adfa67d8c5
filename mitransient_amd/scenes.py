"""Procedural test/benchmark scenes built from the supported plugin subset.

``staircase_like`` is a stand-in for the reference's ``examples/diff-transient/staircase`` scene
(BASELINE config 5: 262 661 triangles, diffuse / conductor / dielectric / twosided mix, one
rectangle light, max_depth 65, camera_unwarp): the real assets (774 OBJ files, 27 MB) live only in
the reference checkout, so the divergence stress is reproduced with generated geometry of the same
size class and material mix.  Every shape is a `cube` or `rectangle` plugin, so the scene goes
through the ordinary ``load_dict`` path.
"""
from __future__ import annotations

import math

import numpy as np
import os

from .transform import ScalarTransform4f as T


def staircase_like(n_steps=12, balusters=2, tiles=0, width=128, height=128, temporal_bins=256,
                   max_depth=65, spp=16):
    """A stair flight inside a room.  Triangle count = 12*(n_steps*(1 + 2*balusters) + 2) + 2*tiles^2*... ;
    ``tiles`` tessellates the floor into tiles x tiles small two-sided diffuse quads of alternating colour
    (cheap way to reach 10^5 triangles)."""
    d = {
        "type": "scene",
        "integrator": {"type": "transient_path", "max_depth": max_depth, "rr_depth": 5, "camera_unwarp": True},
        "sensor": {
            "type": "perspective", "fov": 55.0, "near_clip": 0.01, "far_clip": 100.0,
            "to_world": T().look_at(origin=[2.6, 2.4, 5.2], target=[0.0, 1.2, 0.0], up=[0, 1, 0]),
            "sampler": {"type": "independent", "sample_count": spp},
            "film": {"type": "transient_hdr_film", "width": width, "height": height, "rfilter": {"type": "box"},
                     "temporal_bins": temporal_bins, "start_opl": 0.0, "bin_width_opl": 40.0 / temporal_bins},
        },
        "wall": {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.75, 0.72, 0.68]}}},
        "wood": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.42, 0.26, 0.13]}},
        "tile_a": {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.8, 0.8, 0.8]}}},
        "tile_b": {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.15, 0.15, 0.2]}}},
        "steel": {"type": "conductor", "eta": [2.76, 2.54, 2.27], "k": [3.83, 3.43, 3.04]},
        "brass": {"type": "twosided", "bsdf": {"type": "conductor", "eta": [0.44, 0.53, 1.03], "k": [3.7, 2.77, 1.97]}},
        "glass": {"type": "dielectric", "int_ior": 1.5, "ext_ior": 1.0},
        "light": {"type": "rectangle", "to_world": T().translate([0.0, 3.95, 0.5]).rotate([1, 0, 0], 90).scale([0.9, 0.6, 1.0]),
                  "bsdf": {"type": "ref", "id": "wall"},
                  "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [30.0, 30.0, 30.0]}}},
        # room: 6 x 4 x 8, open towards the camera (+z)
        "floor": {"type": "rectangle", "to_world": T().rotate([1, 0, 0], -90).scale([3.0, 4.0, 1.0]), "bsdf": {"type": "ref", "id": "wall"}},
        "ceiling": {"type": "rectangle", "to_world": T().translate([0, 4, 0]).rotate([1, 0, 0], 90).scale([3.0, 4.0, 1.0]), "bsdf": {"type": "ref", "id": "wall"}},
        "back": {"type": "rectangle", "to_world": T().translate([0, 2, -4]).scale([3.0, 2.0, 1.0]), "bsdf": {"type": "ref", "id": "wall"}},
        "left": {"type": "rectangle", "to_world": T().translate([-3, 2, 0]).rotate([0, 1, 0], 90).scale([4.0, 2.0, 1.0]), "bsdf": {"type": "ref", "id": "wall"}},
        "right": {"type": "rectangle", "to_world": T().translate([3, 2, 0]).rotate([0, 1, 0], -90).scale([4.0, 2.0, 1.0]), "bsdf": {"type": "ref", "id": "wall"}},
        "glass-pane": {"type": "cube", "to_world": T().translate([-1.6, 1.2, 1.5]).scale([0.02, 1.2, 1.0]), "bsdf": {"type": "ref", "id": "glass"}},
        "mirror": {"type": "cube", "to_world": T().translate([2.9, 1.6, -1.0]).scale([0.03, 1.0, 1.4]), "bsdf": {"type": "ref", "id": "steel"}},
    }
    rise, run, sw = 3.0 / n_steps, 5.0 / n_steps, 0.9
    for i in range(n_steps):
        y, z = rise * (i + 0.5), 2.5 - run * (i + 0.5)
        d[f"step{i}"] = {"type": "cube", "to_world": T().translate([0.6, y, z]).scale([sw, rise * 0.5, run * 0.5]),
                         "bsdf": {"type": "ref", "id": "wood"}}
        for b in range(balusters):
            for side, x in ((0, 0.6 - sw + 0.05), (1, 0.6 + sw - 0.05)):
                zz = z + run * ((b + 0.5) / balusters - 0.5)
                d[f"bal{i}_{b}_{side}"] = {
                    "type": "cube", "to_world": T().translate([x, y + rise * 0.5 + 0.45, zz]).rotate([0, 1, 0], 45.0).scale([0.02, 0.45, 0.02]),
                    "bsdf": {"type": "ref", "id": "brass" if (i + b) % 2 else "steel"}}
    for side, x in ((0, 0.6 - sw + 0.05), (1, 0.6 + sw - 0.05)):
        ang = math.degrees(math.atan2(3.0, 5.0))
        d[f"rail{side}"] = {"type": "cube",
                            "to_world": T().translate([x, 1.5 + 0.95, 0.0]).rotate([1, 0, 0], ang).scale([0.03, 0.03, 2.95]),
                            "bsdf": {"type": "ref", "id": "brass"}}
    for a in range(tiles):
        for b in range(tiles):
            x = -3.0 + 6.0 * (a + 0.5) / tiles
            z = -4.0 + 8.0 * (b + 0.5) / tiles
            d[f"tile{a}_{b}"] = {"type": "rectangle",
                                 "to_world": T().translate([x, 0.002, z]).rotate([1, 0, 0], -90).scale([2.9 / tiles, 3.9 / tiles, 1.0]),
                                 "bsdf": {"type": "ref", "id": "tile_a" if (a + b) % 2 else "tile_b"}}
    return d


# ---------------------------------------------------------------------------------------------------------------
# Scenes of the reference's examples as DATA fixtures: the flattened triangles / material table / emitters of an XML
# scene (written by tests/golden/make_golden.py from the example assets) plus its sensor / film / integrator
# dictionaries.  They let the GPU box render the reference's own example scenes without the asset files.
def _jsonable(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out[k] = _jsonable(v)
        elif isinstance(v, T):
            out[k] = {"__matrix__": v.matrix.reshape(-1).tolist()}
        else:
            out[k] = v
    return out


def _unjson(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict) and "__matrix__" in v:
            out[k] = T(np.asarray(v["__matrix__"], dtype=np.float64).reshape(4, 4))
        elif isinstance(v, dict):
            out[k] = _unjson(v)
        else:
            out[k] = v
    return out


def save_fixture(scene, path, **meta):
    """a loaded (perspective-sensor) Scene -> .npz fixture: geometry tables + the sensor and integrator dictionaries"""
    from .scene import save_geometry
    sensor = scene.sensors()[0]
    integ = [v for v in scene.dict_.values() if isinstance(v, dict) and str(v.get("type", "")).startswith("transient")][0]
    save_geometry(scene.data(), path, sensor=_jsonable(sensor.dict_), integrator=_jsonable(integ), **meta)


def from_fixture(path, film=None, integrator=None, spp=None):
    """Scene from a fixture written by ``save_fixture``; ``film`` / ``integrator`` entries override the stored ones"""
    from . import mi
    from .scene import load_geometry
    g = load_geometry(path)
    sensor = _unjson(g["meta"]["sensor"])
    integ = _unjson(g["meta"]["integrator"])
    sensor["film"].update(film or {})
    integ.update(integrator or {})
    if spp is not None:
        sensor.setdefault("sampler", {"type": "independent"})["sample_count"] = spp
    return mi.Scene({"type": "scene", "integrator": integ, "sensor": sensor}, geometry=g)


DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")       # scene data shipped with the package


def staircase(width=720, height=1280, temporal_bins=400, spp=64, max_depth=65, materials="smooth", vertex_normals=False,
              textures=False, **integrator):
    """BASELINE config 5: the reference's examples/diff-transient/staircase/scene.xml ('The Wooden Staircase' by
    Wig42, CC-BY 3.0, Mitsuba version by B. Bitterli), 262,663 triangles.  ``materials="smooth"`` (the bench workload,
    SURVEY section 8d): flattened with approximate_materials="smooth" (roughplastic -> diffuse, roughconductor -> conductor,
    bitmap -> mean colour, bump map ignored); ``materials="rough"``: the GGX lobes of the scene file kept
    (approximate_materials=True: only textures and the bump map are approximated).  ``vertex_normals=True``: the meshes'
    own shading normals (the scene file sets face_normals on 157 of its 774 shapes only; 91.5 % of the triangles are
    smooth-shaded) instead of flat shading everywhere.  ``textures=True``: the nine bitmap textures of the scene's
    reflectances (fixture: box-downsampled to at most 256 pixels a side, 8-bit sRGB) instead of their mean colours —
    with materials="rough" and vertex_normals=True that is the scene as its file describes it (its one bumpmap wraps a BSDF
    that the shapes reference directly by id: it is never instantiated, in mitsuba neither)."""
    scene = from_fixture(os.path.join(DATA_DIR, "staircase_geometry.npz"),
                         film={"width": width, "height": height, "temporal_bins": temporal_bins},
                         integrator=dict(max_depth=max_depth, **integrator), spp=spp)
    if materials == "rough":
        import ctypes as C
        from . import _cabi
        z = np.load(os.path.join(DATA_DIR, "staircase_materials_rough.npz"))
        if int(z["layout"][0]) != C.sizeof(_cabi.mtr_material):
            raise ValueError("staircase_materials_rough.npz was written for another C-ABI; regenerate with tests/golden/make_golden.py")
        g = scene.geometry_
        nm = z["materials"].size // C.sizeof(_cabi.mtr_material)
        if nm != g["n_materials"]:
            raise ValueError("staircase_materials_rough.npz does not belong to staircase_geometry.npz")
        g["materials"] = (_cabi.mtr_material * nm).from_buffer_copy(z["materials"].tobytes())
    elif materials != "smooth":
        raise ValueError("materials: 'smooth' or 'rough'")
    if textures:
        from .scene import decode_texture_u8
        z = np.load(os.path.join(DATA_DIR, "staircase_textures.npz"))
        g = scene.geometry_
        if len(z["tex_of_mat"]) != g["n_materials"]:
            raise ValueError("staircase_textures.npz does not belong to staircase_geometry.npz")
        g["textures"] = [decode_texture_u8(z[f"tex{i}"]) for i in range(int(z["n"][0]))]
        for i, t in enumerate(z["tex_of_mat"]):
            g["materials"][i].albedo_texture = int(t)
    if vertex_normals:
        tn = np.load(os.path.join(DATA_DIR, "staircase_normals.npz"))["tri_normals"]
        if tn.shape != (scene.geometry_["tri_verts"].shape[0], 9):
            raise ValueError("staircase_normals.npz does not belong to staircase_geometry.npz")
        scene.geometry_["tri_normals"] = np.ascontiguousarray(tn, dtype=np.float32)
    return scene

def nlos_z(width=256, height=256, temporal_bins=4096, bin_width_opl=2.0 ** -11, start_opl=1.85, capture="confocal", spp=512,
           irradiance=1.0, film_extra=None, **integrator):
    """BASELINE config 4's scene (examples/transient-nlos/nlos_Z.xml, tests/integration/test_nlos.py:1-78 of the reference):
    the reference's Z.obj (6 triangles; data fixture ``data/nlos_Z_geometry.npz``, written back to an .obj so that it goes
    through the ``obj`` shape plugin) at z = 1, a 2 x 2 relay ``rectangle`` at the origin carrying a nlos_capture_meter,
    projector + sensor origin at (-0.5, 0, 0.25), fov 0.2, laser + hidden-geometry sampling on,
    account_first_and_last_bounces off, max_depth -1 / rr_depth 5."""
    import tempfile
    import mitransient_amd as mitr
    from . import mi
    tris = np.load(os.path.join(DATA_DIR, "nlos_Z_geometry.npz"))["tris"]
    lines = [f"v {float(v[0])!r} {float(v[1])!r} {float(v[2])!r}" for v in tris.reshape(-1, 3)]
    lines += [f"f {3 * i + 1} {3 * i + 2} {3 * i + 3}" for i in range(len(tris))]
    with tempfile.TemporaryDirectory() as tmp:
        obj = os.path.join(tmp, "Z.obj")
        with open(obj, "w") as fh:
            fh.write("\n".join(lines) + "\n")
        white = {"type": "diffuse", "reflectance": {"type": "rgb", "value": [1.0, 1.0, 1.0]}}
        relay = mi.load_dict({
            "type": "rectangle", "bsdf": white,
            "nlos_sensor": {"type": "nlos_capture_meter", "sampler": {"type": "independent", "sample_count": spp, "seed": 0},
                            "sensor_origin": [-0.5, 0.0, 0.25],
                            "film": dict({"type": "transient_hdr_film", "width": width, "height": height, "temporal_bins": temporal_bins,
                                          "bin_width_opl": bin_width_opl, "start_opl": start_opl, "rfilter": {"type": "box"}},
                                         **(film_extra or {}))}})
        laser = mi.load_dict({"type": "projector", "to_world": T().translate([-0.5, 0.0, 0.25]),
                              "irradiance": {"type": "rgb", "value": [irradiance] * 3}, "fov": 0.2})
        idict = {"type": "transient_nlos_path", "max_depth": -1, "rr_depth": 5, "nlos_laser_sampling": True,
                 "nlos_hidden_geometry_sampling": True, "account_first_and_last_bounces": False,
                 "capture_type": capture, "temporal_filter": "box"}
        idict.update(integrator)
        scene = mi.load_dict({"type": "scene", "integrator": idict, "laser": laser, "relay_wall": relay,
                              "Z": {"type": "obj", "filename": obj, "to_world": T().translate([0.0, 0.0, 1.0]), "bsdf": white}})
        scene.data()                      # flatten now: the mesh file goes away with the temporary directory
    mitr.nlos.focus_emitter_at_relay_wall_pixel((width / 2, height / 2), relay, laser)
    return scene
