"""Generates the committed fixtures under tests/golden/.

Nothing here imports the reference: mitransient needs Mitsuba 3, which is not installable in the
authoring container (no network, no wheel), and its only test holds no numeric vectors
(tests/integration/test_nlos.py:117-118).  The fixtures are therefore
  * pcg32_kat.json       — the published known-answer stream of the PCG32 reference implementation
                           (pcg32-demo, seed 42 / stream 54), typed in from pcg-random.org;
  * bin_mapping_kat.json — the f32 bin mapping of transient_hdr_film.py:263-265 /
                           transient_image_block.py:131-146, evaluated with numpy float32
                           (independent of the C oracle and of the HIP code);
  * cornell_c1_oracle.npz — a regression snapshot of the oracle on BASELINE config 1
                           (per-pixel and per-bin marginals + checksum), NOT a reference output.
  * cbox_diffuse_scene.npz, cbox_mirror_scene.npz, nlos_Z_geometry.npz, ../../mitransient_amd/data/staircase_geometry.npz —
                           DATA of the reference's example scenes (examples/transient/cornell-box/*.xml,
                           examples/transient-nlos/Z.obj, examples/diff-transient/staircase/scene.xml = BASELINE
                           config 5): the triangles in world space, material / emitter tables and the sensor /
                           film / integrator dictionaries, flattened by mitransient_amd's XML loader from the
                           asset files under /root/reference (only written when that directory exists).  No
                           expected outputs: Mitsuba cannot run here.  Staircase: 'The Wooden Staircase' by Wig42,
                           CC-BY 3.0, Mitsuba version from benedikt-bitterli.me/resources.
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def pcg32_kat():
    return {"initstate": 42, "initseq": 54,
            "u32": ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]}


def bin_mapping_kat():
    rows = []
    cases = [(3.5, 0.02, 300), (3.5, 6.0 / 1024, 1024), (3.5, 6.0 / 64, 64), (0.0, 0.003, 2048), (1.85, 0.006, 300)]
    for start, width, T in cases:
        s32, w32 = np.float32(start), np.float32(width)
        ds = [start, start + 0.995 * width, start + width, start + 2 * width, start + (T - 0.005) * width,
              start + T * width, start - 0.005 * width, start + 0.5 * T * width, np.inf, -np.inf, np.nan]
        if (start, width, T) == (3.5, 0.02, 300):
            ds += [3.5199, 3.52, 3.54, 9.4999, 9.5, 3.4999]
        for d in ds:
            d32 = np.float32(d)
            with np.errstate(invalid="ignore", over="ignore"):
                pos = (d32 - s32) / w32                       # f32 sub, f32 div
                ok = bool(pos >= 0) and bool(pos < np.float32(T))
                b = int(np.floor(pos)) if ok else -1
            rows.append({"start": float(s32), "width": float(w32), "T": T,
                         "d_bits": int(np.float32(d32).view(np.uint32)), "bin": b})
    return rows


def cornell_c1():
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    from oracle import oracle
    mi.set_variant("llvm_ad_rgb")
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=64, height=64, temporal_bins=64, bin_width_opl=6.0 / 64)
    scene = mi.load_dict(d)
    sd = scene.data()
    params = scene.integrator().render_params(scene.sensors()[0].film(), 0, 16)
    t4, s4, cnt = oracle.render(sd, params, n_threads=1)
    t3, s3 = oracle.develop(sd.film, t4, s4)
    return {"per_bin": t3.sum(axis=(0, 1)).astype(np.float64), "per_pixel": t3.sum(axis=2).astype(np.float32),
            "steady": s3, "counters": np.array([cnt[k] for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces")], np.int64),
            "nonzero_cells": np.int64(np.count_nonzero(t3))}


def example_scenes():
    ref = "/root/reference/examples"
    if not os.path.isdir(ref):
        print("reference examples not present: scene fixtures left as they are")
        return
    import mitransient_amd.mi as mi
    from mitransient_amd.scenes import save_fixture
    from mitransient_amd.scene import load_obj
    mi.set_variant("llvm_ad_rgb")
    for name in ("cbox_diffuse", "cbox_mirror"):
        sc = mi.load_file(f"{ref}/transient/cornell-box/{name}.xml")
        save_fixture(sc, os.path.join(HERE, f"{name}_scene.npz"), source=f"examples/transient/cornell-box/{name}.xml")
    sc = mi.load_file(f"{ref}/diff-transient/staircase/scene.xml", approximate_materials="smooth")
    # the geometry fixture is the SURVEY section-8d workload: flat shading (the vertex normals travel in staircase_normals.npz)
    sc.data().tri_normals = None
    save_fixture(sc, os.path.join(ROOT, "mitransient_amd", "data", "staircase_geometry.npz"), source="examples/diff-transient/staircase/scene.xml",
                 approximate_materials="smooth")
    # the same scene with its GGX lobes kept (roughplastic / roughconductor; textures -> mean colour, bump map ignored): only
    # the material table differs, so it travels as a small side file next to the geometry
    import ctypes as C
    from mitransient_amd import _cabi
    sr = mi.load_file(f"{ref}/diff-transient/staircase/scene.xml", approximate_materials=True).data()
    assert sr.n_materials == sc.data().n_materials and np.array_equal(sr.tri_material, sc.data().tri_material)
    np.savez_compressed(os.path.join(ROOT, "mitransient_amd", "data", "staircase_materials_rough.npz"),
                        materials=np.frombuffer(bytes(sr.materials), dtype=np.uint8)[:sr.n_materials * C.sizeof(_cabi.mtr_material)],
                        layout=np.asarray([C.sizeof(_cabi.mtr_material)]))
    # ... and the meshes' vertex normals (face_normals is set on 157 of the 774 shapes only), same triangle order
    assert np.array_equal(sr.tri_verts, sc.data().tri_verts)
    np.savez_compressed(os.path.join(ROOT, "mitransient_amd", "data", "staircase_normals.npz"), tri_normals=sr.tri_normals)
    # ... and its nine bitmap textures, box-downsampled to at most 256 pixels a side (8-bit sRGB), with the material -> texture table
    from PIL import Image
    from mitransient_amd import scene as S
    names, orig = [], S.load_bitmap_texture
    S.load_bitmap_texture = lambda path, raw=False, max_size=None: (names.append(path), orig(path, raw, max_size))[1]
    sf = mi.load_file(f"{ref}/diff-transient/staircase/scene.xml").data()
    S.load_bitmap_texture = orig
    arrs = {}
    for i, path in enumerate(names):
        with Image.open(path) as im:
            im = im.convert("RGB")
            k = 256 / float(max(im.size))
            if k < 1:
                im = im.resize((max(1, round(im.size[0] * k)), max(1, round(im.size[1] * k))), Image.BOX)
            arrs[f"tex{i}"] = np.asarray(im, dtype=np.uint8)
    np.savez_compressed(os.path.join(ROOT, "mitransient_amd", "data", "staircase_textures.npz"), n=np.asarray([len(names)]),
                        names=np.asarray([os.path.basename(n) for n in names]),
                        tex_of_mat=np.array([sf.materials[i].albedo_texture for i in range(sf.n_materials)], np.uint32), **arrs)
    np.savez_compressed(os.path.join(HERE, "nlos_Z_geometry.npz"), tris=load_obj(f"{ref}/transient-nlos/Z.obj").astype(np.float32))


if __name__ == "__main__":
    example_scenes()
    json.dump(pcg32_kat(), open(os.path.join(HERE, "pcg32_kat.json"), "w"), indent=1)
    json.dump(bin_mapping_kat(), open(os.path.join(HERE, "bin_mapping_kat.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "cornell_c1_oracle.npz"), **cornell_c1())
    print("golden fixtures written")
