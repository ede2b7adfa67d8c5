"""tools/size_sweep.py [spp] — where is the cliff between the two kernel organisations?  

Config 2's film (512 x 512 x 1024 bins, max_depth 8) over the Cornell box with its two boxes TESSELLATED: every face of
either box split into n x n cells of two triangles (n = 1: the `cube` shapes of cornell_box() themselves, 36 triangles),
written as OBJ meshes with the boxes' own to_world applied, so that the image, the path lengths and the ray counts stay
those of config 2 while the scene grows 36 -> 260k triangles.  For every size: the organisation mtr_render_plan picks, its
time, and the other organisation forced (amd_mode) beside it.  The staircase (262,663 triangles) rides along at
1 / 64 ... 1 / 1 of its triangles (every k-th triangle kept: a workload with holes, not a picture).
"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mitransient_amd as mitr
import mitransient_amd.mi as mi

mi.set_variant("llvm_ad_rgb")
SPP = int(sys.argv[1]) if __name__ == "__main__" and len(sys.argv) > 1 else 1024
TMP = tempfile.mkdtemp(prefix="mtr_sweep_")


def box_obj(path, to_world, n):
    """the cube [-1,1]^3, every face n x n cells of two triangles (outward winding), vertices mapped by to_world"""
    M = np.asarray(to_world.matrix if hasattr(to_world, "matrix") else to_world, np.float64).reshape(4, 4)
    verts, faces = [], []
    t = np.linspace(-1.0, 1.0, n + 1)
    for axis in range(3):
        for sign in (-1.0, 1.0):
            u, v = (axis + 1) % 3, (axis + 2) % 3
            base = len(verts)
            for a in t:
                for b in t:
                    p = np.zeros(3)
                    p[axis] = sign; p[u] = a; p[v] = b
                    verts.append(M[:3, :3] @ p + M[:3, 3])
            for i in range(n):
                for j in range(n):
                    v00 = base + i * (n + 1) + j; v01 = v00 + 1; v10 = v00 + n + 1; v11 = v10 + 1
                    quad = [(v00, v10, v11), (v00, v11, v01)]
                    for tri in quad:
                        faces.append(tri if sign > 0 else tri[::-1])
    with open(path, "w") as fh:
        for p in verts:
            fh.write("v %.9g %.9g %.9g\n" % tuple(p))
        for f in faces:
            fh.write("f %d %d %d\n" % (f[0] + 1, f[1] + 1, f[2] + 1))


def cornell(n, width=512, height=512, bins=1024):
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=width, height=height, temporal_bins=bins, start_opl=3.5, bin_width_opl=6.0 / bins)
    d["integrator"]["max_depth"] = 8
    if n > 1:
        for name in ("small-box", "large-box"):
            fn = os.path.join(TMP, f"{name}_{n}.obj")
            box_obj(fn, d[name]["to_world"], n)
            d[name] = {"type": "obj", "filename": fn, "face_normals": True, "bsdf": d[name]["bsdf"]}
    return d


def timed(scene, mode):
    import torch
    integ = scene.integrator()
    integ.amd_mode = mode
    integ.collect_stats = True
    try:
        for _ in range(2):
            integ.render(scene, spp=SPP)
        best = None
        for _ in range(3):
            integ.render(scene, spp=SPP)
            torch.cuda.synchronize()
            ms = integ.last_times["total_ms"]
            best = ms if best is None else min(best, ms)
    except Exception as e:      # noqa: BLE001
        return None, None, str(e)[:80]
    c, tm = integ.last_counters, integ.last_times
    return best, (c["rays_closest"] + c["rays_shadow"]) / best / 1e6, ("wavefront" if tm["scatter_launches"] else "fused")


def report(label, scene):
    sd = scene.data()
    ntri = sd.tri_verts.shape[0]
    row = [f"{label:28s} {ntri:8d} tris"]
    res = {}
    for mode in ("auto", "fused", "wavefront"):
        ms, gray, what = timed(scene, mode)
        res[mode] = (ms, gray, what)
        row.append(f"{mode}: " + (f"{ms:8.2f} ms {gray:6.2f} Gray/s ({what})" if ms else f"-- ({what})"))
    print("  |  ".join(row), flush=True)
    return ntri, res


def main():
    global SPP
    print(f"# config-2 film (512x512x1024 bins), {SPP} spp, max_depth 8; per size: auto / forced fused / forced wavefront")
    sweep()


def sweep():
    global SPP
    import torch
    for n in (1, 2, 6, 13, 26, 52, 104):
        t0 = time.time()
        scene = mi.load_dict(cornell(n))
        report(f"cornell, boxes {n}x{n} per face", scene)
        del scene
        torch.cuda.empty_cache()

    from mitransient_amd.scenes import staircase
    for k in (64, 16, 4, 1):
        sc = staircase(width=512, height=512, temporal_bins=2048, max_depth=65, materials="smooth")
        film = sc.sensors()[0].film()
        film.start_opl, film.bin_width_opl = 0.0, 40.0 / 2048
        if k > 1:
            # every k-th triangle of every shape (at least one; emitting shapes and rectangles stay whole), ranges rebuilt
            g = sc.geometry_
            nt = g["tri_verts"].shape[0]
            em = np.asarray(g["tri_emitter"])
            keep = np.zeros(nt, bool)
            ranges = []
            for i in range(g["n_shapes"]):
                sh = g["shapes"][i]
                f, n = int(sh.first_tri), int(sh.n_tris)
                whole = bool(sh.is_rectangle) or bool((em[f:f + n] >= 0).any())
                m = np.ones(n, bool) if whole else (np.arange(n) % k == 0)
                keep[f:f + n] = m
                ranges.append((i, f, int(m.sum())))
            new_index = np.cumsum(keep) - 1
            for i, f, n in ranges:
                g["shapes"][i].first_tri = int(new_index[f]) if n else 0
                g["shapes"][i].n_tris = n
            for e in range(g["n_emitters"]):
                E = g["emitters"][e]
                if E.is_mesh:
                    E.first_tri = int(new_index[int(E.first_tri)])
            for key in ("tri_verts", "tri_material", "tri_emitter", "tri_normals", "tri_uv"):
                if g.get(key) is not None:
                    g[key] = np.ascontiguousarray(np.asarray(g[key])[keep])
        SPP_SAVE = SPP
        SPP = min(SPP, 256)
        report(f"staircase, 1/{k} of the triangles", sc)
        SPP = SPP_SAVE
        del sc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
