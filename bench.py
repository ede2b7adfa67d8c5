#!/usr/bin/env python
"""bench.py — headline benchmark: Cornell box 512x512, 1024 time bins, 1024 spp per GPU
(BASELINE.json configs[1]; with N GPUs: N*1024 spp sharded by samples + one RCCL film reduction,
configs[2] at N=8).

    python bench.py --gpus N --steps K --warmup W

A "step" is one complete transient render on every rank: film clear, the path kernel(s) over
rank's sample slice of all 512^2 pixels, (N>1) reduce-scatter of the raw (H,W,T,4) film over
RCCL + all-gather of the developed tensor, develop.  Scene, BVH and film live in HBM before the
timed region starts.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SPLAT_BYTES = 24.0         # algorithmic bytes per issued time-bin contribution
# f32 vector (VALU) roof: 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6e12 lane-operations/s (one v_fma_f32 of a wave64 occupies its
# SIMD for 2 cycles; x2 flops per fma = the 157.3 TFLOP/s vector peak of MI355X_MICROARCH.md)
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12


def source_hash():
    """sha256 (first 16 hex digits) over the sources the HIP library is built from.  tools/profile.sh stores it in
    profiles/traffic.json next to the counters it collects; a roofline that divides committed counters by a live time is
    only printed when the two agree (otherwise the kernel was edited after it was last profiled)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "mitransient_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "mitransient_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "mitransient_amd", "csrc", "*.cpp")) +
                   [os.path.join(ROOT, "mitransient_amd", "csrc", "Makefile"), os.path.join(ROOT, "include", "mitransient_amd.h")])
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (the form the driver uses for N = 1): start the N ranks here, exactly
    as the driver would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), and
    pass rank 0's JSON line through."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


MATERIALS = "smooth"       # --materials (staircase only): "smooth" = the SURVEY section-8d mapping (default) | "rough" = GGX lobes kept
SCENE = "cornell"          # --scene: "cornell" (BASELINE configs[1], the default) | "staircase" (configs[4] geometry) | "nlos" (configs[3]: one GPU's share)


def build_scene(width, height, bins, max_depth=8, mode=None):
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
    if SCENE == "nlos":          # BASELINE configs[3]: NLOS confocal Z scene, 256^2 x 4096 bins; --spp = one GPU's share of the 4096
        from mitransient_amd.scenes import nlos_z
        kw = {"amd_mode": mode} if mode else {}
        return nlos_z(width=width, height=height, temporal_bins=bins, **kw)
    if SCENE == "staircase":
        from mitransient_amd.scenes import staircase
        kw = {"amd_mode": mode} if mode else {}
        sc = staircase(width=width, height=height, temporal_bins=bins, max_depth=65, materials=MATERIALS,
                       vertex_normals=(MATERIALS == "rough"), textures=(MATERIALS == "rough"), **kw)
        film = sc.sensors()[0].film()
        film.start_opl, film.bin_width_opl = 0.0, 40.0 / bins     # the reference's 0..40 window (400 x 0.1), SURVEY §8d
        return sc
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=width, height=height, temporal_bins=bins, start_opl=3.5,
                               bin_width_opl=6.0 / bins)
    d["integrator"]["max_depth"] = max_depth
    if os.environ.get("MTR_BENCH_CLASSIC_FILM"):             # experiments: clear + accumulate + develop instead of developed rows
        d["integrator"]["amd_direct_develop"] = False
    if os.environ.get("MTR_BENCH_DETERMINISTIC"):            # experiments: fixed-point (order-independent) LDS rows in k_fused
        d["integrator"]["amd_deterministic"] = True
    if mode:
        d["integrator"]["amd_mode"] = mode
    return mi.load_dict(d)


def cpu_baseline(width, height, bins, spp_total, target_s=15.0):
    """Times the CPU oracle (the build's C restatement, OpenMP over all host cores) on a bounded
    sample of the SAME workload: all width x height pixels, the first k of spp_total samples."""
    from oracle import oracle
    scene = build_scene(width, height, bins)
    sd = scene.data()
    integ = scene.integrator()
    film = scene.sensors()[0].film()
    cores = oracle.num_threads()
    bufs = oracle.alloc_film(sd.film, prefault=True)      # film allocation/page faults are NOT timed
    # warm-up (thread pool, caches), then a 2-sample calibration of both intersection modes of the oracle
    # (brute force over the 36 triangles / its own BVH); the faster one runs the bounded sample
    oracle.render(sd, integ.render_params(film, 0, spp_total, 0, 1), use_bvh=True, out=bufs)
    best = None
    # (brute force is only a candidate for tiny scenes: it is O(triangles) per ray)
    for use_bvh in ((True, False) if sd.tri_verts.shape[0] <= 256 else (True,)):
        t0 = time.perf_counter()
        oracle.render(sd, integ.render_params(film, 0, spp_total, 1, 3), use_bvh=use_bvh, out=bufs)
        dt2 = max(time.perf_counter() - t0, 1e-3) / 2.0
        if best is None or dt2 < best[0]:
            best = (dt2, use_bvh)
    dt_per_spp, use_bvh = best
    k = int(max(1, min(spp_total - 3, target_s / dt_per_spp)))
    p = integ.render_params(film, 0, spp_total, 3, 3 + k)
    t0 = time.perf_counter()
    bufs[0].fill(0.0)                     # TransientImageBlock.clear is part of a render
    bufs[1].fill(0.0)
    t_clear = time.perf_counter() - t0
    _, _, c = oracle.render(sd, p, use_bvh=use_bvh, out=bufs)
    dt = time.perf_counter() - t0
    rays = c["rays_closest"] + c["rays_shadow"]
    return {"value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
            "time_bins_per_s": c["splats_issued"] / dt,
            "sample": f"{width}x{height} px, {bins} bins, samples 3..{2 + k} of {spp_total} per pixel "
                      f"({c['paths']} paths in {dt:.1f} s incl. {t_clear:.1f} s film clear; oracle {'BVH' if use_bvh else 'brute-force'} "
                      f"intersection, OpenMP {cores} threads, film pre-faulted)",
            "note": "the build's own scalar C restatement (libm, one lane at a time, no SIMD packets): NOT Mitsuba's Embree / "
                    "Dr.Jit-LLVM path, which is not installable here; the GPU/CPU ratio says little about kernel quality"}


_TRAFFIC = None


def traffic_file():
    global _TRAFFIC
    if _TRAFFIC is None:
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
                _TRAFFIC = json.load(fh)
        except Exception:
            _TRAFFIC = {}
    return _TRAFFIC


def profile_is_current(section=None):
    """the committed counters were collected from THIS source tree (tools/profile.sh records bench.py's source_hash())"""
    t = traffic_file()
    t = t.get(section, {}) if section else t
    return bool(t) and t.get("source_hash") == source_hash()


def pmc_from_profiles(kernel, section=None):
    """Counters per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/profile.sh: separate --pmc passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    ``section``: None = the config-2 passes (top level), "staircase" = the config-5 passes.
    {} when no profile of this kernel is committed: bench.py itself cannot collect PMC counters."""
    t = traffic_file()
    if section:
        t = t.get(section, {})
    v = t.get(kernel, {})
    return v if isinstance(v, dict) else {}


def traffic_from_profiles(kernel, section=None):
    """HBM bytes per launch from the PMC passes — None when they were not collected from this source tree"""
    if not profile_is_current(section):
        return None
    return pmc_from_profiles(kernel, section).get("hbm_bytes_per_launch")


def valu_roofline(kernel, avg_launch_ms, workload_matches, launches_per_render=1.0, section=None):
    """The path kernel is bound by VALU issue + SIMT divergence, not by HBM: achieved = VALU lane-operations
    per launch (SQ_INSTS_VALU x active lanes per instruction, from the committed PMC pass of the SAME workload) / the
    launch time measured live with HIP events; peak = the f32 vector roof."""
    c = pmc_from_profiles(kernel, section)
    if not workload_matches or "valu_insts_per_launch" not in c or avg_launch_ms <= 0 or not profile_is_current(section):
        return None
    # the profile is of ONE launch per render; a multi-GPU step issues the same per-rank work as `launches_per_render` band launches
    insts = c["valu_insts_per_launch"] / launches_per_render
    lane_ops = insts * c["valu_lanes_per_inst"]
    achieved = lane_ops / (avg_launch_ms * 1e-3) / 1e12
    return {"kernel": kernel, "bound": "valu", "achieved": achieved, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s",
            "frac": achieved / VALU_PEAK_TLANEOPS,
            "valu_issue_frac": insts * 2.0 / (1024 * avg_launch_ms * 1e-3 * 2.4e9),
            "lanes_per_valu_inst": c["valu_lanes_per_inst"], "valu_insts_per_launch": insts,
            "traffic": c.get("hbm_bytes_per_launch"), "avg_launch_ms": avg_launch_ms,
            "traffic_note": "2 x FETCH_SIZE + WRITE_SIZE of the PMC passes",
            "profile_source_hash": source_hash(),
            "note": "frac = executed VALU instructions x active lanes / time: a utilisation figure — it also rises when a kernel "
                    "executes MORE instructions, so read it beside avg_launch_ms and valu_insts_per_launch",
            "source": "profiles/traffic.json (rocprofv3 --pmc SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU; "
                      + str(c.get("profile", "")) + "; same source hash as the library that ran) over the launch time measured live (HIP events)"}


WF_STATE_BYTES_PER_BOUNCE = 216.0      # un-fused wavefront state per live path and bounce
WF_SHADOW_RAY_BYTES = 48.0             # ... + 48 B per shadow ray


def wavefront_rooflines(counters, times, n_renders, ms_per_render, profile_ok, section="staircase"):
    """Roofline blocks of a render in the wavefront organisation (config 5): the dominant kernel k_wf_trace (VALU issue x active
    lanes: instructions of the committed PMC pass of the SAME workload and sources over its own live HIP-event time), the
    HBM-bound k_wf_shade (SURVEY section 8d's algorithmic bytes over its live time) and the time-bin
    scatter-add k_wf_scatter (24 B per contribution over its live time).  counters / times: sums over ``n_renders`` renders."""
    out = {}
    trace_ms = times.get("wf_trace_ms", 0.0) / n_renders
    shade_ms = times.get("wf_shade_ms", 0.0) / n_renders
    scat_ms = times.get("scatter_ms", 0.0) / n_renders
    c5 = pmc_from_profiles("k_wf_trace", section)
    if trace_ms > 0:
        blk = {"kernel": "k_wf_trace", "bound": "valu", "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s", "kernel_ms_per_render": trace_ms,
               "launches_per_render": times.get("wf_trace_kernel_launches", 0) / n_renders, "share_of_render": trace_ms / ms_per_render}
        if profile_ok and c5.get("valu_insts_per_render"):
            lane_ops = c5["valu_insts_per_render"] * c5["valu_lanes_per_inst"]
            ach = lane_ops / (trace_ms * 1e-3) / 1e12
            blk.update({"achieved": ach, "frac": ach / VALU_PEAK_TLANEOPS,
                        "valu_issue_frac": c5["valu_insts_per_render"] * 2.0 / (1024 * trace_ms * 1e-3 * 2.4e9),
                        "lanes_per_valu_inst": c5["valu_lanes_per_inst"], "valu_insts_per_render": c5["valu_insts_per_render"],
                        "traffic": c5.get("hbm_bytes_per_render"), "wait_any_frac": c5.get("wait_any_frac"),
                        "profile_source_hash": source_hash(),
                        "note": "k_wf_trace waits on divergent 16-byte loads from L1/L2 (wait_any_frac), so its issue fraction is the "
                                "honest utilisation figure; HBM traffic is a small fraction of the roof (scene resident in L2)",
                        "source": "profiles/traffic.json[" + section + "] (rocprofv3 --pmc, same sources) over k_wf_trace's own HIP-event time, live"})
            if c5.get("vmem_insts_per_render"):
                # the kernel's divergent loads as rates.  Vector-memory wave-instructions x active lanes = lane addresses (a node step
                # costs a lane 4 loads, a visited child 1 — its reference —, a triangle pair 5: tools/walk_stats.py); the lines they miss
                # in L1 go to L2 one request each; the round-1 microbenchmark of this access pattern
                # (profiles/r01_divergent_load_microbench.txt) reaches 0.10 - 0.16 L2 lines per clock and CU on a 32 - 8 MB set (the
                # staircase: 5.5 MB of nodes + 22 MB of triangle pairs).  NOT a roof the kernel leans on: fetching a quad's four nodes
                # with quad-coalesced loads (a quarter of the node steps' lane addresses, transposed through LDS) made it 23 % slower
                # (HISTORY.md: "quad fetch") — what it waits for is the latency of DEPENDENT loads (wait_any_frac)
                lane_loads = c5["vmem_insts_per_render"] * c5["valu_lanes_per_inst"]
                clk_cu = trace_ms * 1e-3 * 2.4e9 * 256.0
                blk["divergent_loads"] = {"vmem_insts_per_render": c5["vmem_insts_per_render"], "lane_loads_per_render": lane_loads,
                                          "lane_loads_per_clk_per_cu": lane_loads / clk_cu,
                                          "l2_requests_per_render": c5.get("l2_requests_per_render"),
                                          "l2_requests_per_clk_per_cu": (c5["l2_requests_per_render"] / clk_cu) if c5.get("l2_requests_per_render") else None,
                                          "microbench_l2_lines_per_clk_per_cu": [0.10, 0.16],
                                          "note": "lane addresses = vector-memory wave-instructions x active lanes (the VALU average); L2 requests = "
                                                  "TCC_HIT + TCC_MISS of the PMC pass; both over the kernel's live time, 256 CUs at 2.4 GHz; "
                                                  "diagnostic rates, not roofs: the kernel waits on dependent-load latency"}
        else:
            blk.update({"achieved": None, "frac": None, "roofline_stale": True,
                        "note": "instruction counts need the PMC pass of THESE sources (tools/profile_all.sh); the live kernel time stands"})
        out["roofline_valu"] = blk
        # the contract form for the same kernel: the bytes a trace launch must move (a 32-byte ray read and a 16-byte
        # hit written per closest-hit ray, 32 + 1 per shadow ray) over its live time; `traffic` = L2-miss bytes of the PMC pass
        alg = (48.0 * counters["rays_closest"] + 33.0 * counters["rays_shadow"]) / n_renders
        ach_h = alg / (trace_ms * 1e-3) / 1e9
        out["roofline"] = {"kernel": "k_wf_trace", "bound": "hbm", "achieved": ach_h, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach_h / HBM_PEAK_GBS, "traffic": c5.get("hbm_bytes_per_render") if profile_ok else None,
                           "kernel_ms_per_render": trace_ms, "algorithmic_bytes_per_render": alg,
                           "launches_per_render": times.get("wf_trace_kernel_launches", 0) / n_renders,
                           "note": "rays and hits stream through once; the kernel is bound by instructions per ray at ~37 of 64 lanes and by "
                                   "L2 misses on the scene (traffic >> algorithmic bytes: the 10 MB of triangle pairs do not fit an L2 slice), "
                                   "see roofline_valu"}
    if shade_ms > 0:
        alg = WF_STATE_BYTES_PER_BOUNCE * counters["bounces"] / n_renders + WF_SHADOW_RAY_BYTES * counters["rays_shadow"] / n_renders
        ach = alg / (shade_ms * 1e-3) / 1e9
        traffic = None
        if profile_ok:
            traffic = pmc_from_profiles("k_wf_shade", section).get("hbm_bytes_per_render")
        out["roofline_shade"] = {"kernel": "k_wf_shade", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms_per_render": shade_ms,
                                 "algorithmic_bytes_per_render": alg, "share_of_render": shade_ms / ms_per_render,
                                 "note": "algorithmic bytes = SURVEY section 8d: 216 B per live path and bounce + 48 B per shadow ray; "
                                         "traffic = 2 x FETCH_SIZE + WRITE_SIZE of the kernel per render (PMC passes)"}
    if scat_ms > 0:
        alg = SPLAT_BYTES * counters["splats_issued"] / n_renders
        ach = alg / (scat_ms * 1e-3) / 1e9
        out["scatter_add"] = {"kernel": "k_wf_scatter", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                              "traffic": (pmc_from_profiles("k_wf_scatter", section).get("hbm_bytes_per_render") if profile_ok else None),
                              "kernel_ms_per_render": scat_ms, "algorithmic_bytes_per_render": alg,
                              "launches_per_render": times.get("scatter_launches", 0) / n_renders}
    return out


def extra_config_legs(spp4, spp5):
    """Side legs of the driver-run line (rank 0, N = 1), each with its own HIP-event times, counters and roofline blocks:
    BASELINE configs[3] — the NLOS confocal Z scene, ONE GPU's share (512 of the 4096 spp; the config is stated for 8 GPUs) —
    and configs[4] — the staircase AT ITS STATED SIZE (512^2 x 2048 bins x 2048 spp, max_depth 65).  The staircase's wavefront
    workspace is sized by the library from the free device memory (up to 82 GB for a 2^28-slot tile)."""
    import torch
    import mitransient_amd.mi as mi
    from mitransient_amd.scenes import nlos_z, staircase
    out = {}

    def run(scene, spp, reps):
        integ = scene.integrator()
        integ.collect_stats = True
        integ.render(scene, spp=spp, seed=0)                   # the first render pays the workspace allocation: not counted
        torch.cuda.synchronize()
        acc_c, acc_t, wall = None, None, 0.0
        for _ in range(reps):
            t0 = time.perf_counter()
            integ.render(scene, spp=spp, seed=0)
            torch.cuda.synchronize()
            wall += time.perf_counter() - t0
            c, tm = integ.total_counters, integ.total_times
            acc_c = dict(c) if acc_c is None else {k: acc_c[k] + c[k] for k in acc_c}
            acc_t = dict(tm) if acc_t is None else {k: acc_t[k] + tm[k] for k in acc_t}
        rays = acc_c["rays_closest"] + acc_c["rays_shadow"]
        ms = acc_t["total_ms"] / reps
        r = {"ms": ms, "wall_ms": wall / reps * 1e3, "renders_timed": reps, "Mray_per_s": rays / acc_t["total_ms"] / 1e3,
             "time_bins_per_s": acc_c["splats_issued"] / acc_t["total_ms"] * 1e3,
             "counters": {k: acc_c[k] / reps for k in ("paths", "rays_closest", "rays_shadow", "splats_issued", "bounces")},
             "mode": "wavefront" if acc_t["scatter_launches"] else "fused"}
        return r, acc_c, acc_t
    mi.set_variant("llvm_ad_rgb")
    t0 = time.perf_counter()
    sc4 = nlos_z(width=256, height=256, temporal_bins=4096, spp=spp4)
    r4, c4, t4 = run(sc4, spp4, 3)
    r4["workload"] = (f"NLOS confocal Z scene (reference Z.obj), 256x256 px, 4096 time bins (start_opl 1.85, width 2^-11), {spp4} of 4096 spp "
                      f"(one GPU's share of 8), max_depth -1, rr_depth 5")
    if r4["mode"] == "fused":
        avg = t4["trace_ms"] / max(1, t4["trace_launches"])
        v = valu_roofline("k_fused", avg, spp4 == 512, 1.0, section="nlos")
        alg = SPLAT_BYTES * c4["splats_issued"] / max(1, t4["trace_launches"])
        hbm = {"kernel": "k_fused<NLOS>", "bound": "hbm", "achieved": alg / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": avg, "algorithmic_bytes_per_launch": alg,
               "traffic": traffic_from_profiles("k_fused", "nlos") if spp4 == 512 else None}
        r4["roofline"] = hbm
        if v is not None:
            r4["roofline_valu"] = v
        else:
            r4["roofline_stale"] = not profile_is_current("nlos")
    out["config4_share"] = r4
    del sc4
    torch.cuda.empty_cache()
    sc5 = staircase(width=512, height=512, temporal_bins=2048, max_depth=65, materials="smooth")
    film = sc5.sensors()[0].film()
    film.start_opl, film.bin_width_opl = 0.0, 40.0 / 2048
    r5, c5, t5 = run(sc5, spp5, 2)
    r5["workload"] = (f"staircase scene.xml geometry (262,663 triangles, approximate materials), 512x512 px, 2048 time bins (start_opl 0, "
                      f"width 40/2048), {spp5} of 2048 spp, max_depth 65, camera_unwarp")
    r5.update(wavefront_rooflines(c5, t5, 2, r5["ms"], spp5 == 2048 and profile_is_current("staircase")))
    out["config5" if spp5 == 2048 else "config5_reduced"] = r5
    del sc5
    torch.cuda.empty_cache()
    # ... and the scene AS ITS FILE DESCRIBES IT — what a user of the reference's scene.xml gets: GGX lobes (6 roughplastic, 2
    # roughconductor), interpolated vertex normals, the nine bitmap textures (256-px fixtures) — instead of the section-8d mapping
    sc5r = staircase(width=512, height=512, temporal_bins=2048, max_depth=65, materials="rough", vertex_normals=True, textures=True)
    film = sc5r.sensors()[0].film()
    film.start_opl, film.bin_width_opl = 0.0, 40.0 / 2048
    r5r, c5r, t5r = run(sc5r, spp5, 2)
    r5r["workload"] = (f"staircase scene.xml as written (262,663 triangles, GGX roughplastic / roughconductor lobes, vertex normals, bitmap "
                       f"textures), 512x512 px, 2048 time bins (start_opl 0, width 40/2048), {spp5} of 2048 spp, max_depth 65, camera_unwarp")
    r5r.update(wavefront_rooflines(c5r, t5r, 2, r5r["ms"], spp5 == 2048 and profile_is_current("staircase_rough"), section="staircase_rough"))
    out["config5_rough" if spp5 == 2048 else "config5_rough_reduced"] = r5r
    del sc5r
    torch.cuda.empty_cache()
    out["wall_s"] = time.perf_counter() - t0
    return out


def splat_microbench(log2_s=30):
    """SURVEY section 8d's micro-benchmark of the scatter-add ALONE, as stated: S = 2^30 synthetic contributions into a
    512 x 512 x 1024-bin film through the film's own `add_transient_data` (-> mtr_splat_add, variant 1), pixel ~ U[0, 2^18)
    either pixel-major sorted (the order the reference's own call produces: lanes are pixel-major) or in arbitrary order,
    bin ~ clipped Normal(400, 120), rgb ~ U(0, 1), seed 1234.  frac = 24 B x S / time / 8 TB/s, time = the kernels' HIP events
    (best of 3).  Falls back to 2^28 when 2^30 does not fit beside what the process already holds."""
    import torch
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    from mitransient_amd.scene import Properties
    mi.set_variant("llvm_ad_rgb")
    W = H = 512
    T = 1024

    def run(log2):
        S = 1 << log2
        film = mitr.TransientHDRFilm(Properties("transient_hdr_film", {"width": W, "height": H, "temporal_bins": T, "start_opl": 3.5,
                                                                       "bin_width_opl": 6.0 / T, "rfilter": {"type": "box"}}))
        film.prepare([])
        g = torch.Generator(device="cuda")
        g.manual_seed(1234)
        pix = torch.randint(0, W * H, (S,), device="cuda", generator=g, dtype=torch.int32)
        bins = torch.clamp(torch.normal(400.0, 120.0, (S,), device="cuda", generator=g), 0, T - 1).floor_()
        opl = bins.add_(0.5).mul_(6.0 / T).add_(3.5)
        del bins
        rgb = torch.rand((S, 3), device="cuda", generator=g)
        out = {"contributions": S, "film": f"{W}x{H}x{T}", "algorithmic_bytes": SPLAT_BYTES * S, "peak": HBM_PEAK_GBS, "unit": "GB/s"}

        def leg(name, pixels, variant, what):
            pos = torch.stack(((pixels % W).float() + 0.5, (pixels // W).float() + 0.5), dim=1)
            best = None
            for _ in range(3):
                film.clear()
                ms = film.add_transient_data(pos, opl, None, rgb, 1.0, None, variant=variant)
                best = ms if best is None else min(best, ms)
            gbs = SPLAT_BYTES * S / (best * 1e-3) / 1e9
            out[name] = {"ms": best, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "what": what}
        leg("uniform", pix, 1, "arbitrary order: device-side partition by pixel (two scatter passes over 16-byte records) + LDS rows")
        # (opl and rgb are i.i.d. and independent of the pixel: sorting the pixel column alone gives the sorted benchmark's distribution)
        pix = torch.sort(pix)[0]
        leg("sorted", pix, 1, "pixel-major sorted (the order of the reference's own call): LDS row per pixel run, 16-byte read-modify-write of the touched bins")
        leg("sorted_zero_film", pix, 1 | 0x100, "the same onto a film the caller vouches is zero (MTR_SPLAT_FILM_ZERO): whole-row stores")
        film.clear()
        return out
    for log2 in (log2_s, 28):
        try:
            r = run(log2)
            torch.cuda.empty_cache()
            return r
        except (RuntimeError, torch.OutOfMemoryError, Exception) as e:      # noqa: BLE001 — a side leg must never take the line down
            err = f"{type(e).__name__}: {e}"[:200]
            torch.cuda.empty_cache()
    return {"error": err}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--materials", default="smooth", choices=["smooth", "rough"],
                    help="staircase only: 'smooth' = roughplastic -> diffuse, roughconductor -> conductor (the SURVEY section-8d "
                         "workload); 'rough' = the scene as its file describes it: GGX lobes (roughplastic, roughconductor), vertex normals, bitmap textures")
    ap.add_argument("--scene", default="cornell", choices=["cornell", "staircase", "nlos"],
                    help="staircase: BASELINE configs[4] (512x512, 2048 bins over OPL 0..40, 2048 spp, max_depth 65; "
                         "the reference's scene.xml geometry with approximate materials); nlos: BASELINE configs[3] (NLOS confocal Z scene, "
                         "256x256, 4096 bins of 2^-11 from 1.85; 512 spp per GPU = one GPU's share of 4096 over 8)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--bins", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel PER GPU (weak scaling)")
    ap.add_argument("--mode", default=None, choices=[None, "auto", "fused", "wavefront"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scatter-leg", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the config-4 / config-5 side legs")
    ap.add_argument("--no-splat-microbench", action="store_true", help="skip SURVEY 8d's stand-alone scatter-add micro-benchmark (2^30 contributions)")
    ap.add_argument("--splat-log2", type=int, default=30, help="log2 of the micro-benchmark's contribution count")
    ap.add_argument("--reduced-config5", action="store_true", help="config-5 side leg at 128 of its 2048 spp (quick runs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--reserve-cus", type=int, default=None,
                    help="compute units the persistent path kernel leaves free for RCCL's kernels (mtr_render_params.reserve_cus); "
                         "default: 8 with N > 1, 0 with one GPU")
    ap.add_argument("--comm-only", action="store_true",
                    help="N > 1: time ONLY the communication of a step (reduce-scatter + all-gather of a film-sized tensor in 8 row "
                         "bands + the steady all-reduce, no path kernel) and print that as the line's value; without the flag the same "
                         "measurement rides along as `comm_only`")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    global SCENE, MATERIALS
    SCENE = args.scene
    MATERIALS = args.materials
    dflt = {"cornell": (512, 512, 1024, 1024), "staircase": (512, 512, 2048, 2048), "nlos": (256, 256, 4096, 512)}[SCENE]
    args.width, args.height, args.bins, args.spp = [d if a is None else a
                                                    for a, d in zip((args.width, args.height, args.bins, args.spp), dflt)]

    import torch
    import torch.distributed as dist
    from mitransient_amd import distributed as mdist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run --nproc-per-node N for --gpus N "
                         "(or with no launcher at all: bench.py then starts its ranks itself)")
    # dry-run hooks (single-GPU box): MTR_BENCH_DEVICE pins every rank to one device, MTR_BENCH_BACKEND=gloo
    # replaces RCCL (which needs one device per rank) so that the N>1 code path can be exercised end to end
    device_index = int(os.environ.get("MTR_BENCH_DEVICE", local_rank))
    backend = os.environ.get("MTR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(device_index)
    if world > 1:
        if backend == "nccl":
            # RCCL's kernels on a HIGH-PRIORITY stream: k_fused is persistent and fills every CU's LDS, so the film
            # reduction of band b can only get onto the chip at the boundary between the kernels of bands b and b+1 —
            # where, with equal priority, the next path kernel would take every slot first and the communication of all
            # bands would pile up behind the last one.  (The path kernel does not mind starting a few workgroups late:
            # its work is drawn from a ticket counter.)
            opts = dist.ProcessGroupNCCL.Options()
            opts.is_high_priority_stream = True
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index), pg_options=opts)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")

    # the CPU baseline runs FIRST (rank 0, N = 1): the GPU legs then sit at the end of the command, where a coarse
    # utilisation sampler cannot miss them behind ~20 s of host work
    cpu_res = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_res = cpu_baseline(args.width, args.height, args.bins, args.spp, args.cpu_seconds)

    scene = build_scene(args.width, args.height, args.bins, mode=args.mode)
    integ = scene.integrator()
    integ.collect_stats = True
    spp_total = args.spp * world

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(renderer):
        """W untimed + K timed steps of `renderer`, fenced on both sides; returns (elapsed s = max over ranks, counters summed
        over ranks, per-step kernel ms of this rank, launches, wavefront seen, sums of this rank's per-kernel HIP-event times)"""
        totals = {"paths": 0, "rays_closest": 0, "rays_shadow": 0, "splats_issued": 0, "bounces": 0}
        kernel_ms, launches, wf_seen = [], 0, False
        wf = {"wf_trace_ms": 0.0, "wf_trace_kernel_launches": 0, "wf_shade_ms": 0.0, "scatter_ms": 0.0, "scatter_launches": 0}
        for _ in range(args.warmup):
            renderer.render(spp=spp_total, seed=0)
        fence()
        t0 = time.perf_counter()
        out = None
        for _ in range(args.steps):
            # the previous step's result is dropped BEFORE the next render allocates its own (a render's result belongs to the
            # caller since round 4): the caching allocator hands the same 3 GiB block back.  Holding it across the call made the
            # second timed step hipMalloc a second block — 1 ms on most boxes, 90 ms on one (97 instead of 66.6 ms per step)
            out = None
            out = renderer.render(spp=spp_total, seed=0)
            # (counters and HIP-event times were read back inside render(): collect_stats)
            for k in totals:
                totals[k] += integ.total_counters[k]
            kernel_ms.append(integ.total_times["trace_ms"])          # sum over the launches of this step
            launches += integ.total_times["trace_launches"]
            wf_seen = wf_seen or integ.total_times["scatter_launches"] > 0
            for k in wf:
                wf[k] += integ.total_times.get(k, 0)
        fence()
        elapsed = time.perf_counter() - t0
        del out
        rank_totals = dict(totals)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            c = torch.tensor([totals[k] for k in sorted(totals)], dtype=torch.int64, device="cuda")
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            for k, v in zip(sorted(totals), c.tolist()):
                totals[k] = int(v)
        wf["rank_totals"] = rank_totals
        return elapsed, totals, kernel_ms, launches, wf_seen, wf

    reserve = args.reserve_cus if args.reserve_cus is not None else (8 if world > 1 else 0)
    integ.reserve_cus = reserve

    def comm_only_leg(gather):
        """the communication of one step WITHOUT the path kernel: per band one reduce-scatter of a (rows, W, T, 3) slab of a
        film-sized tensor (+ one all-gather of the reduced rows), then the all-reduce of the (H, W, 4) steady sums — what
        `DistributedRenderer._render_pipelined` issues, on one stream, nothing to overlap with.  step - comm_only = the part of
        the communication the pipeline hides (or fails to)."""
        H, W, T, nb = args.height, args.width, args.bins, 8
        if H % (nb * world):
            return None
        buf = torch.zeros((H, W, T, 3), dtype=torch.float32, device="cuda")
        outb = torch.empty_like(buf) if gather else None
        steady = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        rows_b = H // nb

        def step():
            for b in range(nb):
                slab = mdist.reduce_scatter_rows(buf[b * rows_b:(b + 1) * rows_b])
                if gather:
                    outb[b * rows_b:(b + 1) * rows_b].copy_(mdist.all_gather_rows(slab, rows_b))
            dist.all_reduce(steady)
        for _ in range(max(1, args.warmup)):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item()) / args.steps * 1e3
        gib = buf.numel() * 4 / 2 ** 30
        return {"ms_per_step": ms, "bands": nb, "collectives_per_step": nb * (2 if gather else 1) + 1, "film_GiB": gib,
                "reduce_scatter_GBps_per_rank": (world - 1) / world * buf.numel() * 4 / (ms * 1e-3) / 1e9 if not gather else None,
                "what": "reduce_scatter" + (" + all_gather" if gather else "") + " of a film-sized f32 tensor in 8 row bands + all_reduce of the "
                        "steady sums; no path kernel"}

    if world > 1 and args.comm_only:
        c_full, c_rs = comm_only_leg(True), comm_only_leg(False)
        if rank == 0:
            print(json.dumps({"metric": "ms per step, communication only (no path kernel)", "value": c_full["ms_per_step"] if c_full else None,
                              "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": c_full["ms_per_step"] if c_full else None,
                              "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": f"film {args.height}x{args.width}x{args.bins}x3 f32, 8 row bands"},
                              "comm_only": c_full, "comm_only_reduce_scatter": c_rs,
                              "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0, "comm_backend": backend}))
        dist.destroy_process_group()
        return

    renderer = mdist.DistributedRenderer(scene, partition="spp", gather=True)
    elapsed, totals, kernel_ms, trace_launches, wf_seen, wf_t = timed_run(renderer)
    wft_ms, wft_n, wfs_ms, wsc_ms, wsc_n = (wf_t["wf_trace_ms"], wf_t["wf_trace_kernel_launches"], wf_t["wf_shade_ms"], wf_t["scatter_ms"],
                                            wf_t["scatter_launches"])
    totals_rank0 = wf_t["rank_totals"]
    path_taken = renderer.last_path
    renderer_collectives = renderer.last_collectives
    # N > 1: the same steps with the film reduction ALONE (reduce-scatter, every rank keeps the developed rows it owns —
    # north_star's "single RCCL reduce"); the headline keeps the all-gather that hands every rank the whole tensor
    rs_only = rows_leg = single_leg = None
    if world > 1:
        r2 = mdist.DistributedRenderer(scene, partition="spp", gather=False)
        e2, t2, *_ = timed_run(r2)
        rs_only = {"ms_per_step": e2 / args.steps * 1e3, "value": (t2["rays_closest"] + t2["rays_shadow"]) / e2 / 1e6,
                   "unit": "Mray/s", "what": "reduce_scatter(film) only: every rank develops and keeps its rows of each band "
                                             "(3 GiB all-gather of the developed tensor left out)", "path": r2.last_path}
        del r2
        # ... and the partition that needs no reduction at all: every rank renders ITS rows with all
        # spp x N samples — the same work per rank — and only the developed rows are gathered
        r3 = mdist.DistributedRenderer(scene, partition="rows", gather=True)
        e3, t3, *_ = timed_run(r3)
        rows_leg = {"ms_per_step": e3 / args.steps * 1e3, "value": (t3["rays_closest"] + t3["rays_shadow"]) / e3 / 1e6,
                                  "unit": "Mray/s", "what": "pixel rows sharded instead of samples: no film reduction, all_gather of the developed rows",
                                  "path": r3.last_path}
        del r3
        comm_full, comm_rs = comm_only_leg(True), comm_only_leg(False)
        # opt-in (MTR_BENCH_SINGLE_LAUNCH=1): the headline's pipeline with ONE launch of the path kernel per step — band completion
        # words + hipStreamWaitValue32 on the communication stream (DistributedRenderer(single_launch=True)) instead of a launch
        # per band (+4.7 % on one GPU; tools/bands.py).  Not in the default line: RCCL beside a stream parked on a memory word has
        # not run on more than one GPU anywhere.
        if os.environ.get("MTR_BENCH_SINGLE_LAUNCH") == "1":
            r4 = mdist.DistributedRenderer(scene, partition="spp", gather=True, single_launch=True)
            e4, t4, *_ = timed_run(r4)
            single_leg = {"ms_per_step": e4 / args.steps * 1e3, "value": (t4["rays_closest"] + t4["rays_shadow"]) / e4 / 1e6, "unit": "Mray/s",
                          "band_launches_per_step": r4.last_band_launches, "path": r4.last_path,
                          "what": "the headline's reduce_scatter + all_gather pipeline behind ONE launch with band completion words"}
            del r4

    # ---- untimed extra leg (rank 0, N=1): the same render in wavefront mode, to time the stand-alone
    # time-bin scatter-add kernel (k_wf_scatter) with HIP events on its stream
    scatter = None
    if rank == 0 and world == 1 and not args.no_scatter_leg:
        sc2 = build_scene(args.width, args.height, args.bins, mode="wavefront")
        i2 = sc2.integrator()
        i2.collect_stats = True
        for _ in range(2):
            i2.render(sc2, spp=args.spp, seed=0)
        tm, cn = i2.last_times, i2.last_counters
        n_l = max(1, tm["scatter_launches"])
        b_l = SPLAT_BYTES * cn["splats_issued"] / n_l
        avg = tm["scatter_ms"] / n_l
        default_wl_sc = (SCENE == "cornell" and (args.width, args.height, args.bins, args.spp) == dflt)
        scatter = {"kernel": "k_wf_scatter (MTR_MODE_WAVEFRONT, untimed extra leg)", "bound": "hbm",
                   "achieved": b_l / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": b_l / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "traffic": traffic_from_profiles("k_wf_scatter") if default_wl_sc else None,
                   "avg_launch_ms": avg, "launches_per_render": n_l, "algorithmic_bytes_per_launch": b_l,
                   "render_ms_wavefront": tm["total_ms"]}
        del sc2, i2

    extra = None
    if rank == 0 and world == 1 and SCENE == "cornell" and not args.no_extra_configs:
        del renderer, scene
        torch.cuda.empty_cache()
        extra = extra_config_legs(512, 2048 if not args.reduced_config5 else 128)

    splat_mb = None
    if rank == 0 and world == 1 and SCENE == "cornell" and not args.no_splat_microbench and not args.no_extra_configs:
        try:
            del renderer, scene
        except NameError:
            pass
        torch.cuda.empty_cache()
        splat_mb = splat_microbench(args.splat_log2)

    if rank == 0:
        rays = totals["rays_closest"] + totals["rays_shadow"]
        ms_per_step = elapsed / args.steps * 1e3
        # roofline of the dominant kernel (the path kernel), rank 0's launches, HIP events on its stream:
        # algorithmic bytes per launch = 24 B x contributions one launch issues
        n_launch = max(1, trace_launches)
        fused = not wf_seen
        avg_ms = sum(kernel_ms) / max(1, n_launch) if fused else sum(kernel_ms) / max(1, len(kernel_ms))
        splats_rank0 = totals["splats_issued"] / world
        bytes_per_launch = SPLAT_BYTES * splats_rank0 / (max(1, n_launch) if fused else max(1, len(kernel_ms)))
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        kname = "k_fused" if fused else "k_wf_trace+k_wf_shade+k_wf_scatter (whole render)"
        res = {
            "metric": ("Mray/s (closest-hit + shadow rays), Cornell-box 512^2 x 1024 bins x 1024 spp per GPU" if SCENE == "cornell"
                       else "Mray/s (closest-hit + shadow rays), NLOS confocal Z scene 256^2 x 4096 bins x 512 spp per GPU" if SCENE == "nlos"
                       else "Mray/s (closest-hit + shadow rays), staircase 512^2 x 2048 bins x 2048 spp per GPU"),
            "value": rays / elapsed / 1e6,
            "unit": "Mray/s",
            "time_bins_per_s": totals["splats_issued"] / elapsed,
            "paths_per_s": totals["paths"] / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"cornell_box() diffuse, {args.width}x{args.height} px, {args.bins} time bins "
                                    f"(start_opl 3.5, width 6/{args.bins}), {args.spp} spp per GPU "
                                    f"({spp_total} spp total), max_depth 8, rr_depth 5, seed 0") if SCENE == "cornell" else
                                   (f"NLOS confocal Z scene (the reference's Z.obj behind a 2 x 2 relay wall, nlos_capture_meter + projector, laser and "
                                    f"hidden-geometry sampling), {args.width}x{args.height} px, {args.bins} time bins (start_opl 1.85, width 2^-11), "
                                    f"{args.spp} spp per GPU ({spp_total} spp total), max_depth -1, rr_depth 5, seed 0") if SCENE == "nlos" else
                                   (f"examples/diff-transient/staircase/scene.xml geometry (262,663 triangles, " + ("approximate materials" if MATERIALS == "smooth" else "GGX lobes, vertex normals and (256-px) bitmap textures as in the scene file") + f"), "
                                    f"{args.width}x{args.height} px, {args.bins} time bins (start_opl 0, width 40/{args.bins}), {args.spp} spp per GPU "
                                    f"({spp_total} spp total), max_depth 65, rr_depth 5, camera_unwarp, seed 0"),
                       "parallelism": (f"spp-shard x{world} + RCCL reduce_scatter(film) + all_gather, 8 row bands pipelined against the path kernel "
                                       f"({reserve} CUs left to RCCL)" if world > 1 else "1 GPU"),
                       "mode": args.mode or ("auto (fused: scene + per-pixel time histograms in LDS)" if SCENE in ("cornell", "nlos")
                                             else "auto (wavefront: scene in HBM)")},
            # the fused kernel absorbs the scatter-add in LDS: its HBM fraction is small BY DESIGN;
            # `scatter_add` below is the stand-alone scatter-add kernel of the wavefront organisation
            "counters_per_step": {k: v / args.steps for k, v in totals.items()},
            "source_hash": source_hash(),
        }
        if reserve:
            res["reserve_cus"] = reserve
        default_wl = (args.width, args.height, args.bins, args.spp) == dflt and args.mode in (None, "auto")
        hbm_line = {"kernel": kname, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic_from_profiles("k_fused", {"cornell": None, "nlos": "nlos"}[SCENE]) if (kname == "k_fused" and SCENE in ("cornell", "nlos") and default_wl) else None,
                    "avg_launch_ms": avg_ms, "launches_per_step": n_launch / args.steps,
                    "algorithmic_bytes_per_launch": bytes_per_launch}
        # the dominant kernel's bound: k_fused keeps the scatter-add in LDS, so the HBM line only says how little it moves;
        # what bounds it is VALU issue at ~30 of 64 active lanes (PMC passes of the same workload and the same sources, profiles/)
        vline = None
        if fused and SCENE == "cornell":
            vline = valu_roofline("k_fused", avg_ms, default_wl, n_launch / args.steps)
        elif fused and SCENE == "nlos":       # k_fused<NLOS>: counters of `tools/profile.sh <tag> --scene nlos`, section "nlos"
            vline = valu_roofline("k_fused", avg_ms, default_wl, n_launch / args.steps, section="nlos")
        elif not fused and SCENE == "staircase" and MATERIALS == "smooth" and wft_n:
            # config 5: the dominant kernel is k_wf_trace (closest-hit and any-hit runs), timed alone with HIP events; beside it the
            # HBM-bound k_wf_shade and the scatter-add
            blocks = wavefront_rooflines(totals_rank0, {"wf_trace_ms": wft_ms, "wf_trace_kernel_launches": wft_n, "wf_shade_ms": wfs_ms,
                                                        "scatter_ms": wsc_ms, "scatter_launches": wsc_n},
                                         args.steps, ms_per_step, default_wl and profile_is_current("staircase"))
            if blocks.get("roofline_valu", {}).get("frac") is not None:
                vline = blocks["roofline_valu"]
            if "roofline" in blocks:
                hbm_line["kernel"] = "k_wf_trace+k_wf_shade+k_wf_scatter (whole render: 24 B x contributions)"
                res["roofline_render"] = dict(hbm_line)
                hbm_line = blocks["roofline"]
            for k in ("roofline_shade", "scatter_add"):
                if k in blocks:
                    res[k] = blocks[k]
        # `roofline` is the CONTRACT form: algorithmic bytes of the scatter-add per launch of the dominant kernel over
        # its live launch time against the HBM peak, `traffic` = the PMC passes' HBM bytes.  The fused kernel keeps the scatter-add in
        # LDS, so that fraction is small by construction; what bounds the kernel is VALU issue at ~30 of 64 active lanes, which
        # `roofline_valu` states beside it (a utilisation figure, rounds 2-3 printed it as `roofline`)
        hbm_line["note"] = ("algorithmic bytes = 24 B x time-bin contributions (SURVEY 8d); the scatter-add itself lives in LDS rows "
                            "(k_fused) / in k_wf_scatter (`scatter_add`): the dominant kernel is instruction-bound, see roofline_valu")
        res["roofline"] = hbm_line
        if vline is not None:
            vline["launches_per_step"] = n_launch / args.steps
            res["roofline_valu"] = vline
        else:
            if (SCENE == "cornell" and fused and default_wl and not profile_is_current()) or \
               (SCENE == "nlos" and fused and default_wl and not profile_is_current("nlos")) or \
               (SCENE == "staircase" and default_wl and not profile_is_current("staircase")):
                res["roofline_stale"] = True      # profiles/traffic.json was collected from other sources: counter-based lines omitted
        if not fused and wft_n:
            res["k_wf_trace_ms_per_step"] = wft_ms / args.steps
        if world > 1:
            res["rccl_ranks"] = dist.get_world_size() if backend == "nccl" else 0
            res["comm_backend"] = backend
            res["render_path"] = path_taken
            res["reduce_scatter_only"] = rs_only
            res["row_sharded"] = rows_leg
            if single_leg:
                res["single_launch_bands"] = single_leg
            res["comm_only"] = comm_full
            res["comm_only_reduce_scatter"] = comm_rs
            res["reserve_cus"] = reserve
            res["collectives_per_step"] = renderer_collectives
        if scatter:
            res["scatter_add"] = scatter
        if extra:
            res["extra_configs"] = extra
        if splat_mb:
            res["splat_microbench"] = splat_mb
        if cpu_res is not None:
            res["cpu_baseline"] = cpu_res
            res["gpu_over_cpu"] = res["value"] / cpu_res["value"]
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
