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


MATERIALS = "smooth"       # --materials (staircase only): "smooth" = the SURVEY section-8d mapping (default) | "rough" = GGX lobes kept
SCENE = "cornell"          # --scene: "cornell" (BASELINE configs[1], the default) | "staircase" (configs[4] geometry)


def build_scene(width, height, bins, max_depth=8, mode=None):
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
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


def pmc_from_profiles(kernel):
    """Counters per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/profile.sh: separate --pmc passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    {} when no profile of this kernel is committed: bench.py itself cannot collect PMC counters."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh).get(kernel, {}) or {}
    except Exception:
        return {}


def traffic_from_profiles(kernel):
    return pmc_from_profiles(kernel).get("hbm_bytes_per_launch")


def valu_roofline(kernel, avg_launch_ms, workload_matches, launches_per_render=1.0):
    """The path kernel is bound by VALU issue + SIMT divergence, not by HBM: achieved = VALU lane-operations
    per launch (SQ_INSTS_VALU x active lanes per instruction, from the committed PMC pass of the SAME workload) / the
    launch time measured live with HIP events; peak = the f32 vector roof."""
    c = pmc_from_profiles(kernel)
    if not workload_matches or "valu_insts_per_launch" not in c or avg_launch_ms <= 0:
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
            "traffic_note": "2 x FETCH_SIZE + WRITE_SIZE; algorithmic 24 B x contributions = 10.7 GB per config-2 launch — the excess is "
                            "WRITE_SIZE: register-spill scratch evicted from L2 (DESIGN.md section 6)",
            "source": "profiles/traffic.json (rocprofv3 --pmc SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU; "
                      + str(c.get("profile", "")) + ") over the launch time measured live (HIP events)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--materials", default="smooth", choices=["smooth", "rough"],
                    help="staircase only: 'smooth' = roughplastic -> diffuse, roughconductor -> conductor (the SURVEY section-8d "
                         "workload); 'rough' = the scene as its file describes it: GGX lobes (roughplastic, roughconductor), vertex normals, bitmap textures")
    ap.add_argument("--scene", default="cornell", choices=["cornell", "staircase"],
                    help="staircase: BASELINE configs[4] (512x512, 2048 bins over OPL 0..40, 2048 spp, max_depth 65; "
                         "the reference's scene.xml geometry with approximate materials)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--bins", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel PER GPU (weak scaling)")
    ap.add_argument("--mode", default=None, choices=[None, "auto", "fused", "wavefront"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scatter-leg", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    global SCENE, MATERIALS
    SCENE = args.scene
    MATERIALS = args.materials
    dflt = {"cornell": (512, 512, 1024, 1024), "staircase": (512, 512, 2048, 2048)}[SCENE]
    args.width, args.height, args.bins, args.spp = [d if a is None else a
                                                    for a, d in zip((args.width, args.height, args.bins, args.spp), dflt)]

    import torch
    import torch.distributed as dist
    from mitransient_amd import distributed as mdist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run --nproc-per-node N for --gpus N")
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
    renderer = mdist.DistributedRenderer(scene, partition="spp", gather=True)

    totals = {"paths": 0, "rays_closest": 0, "rays_shadow": 0, "splats_issued": 0, "bounces": 0}
    kernel_ms = []
    trace_launches = 0
    wf_seen = False                      # MTR_MODE_AUTO resolves inside the library: wavefront runs scatter launches

    def step(timed):
        nonlocal trace_launches, wf_seen
        steady, transient = renderer.render(spp=spp_total, seed=0)
        if timed:
            for k in totals:
                totals[k] += integ.total_counters[k]
            kernel_ms.append(integ.total_times["trace_ms"])          # sum over the launches of this step
            trace_launches += integ.total_times["trace_launches"]
            wf_seen = wf_seen or integ.total_times["scatter_launches"] > 0
        return steady, transient

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    fence()
    elapsed = time.perf_counter() - t0
    del out

    # max over ranks of the elapsed time; sums of the counters
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([totals[k] for k in sorted(totals)], dtype=torch.int64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        for k, v in zip(sorted(totals), c.tolist()):
            totals[k] = int(v)

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
        scatter = {"kernel": "k_wf_scatter (MTR_MODE_WAVEFRONT, untimed extra leg)", "bound": "hbm",
                   "achieved": b_l / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": b_l / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic_from_profiles("k_wf_scatter"),
                   "avg_launch_ms": avg, "launches_per_render": n_l, "algorithmic_bytes_per_launch": b_l,
                   "render_ms_wavefront": tm["total_ms"]}
        del sc2, i2

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
                                   (f"examples/diff-transient/staircase/scene.xml geometry (262,663 triangles, " + ("approximate materials" if MATERIALS == "smooth" else "GGX lobes, vertex normals and (256-px) bitmap textures as in the scene file") + f"), "
                                    f"{args.width}x{args.height} px, {args.bins} time bins (start_opl 0, width 40/{args.bins}), {args.spp} spp per GPU "
                                    f"({spp_total} spp total), max_depth 65, rr_depth 5, camera_unwarp, seed 0"),
                       "parallelism": f"spp-shard x{world} + RCCL reduce_scatter(film) + all_gather" if world > 1 else "1 GPU",
                       "mode": args.mode or ("auto (fused: scene + per-pixel time histograms in LDS)" if SCENE == "cornell"
                                             else "auto (wavefront: scene in HBM)")},
            # the fused kernel absorbs the scatter-add in LDS: its HBM fraction is small BY DESIGN;
            # `scatter_add` below is the stand-alone scatter-add kernel of the wavefront organisation
            "counters_per_step": {k: v / args.steps for k, v in totals.items()},
        }
        hbm_line = {"kernel": kname, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_from_profiles("k_fused") if kname == "k_fused" else None,
                    "avg_launch_ms": avg_ms, "launches_per_step": n_launch / args.steps,
                    "algorithmic_bytes_per_launch": bytes_per_launch}
        # the dominant kernel's bound: k_fused keeps the scatter-add in LDS, so the HBM line only says how little it moves;
        # what bounds it is VALU issue at ~27 of 64 active lanes (PMC passes of the same workload, profiles/)
        default_wl = (SCENE == "cornell" and (args.width, args.height, args.bins, args.spp) == dflt and world >= 1)
        vline = valu_roofline("k_fused", avg_ms, default_wl, n_launch / args.steps) if fused else None
        if vline is not None:
            vline["launches_per_step"] = n_launch / args.steps
            res["roofline"] = vline
            res["roofline_hbm"] = hbm_line
        else:
            res["roofline"] = hbm_line
        if world > 1:
            res["rccl_ranks"] = dist.get_world_size() if backend == "nccl" else 0
            res["comm_backend"] = backend
        if scatter:
            res["scatter_add"] = scatter
        if cpu_res is not None:
            res["cpu_baseline"] = cpu_res
            res["gpu_over_cpu"] = res["value"] / cpu_res["value"]
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
