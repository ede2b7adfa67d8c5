"""Multi-process render on the GPU box: two ranks (both on cuda:0, gloo backend — the box has one GPU, RCCL
needs one device per rank) run DistributedRenderer end to end: sample sharding, reduce-scatter of the raw
film, develop of the row slab, all-gather.  The result must equal the single-process render."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(kind, height):
    from conftest import make_cornell
    if kind == "cornell":
        return make_cornell(width=24, height=height, bins=48)
    if kind == "cornell_wf":
        return make_cornell(width=24, height=height, bins=48, amd_mode="wavefront")
    import mitransient_amd.mi as mi
    from test_rough_bsdf import _rough_cornell
    return mi.load_dict(_rough_cornell(width=24, height=height, temporal_bins=48, bin_width_opl=6.0 / 48))


def _worker(rank, world, port, tmp, partition, height=18, kind="cornell"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mitransient_amd import distributed as md
        gather = kind != "cornell_nogather"
        single = kind == "cornell_single"            # ONE fused launch with band completion words instead of a launch per band
        scene = _scene("cornell" if (single or not gather) else kind, height)
        r = md.DistributedRenderer(scene, partition=partition, gather=gather, single_launch=single)
        steady, transient = r.render(spp=10, seed=3)
        torch.cuda.synchronize()
        np.save(os.path.join(tmp, f"t{rank}.npy"), np.array(transient))
        np.save(os.path.join(tmp, f"s{rank}.npy"), np.array(steady))
        with open(os.path.join(tmp, f"ch{rank}.txt"), "w") as fh:
            fh.write(str(getattr(r, "last_reduced_channels", 0)))
        with open(os.path.join(tmp, f"info{rank}.txt"), "w") as fh:
            fh.write(f"{r.last_path} {getattr(r, 'last_band_streams', 0)} {' '.join(map(str, getattr(r, 'owned_rows', None) or []))}")
        with open(os.path.join(tmp, f"launches{rank}.txt"), "w") as fh:
            fh.write(str(getattr(r, "last_band_launches", 0)))
        with open(os.path.join(tmp, f"coll{rank}.txt"), "w") as fh:
            fh.write(str(r.last_collectives))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("partition", ["spp", "rows"])
def test_two_rank_render_equals_single(tmp_path, partition):
    from conftest import make_cornell, rel_l2
    scene = make_cornell(width=24, height=18, bins=48)
    s_ref, t_ref = scene.integrator().render(scene, seed=3, spp=10)
    s_ref, t_ref = np.array(s_ref), np.array(t_ref)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), partition), nprocs=2, join=True)
    for r in range(2):
        t = np.load(tmp_path / f"t{r}.npy")
        s = np.load(tmp_path / f"s{r}.npy")
        assert t.shape == t_ref.shape and s.shape == s_ref.shape
        assert rel_l2(t, t_ref) <= 1e-6 and rel_l2(s, s_ref) <= 1e-6


def test_two_rank_pipelined_band_reduction(tmp_path):
    """H = 32 = 8 bands x 2 ranks x 2 rows: the band-pipelined reduce-scatter / develop / all-gather path."""
    from conftest import make_cornell, rel_l2
    scene = make_cornell(width=24, height=32, bins=48)
    s_ref, t_ref = scene.integrator().render(scene, seed=3, spp=10)
    s_ref, t_ref = np.array(s_ref), np.array(t_ref)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), "spp", 32), nprocs=2, join=True)
    for r in range(2):
        t = np.load(tmp_path / f"t{r}.npy")
        s = np.load(tmp_path / f"s{r}.npy")
        assert t.shape == t_ref.shape == (32, 24, 48, 3)
        assert rel_l2(t, t_ref) <= 1e-6 and rel_l2(s, s_ref) <= 1e-6
        # one reduce-scatter + one all-gather per band, one all-reduce of the steady sums per render (round 3: four per band)
        assert int((tmp_path / f"coll{r}.txt").read_text()) == 2 * 8 + 1


def test_two_rank_pipelined_single_launch_with_band_words(tmp_path):
    """the same pipeline with ONE launch of the fused kernel per render: the communication stream is parked on band b's
    completion word (mtr_render_params.n_bands, hipStreamWaitValue32) while the launch renders the later bands"""
    from conftest import make_cornell, rel_l2
    scene = make_cornell(width=24, height=32, bins=48)
    s_ref, t_ref = scene.integrator().render(scene, seed=3, spp=10)
    s_ref, t_ref = np.array(s_ref), np.array(t_ref)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), "spp", 32, "cornell_single"), nprocs=2, join=True)
    for r in range(2):
        t = np.load(tmp_path / f"t{r}.npy")
        s = np.load(tmp_path / f"s{r}.npy")
        assert rel_l2(t, t_ref) <= 1e-6 and rel_l2(s, s_ref) <= 1e-6
        assert int((tmp_path / f"launches{r}.txt").read_text()) == 1
        assert int((tmp_path / f"coll{r}.txt").read_text()) == 2 * 8 + 1


def test_band_completion_words_on_one_gpu():
    """mtr_render_params.n_bands: one fused launch publishes a word per band; a stream parked on the LAST band's word runs
    its work only after the whole film is there, the film equals an ordinary render's, a second render needs a new epoch,
    and the wavefront organisation refuses band words"""
    from conftest import make_cornell, rel_l2
    from mitransient_amd import distributed as md
    scene = make_cornell(width=40, height=32, bins=64)
    integ, sens = scene.integrator(), scene.sensors()[0]
    film = sens.film()
    s_ref, t_ref = (np.array(x) for x in integ.render(scene, seed=5, spp=12))
    integ.direct_develop = False
    words = torch.zeros(8, dtype=torch.int32, device="cuda")
    seen = torch.zeros(1, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    for epoch in (1, 2):
        passes = integ.prepare(scene, sens, 5, 12, [])
        integ.accumulate(scene, sens, passes, 12, bands=(8, epoch, words.data_ptr()))
        raw = film.transient_storage.torch_tensor()
        with torch.cuda.stream(side):
            md.stream_wait_value(side, words.data_ptr() + 4 * 7, epoch)
            seen.copy_(raw[-1].abs().sum().reshape(1))              # the last band's rows, read behind its word
        side.synchronize()
        torch.cuda.synchronize()
        assert words.cpu().tolist() == [epoch] * 8
        s, t = (np.array(x) for x in film.develop())
        assert rel_l2(t, t_ref) <= 1e-6 and rel_l2(s, s_ref) <= 1e-6
        assert abs(float(seen.item()) - float(raw[-1].abs().sum().item())) <= 1e-3 * max(1.0, float(seen.item()))
    wf = make_cornell(width=40, height=32, bins=64, amd_mode="wavefront")
    iw, sw = wf.integrator(), wf.sensors()[0]
    passes = iw.prepare(wf, sw, 5, 12, [])
    with pytest.raises(Exception, match="band completion words"):
        iw.accumulate(wf, sw, passes, 12, bands=(8, 3, words.data_ptr()))


@pytest.mark.parametrize("w,h,nb", [(10, 10, 16), (3, 3, 4), (7, 5, 35), (9, 7, 2)])
def test_band_words_when_the_bands_do_not_divide_the_pixels(w, h, nb):
    """bands of floor(n / n_bands) pixels, the last takes the remainder (include/mitransient_amd.h): no band is empty, every
    word is published — 100 pixels in 16 bands and 9 in 4 left trailing bands EMPTY under the former ceil rule, and a stream
    parked on such a word would never have run"""
    from conftest import make_cornell, rel_l2
    scene = make_cornell(width=w, height=h, bins=32)
    integ, sens = scene.integrator(), scene.sensors()[0]
    s_ref, t_ref = (np.array(x) for x in integ.render(scene, seed=2, spp=9))
    integ.direct_develop = False
    words = torch.zeros(nb, dtype=torch.int32, device="cuda")
    passes = integ.prepare(scene, sens, 2, 9, [])
    integ.accumulate(scene, sens, passes, 9, bands=(nb, 5, words.data_ptr()))
    torch.cuda.synchronize()
    assert words.cpu().tolist() == [5] * nb
    s, t = (np.array(x) for x in sens.film().develop())
    assert rel_l2(t, t_ref) <= 1e-6 and rel_l2(s, s_ref) <= 1e-6


def test_two_rank_pipelined_with_rough_materials(tmp_path):
    """the band-pipelined path with a scene that runs the extended-shading kernels (GGX lobes on four shapes)"""
    from conftest import rel_l2
    scene = _scene("rough", 32)
    s_ref, t_ref = scene.integrator().render(scene, seed=3, spp=10)
    s_ref, t_ref = np.array(s_ref), np.array(t_ref)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), "spp", 32, "rough"), nprocs=2, join=True)
    for r in range(2):
        t = np.load(tmp_path / f"t{r}.npy")
        s = np.load(tmp_path / f"s{r}.npy")
        assert t.shape == t_ref.shape == (32, 24, 48, 3)
        assert rel_l2(t, t_ref) <= 1e-6 and rel_l2(s, s_ref) <= 1e-6
    assert np.count_nonzero(t_ref) > 2000


def test_rccl_api_path_with_one_rank():
    """the `nccl` (= RCCL) backend itself, on one GPU: a 1-rank process group created as bench.py creates it, the film
    reduction helpers and the band-pipelined renderer forced through its multi-rank branch (tools/rccl_one_rank.py).
    RCCL moves nothing with one rank, but every call, option and layout of the N > 1 path runs."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_one_rank.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL one-rank path: OK" in r.stdout, r.stdout[-2000:]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_pipelined_wavefront_bands_stay_on_one_stream(tmp_path):
    """the wavefront organisation has ONE workspace per scene, so its row bands must not overlap on two
    streams.  2 ranks x 8 bands with amd_mode='wavefront' against the single-process film."""
    from conftest import make_cornell, rel_l2
    scene = make_cornell(width=24, height=32, bins=48, amd_mode="wavefront")
    s_ref, t_ref = scene.integrator().render(scene, seed=3, spp=10)
    s_ref, t_ref = np.array(s_ref), np.array(t_ref)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "spp", 32, "cornell_wf"), nprocs=2, join=True)
    for r in range(2):
        t = np.load(tmp_path / f"t{r}.npy")
        s = np.load(tmp_path / f"s{r}.npy")
        assert rel_l2(t, t_ref) <= 1e-6 and rel_l2(s, s_ref) <= 1e-6
        path, streams = (tmp_path / f"info{r}.txt").read_text().split()[:2]
        assert path == "pipelined" and streams == "1"
        assert (tmp_path / f"ch{r}.txt").read_text() == "4"            # the wavefront organisation reduces the 4-channel block
    # ... and the fused organisation does use both, and reduces DEVELOPED partial rows: 3 channels, no clear, no develop
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "spp", 32, "cornell"), nprocs=2, join=True)
    assert (tmp_path / "info0.txt").read_text().split()[:2] == ["pipelined", "2"]
    assert (tmp_path / "ch0.txt").read_text() == "3"
    for r in range(2):
        assert rel_l2(np.load(tmp_path / f"t{r}.npy"), t_ref) <= 1e-6 and rel_l2(np.load(tmp_path / f"s{r}.npy"), s_ref) <= 1e-6


def test_two_rank_reduce_scatter_only(tmp_path):
    """gather=False: the film reduction alone — every rank keeps the developed rows it owns (owned_rows), which together
    are the single-process render"""
    from conftest import make_cornell, rel_l2
    scene = make_cornell(width=24, height=32, bins=48)
    s_ref, t_ref = scene.integrator().render(scene, seed=3, spp=10)
    s_ref, t_ref = np.array(s_ref), np.array(t_ref)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "spp", 32, "cornell_nogather"), nprocs=2, join=True)
    seen = np.zeros(32, bool)
    for r in range(2):
        t = np.load(tmp_path / f"t{r}.npy")
        s = np.load(tmp_path / f"s{r}.npy")
        info = (tmp_path / f"info{r}.txt").read_text().split()
        rows = np.array([int(x) for x in info[2:]])
        assert info[0] == "pipelined" and t.shape[0] == len(rows) == 16 and not seen[rows].any()
        assert int((tmp_path / f"coll{r}.txt").read_text()) == 8 + 1         # the film reduction alone: one collective per band + the steady image
        seen[rows] = True
        assert rel_l2(t, t_ref[rows]) <= 1e-6 and rel_l2(s, s_ref[rows]) <= 1e-6
    assert seen.all()


def _worker8(rank, world, port, tmp, partition, gather, single, bands):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import make_cornell
        from mitransient_amd import distributed as md
        scene = make_cornell(width=64, height=64, bins=64)
        r = md.DistributedRenderer(scene, partition=partition, gather=gather, bands=bands, single_launch=single)
        steady, transient = r.render(spp=20, seed=3)          # 20 samples over 8 ranks: shards of 3 and 2
        torch.cuda.synchronize()
        np.save(os.path.join(tmp, f"t{rank}.npy"), np.array(transient))
        np.save(os.path.join(tmp, f"s{rank}.npy"), np.array(steady))
        with open(os.path.join(tmp, f"info{rank}.txt"), "w") as fh:
            fh.write(f"{r.last_path} {getattr(r, 'last_band_launches', 0)} {r.last_collectives} {' '.join(map(str, getattr(r, 'owned_rows', None) or []))}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("partition,gather,single", [("spp", True, False), ("spp", False, True), ("spp", True, True), ("spp", False, False),
                                                     ("rows", True, False), ("rows", False, False)])
def test_eight_ranks_on_one_gpu(tmp_path, partition, gather, single):
    """WORLD SIZE 8 (the node the 1/2/4/8 curve is measured on) through DistributedRenderer on this box's one GPU, gloo instead of
    RCCL: 64 rows = 8 bands x 8 ranks x 1 row, 20 samples in shards of 3 and 2, both partitions, with and without the final
    all-gather, per-band launches and the single launch with band completion words — the first 8-rank execution of this code
    should not be the driver's"""
    from conftest import make_cornell, rel_l2
    scene = make_cornell(width=64, height=64, bins=64)
    s_ref, t_ref = scene.integrator().render(scene, seed=3, spp=20)
    s_ref, t_ref = np.array(s_ref), np.array(t_ref)
    mp.spawn(_worker8, args=(8, _free_port(), str(tmp_path), partition, gather, single, 8), nprocs=8, join=True)
    seen = np.zeros(64, bool)
    for r in range(8):
        t = np.load(tmp_path / f"t{r}.npy")
        s = np.load(tmp_path / f"s{r}.npy")
        info = (tmp_path / f"info{r}.txt").read_text().split()
        if partition == "spp":
            assert info[0] == "pipelined" and int(info[1]) == (1 if single else 8)
            assert int(info[2]) == 8 * (2 if gather else 1) + 1
        if gather:
            assert t.shape == t_ref.shape and rel_l2(t, t_ref) <= 1e-6 and rel_l2(s, s_ref) <= 1e-6
        else:
            rows = np.array([int(x) for x in info[3:]])
            assert len(rows) == 8 == t.shape[0] and not seen[rows].any()
            seen[rows] = True
            assert rel_l2(t, t_ref[rows]) <= 1e-6 and rel_l2(s, s_ref[rows]) <= 1e-6
    assert gather or seen.all()


def test_bench_with_eight_ranks_dry_run():
    """`python bench.py --gpus 8` in the dry-run form (every rank on this GPU, gloo): one JSON line, weak scaling, 17 collectives"""
    res = _bench_line({"MTR_BENCH_BACKEND": "gloo", "MTR_BENCH_DEVICE": "0", "OMP_NUM_THREADS": "1"},
                      ["--gpus", "8", "--steps", "2", "--warmup", "1", "--spp", "8", "--width", "128", "--height", "128", "--bins", "128", "--no-cpu-baseline"], timeout=900)
    assert res["n_gpus"] == 8 and res["comm_backend"] == "gloo" and res["scaling"] == "weak"
    assert res["counters_per_step"]["paths"] == 128 * 128 * 8 * 8
    assert res["render_path"] == "pipelined" and res["collectives_per_step"] == 17
    assert res["row_sharded"]["path"] == "rows" and res["value"] > 0


def _bench_line(env_extra, args, timeout=600):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]              # rank 0 prints ONE JSON line
    return json.loads(lines[0])


BENCH_SMALL = ["--steps", "2", "--warmup", "1", "--spp", "8", "--width", "128", "--height", "128", "--bins", "256", "--no-cpu-baseline"]


def test_bench_self_launches_its_ranks():
    """the driver's command form for N > 1 WITHOUT a launcher — `python bench.py --gpus 2 ...` — starts its two ranks
    itself and prints one JSON line (both ranks on this box's GPU, gloo instead of RCCL: MTR_BENCH_* dry-run hooks)"""
    res = _bench_line({"MTR_BENCH_BACKEND": "gloo", "MTR_BENCH_DEVICE": "0", "MTR_BENCH_SINGLE_LAUNCH": "1"}, ["--gpus", "2"] + BENCH_SMALL)
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["comm_backend"] == "gloo" and res["rccl_ranks"] == 0
    assert res["single_launch_bands"]["band_launches_per_step"] == 1 and res["single_launch_bands"]["value"] > 0      # the opt-in leg
    assert res["render_path"] == "pipelined" and res["scaling"] == "weak"
    assert res["counters_per_step"]["paths"] == 128 * 128 * 8 * 2            # weak scaling: 8 spp per rank
    assert res["reduce_scatter_only"]["path"] == "pipelined" and res["reduce_scatter_only"]["ms_per_step"] > 0
    assert res["row_sharded"]["path"] == "rows" and res["row_sharded"]["value"] > 0
    assert res["value"] > 0 and res["roofline"]["bound"] == "hbm"           # the contract form; the utilisation figure rides along as roofline_valu
    # the communication alone (no path kernel), the CUs left to it, and how many collectives a step issues
    assert res["comm_only"]["ms_per_step"] > 0 and res["comm_only"]["collectives_per_step"] == 17
    assert res["comm_only_reduce_scatter"]["collectives_per_step"] == 9 and res["collectives_per_step"] == 17
    assert res["reserve_cus"] == 8


def test_bench_comm_only_line():
    """`bench.py --gpus 2 --comm-only`: the line's value is the communication of a step alone"""
    res = _bench_line({"MTR_BENCH_BACKEND": "gloo", "MTR_BENCH_DEVICE": "0"}, ["--gpus", "2", "--comm-only"] + BENCH_SMALL)
    assert res["n_gpus"] == 2 and res["unit"] == "ms" and res["higher_is_better"] is False
    assert res["value"] == res["comm_only"]["ms_per_step"] > 0 and res["comm_only_reduce_scatter"]["ms_per_step"] > 0


def test_reserved_compute_units_change_nothing_but_the_grid():
    """mtr_render_params.reserve_cus (amd_reserve_cus): the persistent kernel leaves CUs to other streams; same samples, same film"""
    from conftest import make_cornell, rel_l2
    a = make_cornell(width=64, height=64, bins=64)
    b = make_cornell(width=64, height=64, bins=64, amd_reserve_cus=16)
    c = make_cornell(width=64, height=64, bins=64, amd_reserve_cus=100000)          # clamped: at least one CU renders
    ia = a.integrator(); ia.collect_stats = True
    sa, ta = ia.render(a, seed=2, spp=32)
    ref = np.array(ta)
    for sc in (b, c):
        integ = sc.integrator(); integ.collect_stats = True
        s_, t_ = integ.render(sc, seed=2, spp=32)
        torch.cuda.synchronize()
        assert rel_l2(np.array(t_), ref) <= 1e-6 and integ.last_counters == ia.last_counters


def test_bench_config4_share_two_ranks():
    """`bench.py --scene nlos` — BASELINE config 4's per-GPU share as a bench workload — with two ranks sharing this GPU: the NLOS
    film goes through the same sample-sharded reduce as config 3's"""
    res = _bench_line({"MTR_BENCH_BACKEND": "gloo", "MTR_BENCH_DEVICE": "0"},
                      ["--gpus", "2", "--scene", "nlos", "--steps", "2", "--warmup", "1", "--width", "32", "--height", "32", "--bins", "512",
                       "--spp", "16", "--no-cpu-baseline", "--no-scatter-leg"])
    assert res["n_gpus"] == 2 and "NLOS" in res["metric"] and "nlos_capture_meter" in res["config"]["workload"]
    assert res["counters_per_step"]["paths"] == 32 * 32 * 16 * 2 and res["counters_per_step"]["splats_issued"] > 0
    assert res["value"] > 0 and res["render_path"] in ("pipelined", "whole")


def test_bench_under_torchrun_matches_the_contract():
    """the same branch launched as the driver launches it: python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MTR_BENCH_BACKEND="gloo", MTR_BENCH_DEVICE="0", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), "bench.py", "--gpus", "2"] + BENCH_SMALL, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["counters_per_step"]["paths"] == 128 * 128 * 8 * 2


def test_bench_two_ranks_over_rccl():
    """... and over RCCL proper wherever two GPUs are visible (skips on the one-GPU box)"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    res = _bench_line({}, ["--gpus", "2"] + BENCH_SMALL)
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["comm_backend"] == "nccl" and res["render_path"] == "pipelined"
