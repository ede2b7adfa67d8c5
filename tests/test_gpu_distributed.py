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
        scene = _scene(kind, height)
        steady, transient = md.DistributedRenderer(scene, partition=partition, gather=True).render(spp=10, seed=3)
        torch.cuda.synchronize()
        np.save(os.path.join(tmp, f"t{rank}.npy"), np.array(transient))
        np.save(os.path.join(tmp, f"s{rank}.npy"), np.array(steady))
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
