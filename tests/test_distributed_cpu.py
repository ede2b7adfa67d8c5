"""N>1 path on CPU: world_size-2 gloo processes.  Each rank produces its sample shard of the film
(with the ORACLE as the stand-in renderer — the HIP path needs a GPU), the product's
reduce-scatter / all-gather helpers combine them, and the result equals the single-process render."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition():
    from mitransient_amd.distributed import shard_range, row_slab
    for n in (1, 7, 8, 1024, 8192, 13):
        for world in (1, 2, 3, 4, 8):
            parts = [shard_range(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1
            slabs = [row_slab(n, world, r) for r in range(world)]
            assert slabs[0][0] == 0 and slabs[-1][1] == n


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import make_cornell
        from mitransient_amd import distributed as md
        from oracle import oracle
        scene = make_cornell(width=12, height=10, bins=32)      # 10 rows over 2 (and padded) ranks
        sd = scene.data()
        integ, film = scene.integrator(), scene.sensors()[0].film()
        spp = 6
        full_t, full_s, _ = oracle.render(sd, integ.render_params(film, 0, spp), n_threads=1)
        # --- partition "spp": shard samples, reduce-scatter rows, all-gather
        s0, s1 = md.shard_range(spp, world, rank)
        t4, s4, _ = oracle.render(sd, integ.render_params(film, 0, spp, s0, s1), n_threads=1)
        slab_t = md.reduce_scatter_rows(torch.from_numpy(t4))
        slab_s = md.reduce_scatter_rows(torch.from_numpy(s4))
        lo, hi = md.row_slab(10, world, rank)
        assert slab_t.shape[0] == hi - lo
        assert np.allclose(slab_t.numpy(), full_t[lo:hi], rtol=1e-5, atol=1e-9)
        got_t = md.all_gather_rows(slab_t, 10).numpy()
        got_s = md.all_gather_rows(slab_s, 10).numpy()
        assert np.linalg.norm(got_t - full_t) / np.linalg.norm(full_t) < 1e-6
        assert np.linalg.norm(got_s - full_s) / np.linalg.norm(full_s) < 1e-6
        # --- partition "rows": disjoint slabs, no reduction
        r0, r1 = md.shard_range(10, world, rank)
        t4r, s4r, _ = oracle.render(sd, integ.render_params(film, 0, spp, 0, spp, r0 * 12, r1 * 12), n_threads=1)
        assert np.array_equal(t4r[r0:r1], full_t[r0:r1])
        assert np.all(t4r[:r0] == 0) and np.all(t4r[r1:] == 0)
        open(os.path.join(tmp, f"ok{rank}"), "w").write("1")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])      # 8: the node the scaling curve is measured on — 10 rows in 8 padded slabs, 6 samples over 8 ranks (two render nothing)
def test_gloo_spp_shard_reduce(tmp_path, world):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
