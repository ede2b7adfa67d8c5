"""Sanity of the RCCL code path on a ONE-GPU box: a 1-rank `nccl` process group created exactly as bench.py creates it
(high-priority stream option), the film reduction helpers (reduce_scatter_tensor / all_gather_into_tensor) and the
band-pipelined DistributedRenderer forced through its multi-rank branch.  RCCL moves nothing with one rank, but every API
call, option and tensor layout of the N > 1 path executes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
opts = dist.ProcessGroupNCCL.Options(); opts.is_high_priority_stream = True
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=opts)
from mitransient_amd import distributed as md
from conftest import make_cornell
t = torch.rand((10, 12, 8, 4), device="cuda")
slab = md.reduce_scatter_rows(t); full = md.all_gather_rows(slab, 10)
assert torch.equal(full, t), "reduce_scatter / all_gather round trip"
scene = make_cornell(width=64, height=64, bins=64)
ref_s, ref_t = scene.integrator().render(scene, seed=0, spp=32)
r = md.DistributedRenderer(scene, partition="spp", gather=True, bands=8)
integ = scene.integrator(); sens = scene.sensors()[0]; film = sens.film()
passes = integ.prepare(scene, sens, 0, 32, [])
s, tt = r._render_pipelined(integ, sens, film, passes, 32, (0, 32), 8, 1)
torch.cuda.synchronize()
e = float((tt.torch() - ref_t.torch()).norm() / ref_t.torch().norm())
print("rccl ranks", dist.get_world_size(), "backend", dist.get_backend(), "pipelined vs plain rel-L2 %.2e" % e)
assert e <= 1e-6
dist.destroy_process_group()
print("RCCL one-rank path: OK")
