"""Soak of mtr_splat_add's partition path: 40 calls with random film shapes, sizes (across tile multiples), orders and window
fractions against a float64 index_add on the GPU — the workspace is reused, grown and trimmed in between.
usage: python tools/soak_splat.py [seed] [calls = 40]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mitransient_amd as mitr, mitransient_amd.mi as mi
from mitransient_amd.scene import Properties
from mitransient_amd import _cabi
from mitransient_amd.runtime import get_context
mi.set_variant('llvm_ad_rgb')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N_CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = 0.0
for it in range(N_CALLS):
    W, H = int(rng.integers(1, 200)), int(rng.integers(1, 120))
    if W * H < 2: W = 2
    T = int(rng.choice([1, 7, 64, 300, 1024, 2048, 4096]))
    n = int(rng.choice([1, 2, 255, 4095, 4096, 4097, 8193, 100000, 1 << 20, 3000001]))
    film = mitr.TransientHDRFilm(Properties('transient_hdr_film', {'width': W, 'height': H, 'temporal_bins': T, 'start_opl': 1.0,
                                                                   'bin_width_opl': 2.0 / T, 'rfilter': {'type': 'box'}}))
    film.prepare([])
    g = torch.Generator(device='cuda'); g.manual_seed(int(rng.integers(1 << 30)))
    pix = torch.randint(0, W * H + 2, (n,), device='cuda', generator=g, dtype=torch.int32)
    if rng.random() < 0.2:
        pix = torch.sort(pix).values
    opl = (0.8 + 2.4 * torch.rand((n,), device='cuda', generator=g)).float()          # 1/6 of them outside the window
    rgb = torch.rand((n, 3), device='cuda', generator=g) * (10.0 if rng.random() < 0.5 else 1e-3)
    zero = rng.random() < 0.5
    variant = 1 | (_cabi.MTR_SPLAT_FILM_ZERO if zero else 0)
    if not zero:                                                                      # something already on the film
        film.transient_storage.torch_tensor().uniform_(0.0, 1.0)
        film.transient_storage.torch_tensor()[..., 3] = 0
    before = film.transient_storage.torch_tensor().double().clone()
    film.transient_storage.put_opl(pix, opl, rgb[:, 0].contiguous(), rgb[:, 1].contiguous(), rgb[:, 2].contiguous(), film.desc(), variant)
    torch.cuda.synchronize()
    got = film.transient_storage.torch_tensor().double()
    # reference: the contract form (variant 0, f32 atomics wherever a contribution lands — an independent code path with the
    # library's own f32 bin arithmetic; a float64 index_add disagrees on the few path lengths that sit on a bin edge)
    film0 = mitr.TransientHDRFilm(Properties('transient_hdr_film', {'width': W, 'height': H, 'temporal_bins': T, 'start_opl': 1.0,
                                                                    'bin_width_opl': 2.0 / T, 'rfilter': {'type': 'box'}}))
    film0.prepare([])
    film0.transient_storage.torch_tensor().copy_(before.float())
    film0.transient_storage.put_opl(pix, opl, rgb[:, 0].contiguous(), rgb[:, 1].contiguous(), rgb[:, 2].contiguous(), film0.desc(), 0)
    torch.cuda.synchronize()
    ref = film0.transient_storage.torch_tensor().double()
    err = float((got - ref).norm() / max(float(ref.norm()), 1e-30))
    same_support = bool(((got != 0) == (ref != 0)).all())
    tol = 2e-6
    worst = max(worst, err)
    print(f'{it:2d} film {W}x{H}x{T} n {n} zero {zero} rel-L2 vs the contract form {err:.2e} same support {same_support}', 'OK' if err <= tol and same_support else 'FAIL')
    assert err <= tol and same_support
    del film0
    if rng.random() < 0.3:
        ctx = get_context(); ctx.check(ctx.lib.mtr_ctx_trim(ctx.handle), 'mtr_ctx_trim')
print('soak_splat: %d calls OK, worst rel-L2 %.2e' % (N_CALLS, worst))
