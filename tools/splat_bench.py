"""SURVEY §8d micro-benchmark of the stand-alone time-bin scatter-add (`mtr_splat_add` = film.add_transient_data from
Python): S synthetic contributions into a 512 x 512 x 1024 film, pixel uniform or pixel-major sorted in [0, 2^18),
bin ~ clipped Normal(400, 120), rgb ~ U(0, 1).  Reports contributions/s and algorithmic GB/s (24 B per contribution)
against the 8 TB/s HBM peak.   usage: python tools/splat_bench.py [log2 S = 28]"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mitransient_amd as mitr
import mitransient_amd.mi as mi
from mitransient_amd.scene import Properties

mi.set_variant('llvm_ad_rgb')
S = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
W = H = 512
T = 1024
film = mitr.TransientHDRFilm(Properties('transient_hdr_film', {'width': W, 'height': H, 'temporal_bins': T, 'start_opl': 3.5,
                                                               'bin_width_opl': 6.0 / T, 'rfilter': {'type': 'box'}}))
film.prepare([])
g = torch.Generator(device='cuda'); g.manual_seed(1234)
pix = torch.randint(0, W * H, (S,), device='cuda', generator=g, dtype=torch.int32)
bins = torch.clamp(torch.normal(400.0, 120.0, (S,), device='cuda', generator=g), 0, T - 1).floor()
opl = (3.5 + (bins + 0.5) * (6.0 / T)).float()
rgb = torch.rand((S, 3), device='cuda', generator=g)
del bins
for order in ('uniform', 'pixel-sorted'):
    if order == 'pixel-sorted':
        idx = torch.argsort(pix)
        pix, opl, rgb = pix[idx].contiguous(), opl[idx].contiguous(), rgb[idx].contiguous()
        del idx
    pos = torch.stack(((pix % W).float() + 0.5, (pix // W).float() + 0.5), dim=1)
    for variant in (0, 1, 1 | 0x100):                 # 0x100 = MTR_SPLAT_FILM_ZERO: the film is zero on entry (store-only flush)
        best = 1e30
        for _ in range(3):
            film.clear()
            ms = film.add_transient_data(pos, opl, None, rgb, 1.0, None, variant=variant)
            best = min(best, ms)
        gbs = 24.0 * S / (best * 1e-3) / 1e9
        what = {0: "f32 atomics to HBM", 1: "LDS rows per pixel run" if order != 'uniform' else "partition by pixel + LDS rows",
                0x101: ("LDS rows" if order != 'uniform' else "partition + LDS rows") + ", zero film: store-only flush"}[variant]
        print(f'S = 2^{int(np.log2(S))}  {order:12s} variant {variant:#x} ({what}): '
              f'{best:9.2f} ms  {S / best / 1e6:8.2f} G contributions/s  {gbs:8.1f} GB/s = {gbs / 8000 * 100:5.1f} % of the HBM peak')
    del pos
