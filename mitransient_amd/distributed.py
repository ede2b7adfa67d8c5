"""Multi-GPU rendering: one process per GPU, lanes sharded, ONE film reduction over RCCL/xGMI.

The reference is single-device; this is the north-star's extension.  Lanes are
independent and the RNG stream depends only on (seed, lane = pixel*spp_total + s), so any
partition of the samples over ranks reproduces the single-GPU sample set exactly; only the
f32 summation order of the film changes.

Two partitions:
  * ``spp``  — every rank renders all pixels with its slice of the samples, then the raw
    (H,W,T,4) films are summed with ``reduce_scatter`` along H: all 7 xGMI links of every GPU
    carry 1/8 of the film at once (a ring all-reduce would push 2*(7/8)*4 GiB through one link).
    Each rank develops its row slab; ``all_gather`` (optional) rebuilds the full tensor.
  * ``rows`` — every rank renders a contiguous block of image rows with all samples: film slabs
    are disjoint, no reduction at all (gather only).
"""
from __future__ import annotations

import os

from typing import Optional, Tuple


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced partition of range(n) over ``world`` ranks (first n % world ranks get +1)."""
    base, rem = divmod(n, world)
    b = rank * base + min(rank, rem)
    return b, b + base + (1 if rank < rem else 0)


def row_slab(height: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows owned by ``rank`` after the reduce-scatter (equal slabs; the film is padded to a multiple of world)."""
    per = (height + world - 1) // world
    return min(height, rank * per), min(height, (rank + 1) * per)


def reduce_scatter_rows(t, group=None):
    """Sum a (H, ...) tensor over ranks and return this rank's row slab (rows padded to world*per).
    Works with any backend torch.distributed offers (nccl = RCCL on ROCm; gloo on CPU)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    H = t.shape[0]
    per = (H + world - 1) // world
    if per * world != H:
        pad = torch.zeros((per * world - H,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], dim=0)
    t = t.contiguous()
    out = torch.empty((per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if dist.get_backend(group) == "gloo" and t.device.type != "cpu":
        # gloo reduce-scatters host tensors only (the CPU tests run the same reduce_scatter_tensor call as RCCL, world size 2
        # and 3); device tensors over gloo — the 2-rank GPU tests on a one-GPU box — go through all_reduce + slice
        tt = t.clone()
        dist.all_reduce(tt, group=group)
        out.copy_(tt[rank * per:(rank + 1) * per])
    else:
        try:
            dist.reduce_scatter_tensor(out, t, op=dist.ReduceOp.SUM, group=group)
        except (RuntimeError, NotImplementedError):
            if dist.get_backend(group) != "gloo":
                raise
            tt = t.clone()                   # older torch: gloo has no reduce_scatter_tensor — the same sum through all_reduce + slice
            dist.all_reduce(tt, group=group)
            out.copy_(tt[rank * per:(rank + 1) * per])
    lo, hi = row_slab(H, world, rank)
    return out[:hi - lo]


def all_gather_rows(slab, height: int, group=None):
    """Inverse of the scatter: every rank receives the full (H, ...) tensor."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = (height + world - 1) // world
    if slab.shape[0] != per:
        pad = torch.zeros((per - slab.shape[0],) + tuple(slab.shape[1:]), dtype=slab.dtype, device=slab.device)
        slab = torch.cat([slab, pad], dim=0)
    slab = slab.contiguous()
    full = torch.empty((per * world,) + tuple(slab.shape[1:]), dtype=slab.dtype, device=slab.device)
    if dist.get_backend(group) == "gloo" and slab.device.type != "cpu":      # (as above: host tensors take the call RCCL takes)
        parts = [torch.empty_like(slab) for _ in range(world)]
        dist.all_gather(parts, slab, group=group)
        full = torch.cat(parts, dim=0)
    else:
        try:
            dist.all_gather_into_tensor(full, slab, group=group)
        except (RuntimeError, NotImplementedError):
            if dist.get_backend(group) != "gloo":
                raise
            parts = [torch.empty_like(slab) for _ in range(dist.get_world_size(group))]      # older torch over gloo
            dist.all_gather(parts, slab, group=group)
            full = torch.cat(parts, dim=0)
    return full[:height]


_HIP = None


def stream_wait_value(stream, word_ptr: int, value: int):
    """park ``stream`` until the uint32 at device address ``word_ptr`` EQUALS ``value`` (hipStreamWaitValue32): what follows on
    the stream — the film reduction of a band — starts when the path kernel has published that band's completion word"""
    global _HIP
    import ctypes as C
    if _HIP is None:
        _HIP = C.CDLL("libamdhip64.so")             # already mapped by torch
        _HIP.hipStreamWaitValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_uint32]
        _HIP.hipStreamWaitValue32.restype = C.c_int
    rc = _HIP.hipStreamWaitValue32(C.c_void_p(stream.cuda_stream), C.c_void_p(word_ptr), C.c_uint32(value & 0xFFFFFFFF), 1, 0xFFFFFFFF)   # 1 = hipStreamWaitValueEq
    if rc != 0:
        raise RuntimeError(f"hipStreamWaitValue32 failed with {rc}")


def _develop_rows(film, raw_rows, out=None):
    """develop() of a slab of raw film rows (rows, W, T, 4) -> (rows, W, T, 3), without a steady image"""
    import torch
    rows = int(raw_rows.shape[0])
    dummy_s = torch.zeros((rows, film.size()[0], 4), dtype=torch.float32, device=raw_rows.device)
    if out is None:
        return film.develop_slab(raw_rows, dummy_s)[0]
    dummy_o = torch.empty((rows, film.size()[0], 3), dtype=torch.float32, device=raw_rows.device)
    film.develop_slab(raw_rows, dummy_s, out=(out, dummy_o))
    return out


class DistributedRenderer:
    """Shards one render over the ranks of a process group.

    ``spp`` is the TOTAL sample count; each rank renders ``shard_range(spp, world, rank)`` (partition
    "spp") or all samples of its rows (partition "rows").

    With partition "spp" the film reduction is PIPELINED against the path kernel: the image is rendered
    in ``bands`` horizontal bands; as soon as a band's kernel has finished (event), its slab of the raw
    film is reduce-scattered on a side stream while the next band renders, then developed and
    all-gathered straight into its rows of the output.  Only the last band's communication is exposed.
    Collectives per render: ONE reduce-scatter (+ one all-gather with ``gather``) per band and one all-reduce of the
    4 MB steady accumulator at the end — 2 * bands + 1 (``last_collectives``); round 3 issued 4 per band.

    ``gather=False`` ("the single RCCL reduce"): every rank keeps the developed rows it OWNS.  Through the band pipeline those
    are not one contiguous slab: band b's rows ``b * H/bands + rank * per ... + per`` with ``per = H / (bands * world)``,
    stacked in band order — ``owned_rows`` lists the film row of every returned row (the non-pipelined fallback returns the
    contiguous slab ``row_slab(H, world, rank)`` and leaves ``owned_rows`` = that range).

    ``reserve_cus``: compute units the persistent fused kernel leaves free while ``world > 1`` (mtr_render_params.reserve_cus),
    so that RCCL's kernels of band b can run WHILE band b + 1 renders instead of waiting for a launch boundary; None keeps the
    integrator's own ``amd_reserve_cus``."""

    def __init__(self, scene, partition: str = "spp", group=None, gather: bool = True, bands: int = 8, reserve_cus=None,
                 single_launch=None):
        if partition not in ("spp", "rows"):
            raise ValueError("partition must be 'spp' or 'rows'")
        self.scene, self.partition, self.group, self.gather, self.bands = scene, partition, group, gather, int(bands)
        self.reserve_cus = reserve_cus
        # ONE launch of the fused kernel per render with band completion words (mtr_render_params.n_bands) instead of one launch
        # per band: the communication stream is parked on band b's word (hipStreamWaitValue32) while the same launch renders
        # band b + 1.  Opt-in (or MTR_SINGLE_LAUNCH_BANDS=1): it removes the +4 % of eight launches on one GPU, but RCCL next to a
        # stream parked on a memory word has not run on more than one GPU anywhere — the per-band launches stay the default.
        self.single_launch = (os.environ.get("MTR_SINGLE_LAUNCH_BANDS", "") == "1") if single_launch is None else bool(single_launch)
        self.last_band_launches = 0
        self._band_epoch = 0
        self.last_path = None            # "single" | "pipelined" | "spp" | "rows": which code path the last render took
        self.last_collectives = 0        # collectives the last render issued on the data path
        self.owned_rows = None

    def render(self, spp: int, seed: int = 0, sensor: int = 0):
        import torch
        import torch.distributed as dist
        from .tensor import TensorXf
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        scene = self.scene
        integ = scene.integrator()
        sens = scene.sensors()[sensor]
        film = sens.film()
        integ.check_transient_(scene, sens)
        if world == 1:
            self.last_path = "single"
            self.last_collectives = 0
            return integ.render(scene, sens, seed=seed, spp=spp)      # (incl. the single-pass film lifecycle when it applies)
        if self.reserve_cus is not None:
            keep = integ.reserve_cus
            integ.reserve_cus = int(self.reserve_cus)
            try:
                return self._render_sharded(integ, sens, film, spp, seed, world, rank)
            finally:
                integ.reserve_cus = keep
        return self._render_sharded(integ, sens, film, spp, seed, world, rank)

    def _render_sharded(self, integ, sens, film, spp, seed, world, rank):
        import torch
        import torch.distributed as dist
        from .tensor import TensorXf
        scene = self.scene
        W, H = film.size()
        cw, ch = film.crop_size()
        # sample shards through the band pipeline: when the fused kernel can store DEVELOPED rows (MTR_FLAG_DEVELOPED_ROWS), every
        # rank renders its partial sums straight into an (H,W,T,3) tensor — developing is linear, the weight channel is 0 — so the
        # film reduction moves 3 channels instead of 4 and neither the 4-channel clear nor a develop pass runs on any rank
        nb = self.bands
        pipelined = (self.partition == "spp" and nb > 1 and ch == H and cw == W and H % (nb * world) == 0
                     and getattr(film, "frequencies_f32", None) is None)
        n_spp = spp if spp != 0 else sens.sampler().sample_count()
        dev3 = bool(pipelined and not film.exhaustive_scan and
                    integ.developed_rows_ok(scene, sens, n_spp, shard_range(n_spp, world, rank), (0, (H // nb) * W)))
        passes = integ.prepare(scene, sens, seed, spp, integ.aov_names(), _direct_develop=dev3)
        total_spp = sum(s for _, s in passes)
        if dev3 and len(passes) > 1:              # (a split render accumulates in the block)
            dev3 = False
            film._ensure_raw()
        self.last_path = self.partition
        # reject what the slab develop cannot do BEFORE any rendering (a phasor film has no time rows to scatter;
        # an exhaustive film's "steady" image is a mean over channels of the gathered tensor)
        if getattr(film, "frequencies_f32", None) is not None:
            raise NotImplementedError("DistributedRenderer: phasor_hdr_film is single-GPU only")
        if self.partition == "spp":
            my_spp = shard_range(total_spp, world, rank)
            if pipelined:
                self.last_path = "pipelined"
                return self._render_pipelined(integ, sens, film, passes, total_spp, my_spp, nb, world, dev3)
            integ.accumulate(scene, sens, passes, total_spp, spp_range=my_spp)
        else:
            r0, r1 = shard_range(ch, world, rank)
            integ.accumulate(scene, sens, passes, total_spp, pixel_range=(r0 * cw, r1 * cw))
        raw_t = film.transient_storage.torch_tensor()
        raw_s = film.steady_accum()
        lo_, hi_ = row_slab(H, world, rank) if self.partition == "spp" else shard_range(ch, world, rank)
        self.owned_rows = list(range(lo_, hi_))
        self.last_collectives = (2 if self.partition == "spp" else 0) + (2 if self.gather else 0)
        if self.partition == "spp":
            slab_t = reduce_scatter_rows(raw_t, self.group)      # THE film reduction
            slab_s = reduce_scatter_rows(raw_s, self.group)
        else:
            # rows: crop rows == film rows (transient_image_block.py:132 subtracts the crop offset)
            lo, hi = shard_range(ch, world, rank)
            slab_t, slab_s = raw_t[lo:hi], raw_s[lo:hi]
        dev_t, dev_s = film.develop_slab(slab_t, slab_s)
        exh = film.exhaustive_scan                # its "steady" is mean(transient, axis=-1) (transient_hdr_film.py:213-214)
        if not self.gather:
            return TensorXf(dev_t.mean(dim=-1) if exh else dev_s), TensorXf(dev_t)
        if self.partition == "spp":
            full_t = all_gather_rows(dev_t, H, self.group)
            return (TensorXf(full_t.mean(dim=-1) if exh else all_gather_rows(dev_s, H, self.group)), TensorXf(full_t))
        # rows: uneven slabs are padded to the largest before the gather
        per = (ch + world - 1) // world

        def pad(x):
            if x.shape[0] == per:
                return x
            return torch.cat([x, torch.zeros((per - x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)])
        parts_t = all_gather_rows(pad(dev_t), per * world, self.group)
        parts_s = all_gather_rows(pad(dev_s), per * world, self.group)
        rows_t, rows_s = [], []
        for r in range(world):
            a, b = shard_range(ch, world, r)
            rows_t.append(parts_t[r * per:r * per + (b - a)])
            rows_s.append(parts_s[r * per:r * per + (b - a)])
        full_t, full_s = torch.cat(rows_t), torch.cat(rows_s)
        if full_t.shape[0] < H:
            full_t = torch.cat([full_t, torch.zeros((H - full_t.shape[0],) + tuple(full_t.shape[1:]), dtype=full_t.dtype, device=full_t.device)])
            full_s = torch.cat([full_s, torch.zeros((H - full_s.shape[0],) + tuple(full_s.shape[1:]), dtype=full_s.dtype, device=full_s.device)])
        return TensorXf(full_t.mean(dim=-1) if exh else full_s), TensorXf(full_t)

    def _render_pipelined(self, integ, sens, film, passes, total_spp, my_spp, nb, world, dev3=False):
        """bands of rows: render band b | reduce-scatter + develop + all-gather band b-1 on a side stream"""
        import torch
        import torch.distributed as dist
        from .tensor import TensorXf
        scene = self.scene
        W, H = film.size()
        T = film.temporal_bins
        # dev3: this rank's partial sums, already in the developed (H, W, T, 3) layout
        raw_t = film.developed_storage() if dev3 else film.transient_storage.torch_tensor()
        raw_s = film.steady_accum()
        dev = raw_t.device
        rows_b = H // nb
        gather = self.gather
        self.last_reduced_channels = 3 if dev3 else 4
        rank = dist.get_rank(self.group)
        per = rows_b // world                       # rows of a band this rank owns after the reduce-scatter
        # gather=False: "the single RCCL reduce" only — every rank keeps the developed rows it owns (band b: rows
        # b*rows_b + rank*per ... + per), stacked in band order; ``owned_rows`` lists them
        n_out = H if gather else nb * per
        out_t = torch.empty((n_out,) + tuple(film.raw_shape()[1:-1]) + (3,), dtype=torch.float32, device=dev)
        self.owned_rows = None if gather else [b * rows_b + rank * per + i for b in range(nb) for i in range(per)]
        self.last_collectives = nb * (2 if gather else 1) + 1
        main = torch.cuda.current_stream(dev)
        side = getattr(self, "_side_stream", None)
        if side is None:
            # high priority: its kernels (develop, copies; RCCL's own stream takes the process group's priority option) must
            # get onto the chip at the boundary between two persistent path kernels, which otherwise fill every CU first
            side = self._side_stream = torch.cuda.Stream(device=dev, priority=-1)
        gloo = dist.get_backend(self.group) == "gloo"
        # consecutive bands are launched on TWO alternating streams: the path kernel is persistent, so a band's launch ends
        # with a tail in which its workgroups run out of pixels one by one — the next band's workgroups fill those slots at
        # once instead of waiting for the whole launch to end (8 bands on one stream: +3.8 ms per render, measured)
        lanes = getattr(self, "_render_streams", None)
        if lanes is None:
            lanes = self._render_streams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        # ... but ONLY launches of the fused kernel may overlap (each takes its own work-ticket slot): the wavefront
        # organisation keeps one workspace per scene — path state, queues, records, segment tickets — so its bands stay
        # on ONE stream, in order
        fused = integ.resolved_mode(scene, sens, total_spp, my_spp, (0, rows_b * W)) == "fused"
        if not fused:
            lanes = (lanes[0], lanes[0])
        # (band completion words exist for a render that is ONE pass: accumulate() refuses them for a split one — and by then the
        # streams of this pipeline are set up on every rank — so a split render keeps the per-band launches)
        single = self.single_launch and fused and dev.type == "cuda" and H % nb == 0 and len(passes) == 1
        if single:
            lanes = (lanes[0], lanes[0])
        self.last_band_streams = 1 if lanes[0] is lanes[1] else 2
        self.last_band_launches = 1 if single else nb
        # the device counters are zeroed ONCE, before any band starts (a reset issued by band 0 on its own stream could
        # land after band 1, on the other stream, had begun to count)
        integ.reset_counters(film)
        start = torch.cuda.Event()
        start.record(main)
        for st in set(lanes):
            st.wait_event(start)                    # the clear of prepare() and the counter reset ran on the main stream
        band_ev = []
        band_words = None
        if single:
            # ONE launch over all rows; word b of band_words receives this render's epoch when band b is in the film
            if getattr(self, "_band_words", None) is None or self._band_words.numel() < nb:
                self._band_words = torch.zeros(nb, dtype=torch.int32, device=dev)
                self._band_epoch = 0
            band_words = self._band_words
            self._band_epoch = (self._band_epoch % 0x7FFFFFFE) + 1
            begin = torch.cuda.Event(enable_timing=True)
            done = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(lanes[0]):
                begin.record(lanes[0])
                integ.accumulate(scene, sens, passes, total_spp, spp_range=my_spp, pixel_range=(0, H * W),
                                 rows_are_zero=True, defer_stats="more", developed_partial=dev3,
                                 bands=(nb, self._band_epoch, band_words.data_ptr()))
                done.record(lanes[0])
            band_ev.append((begin, done))
        for b in range(nb):
            r0, r1 = b * rows_b, (b + 1) * rows_b
            ready = torch.cuda.Event(enable_timing=True)
            begin = torch.cuda.Event(enable_timing=True)
            if single:
                with torch.cuda.stream(side):
                    side.wait_event(start)
                    stream_wait_value(side, band_words.data_ptr() + 4 * b, self._band_epoch)
            else:
                with torch.cuda.stream(lanes[b & 1]):
                    begin.record(lanes[b & 1])
                    # every band owns its rows: they are still zero from prepare()'s clear when the band's only pass flushes
                    # them; no read-back per band (the launches stay asynchronous): counters sum on the device
                    integ.accumulate(scene, sens, passes, total_spp, spp_range=my_spp, pixel_range=(r0 * W, r1 * W),
                                     rows_are_zero=True, defer_stats="more", developed_partial=dev3)
                    ready.record(lanes[b & 1])
                band_ev.append((begin, ready))
            with torch.cuda.stream(side):
                if not single:
                    side.wait_event(ready)
                if gloo:
                    side.synchronize()              # gloo collectives are host-driven (CPU test path)
                # ONE collective per band and direction: the steady accumulator (4 MB for the whole image) is reduced once, below
                slab_t = reduce_scatter_rows(raw_t[r0:r1], self.group)          # THE film reduction, band b
                if dev3:                            # the sum of developed partial rows IS the developed row
                    d_t = slab_t
                    if not gather:
                        out_t[b * per:(b + 1) * per].copy_(d_t)
                elif gather:
                    d_t = _develop_rows(film, slab_t)
                else:
                    _develop_rows(film, slab_t, out=out_t[b * per:(b + 1) * per])
                if gather:
                    out_t[r0:r1].copy_(all_gather_rows(d_t, rows_b, self.group))
        # the steady image: every band has rendered -> one all-reduce of the (H, W, 4) sums, developed whole on every rank
        with torch.cuda.stream(side):
            for st in set(lanes):
                side.wait_stream(st)
            if gloo:
                side.synchronize()
            # a COPY: raw_s is the film's own steady accumulator and must keep this rank's partial sums, like transient_storage
            sums = raw_s.clone()
            dist.all_reduce(sums, group=self.group)
            full_s = film.develop_slab(None, sums)[1]
            out_s = full_s if gather else full_s[torch.as_tensor(self.owned_rows, device=dev)]
            # allocated under the side stream, read on the main one: the caching allocator must not hand the block to later
            # side-stream work while main still reads it
            if dev.type == "cuda":
                out_s.record_stream(main)
        main.wait_stream(side)
        for st in lanes:
            main.wait_stream(st)
        if integ.collect_stats:
            for st in lanes:
                st.synchronize()
            integ.fetch_counters(film)
            ms = [a.elapsed_time(b_) for a, b_ in band_ev]
            integ.total_times = {"total_ms": sum(ms), "trace_ms": sum(ms), "scatter_ms": 0.0, "trace_launches": len(band_ev),
                                 "scatter_launches": 0}
        if dev3:                                 # the film held this rank's PARTIAL sums: not a result anyone should develop
            film._developed = None
            film.direct_develop = False
        if film.exhaustive_scan:                 # transient_hdr_film.py:213-214
            return TensorXf(out_t.mean(dim=-1)), TensorXf(out_t)
        return TensorXf(out_s), TensorXf(out_t)
