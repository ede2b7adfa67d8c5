"""Host-side sensor + sampler objects (only what the transient_path hot path touches)."""
from __future__ import annotations

from ..scene import Properties


class IndependentSampler:
    """``independent`` sampler [mitsuba3: src/samplers/independent.cpp]: PCG32 streams seeded by
    TEA(base_seed + seed, lane).  The stream arithmetic itself is in the HIP kernels."""

    def __init__(self, props: Properties):
        self.sample_count_ = int(props.get("sample_count", 4))
        self.base_seed = int(props.get("seed", 0))
        self._wavefront_size = 0
        self._seed_value = self.base_seed
        self.samples_per_wavefront = 1

    def clone(self):
        s = IndependentSampler(Properties("independent", {"sample_count": self.sample_count_, "seed": self.base_seed}))
        return s

    def sample_count(self):
        return self.sample_count_

    def set_sample_count(self, spp):
        self.sample_count_ = int(spp)

    def set_samples_per_wavefront(self, n):
        self.samples_per_wavefront = int(n)

    def seed(self, seed, wavefront_size):
        self._seed_value = (self.base_seed + int(seed)) & 0xFFFFFFFF
        self._wavefront_size = int(wavefront_size)

    def seed_value(self):
        return self._seed_value

    # -- the few host-side draws the plugin layer makes itself (the seeder of multi-pass renders, common.py:72-75) --
    @staticmethod
    def _tea(v0, v1, rounds=4):
        M = 0xFFFFFFFF
        s = 0
        for _ in range(rounds):
            s = (s + 0x9E3779B9) & M
            v0 = (v0 + ((((v1 << 4) & M) + 0xA341316C) ^ ((v1 + s) & M) ^ ((v1 >> 5) + 0xC8013EA4))) & M
            v1 = (v1 + ((((v0 << 4) & M) + 0xAD90777D) ^ ((v0 + s) & M) ^ ((v0 >> 5) + 0x7E95761E))) & M
        return v0, v1

    def next_1d_first(self):
        """The FIRST ``next_1d()`` of every lane of the seeded wavefront, as float32 values: lane j's PCG32 stream is
        seeded with TEA(seed_value, j) [mitsuba3: independent.cpp seed(); drjit PCG32]; same integers as the kernels."""
        import numpy as np
        M64 = (1 << 64) - 1
        out = np.empty(self._wavefront_size, np.float32)
        for j in range(self._wavefront_size):
            v0, v1 = self._tea(self._seed_value, j)
            inc = ((v1 << 1) | 1) & M64
            state = (0 * 0x5851F42D4C957F2D + inc) & M64                 # state = 0; next(); state += initstate; next()
            state = (state + v0) & M64
            old = state = (state * 0x5851F42D4C957F2D + inc) & M64
            xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
            rot = old >> 59
            u = ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF
            out[j] = np.array([(u >> 9) | 0x3F800000], np.uint32).view(np.float32)[0] - np.float32(1.0)
        return out


class PerspectiveSensor:
    def __init__(self, sensor_dict, film, sampler):
        self.dict_ = sensor_dict
        self.film_ = film
        self.sampler_ = sampler

    def film(self):
        return self.film_

    def sampler(self):
        return self.sampler_
