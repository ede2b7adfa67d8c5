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


class PerspectiveSensor:
    def __init__(self, sensor_dict, film, sampler):
        self.dict_ = sensor_dict
        self.film_ = film
        self.sampler_ = sampler

    def film(self):
        return self.film_

    def sampler(self):
        return self.sampler_
