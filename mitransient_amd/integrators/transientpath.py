"""``transient_path`` plugin (mitransient/integrators/transientpath.py).

``TransientPath.sample`` — the unidirectional path tracer with emitter sampling, BSDF
sampling, MIS and Russian roulette that splats every contribution into the time-resolved
film (:88-326) — is implemented by the HIP kernels in ``csrc/`` (see DESIGN.md for the
step-by-step map to the reference's loop).  This class carries the plugin's properties.
"""
from __future__ import annotations

from .common import TransientADIntegrator


class TransientPath(TransientADIntegrator):
    def sample(self, *args, **kwargs):
        raise NotImplementedError(
            "TransientPath.sample() is not traced in Python here: the whole loop of "
            "transientpath.py:140-319 runs inside mtr_render (HIP). Use render().")


def register():
    from ..plugins import register_integrator
    register_integrator("transient_path", lambda props: TransientPath(props))
