"""Plugin registry keyed by the reference's type strings (``mi.register_integrator`` /
``mi.register_film``; mitransient/integrators/transientpath.py:329, films/transient_hdr_film.py:314)."""
from __future__ import annotations

_integrators = {}
_films = {}


def register_integrator(name, ctor):
    _integrators[name] = ctor


def register_film(name, ctor):
    _films[name] = ctor


def _unknown(kind, name, table):
    raise ValueError(f"failed to instantiate unknown plugin of type \"{name}\" "
                     f"(registered {kind}s: {sorted(table)})")


def create_integrator(name, props):
    if name not in _integrators:
        _unknown("integrator", name, _integrators)
    return _integrators[name](props)


def create_film(name, props):
    if name not in _films:
        _unknown("film", name, _films)
    return _films[name](props)
