import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; built from oracle/mtr_oracle.c)."""
    from oracle import oracle as o
    o.build()
    o.lib()
    return o


@pytest.fixture(scope="session")
def host_harness():
    """TEST-ONLY host build of the product's mtr_core.h arithmetic (tests/host_harness.cpp)."""
    import __graft_entry__ as g
    path = g.build_host_harness()
    lib = C.CDLL(path)
    return lib


def make_cornell(width=64, height=64, bins=64, start=3.5, window=6.0, **integrator):
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    mi.set_variant("llvm_ad_rgb")
    d = mitr.cornell_box()
    d["sensor"]["film"].update(width=width, height=height, temporal_bins=bins, start_opl=start,
                               bin_width_opl=window / bins)
    d["integrator"].update(integrator)
    return mi.load_dict(d)


@pytest.fixture
def cornell_c1():
    """BASELINE config 1: Cornell box 64x64, 64 bins, (16 spp)."""
    return make_cornell()


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def hh_render(lib, sd, params):
    from mitransient_amd import _cabi
    f = sd.film
    t4 = np.zeros((f.height, f.width, f.temporal_bins, 4), np.float32)
    s4 = np.zeros((f.height, f.width, 4), np.float32)
    cnt = _cabi.mtr_counters()
    d = sd.desc()
    fp = C.POINTER(C.c_float)
    rc = lib.hh_render(C.byref(d), C.byref(params), t4.ctypes.data_as(fp), s4.ctypes.data_as(fp), C.byref(cnt))
    assert rc == 0
    return t4, s4, cnt.as_dict()
