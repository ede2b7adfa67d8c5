"""The C-ABI library loads and exports every symbol include/mitransient_amd.h declares.
(No compute calls here: there is no GPU in the CPU test environment and no CPU fallback.)"""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mitransient_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mtr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    from mitransient_amd import _cabi
    assert os.path.exists(_cabi.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_cabi.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(_cabi.EXPORTS) == declared
    lib.mtr_abi_version.restype = ctypes.c_int
    assert lib.mtr_abi_version() == _cabi.MTR_ABI_VERSION


def test_struct_layouts_match_header():
    """sizeof() of the ctypes mirrors against the C compiler's view of the header."""
    import subprocess
    import tempfile
    from mitransient_amd import _cabi
    names = ["mtr_material", "mtr_emitter", "mtr_camera", "mtr_film_desc", "mtr_scene_desc", "mtr_shape", "mtr_nlos_desc", "mtr_texture",
             "mtr_render_params", "mtr_counters", "mtr_splat_soa", "mtr_kernel_times"]
    prog = '#include <stdio.h>\n#include "mitransient_amd.h"\nint main(){' + "".join(
        f'printf("%zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(prog)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    for n, sz in zip(names, sizes):
        assert ctypes.sizeof(getattr(_cabi, n)) == sz, n


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the product refuses to run (it must never fall back to the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import mitransient_amd as mitr
    import mitransient_amd.mi as mi
    from mitransient_amd._cabi import MitransientAMDError
    mi.set_variant("llvm_ad_rgb")
    scene = mi.load_dict(mitr.cornell_box())
    with pytest.raises(MitransientAMDError):
        mi.render(scene, spp=1)
    from mitransient_amd import _cabi
    lib = _cabi.load_library()
    h = ctypes.c_void_p()
    assert lib.mtr_ctx_create(0, ctypes.byref(h)) == -2          # MTR_ERR_NO_DEVICE
    assert b"no HIP device" in lib.mtr_last_error(None)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mitransient_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "libmtr_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def _build_abi_smoke():
    """tests/abi_smoke.c: the boundary from plain C (gcc, the header, the HIP runtime's C API) — no Python, no torch"""
    import subprocess
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "abi_smoke")
    libdir = os.path.join(ROOT, "mitransient_amd", "csrc")
    cmd = ["gcc", "-O1", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "abi_smoke.c"),
           "-o", exe, "-L", libdir, "-lmitransient_amd", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_c_smoke_links_against_the_header_and_fails_loudly_without_a_gpu():
    import subprocess
    import torch
    exe = _build_abi_smoke()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_c_smoke_renders_on_the_gpu")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 3 and "no CPU path" in r.stdout, r.stdout


@pytest.mark.gpu
def test_c_smoke_renders_on_the_gpu():
    import subprocess
    exe = _build_abi_smoke()
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "abi_smoke: PASS" in r.stdout, r.stdout


def test_release_library_reads_no_experiment_knobs():
    """the MTR_* environment knobs of the measurement scripts exist only in builds with -DMTR_EXPERIMENTS (mtr_knobs.h): the
    product must not change its performance or its results with a stray environment variable"""
    from mitransient_amd import _cabi
    blob = open(os.path.join(os.path.dirname(_cabi.__file__), "csrc", "libmitransient_amd.so"), "rb").read()
    for name in (b"MTR_FUSED_PER_CU", b"MTR_FUSED_CHUNK", b"MTR_NO_WIDE8Q", b"MTR_NO_BOX_NODES", b"MTR_WF_SEG", b"MTR_WF_TILE_LOG2",
                 b"MTR_BVH_LEAF", b"MTR_WIDE_WIDTH", b"MTR_NO_FLAT", b"MTR_NO_GREY", b"MTR_FUSED_G", b"MTR_FUSED_OLD_PLAN", b"MTR_FUSED_VERBOSE", b"MTR_WF_SORT", b"MTR_BVH_THREADS", b"MTR_BVH_DEPTH_BUDGET"):
        assert name not in blob, name
    src = os.path.join(os.path.dirname(_cabi.__file__), "csrc")
    for f in os.listdir(src):
        if f.endswith((".hip", ".cpp", ".h")) and f != "mtr_knobs.h":
            assert "getenv" not in open(os.path.join(src, f)).read(), f
