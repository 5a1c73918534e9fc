"""C-ABI library: loads, exports every symbol include/rsr_mjx.h declares, host-side entry points work without
a GPU, and compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, make_blob


@pytest.fixture(scope="module")
def lib():
    from rsr_mjx_amd import _lib, build
    build.build()
    return _lib.lib()


def test_header_symbols_are_exported(lib):
    from rsr_mjx_amd import _lib
    header = open(os.path.join(ROOT, "include", "rsr_mjx.h")).read()
    declared = set(re.findall(r"\b(rsr_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert getattr(lib, s) is not None


def test_model_create_and_dims_on_host(lib, cube_model):
    from rsr_mjx_amd import _lib
    blob = make_blob(cube_model, episode_length=1200, auto_reset=True)
    h = C.c_void_p()
    _lib.check(lib.rsr_model_create(C.create_string_buffer(blob, len(blob)), len(blob), C.byref(h)))
    d = _lib.Dims()
    _lib.check(lib.rsr_model_dims(h, C.byref(d)))
    assert (d.nq, d.nv, d.nu, d.nbody, d.npair, d.obs_dim, d.n_frames, d.episode_length) == (22, 20, 5, 14, 45, 23, 4, 1200)
    assert d.rec_floats % 16 == 0 and d.rec_floats >= 2 * (22 + 20 + 5 + 20 + 1 + 42 + 3) + 2 * 23
    assert d.ncon_max >= 16 and d.nefc_max == 17 + 6 * d.ncon_max and 0 < d.lds_bytes <= 64 * 1024
    lib.rsr_model_destroy(h)


def test_errors_are_reported_not_thrown(lib, cube_model):
    h = C.c_void_p()
    assert lib.rsr_model_create(b"junkjunkjunkjunkjunkjunkjunkjunkjunk", 36, C.byref(h)) == -1
    assert b"RSRM" in lib.rsr_last_error()
    # a model whose dims have no compiled kernel
    from rsr_mjx_amd.model import model_fields, pack_blob
    from rsr_mjx_amd.envs.config import cube_env_fields
    f = model_fields(cube_model)
    f.update(cube_env_fields(cube_model))
    f["dims"] = f["dims"].copy(); f["dims"][1] = 19
    blob = pack_blob(f)
    assert lib.rsr_model_create(C.create_string_buffer(blob, len(blob)), len(blob), C.byref(h)) == -2
    assert lib.rsr_step(None, None, None) == -1


def test_no_cpu_fallback(lib, cube_model):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the failure path needs a GPU-less host")
    from rsr_mjx_amd import _lib
    blob = make_blob(cube_model)
    h, b = C.c_void_p(), C.c_void_p()
    _lib.check(lib.rsr_model_create(C.create_string_buffer(blob, len(blob)), len(blob), C.byref(h)))
    rc = lib.rsr_batch_create(h, 8, 0, None, C.byref(b))
    assert rc == -3 and b"no HIP device" in lib.rsr_last_error()
    lib.rsr_model_destroy(h)
    from rsr_mjx_amd.envs.airbot import AirbotPlayBase
    with pytest.raises(RuntimeError):
        AirbotPlayBase(device="cpu").batched(4)


def test_product_does_not_import_the_oracle():
    """The shipped package must not import, call or link anything under oracle/."""
    pkg = os.path.join(ROOT, "rsr_mjx_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "from oracle" not in text and "import oracle" not in text and "liboracle" not in text, fn
                assert "rsr_oracle.c" not in text, fn
