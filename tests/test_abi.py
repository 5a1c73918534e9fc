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
    assert lib.rsr_batch_check(None, None, None) == -1 and lib.rsr_batch_set_fault_injection(None, 0, -1) == -1


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


def test_malformed_blobs_are_refused(lib, cube_model):
    """rsr_model_create validates the blob's directory before reading through it: truncated blobs, entries pointing outside
    the blob, absurd entry counts and missing fields all return RSR_ERR_ARG (-1) with a message, never a crash."""
    import struct
    blob = make_blob(cube_model, episode_length=1200, auto_reset=True)
    h = C.c_void_p()
    ok = lambda b: lib.rsr_model_create(C.create_string_buffer(bytes(b), len(b)), len(b), C.byref(h))
    assert ok(blob) == 0
    lib.rsr_model_destroy(h)
    nent = struct.unpack_from("<i", blob, 8)[0]
    assert nent > 50
    # (1) cut in the middle of the entry table / of the data
    assert ok(blob[:16 + 56 * 3 + 10]) == -1
    assert ok(blob[:len(blob) // 2]) == -1
    # (2) entry count far larger than the blob
    b = bytearray(blob); struct.pack_into("<i", b, 8, 10 ** 8)
    assert ok(b) == -1 and b"entry table" in lib.rsr_last_error()
    # (3) one entry's offset / count pointing past the end
    for field_off, val in ((48, len(blob)), (44, 10 ** 8), (48, -4)):       # blob_entry: name[40], dtype, count, offset, reserved
        b = bytearray(blob); struct.pack_into("<i", b, 16 + 56 * 5 + field_off, val)
        assert ok(b) == -1, (field_off, val)
    # (4) a required field renamed away
    b = bytearray(blob)
    idx = bytes(b).find(b"opt_integrator\0")
    assert idx > 0
    b[idx:idx + 3] = b"xxx"
    assert ok(b) == -1 and b"opt_integrator" in lib.rsr_last_error()
    # (5) a name without terminator
    b = bytearray(blob); b[16:16 + 40] = b"a" * 40
    assert ok(b) == -1


def test_go2_model_create_checks_the_block_arrow_structure(lib):
    """The Go2 kernels factor M and H in block-arrow form (trunk of 6 dofs, legs of 3 that couple only through the trunk):
    rsr_model_create accepts the shipped model and refuses one whose body chains or contact pairs tie two legs together."""
    import numpy as np
    from rsr_mjx_amd.envs import config, go2
    from rsr_mjx_amd.model import model_fields, pack_blob
    env = go2.load("Go2JoystickFlatTerrain")
    f = model_fields(env.sys)
    f.update(config.go2_env_fields(env.sys, env._config, 1000, True))
    h = C.c_void_p()
    create = lambda fields: (lambda b: lib.rsr_model_create(C.create_string_buffer(b, len(b)), len(b), C.byref(h)))(pack_blob(fields))
    assert create(f) == 0
    lib.rsr_model_destroy(h)
    bad = dict(f); m1 = np.array(f["pair_mask1"]).copy()
    m1[0] = int(m1[0]) | (7 << 6) | (7 << 9)                     # a contact pair whose chain holds the dofs of two legs
    bad["pair_mask1"] = m1
    assert create(bad) == -2 and b"legs" in lib.rsr_last_error()
    bad = dict(f); bm = np.array(f["body_dofmask"]).copy()
    bm[-1] = int(bm[-1]) | (7 << 6)                               # the last calf's chain also runs through the first leg
    bad["body_dofmask"] = bm
    assert create(bad) == -2


def test_airbot_model_create_checks_the_tree_ranges(lib, cube_model):
    """The Airbot kernels factor one kinematic tree per DPP row (arm | target | cube over fixed dof ranges): a model whose body
    chain straddles two ranges is refused."""
    import numpy as np
    from rsr_mjx_amd.envs import config
    from rsr_mjx_amd.model import model_fields, pack_blob
    f = model_fields(cube_model); f.update(config.cube_env_fields(cube_model))
    h = C.c_void_p()
    create = lambda fields: (lambda b: lib.rsr_model_create(C.create_string_buffer(b, len(b)), len(b), C.byref(h)))(pack_blob(fields))
    assert create(f) == 0
    lib.rsr_model_destroy(h)
    bad = dict(f); bm = np.array(f["body_dofmask"]).copy()
    bm[-1] = int(bm[-1]) | 1                                      # the last body's chain also runs through the arm's first dof
    bad["body_dofmask"] = bm
    assert create(bad) == -2 and b"trees" in lib.rsr_last_error()
