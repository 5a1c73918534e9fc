import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REF = "/root/reference"
CUBE_XML = os.path.join(REF, "ppo_train/airbot_training/cube.xml")
ASSETS = os.path.join(ROOT, "rsr_mjx_amd", "assets")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cube_model():
    """Compiled Airbot cube model: the committed asset (compiler output), never the reference XML at run time."""
    from rsr_mjx_amd.mjcf import CompiledModel
    return CompiledModel.load(os.path.join(ASSETS, "airbot_cube.npz"))


@pytest.fixture(autouse=True)
def _oracle_defaults():
    """The oracle's switches are library-wide: every test starts from the defaults and sets what it needs."""
    from oracle import oracle as O
    O.reset_switches()
    yield


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def sf_model():
    from rsr_mjx_amd.mjcf import CompiledModel
    return CompiledModel.load(os.path.join(ASSETS, "airbot_sf.npz"))


@pytest.fixture(scope="session")
def tshape_model():
    from rsr_mjx_amd.mjcf import CompiledModel
    return CompiledModel.load(os.path.join(ASSETS, "airbot_tshape.npz"))


@pytest.fixture(scope="session")
def go2_model():
    from rsr_mjx_amd.envs import config
    from rsr_mjx_amd.mjcf import CompiledModel
    return config.go2_apply_overrides(CompiledModel.load(os.path.join(ASSETS, "go2_flat.npz")), config.GO2_DEFAULT_CONFIG)


def make_go2_blob(model, episode_length=0, auto_reset=False, overrides=None):
    from rsr_mjx_amd.envs import config
    from rsr_mjx_amd.model import model_fields, pack_blob
    c = config._merge(config.GO2_DEFAULT_CONFIG, overrides or {})
    f = model_fields(model)
    f.update(config.go2_env_fields(model, c, episode_length, auto_reset))
    return pack_blob(f)


def make_blob(model, kind="cube", **env_kwargs):
    from rsr_mjx_amd.envs import config
    from rsr_mjx_amd.model import model_fields, pack_blob
    f = model_fields(model)
    f.update({"cube": config.cube_env_fields, "sf": config.sf_env_fields, "tshape": config.tshape_env_fields}[kind](model, **env_kwargs))
    return pack_blob(f)


def random_state(model, rng, spread=1.0):
    """A random but sane configuration around the reset pose of the cube env."""
    nq, nv = model.nq, model.nv
    qpos = model.arrays["qpos0"].copy()
    qpos[:6] += np.array([0, -0.5422302, 0.45173569, 1.5718, -1.4794435, 1.1731174])
    qpos[:8] += rng.uniform(-0.3, 0.3, 8) * spread
    qpos[6], qpos[7] = 0.033, -0.033
    for a in (8, 15):
        qpos[a:a + 3] += rng.uniform(-0.05, 0.05, 3) * spread
        q = qpos[a + 3:a + 7] + rng.uniform(-0.3, 0.3, 4) * spread
        qpos[a + 3:a + 7] = q / np.linalg.norm(q)
    qvel = rng.uniform(-1, 1, nv) * spread
    return qpos, qvel
