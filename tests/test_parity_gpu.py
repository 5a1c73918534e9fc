"""Parity of the HIP stepper (through the C ABI: rsr_reset / rsr_step / rsr_view) with the CPU oracle.

Tolerances are per-field envelopes derived from what was measured on the hardware -- max = 5 x the measured maximum, p99 = 3 x
the measured 99 % quantile; tests/parity_envelopes.py states the clauses for large and small samples and what the numbers mean
against the north_star's 1e-5: exact for done / steps / truncation / time and PRNG-only quantities; obs, reward,
metrics, xpos, info within 1e-5 on every env in a rollout; qvel / qacc_warmstart as documented deviations with quantile
bounds.  Parity with the reference (MJX) itself is unpinned -- see oracle/rsr_oracle.c.
"""
import os

import numpy as np
import pytest

from conftest import ROOT
import parity_envelopes as PE

pytestmark = pytest.mark.gpu

STRICT = ["xpos", "site_xpos", "obs", "reward", "metrics", "info_target_pos", "info_new_cube_pos",
          "info_site_pos", "info_cube_pos", "ctrl", "qpos"]
EXACT = ["done", "info_steps", "info_truncation", "info_episode_done", "time"]
SOLVER = ["qvel", "qacc_warmstart"]


def _check_fields(env, st, kind, phase, fields, tag=""):
    """Every field inside its measured envelope (tests/parity_envelopes.py)."""
    for k in fields:
        PE.check(kind, phase, k, _np(env, k, st[k]), st[k], tag=tag)


SHARED = STRICT + EXACT + SOLVER + ["info_episode_metrics", "first_qpos", "first_qvel", "first_ctrl",
                           "first_warmstart", "first_time", "first_xpos", "first_site_xpos", "first_obs"]


def _np(env, name, like):
    return env.view(name).detach().cpu().numpy().reshape(like.shape)


def _push(env, st):
    import torch
    for k in SHARED:
        env.view(k).copy_(torch.from_numpy(st[k].reshape(st[k].shape[0], -1)))


def _scaled_err(a, b):
    n = a.shape[0]
    a, b = a.reshape(n, -1).astype(np.float64), b.reshape(n, -1).astype(np.float64)
    return (np.abs(a - b) / np.maximum(1.0, np.abs(b).max(axis=1, keepdims=True))).max(axis=1)


@pytest.fixture(scope="module")
def setup(oracle_mod):
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from rsr_mjx_amd import prng
    from rsr_mjx_amd.envs.airbot import AirbotPlayBase, domain_randomize
    n = 1024
    envdef = AirbotPlayBase(device="cuda:0")
    dr = domain_randomize(envdef.sys, prng.split(prng.PRNGKey(5), n))
    env = envdef.batched(n, episode_length=1200, auto_reset=True, randomization=dr)
    orc = oracle_mod.Oracle(env.blob)
    orc.set_ncon_cap(env.dims.ncon_max)
    orc64 = oracle_mod.Oracle(env.blob, "f64")
    orc64.set_ncon_cap(env.dims.ncon_max)
    return dict(n=n, env=env, orc=orc, orc64=orc64, dr=dr, keys=prng.split(prng.PRNGKey(0), n), envdef=envdef)


def test_reset_parity(setup):
    import torch
    env, orc, n = setup["env"], setup["orc"], setup["n"]
    st = orc.new_state(n, setup["dr"])
    orc.reset(st, setup["keys"])
    env.reset(setup["keys"])
    torch.cuda.synchronize()
    for k in ("qvel", "ctrl", "first_qvel", "first_ctrl", "info_new_cube_pos"):     # pure PRNG + constants: bit exact
        np.testing.assert_array_equal(_np(env, k, st[k]), st[k], err_msg=k)
    for k in STRICT + ["first_qpos", "first_xpos", "first_site_xpos", "first_obs"]:
        assert _scaled_err(_np(env, k, st[k]), st[k]).max() <= 1e-5, k     # no dynamics yet: every env
    for k in EXACT:
        np.testing.assert_array_equal(_np(env, k, st[k]), st[k], err_msg=k)
    assert np.quantile(_scaled_err(_np(env, "qacc_warmstart", st["qacc_warmstart"]), st["qacc_warmstart"]), 0.99) < 1e-3


@pytest.mark.parametrize("depth", [0, 7, 60])
def test_teacher_forced_step_parity(setup, depth):
    """State after `depth` oracle steps is copied into the device record; one fused step on both sides."""
    import torch
    env, orc, orc64, n = setup["env"], setup["orc"], setup["orc64"], setup["n"]
    st = orc.new_state(n, setup["dr"])
    orc.reset(st, setup["keys"])
    env.reset(setup["keys"])
    rng = np.random.default_rng(100 + depth)
    for _ in range(depth):
        orc.step(st, np.clip(rng.normal(size=(n, 5)), -1, 1).astype(np.float32))
    _push(env, st)
    st64 = {k: (v.copy() if v is not None else None) for k, v in st.items()}
    act = np.clip(rng.normal(size=(n, 5)), -1, 1).astype(np.float32)
    orc.step(st, act)
    orc64.step(st64, act)
    env.step(None, act)
    torch.cuda.synchronize()
    assert int(env.view("stats")[:, 3].sum()) == 0 and int(st["stats"][:, 3].sum()) == 0, "contact capacity exceeded"
    phase = "reset" if depth == 0 else "rollout"
    _check_fields(env, st, "cube", phase, STRICT + SOLVER + ["info_site_pos"], tag=f"depth {depth}")
    for k in EXACT:
        np.testing.assert_array_equal(_np(env, k, st[k]), st[k], err_msg=k)
    for k in SOLVER:      # and the HIP stepper is as close to the fp64 oracle as the fp32 oracle is
        e_gpu = _scaled_err(_np(env, k, st[k]), st64[k])
        e_cpu = _scaled_err(st[k], st64[k])
        for q in (0.5, 0.9, 0.99):
            assert np.quantile(e_gpu, q) <= 1.5 * np.quantile(e_cpu, q) + 1e-6, (k, q, np.quantile(e_gpu, q), np.quantile(e_cpu, q))
    assert np.isfinite(env.record.cpu().numpy()).all()


def test_truncation_and_autoreset_on_device(setup, oracle_mod):
    """episode_length=5: both sides truncate at step 5 and restore the cached first state (wrapper parity)."""
    import torch
    from rsr_mjx_amd import prng
    n, L = 1024, 5
    env = setup["envdef"].batched(n, episode_length=L, auto_reset=True)
    orc = oracle_mod.Oracle(env.blob)
    keys = prng.split(prng.PRNGKey(2), n)
    st = orc.new_state(n)
    orc.reset(st, keys)
    state = env.reset(keys)
    rng = np.random.default_rng(2)
    first_obs = st["obs"].copy()
    for t in range(1, 2 * L + 1):
        act = rng.uniform(-1, 1, (n, 5)).astype(np.float32)
        _push(env, st)                       # teacher forcing keeps the two sides on the same trajectory
        orc.step(st, act)
        state = env.step(state, act)
        torch.cuda.synchronize()
        for k in EXACT:
            np.testing.assert_array_equal(_np(env, k, st[k]), st[k], err_msg=f"{k} at step {t}")
        _check_fields(env, st, "cube", "reset" if t % L == 1 else "rollout", ["obs"], tag=f"step {t}")
        assert _scaled_err(_np(env, "info_episode_metrics", st["info_episode_metrics"]), st["info_episode_metrics"]).max() <= 1e-5
        if t % L == 0:
            assert float(state.done.min()) == 1.0 and float(state.info["truncation"].min()) == 1.0
            np.testing.assert_array_equal(state.obs.cpu().numpy(), first_obs)
            np.testing.assert_array_equal(_np(env, "qpos", st["qpos"]), _np(env, "first_qpos", st["qpos"]))


def test_sf_variant_parity(setup, oracle_mod):
    """reference test/airbot.py (model test/sf.xml): same kernel, different prologue/epilogue constants and info."""
    import torch
    from rsr_mjx_amd import prng
    from rsr_mjx_amd.envs.airbot import AirbotPlaySF
    n = 1024
    env = AirbotPlaySF(device="cuda:0").batched(n, episode_length=1200, auto_reset=True)
    orc = oracle_mod.Oracle(env.blob)
    orc.set_ncon_cap(env.dims.ncon_max)
    keys = prng.split(prng.PRNGKey(21), n)
    st = orc.new_state(n)
    orc.reset(st, keys)
    state = env.reset(keys)
    rng = np.random.default_rng(21)
    fields = SHARED + ["info_last_action"]
    for t in range(12):
        if t == 6:      # put half of the cubes on their targets: hold / bonus / done branches
            st["qpos"][: n // 2, 15:18] = st["info_target_pos"][: n // 2] + np.float32(0.001)
            st["xpos"][: n // 2, 13] = st["qpos"][: n // 2, 15:18]
        for k in fields:
            env.view(k).copy_(torch.from_numpy(st[k].reshape(n, -1)))
        act = rng.uniform(-1, 1, (n, 5)).astype(np.float32)
        orc.step(st, act)
        state = env.step(state, act)
        torch.cuda.synchronize()
        _check_fields(env, st, "sf", "reset" if t == 0 else "rollout", ["obs", "reward", "metrics", "ctrl", "xpos"], tag=f"sf step {t}")
        assert _scaled_err(_np(env, "info_last_action", st["info_last_action"]), st["info_last_action"]).max() <= 1e-6
        for k in EXACT:
            np.testing.assert_array_equal(_np(env, k, st[k]), st[k], err_msg=f"{k} at step {t}")
    assert "last_action" in state.info


def test_tshape_parity(setup, oracle_mod):
    """BASELINE configs[2] model (T_shape.xml: 60 geom pairs, dt 2.5e-4, 8 Newton iterations): reset + teacher-forced steps."""
    import torch
    from rsr_mjx_amd import prng
    from rsr_mjx_amd.envs.airbot import AirbotTShape
    n = 1024
    env = AirbotTShape(device="cuda:0").batched(n, episode_length=1000, auto_reset=True)
    assert env.observation_size == 16 and env.dims.nv == 14 and env.dims.npair == 60
    orc = oracle_mod.Oracle(env.blob)
    orc.set_ncon_cap(env.dims.ncon_max)
    orc64 = oracle_mod.Oracle(env.blob, "f64")
    orc64.set_ncon_cap(env.dims.ncon_max)
    keys = prng.split(prng.PRNGKey(31), n)
    st = orc.new_state(n)
    orc.reset(st, keys)
    state = env.reset(keys)
    torch.cuda.synchronize()
    tfields = ["info_target_base_pos", "info_target_vertical_pos", "info_target_w", "info_new_T_pos", "info_T_pos", "info_xita"]
    common = [k for k in SHARED if not k.startswith("info_target_pos") and k not in ("info_new_cube_pos", "info_cube_pos")]
    for k in ["qpos", "xpos", "site_xpos", "obs", "first_obs"] + tfields:
        assert _scaled_err(_np(env, k, st[k]), st[k]).max() <= 1e-5, k
    rng = np.random.default_rng(31)
    for depth in (0, 9, 40):
        for _ in range(depth):
            orc.step(st, np.clip(rng.normal(size=(n, 5)), -1, 1).astype(np.float32))
        for k in common + tfields:
            env.view(k).copy_(torch.from_numpy(st[k].reshape(n, -1)))
        st64 = {k: (v.copy() if v is not None else None) for k, v in st.items()}
        act = np.clip(rng.normal(size=(n, 5)), -1, 1).astype(np.float32)
        orc.step(st, act); orc64.step(st64, act)
        state = env.step(state, act)
        torch.cuda.synchronize()
        assert int(env.view("stats")[:, 3].sum()) == 0
        _check_fields(env, st, "tshape", "reset" if depth == 0 else "rollout",
                      ["obs", "reward", "metrics", "xpos", "site_xpos", "qpos", "qvel", "qacc_warmstart", "ctrl", "info_xita", "info_new_T_pos", "info_T_pos"],
                      tag=f"tshape depth {depth}")
        for k in EXACT:
            np.testing.assert_array_equal(_np(env, k, st[k]), st[k], err_msg=k)
    assert set(state.info) >= {"target_base_pos", "target_vertical_pos", "target_w", "new_T_pos", "site_pos", "T_pos", "xita"}
    assert set(state.metrics) == {"push_reward", "siet2cube_reward", "health_reward", "task_complete_reward", "site_z_reward"}


def test_golden_fixture_configs0(setup):
    """BASELINE.json configs[0] (N=4, 200 steps): teacher-forced steps from the committed oracle snapshots."""
    import torch
    g = np.load(os.path.join(ROOT, "tests", "golden", "cube_n4_200.npz"))
    env = setup["envdef"].batched(4)
    env.reset(g["keys"])
    torch.cuda.synchronize()
    assert _scaled_err(env.view("obs").cpu().numpy(), g["reset_obs"]).max() <= 1e-5
    for t in (0, 1, 50, 100, 199):
        for f in ("qpos", "qvel", "ctrl", "qacc_warmstart", "time", "xpos", "site_xpos", "info_target_pos", "info_new_cube_pos"):
            a = g[f"pre{t}_{f}"]
            env.view(f).copy_(torch.from_numpy(a.reshape(4, -1)))
        env.step(None, g["actions"][t])
        torch.cuda.synchronize()
        np.testing.assert_array_equal(_np(env, "done", g[f"post{t}_done"]), g[f"post{t}_done"])
        for f in ("obs", "reward", "xpos", "qpos"):
            want = g[f"post{t}_{f}"]
            assert _scaled_err(_np(env, f, want), want).max() <= PE.bound("cube", "reset" if t == 0 else "rollout", f), (t, f)


def test_full_size_properties(setup):
    """N = 8192 (BASELINE headline size): determinism, shard invariance, finiteness -- size-independent properties."""
    import torch
    from rsr_mjx_amd import prng
    from rsr_mjx_amd.envs.airbot import domain_randomize
    n, sub = 8192, 1024
    keys = prng.split(prng.PRNGKey(3), n)
    dr = domain_randomize(setup["envdef"].sys, prng.split(prng.PRNGKey(4), n))
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    acts = torch.clamp(torch.randn((30, n, 5), generator=gen, device="cuda"), -1, 1)

    def rollout(lo, hi):
        e = setup["envdef"].batched(hi - lo, episode_length=10, auto_reset=True,
                                    randomization={k: v[lo:hi] for k, v in dr.items()})
        s = e.reset(keys[lo:hi])
        for t in range(30):
            s = e.step(s, acts[t, lo:hi])
        torch.cuda.synchronize()
        return e.record.clone()

    a, b = rollout(0, n), rollout(0, n)
    assert torch.equal(a, b), "same inputs must give bit-identical records"
    c = rollout(2048, 2048 + sub)
    assert torch.equal(a[2048:2048 + sub], c), "env i must not depend on the batch it is stepped in"
    assert torch.isfinite(a).all()
    env = setup["envdef"].batched(8)
    steps = a[:, (env.view("info_steps").data_ptr() - env.record.data_ptr()) // 4]
    assert float(steps.min()) == float(steps.max()) == 10.0       # 30 steps = 3 truncated episodes of 10


@pytest.mark.gpu
def test_bench_line_contract():
    """bench.py prints ONE JSON line with the fields the driver reads, the roofline and cpu_baseline objects included."""
    import json, subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--cpu-seconds", "1.0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and d["unit"] == "env-steps/s"
    assert "workload" in d["config"] and "model" not in d["config"] and "num_envs=8192" in d["config"]["workload"]
    assert abs(d["value"] - 8192 * 6 / (d["ms_per_step"] * 6e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["peak"] == 8000.0
    assert abs(r["achieved"] * 1e9 - 1220 * 8192 / (r["avg_launch_ms"] * 1e-3)) / (r["achieved"] * 1e9) < 1e-6
    assert r["avg_launch_ms"] <= d["ms_per_step"] * 1.05                      # HIP-event kernel time vs wall time per step
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "env-steps/s" and "oracle" in c["sample"]
    assert d["value"] > 20 * c["value"]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["cube", "tshape", "go2", "handstand"])
def test_odd_batch_sizes_and_argument_errors(setup, kind):
    """Batch sizes that are not multiples of anything (1, 3, 65): each env is the same env as in a larger batch; wrong
    action shapes and stepping before reset are refused on the host."""
    import torch
    from rsr_mjx_amd import prng
    if kind in ("cube", "tshape"):
        from rsr_mjx_amd.envs.airbot import AirbotTShape
        envdef, nu = (setup["envdef"] if kind == "cube" else AirbotTShape(device="cuda:0")), 5
        mk = lambda n: envdef.batched(n, episode_length=7, auto_reset=True)
    else:
        from rsr_mjx_amd.envs import go2
        envdef, nu = go2.load("Go2Handstand" if kind == "handstand" else "Go2JoystickFlatTerrain"), 12
        mk = lambda n: envdef.batched(n, episode_length=7, auto_reset=True)
    keys = prng.split(prng.PRNGKey(21), 65)
    acts = torch.clamp(torch.randn((12, 65, nu), generator=torch.Generator().manual_seed(1)) * 0.7, -1, 1).cuda()
    def run(idx):
        e = mk(len(idx))
        with pytest.raises(RuntimeError):
            e.step(None, acts[0, idx])
        s = e.reset(keys[idx])
        with pytest.raises(ValueError):
            e.step(s, acts[0, idx][:, :nu - 1])
        for t in range(12):
            s = e.step(s, acts[t, idx])
        torch.cuda.synchronize()
        assert torch.isfinite(e.view("obs")).all() and torch.isfinite(e.view("qpos")).all()
        return e.record.clone().view(torch.int32)      # bit patterns: the Go2 record carries PRNG key words, which are NaNs as floats
    full = run(list(range(65)))
    assert torch.equal(run([0]), full[0:1])
    assert torch.equal(run([3, 4, 5]), full[3:6])
    assert torch.equal(run([64]), full[64:65])


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n", [("tshape", 4096), ("go2", 8192), ("go2rough", 4096)])
def test_full_size_properties_other_workloads(kind, n):
    """BASELINE configs[2] / [3] / [4] at their full per-GPU sizes (T-shape 4096, Go2 flat 8192, Go2 rough 4096 per GPU):
    determinism, shard invariance (env i does not depend on the batch it is stepped in), finiteness, and the Episode /
    AutoReset bookkeeping -- the size-independent properties, since the oracle cannot run these sizes in seconds."""
    import torch
    from rsr_mjx_amd import prng
    if kind == "tshape":
        from rsr_mjx_amd.envs.airbot import AirbotTShape
        envdef, nu, astd = AirbotTShape(device="cuda:0"), 5, 1.0
        mk = lambda lo, hi: envdef.batched(hi - lo, episode_length=10, auto_reset=True)
    else:
        from rsr_mjx_amd.envs import go2
        envdef, nu, astd = go2.load("Go2JoystickRoughTerrain" if kind == "go2rough" else "Go2JoystickFlatTerrain"), 12, 0.5
        dr = go2.domain_randomize(envdef.sys, prng.split(prng.PRNGKey(4), n))
        mk = lambda lo, hi: go2.wrap_for_brax_training(envdef, hi - lo, episode_length=10, randomization_fn=lambda sys: {k: v[lo:hi] for k, v in dr.items()})
    keys = prng.split(prng.PRNGKey(3), n)
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    acts = torch.clamp(torch.randn((30, n, nu), generator=gen, device="cuda") * astd, -1, 1)

    def rollout(lo, hi):
        e = mk(lo, hi)
        s = e.reset(keys[lo:hi])
        for t in range(30):
            s = e.step(s, acts[t, lo:hi])
        torch.cuda.synchronize()
        return e, e.record.clone().view(torch.int32)        # bit patterns (the Go2 record carries PRNG key words)

    ea, a = rollout(0, n)
    _, b = rollout(0, n)
    assert torch.equal(a, b), "same inputs must give bit-identical records"
    sub = 1024
    _, c = rollout(n // 2, n // 2 + sub)
    assert torch.equal(a[n // 2:n // 2 + sub], c), "env i must not depend on the batch it is stepped in"
    for f in ("qpos", "qvel", "obs", "reward", "xpos"):
        assert torch.isfinite(ea.view(f)).all(), f
    steps, trunc = ea.view("info_steps"), ea.view("info_truncation")
    done = ea.view("done")
    assert float(steps.max()) <= 10.0 and float(steps.min()) >= 1.0
    # 30 steps with episode_length 10: an env that never terminated early sits exactly on its third truncation
    never = (steps[:, 0] == 10.0)
    assert bool(never.any()) and bool((done[never, 0] == 1.0).all())
    assert int(ea.view("stats")[:, 3].min()) >= 0, "work-queue hand-off timed out"


@pytest.mark.gpu
def test_tshape_reset_solver_quality(oracle_mod):
    """Why the T-shape env-step straight after reset is the loosest envelope: the reference's reset pose has the T block 6-12 mm
    inside the table, the Newton cost is ~4e4 with row terms up to 1e8, and its minimum is flat at fp32 resolution -- all the
    stages before the solve agree with the oracle to 1e-6 (tools/gpu_tshape_reset_diag.py), the solve's qacc differs by up to
    1e-4 relative between any two fp32 implementations.  What can be pinned is how good a minimiser the kernel returns: its
    optimality gap, cost(qacc_hip) - cost(qacc_fp64) evaluated in double on the fp64 oracle's rows, is as small as the fp32
    oracle's own."""
    import torch
    from rsr_mjx_amd import prng
    from rsr_mjx_amd.envs.airbot import AirbotTShape
    n = 256
    env = AirbotTShape(device="cuda:0", n_frames=1).batched(n, episode_length=1000, auto_reset=True)
    orc = oracle_mod.Oracle(env.blob); orc.set_ncon_cap(env.dims.ncon_max)
    o64 = oracle_mod.Oracle(env.blob, "f64"); o64.set_ncon_cap(env.dims.ncon_max)
    keys = prng.split(prng.PRNGKey(7), n)
    st = orc.new_state(n); orc.reset(st, keys)
    state = env.reset(keys)
    for k in SHARED:
        if k in st and st[k] is not None and k in env._views and env.view(k).numel() > 0 and not k.startswith("info_target_pos") and k not in ("info_new_cube_pos", "info_cube_pos"):
            env.view(k).copy_(torch.from_numpy(st[k].reshape(n, -1)))
    pre = {k: (v.copy() if v is not None else None) for k, v in st.items()}
    dbg = env.enable_debug(True)
    act = np.clip(np.random.default_rng(7).normal(size=(n, 5)), -1, 1).astype(np.float32)
    orc.step(st, act)
    env.step(state, act)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    nv = env.dims.nv
    gap_hip, gap_f32, pen = [], [], []
    for e in range(n):
        o64.forward(pre["qpos"][e], pre["qvel"][e], st["ctrl"][e], pre["qacc_warmstart"][e])
        c64 = o64.cost(o64.get("qacc"))
        gap_hip.append((o64.cost(d[e, 800:800 + nv]) - c64) / c64)
        orc.forward(pre["qpos"][e], pre["qvel"][e], st["ctrl"][e], pre["qacc_warmstart"][e])
        gap_f32.append((o64.cost(orc.get("qacc")) - c64) / c64)
        con = o64.get("contacts").reshape(-1, 10)
        pen.append(con[:, 0].min() if len(con) else 0.0)
    gap_hip, gap_f32 = np.abs(np.array(gap_hip)), np.abs(np.array(gap_f32))
    assert np.min(pen) < -0.005, "the reset pose interpenetrates (the premise of this test)"
    assert gap_hip.max() <= 3e-5, float(gap_hip.max())                      # measured: fp32 oracle up to 9e-6
    assert np.quantile(gap_hip, 0.9) <= 3.0 * np.quantile(gap_f32, 0.9) + 1e-7, (np.quantile(gap_hip, 0.9), np.quantile(gap_f32, 0.9))
    # and every stage before the solve agrees: mass matrix and smooth force of the dump against the oracle's
    for e in (0, n // 2, n - 1):
        orc.forward(pre["qpos"][e], pre["qvel"][e], st["ctrl"][e], pre["qacc_warmstart"][e])
        assert np.abs(d[e, 128:128 + nv * nv] - orc.get("M")).max() <= 1e-6 * max(1.0, np.abs(orc.get("M")).max())
        assert np.abs(d[e, 736:736 + nv] - orc.get("qfrc_smooth")).max() <= 1e-5 * max(1.0, np.abs(orc.get("qfrc_smooth")).max())


@pytest.mark.gpu
def test_contact_capacity_overflow_is_reported(setup, oracle_mod):
    """More active contacts than the kernel's LDS layout holds (24): the kernel keeps the first 24 in pair order, drops the rest
    and SAYS so in stats[3] (on the bench workloads that is 2e-6 of the env-steps).  Poses with 25-31 contacts (the arm folded
    into the table; tests/golden/make_overcap_states.py): the dropped count is exactly the oracle's uncapped count minus the
    capacity, the result equals the oracle's under the same cap, and its distance to the uncapped oracle -- the price of the
    drop -- is finite, visible and bounded."""
    import torch
    g = np.load(os.path.join(ROOT, "tests", "golden", "cube_overcap_states.npz"))
    n = g["qpos"].shape[0]
    from rsr_mjx_amd.envs.airbot import AirbotPlayBase
    env = AirbotPlayBase(device="cuda:0", n_frames=1).batched(n, episode_length=1200, auto_reset=True)    # one substep: stats = that pose's
    cap = env.dims.ncon_max
    orc = oracle_mod.Oracle(env.blob); orc.set_ncon_cap(cap)
    from rsr_mjx_amd import prng
    keys = prng.split(prng.PRNGKey(1), n)
    st = orc.new_state(n); orc.reset(st, keys)
    state = env.reset(keys)
    st["qpos"][:] = g["qpos"]; st["qvel"][:] = 0; st["qacc_warmstart"][:] = 0
    st["ctrl"][:] = g["qpos"][:, [0, 1, 2, 4, 5]]
    _push(env, st)
    free = {k: (v.copy() if v is not None else None) for k, v in st.items()}
    act = np.zeros((n, 5), dtype=np.float32)
    orc.step(st, act)
    try:
        orc.set_ncon_cap(1000)
        orc.step(free, act)
    finally:
        orc.set_ncon_cap(cap)
    env.step(state, act)
    torch.cuda.synchronize()
    stats = env.view("stats").cpu().numpy()
    # the last substep's counts: capacity reached, and the kernel counted what it dropped exactly as the oracle did
    assert (stats[:, 3] > 0).sum() >= n // 2 and (stats[:, 2] <= cap).all(), stats[:, 2:4].tolist()
    np.testing.assert_array_equal(stats[:, 3], st["stats"][:, 3])
    np.testing.assert_array_equal(stats[:, 2], st["stats"][:, 2])
    over = stats[:, 3] > 0
    assert (free["stats"][over, 2] == stats[over, 2] + stats[over, 3]).all(), "dropped = uncapped count - capacity"
    # same cap on both sides: the usual parity (these poses are violent: penetrations of centimetres, |qacc| ~ 1e5)
    for k in ("qpos", "xpos", "obs"):
        assert _scaled_err(_np(env, k, st[k]), st[k]).max() <= 1e-3, k
    # against the uncapped oracle the drop shows, and stays bounded over one env-step
    d = _scaled_err(_np(env, "qpos", free["qpos"]), free["qpos"])
    assert np.isfinite(d).all() and d.max() <= 0.2, float(d.max())
    print(f"capacity overflow: dropped {stats[over, 3].tolist()} contacts; qpos distance to the uncapped oracle after one env-step: "
          f"max {d.max():.2e}, median {np.median(d):.2e}")


@pytest.mark.gpu
def test_arm_touching_the_cube_takes_the_coupled_factorisation(setup, oracle_mod):
    """Dims::ROWTREE: while no contact joins two kinematic trees the Hessian is factored one tree per DPP row; a contact between
    the arm and the cube switches to the coupled layout (0.4 % of the bench's env-steps, so random rollouts hardly test it).
    Here the cube is slid towards the gripper in every env until the oracle reports the first arm-cube contact (penetration
    below 2 mm: deep interpenetration is chaotic on both sides), and the step from there must land on the oracle."""
    import torch
    from rsr_mjx_amd.envs.airbot import AirbotPlayBase
    from rsr_mjx_amd import prng
    n = 64
    env = AirbotPlayBase(device="cuda:0").batched(n, episode_length=1200, auto_reset=True)
    orc = oracle_mod.Oracle(env.blob); orc.set_ncon_cap(env.dims.ncon_max)
    orc.set_ls_cycle(True); orc.set_ls_rule(2, 1.0)                       # the kernel's line-search rules
    keys = prng.split(prng.PRNGKey(7), n)
    st = orc.new_state(n); orc.reset(st, keys)
    state = env.reset(keys)
    cube_b = 13                                                            # body ids: arm 1..10, cube 13
    st["qvel"][:] = 0; st["qacc_warmstart"][:] = 0

    def arm_cube_contact(e):
        orc.forward(st["qpos"][e].astype(np.float64), st["qvel"][e].astype(np.float64), st["ctrl"][e].astype(np.float64), st["qacc_warmstart"][e].astype(np.float64))
        con = orc.get("contacts").reshape(-1, 10)
        b1, b2 = con[:, 7].astype(int), con[:, 8].astype(int)
        hit = (((b1 == cube_b) & (b2 >= 1) & (b2 <= 10)) | ((b2 == cube_b) & (b1 >= 1) & (b1 <= 10))) & (con[:, 0] < 0)
        return bool(hit.any())

    touching = 0
    for e in range(n):
        site = st["site_xpos"][e, 0].copy()
        st["qpos"][e, 18:22] = [1, 0, 0, 0]
        for d in np.arange(0.15, -0.002, -0.002):                          # from clear of the gripper towards the site, along -x
            st["qpos"][e, 15:18] = [site[0] - d, site[1], 0.82]
            if arm_cube_contact(e):
                touching += 1
                break
    assert touching >= n // 2, f"only {touching} of {n} envs reach an arm-cube contact"
    _push(env, st)
    rng = np.random.default_rng(7)
    act = np.clip(rng.normal(0, 1, (n, 5)), -1, 1).astype(np.float32)
    orc.step(st, act)
    env.step(state, torch.from_numpy(act).cuda())
    torch.cuda.synchronize()
    for k in ("qpos", "xpos", "obs", "reward"):
        e = _scaled_err(_np(env, k, st[k]), st[k])
        assert np.quantile(e, 0.9) <= 1e-4 and e.max() <= 2e-2, (k, float(np.quantile(e, 0.9)), float(e.max()))
        print(f"arm-cube contact step, {k}: median {np.median(e):.1e} p90 {np.quantile(e, 0.9):.1e} max {e.max():.1e} ({touching} of {n} envs touching)")


def _queue_env(kind, n, seed):
    from rsr_mjx_amd import prng
    from rsr_mjx_amd.envs.airbot import AirbotPlayBase, AirbotTShape, domain_randomize
    envdef = (AirbotTShape if kind == "tshape" else AirbotPlayBase)(device="cuda:0")
    dr = domain_randomize(envdef.sys, prng.split(prng.PRNGKey(seed), n)) if kind == "cube" else None
    return envdef.batched(n, episode_length=7, auto_reset=True, randomization=dr)


@pytest.mark.parametrize("kind", ["cube", "tshape"])
def test_schedule_changes_are_bit_identical(kind):
    """include/rsr_mjx.h, rsr_batch_set_schedule: "bit-identical for every value and for any sequence of values between
    steps".  One batch runs every step as one unit; the other changes units between steps of ONE rollout -- among them the
    4 -> 3 and 4 -> 1 -> 1 -> 2 orders in which a flag word that depended on `units` would meet a stale equal value -- across
    truncation and auto-reset (episode_length 7) -- and moves the boundary between the envs stepped as one unit and the envs cut
    into phases (rsr_batch_set_whole_envs) at every step.  Records are compared as int32 after every step; no hand-off wait
    times out."""
    import torch
    from rsr_mjx_amd import prng
    n, steps = 2048, 30
    a, b = _queue_env(kind, n, 11), _queue_env(kind, n, 11)
    keys = prng.split(prng.PRNGKey(12), n)
    a.reset(keys); b.reset(keys)
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    acts = torch.clamp(torch.randn((steps, n, 5), generator=gen, device="cuda"), -1, 1)
    order = [4, 3, 2, 4, 1, 1, 2, 4, 4, 3, 1, 2, 2, 4, 3]
    whole = [0, -1, 100, n // 2, n, 0, 1, n - 1, -1, 700]        # rsr_batch_set_whole_envs: envs stepped as one unit each
    a.set_schedule(1)
    for t in range(steps):
        b.set_schedule(order[t % len(order)]); b.set_whole_envs(whole[t % len(whole)])
        a.step(None, acts[t]); b.step(None, acts[t])
        assert torch.equal(a.record.view(torch.int32), b.record.view(torch.int32)), (kind, "step", t, "units", order[t % len(order)])
    assert float(a.view("info_truncation").sum()) >= 0 and float(a.view("info_steps").max()) <= 7.0
    assert a.handoff_timeouts() == 0 and b.handoff_timeouts() == 0
    assert int(b.view("stats")[:, 3].min()) >= 0
    b.check()


def test_handoff_timeout_is_sticky_and_visible():
    """The work queue's error path, exercised once: with the poll bound cut to about two milliseconds (far above a unit's ~50 us,
    so no healthy wait gives up: all 4 x 512 units of this batch are resident at once and every phase 1 waits a whole unit) and
    phase 0 of one env never publishing its flag, that env's phase 1 gives up.  The host must be able to see it -- stats[env, 3] == -1 after the step
    (the later phases inherit the mark; the last phase does not overwrite it), rsr_batch_check counts the timeouts and
    fails -- and every other env must be bit-identical to a clean batch's."""
    import torch
    from rsr_mjx_amd import prng
    n, victim = 512, 137
    clean, hurt = _queue_env("cube", n, 21), _queue_env("cube", n, 21)
    keys = prng.split(prng.PRNGKey(22), n)
    clean.reset(keys); hurt.reset(keys)
    act = torch.clamp(torch.randn((n, 5), generator=torch.Generator(device="cuda").manual_seed(4), device="cuda"), -1, 1)
    clean.set_schedule(4); hurt.set_schedule(4)
    clean.set_whole_envs(0); hurt.set_whole_envs(0)       # (a batch that fits the resident waves is not split by default: no hand-offs)
    hurt.set_fault_injection(spin_cap=8192, withhold_env=victim)
    clean.step(None, act); hurt.step(None, act)
    st = hurt.view("stats").cpu().numpy()
    assert st[victim, 3] == -1, st[victim]
    others = np.arange(n) != victim
    assert (st[others, 3] >= 0).all()
    # (the env's later phases are resident too in so small a batch and started polling at the same moment: they give up as well)
    nt = hurt.handoff_timeouts()
    assert 1 <= nt <= 3 and clean.handoff_timeouts() == 0, nt
    with pytest.raises(RuntimeError, match="hand-off"):
        hurt.check()
    ra, rb = clean.record.view(torch.int32).cpu().numpy(), hurt.record.view(torch.int32).cpu().numpy()
    np.testing.assert_array_equal(ra[others], rb[others])
    # the mark is per step: with the hook off the env steps cleanly again, the batch's count stays (sticky)
    hurt.set_fault_injection()
    hurt.step(None, act)
    assert int(hurt.view("stats")[victim, 3]) >= 0 and hurt.handoff_timeouts() == nt


def test_envelopes_were_measured_on_these_kernel_sources():
    """tests/golden/parity_envelopes.json carries the hash of the kernel sources its numbers were measured on (bench.csrc_sha16:
    rsr_mjx_amd/csrc/*.hip, *.hpp and include/rsr_mjx.h).  A kernel edit without a re-measurement (tools/gpu_round_evidence.sh:
    tools/gpu_parity_stats.py -> tools/make_parity_envelopes.py) fails here, so the bounds the other tests enforce always describe
    the kernel they are enforced on."""
    import bench
    assert PE.ENV["_provenance"]["csrc_sha16"] == bench.csrc_sha16(), (
        "parity envelopes were measured on other kernel sources: re-run tools/gpu_parity_stats.py --json and tools/make_parity_envelopes.py")


from wrappers_np import np_repeat_step as _np_repeat_step      # brax EpisodeWrapper / AutoResetWrapper restated in numpy


class _HipPlainEnv:
    """env.step of the plain (unwrapped) HIP env on a numpy state dict: push, one fused step, pull."""

    def __init__(self, env, fields):
        self.env, self.fields = env, fields

    def step(self, st, act):
        import torch
        for k in self.fields:
            self.env.view(k).copy_(torch.from_numpy(st[k].reshape(st[k].shape[0], -1)))
        self.env.step(None, act)
        torch.cuda.synchronize()
        for k in self.fields:
            st[k][...] = _np(self.env, k, st[k])


@pytest.mark.parametrize("repeat", [2, 3])
def test_action_repeat_on_device(oracle_mod, repeat):
    """include/rsr_mjx.h, rsr_batch_set_action_repeat (brax EpisodeWrapper(env, episode_length, action_repeat), RSR/train.py:224-229):
    episode_length 6 with repeats of 2 and 3, so truncation falls on an outer step, over three truncation boundaries.
    (a) The wrapper kernels, bit for bit: the batch with action_repeat against the numpy restatement of the wrapper
        (_np_repeat_step) around the PLAIN HIP env stepped `repeat` times -- same physics kernels, so every field of the record
        must agree exactly, the reward being the repeats' rewards summed in order.
    (b) The independent side: the same restatement around the plain ORACLE env, teacher-forced per outer step -- counters exact;
        obs / reward within the envelopes' scale (the repeats run open-loop, and straight after reset the arm's ill-conditioned
        solve separates any two fp32 steppers by 1e-2 in qvel: DESIGN.md 2)."""
    import torch
    from rsr_mjx_amd import prng
    from rsr_mjx_amd.envs.airbot import AirbotPlayBase, wrap
    n, L = 1024, 6
    envdef = AirbotPlayBase(device="cuda:0")
    env = wrap(envdef, n, episode_length=L, action_repeat=repeat)
    env_o = wrap(envdef, n, episode_length=L, action_repeat=repeat)        # (b): teacher-forced from the oracle side
    plain = envdef.batched(n)
    orc_w = oracle_mod.Oracle(env.blob)                                     # reset with the wrappers' bookkeeping
    orc_p = oracle_mod.Oracle(plain.blob)                                   # plain env.step
    orc_p.set_ncon_cap(env.dims.ncon_max)                                   # (the oracle's switches are library-wide: set, not assumed)
    keys = prng.split(prng.PRNGKey(31), n)
    st = orc_w.new_state(n)
    orc_w.reset(st, keys)
    env.reset(keys); env_o.reset(keys); plain.reset(keys)
    torch.cuda.synchronize()
    carried = SHARED + ["info_last_action"]
    hs = {k: (v.copy() if v is not None else None) for k, v in st.items()}
    for k in carried:
        hs[k][...] = _np(env, k, hs[k])
    hip_plain = _HipPlainEnv(plain, carried)
    rng = np.random.default_rng(31)
    pipeline = ["qpos", "qvel", "ctrl", "qacc_warmstart", "time", "xpos", "site_xpos", "obs"]
    for t in range(1, 3 * L // repeat + 2):
        act = rng.uniform(-1, 1, (n, 5)).astype(np.float32)
        # (a)
        _np_repeat_step(hip_plain, hs, act, repeat, L, pipeline)
        env.step(None, act)
        torch.cuda.synchronize()
        for k in carried:
            np.testing.assert_array_equal(_np(env, k, hs[k]).view(np.int32), hs[k].view(np.int32), err_msg=f"(a) {k} at outer step {t}")
        # (b)
        _push(env_o, st)
        _np_repeat_step(orc_p, st, act, repeat, L, pipeline)
        state = env_o.step(None, act)
        torch.cuda.synchronize()
        for k in EXACT:
            np.testing.assert_array_equal(_np(env_o, k, st[k]), st[k], err_msg=f"(b) {k} at outer step {t}")
        np.testing.assert_array_equal(_np(env_o, "info_episode_metrics", st["info_episode_metrics"])[:, 1], st["info_episode_metrics"][:, 1])
        e_r = _scaled_err(_np(env_o, "reward", st["reward"]), st["reward"])
        e_o = _scaled_err(_np(env_o, "obs", st["obs"]), st["obs"])
        lim = max(PE.bound("cube", "reset", "obs"), PE.bound("cube", "rollout", "obs"))
        assert np.quantile(e_o, 0.99) <= lim and np.quantile(e_r, 0.99) <= 1e-3, (t, float(np.quantile(e_o, 0.99)), float(np.quantile(e_r, 0.99)))
        if (t * repeat) % L == 0:
            assert float(state.done.min()) == 1.0 and float(state.info["truncation"].min()) == 1.0
            np.testing.assert_array_equal(_np(env_o, "qpos", st["qpos"]), _np(env_o, "first_qpos", st["qpos"]))
            np.testing.assert_array_equal(_np(env_o, "obs", st["obs"]), _np(env_o, "first_obs", st["obs"]))
    assert int(st["info_steps"].max()) <= L and int(st["info_episode_metrics"][:, 1].max()) <= L
    with pytest.raises(RuntimeError):
        env.set_action_repeat(0)
