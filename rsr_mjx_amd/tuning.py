"""Environment-parameter tuning on the batched stepper: the counterpart of reference RSR/rsr_pipeline.py:49-206
(`env_params_tuning`, called from test/rsr_env_params_tuning.py:114-124), RSR's "step 3".

The reference differentiates one `env.step` with `jax.grad` and runs optax Adam (lr 0.005) on
    loss(p) = sum_i | w . (obs_pred_i(p) - next_obs_true_i) |,   w = [1]*6 + [10]*3 + [0]*3 + [10]*5 + [0]*6,
where obs_pred_i is the observation after one step from a state rebuilt from the logged observation i (`obs2state`) with
the friction of the model's LAST geom set to p.  A ctypes stepper is not differentiable by tracing; a batched one does not
need to be: every data point is replicated once per perturbed parameter value (p, p +- eps e_k) in ONE batch, so a
central-difference gradient costs a single `rsr_step` launch per optimiser step.

Data format (README.md:333-357 of the reference): text files with one comma-separated row of floats per line.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional, Sequence, Tuple

import numpy as np

from . import prng

OBS_WEIGHTS = np.array([1] * 6 + [10] * 3 + [0] * 3 + [10] * 5 + [0] * 6, dtype=np.float32)   # rsr_pipeline.py:123


def txt_to_2d_array(path: str) -> np.ndarray:
    """test/rsr_env_params_tuning.py:57-70: comma-separated floats, one row per non-empty line."""
    rows = []
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append([float(x) for x in line.split(",")])
    return np.asarray(rows, dtype=np.float32)


def adam_fd_minimise(loss_many: Callable[[np.ndarray], np.ndarray], p0, p_min, p_max, num_steps: int, lr: float = 0.005,
                     fd_eps: float = 1e-3, log: Optional[Callable[[str], None]] = None) -> Tuple[np.ndarray, Dict[str, list]]:
    """Adam (optax defaults b1 0.9, b2 0.999, eps 1e-8) on a loss whose gradient is taken by central differences.
    `loss_many(P)` evaluates the loss for every row of P [nvar, nparam] at once and returns [nvar]."""
    p = np.atleast_1d(np.asarray(p0, dtype=np.float64)).copy()
    lo = np.broadcast_to(np.asarray(p_min, dtype=np.float64), p.shape)
    hi = np.broadcast_to(np.asarray(p_max, dtype=np.float64), p.shape)
    k = p.size
    m, v = np.zeros(k), np.zeros(k)
    hist: Dict[str, list] = {"loss": [], "params": []}
    for it in range(1, num_steps + 1):
        P = np.tile(p, (2 * k + 1, 1))
        for j in range(k):
            P[1 + 2 * j, j] += fd_eps
            P[2 + 2 * j, j] -= fd_eps
        L = np.asarray(loss_many(P), dtype=np.float64)
        g = np.array([(L[1 + 2 * j] - L[2 + 2 * j]) / (2 * fd_eps) for j in range(k)])
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        step = lr * (m / (1 - 0.9 ** it)) / (np.sqrt(v / (1 - 0.999 ** it)) + 1e-8)
        p = np.clip(p - step, lo, hi)                                  # rsr_pipeline.py:160 clip after the update
        hist["loss"].append(float(L[0])); hist["params"].append(p.copy())
        if log is not None:
            log(f"step {it - 1}: params = {p}. loss = {L[0]}.")
    return p, hist


class _StepLoss:
    """One-step prediction loss of a dataset for many friction values at once, on the batched stepper."""

    def __init__(self, env_def, obs: np.ndarray, actions: np.ndarray, next_obs_true: np.ndarray, nvar: int, geom: int = -1):
        import torch
        self.torch = torch
        self.nd, self.nvar = obs.shape[0], nvar
        n = self.nd * nvar
        self.env = env_def.batched(n)                                   # raw env: no episode / auto-reset wrappers, as the reference
        from .model import unpack_blob
        ids = unpack_blob(self.env.blob)["env_ids"]
        cube_body, boxq, jq = int(ids[0]), int(ids[3]), [int(x) for x in ids[6:12]]
        key0 = np.tile(prng.PRNGKey(0)[None], (n, 1))
        self.state = self.env.reset(key0)                               # rsr_pipeline.py:79-85 obs2state: reset(PRNGKey(0)) ...
        pipe = {f: self.env.view(f).clone() for f in ("qpos", "qvel", "ctrl", "qacc_warmstart", "time", "xpos", "site_xpos")}
        self.env.step(self.state, torch.zeros((n, self.env.dims.nu), device=self.env.device))     # ... and one zero-action step
        for f, v in pipe.items():                                       # pipeline_state = state_0's, info / obs = state_1's
            self.env.view(f).copy_(v)
        o = torch.as_tensor(np.repeat(obs, nvar, axis=0), device=self.env.device)                 # data point i -> rows i*nvar ..
        q = self.env.view("qpos")
        q[:, jq] = o[:, 0:6]
        q[:, boxq:boxq + 3] = o[:, 12:15]
        self.env.view("xpos").view(n, -1, 3)[:, cube_body] = o[:, 12:15]
        self.saved = self.env.record.clone()
        self.actions = torch.as_tensor(np.repeat(actions, nvar, axis=0), device=self.env.device).contiguous()
        self.target = torch.as_tensor(np.repeat(next_obs_true, nvar, axis=0), device=self.env.device)
        self.w = torch.as_tensor(OBS_WEIGHTS, device=self.env.device)
        sys = env_def.sys
        self.geom = geom % sys.ngeom
        self.friction = torch.as_tensor(np.tile(sys.arrays["geom_friction"].astype(np.float32)[None], (n, 1, 1)), device=self.env.device)
        self.env.set_randomization({"geom_friction": self.friction})
        self._fr_view = self.env._dr["geom_friction"].view(n, sys.ngeom, 3)

    def __call__(self, P: np.ndarray) -> np.ndarray:
        torch = self.torch
        assert P.shape[0] == self.nvar
        rows = torch.as_tensor(np.ascontiguousarray(P, dtype=np.float32), device=self.env.device)
        fr = rows if rows.shape[1] == 3 else rows[:, :1].expand(self.nvar, 3)                   # a scalar sets the whole row
        self._fr_view[:, self.geom] = fr.repeat(self.nd, 1)
        self.env.record.copy_(self.saved)
        self.env.step(self.state, self.actions)
        err = (self.env.view("obs") - self.target) @ self.w              # jnp.dot(w, error), then |.|
        return err.abs().view(self.nd, self.nvar).sum(0).cpu().numpy()


def env_params_tuning(init_env, num_steps: int, init_env_params, env_params_min, env_params_max, obs, actions, next_obs_true,
                      log_path: Optional[str] = None, geom: int = -1, fd_eps: float = 1e-3, lr: float = 0.005, verbose: bool = True):
    """Same arguments and return value as the reference function: (tuned_env_params, train_log)."""
    obs, actions, next_obs_true = (np.asarray(x, dtype=np.float32) for x in (obs, actions, next_obs_true))
    k = np.atleast_1d(np.asarray(init_env_params)).size
    loss = _StepLoss(init_env, obs, actions, next_obs_true, nvar=2 * k + 1, geom=geom)

    def log(line: str):
        if verbose:
            print(line)
        if log_path:
            with open(log_path, "a") as f:
                f.write(line + "\n")
    p, hist = adam_fd_minimise(loss, init_env_params, env_params_min, env_params_max, num_steps, lr=lr, fd_eps=fd_eps, log=log)
    out = p if np.ndim(init_env_params) else float(p[0])
    return out, {"loss": hist["loss"], "params": hist["params"]}
