"""Rollout harness over the batched HIP stepper: the counterparts of brax.training.acting's `actor_step`,
`generate_unroll` and `Evaluator` (+ brax.envs.wrappers.training.EvalWrapper) that the reference learners
call at RSR/train.py:310-330 (unroll with extra_fields=('truncation',)) and RSR/train.py:441-447 (evaluator).

What differs from the reference, and why: the reference traces `env.step` under `jax.lax.scan`; here the
scan is a host loop that enqueues one `rsr_step` launch per env-step on the current HIP stream, and the
per-step slices of the Transition are written into preallocated [T, N, ...] device tensors (the env's State
tensors are views into the batch record that the next step overwrites, so slices are copied, never aliased).
Nothing in the loop synchronises with the host.

A policy is `policy(obs, key) -> (action, policy_extras)` with obs / action torch tensors on the env's
device and `key` a uint32[2] numpy key (prng.split chain identical to the reference's
`current_key, next_key = jax.random.split(current_key)`).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, Optional, Sequence, Tuple

import numpy as np

from . import prng

Policy = Callable[[Any, np.ndarray], Tuple[Any, Dict[str, Any]]]


@dataclass
class Transition:
    """brax.training.types.Transition; every leaf has leading dims [unroll_length, num_envs]."""
    observation: Any
    action: Any
    reward: Any
    discount: Any
    next_observation: Any
    extras: Dict[str, Dict[str, Any]] = field(default_factory=dict)


def _obs(state):
    return state.obs


def actor_step(env, env_state, policy: Policy, key: np.ndarray, extra_fields: Sequence[str] = ()):
    """One policy + env step.  Returns (next_state, Transition of [N, ...] tensors, all freshly allocated)."""
    obs = _obs(env_state).clone()
    actions, policy_extras = policy(obs, key)
    nstate = env.step(env_state, actions)
    state_extras = {x: nstate.info[x].clone() for x in extra_fields}
    return nstate, Transition(
        observation=obs, action=actions, reward=nstate.reward.clone(), discount=1.0 - nstate.done,
        next_observation=_obs(nstate).clone(),
        extras={"policy_extras": policy_extras, "state_extras": state_extras})


def generate_unroll(env, env_state, policy: Policy, key: np.ndarray, unroll_length: int,
                    extra_fields: Sequence[str] = (), extra_obs: Optional[Dict[str, Callable[[Any], Any]]] = None):
    """`unroll_length` actor steps; returns (final_state, Transition with leaves [T, N, ...]).

    Matches brax acting.generate_unroll: step t uses `current_key` of the chain
    `current_key, next_key = split(current_key)`, the transition stores obs_t, action_t, reward_{t+1},
    discount = 1 - done_{t+1}, obs_{t+1} (after auto-reset, as the wrapped env returns it) and
    extras.state_extras[x] = info[x] after the step.  `extra_obs` records further observation streams (e.g. the Go2
    `privileged_state` an asymmetric critic reads): extras["extra_obs"][name] before and extras["next_extra_obs"][name]
    after every step."""
    import torch
    extra_obs = extra_obs or {}
    if unroll_length < 1:
        raise ValueError("unroll_length must be >= 1")
    state = env_state
    bufs: Dict[str, Any] = {}

    def put(name, t, value):
        if name not in bufs:
            bufs[name] = torch.empty((unroll_length,) + tuple(value.shape), dtype=value.dtype, device=value.device)
        bufs[name][t].copy_(value)

    cur = np.asarray(key, dtype=np.uint32)
    for t in range(unroll_length):
        ks = prng.split(cur, 2)
        step_key, cur = ks[0], ks[1]
        put("observation", t, _obs(state))
        for name, fn in extra_obs.items():
            put("xobs/" + name, t, fn(state))
        actions, policy_extras = policy(bufs["observation"][t], step_key)
        state = env.step(state, actions)
        put("action", t, actions)
        put("reward", t, state.reward)
        put("done", t, state.done)
        put("next_observation", t, _obs(state))
        for name, fn in extra_obs.items():
            put("nxobs/" + name, t, fn(state))
        for x in extra_fields:
            put("state/" + x, t, state.info[x])
        for k, v in policy_extras.items():
            put("policy/" + k, t, v)
    data = Transition(
        observation=bufs["observation"], action=bufs["action"], reward=bufs["reward"], discount=1.0 - bufs["done"],
        next_observation=bufs["next_observation"],
        extras={"policy_extras": {k[7:]: v for k, v in bufs.items() if k.startswith("policy/")},
                "state_extras": {k[6:]: v for k, v in bufs.items() if k.startswith("state/")},
                **({"extra_obs": {k[5:]: v for k, v in bufs.items() if k.startswith("xobs/")},
                    "next_extra_obs": {k[6:]: v for k, v in bufs.items() if k.startswith("nxobs/")}} if extra_obs else {})})
    return state, data


def generate_unroll_pipelined(envs: Sequence[Any], env_states: Sequence[Any], policy: Policy, key: np.ndarray,
                              unroll_length: int, extra_fields: Sequence[str] = ()):
    """`generate_unroll` over several independent sub-batches (airbot.wrap_sub_batches), each on its own HIP stream:
    policy and env step of sub-batch k are enqueued on stream k, so sub-batch A's policy / launch tail overlaps sub-batch
    B's env step.  Step t of every sub-batch uses the same key of the chain (the reference splits one key per step for
    the whole batch); a policy that draws noise from it should fold in the sub-batch's env offset.  Returns
    (final_states, Transition) with the sub-batches concatenated back along the env axis, env i in the same place as in
    the lock-step unroll."""
    import torch
    parts = len(envs)
    streams = [torch.cuda.Stream() if torch.cuda.is_available() else None for _ in range(parts)]
    main = torch.cuda.current_stream() if torch.cuda.is_available() else None
    bufs = [dict() for _ in range(parts)]
    states = list(env_states)

    def put(k, name, t, value):
        b = bufs[k]
        if name not in b:
            b[name] = torch.empty((unroll_length,) + tuple(value.shape), dtype=value.dtype, device=value.device)
        b[name][t].copy_(value)

    cur = np.asarray(key, dtype=np.uint32)
    for st in streams:
        if st is not None:
            st.wait_stream(main)
    for t in range(unroll_length):
        ks = prng.split(cur, 2)
        step_key, cur = ks[0], ks[1]
        for k in range(parts):
            ctx = torch.cuda.stream(streams[k]) if streams[k] is not None else _NullCtx()
            with ctx:
                state = states[k]
                put(k, "observation", t, _obs(state))
                actions, policy_extras = policy(bufs[k]["observation"][t], step_key)
                state = envs[k].step(state, actions)
                put(k, "action", t, actions); put(k, "reward", t, state.reward); put(k, "done", t, state.done)
                put(k, "next_observation", t, _obs(state))
                for x in extra_fields:
                    put(k, "state/" + x, t, state.info[x])
                for name, v in policy_extras.items():
                    put(k, "policy/" + name, t, v)
                states[k] = state
    for st in streams:
        if st is not None:
            main.wait_stream(st)
    cat = lambda name: torch.cat([b[name] for b in bufs], dim=1)
    names = list(bufs[0])
    data = Transition(
        observation=cat("observation"), action=cat("action"), reward=cat("reward"), discount=1.0 - cat("done"),
        next_observation=cat("next_observation"),
        extras={"policy_extras": {n[7:]: cat(n) for n in names if n.startswith("policy/")},
                "state_extras": {n[6:]: cat(n) for n in names if n.startswith("state/")}})
    return states, data


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


@dataclass
class EvalMetrics:
    """brax.envs.wrappers.training.EvalMetrics."""
    episode_metrics: Dict[str, Any]
    active_episodes: Any
    episode_steps: Any


class EvalWrapper:
    """brax EvalWrapper: accumulates reward and metrics of the FIRST episode of every env (an env stops
    contributing once it has been done), and its length.  info['eval_metrics'] carries the running sums."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, keys):
        import torch
        state = self.env.reset(keys)
        names = ["reward"] + list(state.metrics)
        zero = torch.zeros_like(state.reward)
        state.info["eval_metrics"] = EvalMetrics(
            episode_metrics={k: zero.clone() for k in names}, active_episodes=torch.ones_like(state.reward),
            episode_steps=zero.clone())
        return state

    def step(self, state, action):
        import torch
        em: EvalMetrics = state.info["eval_metrics"]
        if not isinstance(em, EvalMetrics):
            raise ValueError(f"Incorrect type for state_metrics: {type(em)}")
        nstate = self.env.step(state, action)
        cur = dict(nstate.metrics)
        cur["reward"] = nstate.reward
        active = em.active_episodes
        steps = nstate.info.get("steps")
        steps = torch.zeros_like(active) if steps is None else steps.to(active.dtype)
        em.episode_steps = torch.where(active > 0, steps, em.episode_steps)
        for k in em.episode_metrics:
            em.episode_metrics[k] = em.episode_metrics[k] + cur[k] * active
        em.active_episodes = active * (1.0 - nstate.done)
        nstate.info["eval_metrics"] = em
        return nstate


class Evaluator:
    """brax.training.acting.Evaluator on the batched stepper (RSR/train.py:441-447).

    `eval_env` is an already wrapped batch (episode + auto-reset) of num_eval_envs envs; `eval_policy_fn(params)`
    returns a Policy.  `key_fanout(key, n)` makes the per-env reset keys (default prng.split, as the reference does)."""

    def __init__(self, eval_env, eval_policy_fn: Callable[[Any], Policy], num_eval_envs: int, episode_length: int,
                 action_repeat: int, key: np.ndarray):
        self._key = np.asarray(key, dtype=np.uint32)
        self._eval_walltime = 0.0
        self._env = EvalWrapper(eval_env)
        self._policy_fn = eval_policy_fn
        self._num_eval_envs = num_eval_envs
        self._unroll_length = episode_length // action_repeat
        self._steps_per_unroll = episode_length * num_eval_envs

    def _generate_eval_unroll(self, policy_params, key):
        policy = self._policy_fn(policy_params)
        state = self._env.reset(prng.split(key, self._num_eval_envs))
        cur = key
        for _ in range(self._unroll_length):
            ks = prng.split(cur, 2)
            step_key, cur = ks[0], ks[1]
            actions, _ = policy(_obs(state), step_key)
            state = self._env.step(state, actions)
        return state

    def run_evaluation(self, policy_params, training_metrics: Dict[str, Any], aggregate_episodes: bool = True) -> Dict[str, Any]:
        import torch
        ks = prng.split(self._key, 2)
        self._key, unroll_key = ks[0], ks[1]
        t = time.time()
        eval_state = self._generate_eval_unroll(policy_params, unroll_key)
        em: EvalMetrics = eval_state.info["eval_metrics"]
        torch.cuda.synchronize() if em.active_episodes.is_cuda else None
        epoch_eval_time = time.time() - t
        metrics: Dict[str, Any] = {}
        for fn, suffix in ((np.mean, ""), (np.std, "_std")):
            for name, value in em.episode_metrics.items():
                v = value.detach().cpu().numpy()
                metrics[f"eval/episode_{name}{suffix}"] = fn(v) if aggregate_episodes else v
        metrics["eval/avg_episode_length"] = float(np.mean(em.episode_steps.detach().cpu().numpy()))
        metrics["eval/epoch_eval_time"] = epoch_eval_time
        metrics["eval/sps"] = self._steps_per_unroll / epoch_eval_time
        self._eval_walltime += epoch_eval_time
        return {"eval/walltime": self._eval_walltime, **training_metrics, **metrics}
