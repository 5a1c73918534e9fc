"""PPO training loop with the RSR term on the batched stepper (torch-ROCm): the counterpart of reference RSR/train.py:76-503
(`train(environment, num_timesteps, episode_length, past_data, ...)`), itself Brax's PPO trainer with `past_data` and
`rsr_loss_scale` threaded into the loss.

Same schedule as the reference: every training step gathers `batch_size * num_minibatches // num_envs` unrolls of
`unroll_length` steps (train.py:310-324), folds the new observations into the running normaliser (:332-336), then runs
`num_updates_per_batch` passes of `num_minibatches` Adam steps over the shuffled batch (:262-300).  Networks follow
brax.training.agents.ppo.networks.make_ppo_networks defaults: policy MLP (32,)*4 -> 2*action_size, value MLP (256,)*5 -> 1,
swish activations, NormalTanh action distribution.  What differs: the scans are host loops over torch modules, sampling
noise comes from a torch generator seeded from the jax-style key (not bit-compatible with jax.random.normal), one GPU per
process (the reference pmaps over local devices).
"""
from __future__ import annotations

import time
from typing import Any, Callable, Dict, Optional, Sequence, Tuple

import numpy as np

from .. import prng
from ..rollout import Evaluator, Transition, generate_unroll
from . import ppo_losses


class RunningStatistics:
    """brax.training.acme.running_statistics: count / mean / summed_variance with the batched Welford update, std clipped
    to [1e-6, 1e6]."""

    def __init__(self, size: int, device=None):
        import torch
        self.count = torch.zeros((), dtype=torch.float32, device=device)
        self.mean = torch.zeros(size, dtype=torch.float32, device=device)
        self.summed_variance = torch.zeros(size, dtype=torch.float32, device=device)
        self.std = torch.ones(size, dtype=torch.float32, device=device)

    def update(self, batch) -> None:
        import torch
        x = batch.reshape(-1, batch.shape[-1]).to(torch.float32)
        x = x[torch.isfinite(x).all(dim=-1)]            # one non-finite row would poison the statistics for good
        n = x.shape[0]
        if n == 0:
            return
        # in place: a captured HIP graph of the update step reads these tensors by address
        self.count.add_(n)
        diff_old = x - self.mean
        self.mean.add_(diff_old.sum(0) / self.count)
        self.summed_variance.add_((diff_old * (x - self.mean)).sum(0))
        self.std.copy_(torch.sqrt(torch.clamp(self.summed_variance, min=0.0) / self.count).clamp(1e-6, 1e6))

    def normalize(self, x):
        return (x - self.mean) / self.std


def make_mlp(sizes: Sequence[int], device=None):
    """flax MLP with swish between layers, lecun_uniform kernels, zero biases (brax networks.MLP)."""
    import torch
    layers = []
    for i in range(len(sizes) - 1):
        lin = torch.nn.Linear(sizes[i], sizes[i + 1])
        bound = float(np.sqrt(3.0 / sizes[i]))
        torch.nn.init.uniform_(lin.weight, -bound, bound)
        torch.nn.init.zeros_(lin.bias)
        layers.append(lin)
        if i < len(sizes) - 2:
            layers.append(torch.nn.SiLU())
    return torch.nn.Sequential(*layers).to(device)


class PPONetworks:
    def __init__(self, observation_size: int, action_size: int, device=None, policy_hidden_layer_sizes=(32,) * 4,
                 value_hidden_layer_sizes=(256,) * 5):
        self.policy = make_mlp([observation_size, *policy_hidden_layer_sizes, 2 * action_size], device)
        self.value = make_mlp([observation_size, *value_hidden_layer_sizes, 1], device)
        self.action_size = action_size
        self.value_normalizer = None      # RunningStatistics of the critic's own observation key, when it has one (train())

    def parameters(self):
        return list(self.policy.parameters()) + list(self.value.parameters())


def make_inference_fn(networks: PPONetworks, normalizer: Optional[RunningStatistics]):
    """ppo_networks.make_inference_fn: make_policy(params_unused, deterministic) -> policy(obs, key) -> (action, extras)."""
    import torch

    def make_policy(_params=None, deterministic: bool = False):
        def policy(obs, key):
            with torch.no_grad():
                x = normalizer.normalize(obs) if normalizer is not None else obs
                logits = networks.policy(x)
                if deterministic:
                    return ppo_losses.tanh_normal_mode(logits), {}
                loc, scale = ppo_losses._split(logits)
                gen = torch.Generator(device=obs.device)
                gen.manual_seed(int(np.asarray(key, dtype=np.uint32)[0]) << 32 | int(np.asarray(key, dtype=np.uint32)[1]))
                raw = loc + scale * torch.randn(loc.shape, generator=gen, device=obs.device, dtype=loc.dtype)
                return torch.tanh(raw), {"log_prob": ppo_losses.tanh_normal_log_prob(logits, raw), "raw_action": raw}
        return policy
    return make_policy


class _GraphedUpdate:
    """One PPO minibatch update (loss, backward, clipping, non-finite guard, Adam step, metric sums) captured once as a HIP
    graph and replayed: the update is ~500 small kernels (the GAE scan alone is ~10 per time step), so in eager mode its cost is
    launch latency, not arithmetic.  Inputs are copied into static buffers; everything the captured code reads by address
    (parameters, Adam moments, normaliser statistics, RSR reference data) is updated in place elsewhere."""

    def __init__(self, loss_fn, optimizer, params, example: Dict[str, Any], max_grad_norm):
        import torch
        self.static = {k: torch.zeros_like(v) for k, v in example.items()}
        self.acc: Dict[str, Any] = {}
        self._loss_fn, self._opt, self._params, self._gn = loss_fn, optimizer, params, max_grad_norm
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                                  # warm-up outside the capture (allocator, lazy Adam state)
            for k, v in example.items():
                self.static[k].copy_(v)
            snapshot = [p.detach().clone() for p in params]
            for _ in range(3):
                self._one(accumulate=False)
            for p, q in zip(params, snapshot):                         # the warm-up must not count as training
                p.data.copy_(q)
            for st in optimizer.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        torch.cuda.current_stream().wait_stream(side)
        self._one(accumulate=False, dry=True)                          # creates the accumulators
        for v in self.acc.values():
            v.zero_()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self._one(accumulate=True)

    def _one(self, accumulate: bool, dry: bool = False):
        import torch
        loss, m = self._loss_fn(self.static)
        if dry:
            for k, v in m.items():
                self.acc[k] = torch.zeros_like(v.detach())
            self.acc["skipped_updates"] = torch.zeros((), device=loss.device)
            return
        self._opt.zero_grad(set_to_none=True)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(self._params, self._gn if self._gn else float("inf"))
        ok = torch.isfinite(gn)
        for p_ in self._params:
            if p_.grad is not None:
                p_.grad.copy_(torch.where(ok, p_.grad, torch.zeros_like(p_.grad)))
        self._opt.step()
        if accumulate:
            for k, v in m.items():
                self.acc[k].add_(v.detach())
            self.acc["skipped_updates"].add_((~ok).to(torch.float32))

    def __call__(self, batch: Dict[str, Any]):
        for k, v in batch.items():
            self.static[k].copy_(v)
        self.graph.replay()


def train(environment, num_timesteps: int, episode_length: int, past_data: Any = None, action_repeat: int = 1, num_envs: int = 1,
          num_eval_envs: int = 128, learning_rate: float = 1e-4, entropy_cost: float = 1e-4, discounting: float = 0.9, seed: int = 0,
          unroll_length: int = 10, batch_size: int = 32, num_minibatches: int = 16, num_updates_per_batch: int = 2, num_evals: int = 1,
          normalize_observations: bool = False, reward_scaling: float = 1.0, clipping_epsilon: float = 0.3, gae_lambda: float = 0.95,
          rsr_loss_scale: float = 1.0, deterministic_eval: bool = False, progress_fn: Callable[[int, Dict[str, Any]], None] = lambda *a: None,
          normalize_advantage: bool = True, randomization_fn: Optional[Callable[[Any, np.ndarray], Dict[str, Any]]] = None, wrap_fn: Optional[Callable] = None,
          policy_hidden_layer_sizes=(32,) * 4, value_hidden_layer_sizes=(256,) * 5, value_obs_key: Optional[str] = None,
          max_grad_norm: Optional[float] = None, use_graph: Optional[bool] = None, num_resets_per_eval: int = 0,
          policy_params_fn: Callable[..., None] = lambda *a: None, restore_checkpoint_path: Optional[str] = None):
    """Returns (make_policy, (normalizer, networks), metrics) as the reference returns (make_policy, params, metrics).
    `environment` is an env definition with `batched` (AirbotPlayBase, go2.Joystick) or, with `wrap_fn`, anything
    `wrap_fn(environment, num_envs, episode_length, randomization_fn)` turns into a batched env."""
    import torch
    assert batch_size * num_minibatches % num_envs == 0                                   # train.py:160
    if wrap_fn is None:
        from ..envs.airbot import wrap as wrap_fn_
        wrap_fn = lambda e, n, ep, rf: wrap_fn_(e, n, episode_length=ep, action_repeat=action_repeat, randomization_fn=rf)
    xt = time.time()
    env_step_per_training_step = batch_size * unroll_length * num_minibatches * action_repeat
    num_evals_after_init = max(num_evals - 1, 1)
    num_training_steps_per_epoch = int(np.ceil(num_timesteps / (num_evals_after_init * env_step_per_training_step)))   # train.py:176-183
    key = prng.PRNGKey(seed)
    global_key, local_key = prng.split(key, 2)
    local_key, key_env, eval_key = prng.split(local_key, 3)
    key_policy, key_value = prng.split(global_key, 2)
    torch.manual_seed(int(key_policy[0]) << 32 | int(key_policy[1]))
    # train.py:205-217 / :428-438: randomization_fn(sys, rng) gets one key per env, a different key set for the eval envs
    key_env, key_rand = prng.split(key_env, 2)
    rand_for = lambda k, n: (None if randomization_fn is None else (lambda sys: randomization_fn(sys, prng.split(k, n))))
    env = wrap_fn(environment, num_envs, episode_length, rand_for(key_rand, num_envs))
    state = env.reset(prng.split(key_env, num_envs))
    device = state.obs.device
    obs_size, act_size = state.obs.shape[-1], env.action_size
    # value_obs_key: the critic reads another entry of the env's observation dict (locomotion_params.py: "privileged_state")
    extra_obs = {"value": (lambda st: st.obs_dict[value_obs_key])} if value_obs_key else None
    vobs_size = state.obs_dict[value_obs_key].shape[-1] if value_obs_key else obs_size
    networks = PPONetworks(obs_size, act_size, device, policy_hidden_layer_sizes, value_hidden_layer_sizes)
    if value_obs_key:
        networks.value = make_mlp([vobs_size, *value_hidden_layer_sizes, 1], device)
    normalizer = RunningStatistics(obs_size, device) if normalize_observations else None
    vnormalizer = RunningStatistics(vobs_size, device) if (normalize_observations and value_obs_key) else None
    networks.value_normalizer = vnormalizer          # travels with the networks: returned, handed to policy_params_fn, checkpointed
    params_list = networks.parameters()
    if use_graph is None:
        use_graph = device.type == "cuda"
    optimizer = torch.optim.Adam(params_list, lr=learning_rate, eps=1e-8, capturable=bool(use_graph))   # optax.adam defaults
    make_policy = make_inference_fn(networks, normalizer)
    if restore_checkpoint_path:                                                          # train.py:395-404 (npz instead of orbax)
        from .checkpoint import load_params
        load_params(restore_checkpoint_path, (normalizer, networks))
    norm = (lambda o: normalizer.normalize(o)) if normalizer is not None else (lambda o: o)
    policy_fn = lambda o: networks.policy(norm(o))
    vnorm = (lambda o: vnormalizer.normalize(o)) if vnormalizer is not None else norm
    value_fn = lambda o: networks.value(vnorm(o)).squeeze(-1)
    eval_env = wrap_fn(environment, num_eval_envs, episode_length, rand_for(eval_key, num_eval_envs))
    evaluator = Evaluator(eval_env, lambda p: make_policy(p, deterministic=deterministic_eval), num_eval_envs, episode_length, action_repeat, eval_key)
    metrics: Dict[str, Any] = {}
    if num_evals > 1:
        metrics = evaluator.run_evaluation(None, training_metrics={})
        progress_fn(0, metrics)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)

    def loss_from_flat(flat):
        extras: Dict[str, Dict[str, Any]] = {}
        for k, v in flat.items():
            if "/" in k:
                grp, name = k.split("/", 1)
                extras.setdefault(grp, {})[name] = v
        sub = Transition(flat["observation"], flat["action"], flat["reward"], flat["discount"], flat["next_observation"], extras)
        return ppo_losses.compute_ppo_loss(policy_fn, value_fn, sub, flat["noise"], past_data=past_data, entropy_cost=entropy_cost,
                                           discounting=discounting, reward_scaling=reward_scaling, gae_lambda=gae_lambda,
                                           clipping_epsilon=clipping_epsilon, normalize_advantage=normalize_advantage,
                                           rsr_loss_scale=rsr_loss_scale)
    graphed = None
    current_step, training_walltime = 0, 0.0
    nunroll = batch_size * num_minibatches // num_envs
    for it in range(num_evals_after_init):
        t0 = time.time()
        if num_resets_per_eval > 0 and it > 0:                                            # train.py:466-480: fresh episodes between evals
            local_key, key_reset = prng.split(local_key, 2)
            state = env.reset(prng.split(key_reset, num_envs))
        agg: Dict[str, float] = {}
        dev_agg: Dict[str, Any] = {}
        for _ in range(num_training_steps_per_epoch):
            local_key, key_gen = prng.split(local_key, 2)
            chunks = []
            for _u in range(nunroll):                                                     # train.py:310-324
                key_gen, cur = prng.split(key_gen, 2)
                state, data = generate_unroll(env, state, make_policy(), cur, unroll_length, extra_fields=("truncation",), extra_obs=extra_obs)
                chunks.append(data)
            cat = lambda f: torch.cat([f(c).transpose(0, 1) for c in chunks], dim=0)      # -> [batch_size * num_minibatches, unroll_length, ...]
            data = Transition(cat(lambda c: c.observation), cat(lambda c: c.action), cat(lambda c: c.reward), cat(lambda c: c.discount),
                              cat(lambda c: c.next_observation),
                              {"state_extras": {"truncation": cat(lambda c: c.extras["state_extras"]["truncation"])},
                               "policy_extras": {k: cat(lambda c, k=k: c.extras["policy_extras"][k]) for k in ("log_prob", "raw_action")},
                               **({"extra_obs": {"value": torch.nan_to_num(cat(lambda c: c.extras["extra_obs"]["value"]), 0.0, 0.0, 0.0)},
                                   "next_extra_obs": {"value": torch.nan_to_num(cat(lambda c: c.extras["next_extra_obs"]["value"]), 0.0, 0.0, 0.0)}}
                                  if extra_obs else {})})
            # A simulation that blows up (seen about once per 1e7 env-steps under a trained policy) yields non-finite
            # observations until its episode is truncated; such transitions are zeroed and counted instead of being learned from.
            bad = ~(torch.isfinite(data.observation).all(-1) & torch.isfinite(data.next_observation).all(-1) & torch.isfinite(data.reward)
                    & torch.isfinite(data.extras["policy_extras"]["raw_action"]).all(-1))
            if bool(bad.any()):
                agg["nonfinite_transitions"] = agg.get("nonfinite_transitions", 0.0) + float(bad.sum())
                data = Transition(torch.nan_to_num(data.observation, 0.0, 0.0, 0.0), data.action, torch.nan_to_num(data.reward, 0.0, 0.0, 0.0),
                                  data.discount, torch.nan_to_num(data.next_observation, 0.0, 0.0, 0.0),
                                  {**data.extras,
                                   "policy_extras": {k: torch.nan_to_num(v, 0.0, 0.0, 0.0) for k, v in data.extras["policy_extras"].items()}})
            if normalizer is not None:
                normalizer.update(data.observation[~bad] if bool(bad.any()) else data.observation)
            if vnormalizer is not None:
                vnormalizer.update(data.extras["extra_obs"]["value"])
            nb = data.observation.shape[0]
            for _e in range(num_updates_per_batch):
                perm = torch.randperm(nb, generator=gen, device=device)
                for mb in perm.view(num_minibatches, -1):
                    flat = {"observation": data.observation[mb], "action": data.action[mb], "reward": data.reward[mb], "discount": data.discount[mb],
                            "next_observation": data.next_observation[mb],
                            **{f"{grp}/{k}": v[mb] for grp, d in data.extras.items() for k, v in d.items()},
                            "noise": torch.randn((unroll_length, mb.numel(), act_size), generator=gen, device=device)}
                    if use_graph:
                        if graphed is None:
                            try:
                                graphed = _GraphedUpdate(loss_from_flat, optimizer, params_list, flat, max_grad_norm)
                            except Exception as exc:                                     # capture not possible: stay eager
                                print(f"ppo_train: HIP-graph capture failed ({type(exc).__name__}: {exc}); running eager")
                                use_graph = False
                        if use_graph:
                            graphed(flat)
                            continue
                    loss, m = loss_from_flat(flat)
                    optimizer.zero_grad(set_to_none=True)
                    loss.backward()
                    gn = torch.nn.utils.clip_grad_norm_(params_list, max_grad_norm if max_grad_norm else float("inf"))
                    # a non-finite gradient would poison Adam's moments for good: such an update gets zero gradients instead.
                    # Decided on the device (no host round trip per minibatch); metrics are summed on the device as well.
                    ok = torch.isfinite(gn)
                    for p_ in params_list:
                        if p_.grad is not None:
                            p_.grad = torch.where(ok, p_.grad, torch.zeros_like(p_.grad))
                    optimizer.step()
                    dev_agg["skipped_updates"] = dev_agg.get("skipped_updates", 0.0) + (~ok).to(torch.float32)
                    for k, v in m.items():
                        dev_agg[k] = dev_agg.get(k, 0.0) + v.detach()
            current_step += env_step_per_training_step
        nsteps = num_training_steps_per_epoch * num_updates_per_batch * num_minibatches
        for k, v in dev_agg.items():
            agg[k] = agg.get(k, 0.0) + float(v)
        if graphed is not None:
            for k, v in graphed.acc.items():
                agg[k] = agg.get(k, 0.0) + float(v)
                v.zero_()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        epoch_time = time.time() - t0
        training_walltime += epoch_time
        training_metrics = {"training/sps": num_training_steps_per_epoch * env_step_per_training_step / epoch_time,
                            "training/walltime": training_walltime,
                            **{f"training/{k}": (v if k in ("nonfinite_transitions", "skipped_updates") else v / nsteps) for k, v in agg.items()}}
        metrics = evaluator.run_evaluation(None, training_metrics)
        progress_fn(current_step, metrics)
        policy_params_fn(current_step, make_policy, (normalizer, networks))              # train.py:493-495 checkpoint hook
    assert current_step >= num_timesteps
    metrics["walltime"] = time.time() - xt
    return make_policy, (normalizer, networks), metrics
