"""SAC training loop with the RSR term in the actor loss (torch-ROCm): the counterpart of reference RSR/sac_train.py:27-123,
which delegates to Brax 0.12.1 `sac.train` with `RSR/sac_losses.make_losses` injected.

Brax's schedule, restated: prefill the replay buffer until `min_replay_size` transitions; then every training step is ONE env
step of all `num_envs` envs (their transitions go into a uniform ring buffer of `max_replay_size`) followed by
`grad_updates_per_step` updates, each on `batch_size` uniformly sampled transitions: temperature (Adam 3e-4), twin critics,
actor, and the Polyak update of the target critics with `tau`.  Networks follow make_sac_networks: policy MLP (256, 256) ->
2*action_size, two Q MLPs (256, 256) -> 1 on [obs, action], relu activations, NormalTanh policy.  Host loop over torch
modules, one GPU; the replay buffer lives in device memory.
"""
from __future__ import annotations

import time
from typing import Any, Callable, Dict, Optional, Tuple

import numpy as np

from .. import prng
from ..rollout import Evaluator, Transition
from . import ppo_losses, sac_losses
from .ppo_train import RunningStatistics


def _mlp(sizes, device):
    import torch
    layers = []
    for i in range(len(sizes) - 1):
        lin = torch.nn.Linear(sizes[i], sizes[i + 1])
        bound = float(np.sqrt(3.0 / sizes[i]))
        torch.nn.init.uniform_(lin.weight, -bound, bound)
        torch.nn.init.zeros_(lin.bias)
        layers.append(lin)
        if i < len(sizes) - 2:
            layers.append(torch.nn.ReLU())
    return torch.nn.Sequential(*layers).to(device)


class TwinQ:
    def __init__(self, obs_size, action_size, hidden, device):
        self.q1 = _mlp([obs_size + action_size, *hidden, 1], device)
        self.q2 = _mlp([obs_size + action_size, *hidden, 1], device)

    def __call__(self, obs, action):
        import torch
        x = torch.cat([obs, action], dim=-1)
        return torch.cat([self.q1(x), self.q2(x)], dim=-1)

    def parameters(self):
        return list(self.q1.parameters()) + list(self.q2.parameters())


class ReplayBuffer:
    """brax UniformSamplingQueue: ring buffer, uniform sampling with replacement."""

    def __init__(self, capacity: int, obs_size: int, action_size: int, device):
        import torch
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=device)
        self.obs, self.next_obs, self.action = z(capacity, obs_size), z(capacity, obs_size), z(capacity, action_size)
        self.reward, self.discount, self.truncation = z(capacity), z(capacity), z(capacity)
        self.capacity, self.size, self.pos = capacity, 0, 0

    def insert(self, obs, action, reward, discount, next_obs, truncation):
        import torch
        n = obs.shape[0]
        idx = (torch.arange(n, device=obs.device) + self.pos) % self.capacity
        self.obs[idx], self.action[idx], self.reward[idx] = obs, action, reward
        self.discount[idx], self.next_obs[idx], self.truncation[idx] = discount, next_obs, truncation
        self.pos = (self.pos + n) % self.capacity
        self.size = min(self.size + n, self.capacity)

    def sample(self, batch_size: int, gen) -> Transition:
        import torch
        idx = torch.randint(0, self.size, (batch_size,), generator=gen, device=self.obs.device)
        return Transition(self.obs[idx], self.action[idx], self.reward[idx], self.discount[idx], self.next_obs[idx],
                          {"state_extras": {"truncation": self.truncation[idx]}})


def sgd_step(log_alpha, q, target_q, policy_net, qf, tqf, losses, optimizers, transitions, noises, tau: float):
    """One SAC update in brax 0.12.1's order (sac/train.py sgd_step): the three losses are all taken at the training state as it
    entered the step -- alpha = exp(alpha_params) before the alpha update, the actor loss against the critic before the critic
    update -- then the three optimisers step, and the target network tracks the NEW critic.  Returns the three loss tensors."""
    import torch
    alpha_loss, critic_loss, actor_loss = losses
    alpha_opt, q_opt, policy_opt = optimizers
    alpha = torch.exp(log_alpha).detach()
    la = alpha_loss(log_alpha, transitions, noises[0])
    lc = critic_loss(qf, tqf, alpha, transitions, noises[1])
    lp = actor_loss(qf, alpha, transitions, noises[2])
    for opt in optimizers:
        opt.zero_grad(set_to_none=True)
    la.backward(inputs=[log_alpha])
    lc.backward(inputs=list(q.parameters()))
    lp.backward(inputs=list(policy_net.parameters()))
    alpha_opt.step(); q_opt.step(); policy_opt.step()
    with torch.no_grad():
        for tp, p in zip(target_q.parameters(), q.parameters()):
            tp.mul_(1.0 - tau).add_(p, alpha=tau)
    return la, lc, lp


def train(environment, num_timesteps: int, episode_length: int, past_data: Any = None, action_repeat: int = 1, num_envs: int = 1,
          num_eval_envs: int = 128, learning_rate: float = 1e-4, discounting: float = 0.9, seed: int = 0, batch_size: int = 256,
          num_evals: int = 1, normalize_observations: bool = False, reward_scaling: float = 1.0, tau: float = 0.005,
          min_replay_size: int = 0, max_replay_size: Optional[int] = None, grad_updates_per_step: int = 1, deterministic_eval: bool = False,
          progress_fn: Callable[[int, Dict[str, Any]], None] = lambda *a: None, randomization_fn=None, rsr_loss_scale: float = 1.0,
          wrap_fn: Optional[Callable] = None, hidden_layer_sizes=(256, 256)):
    """Returns (make_policy, (normalizer, policy, q), metrics).  Arguments as the reference's (sac_train.py:27-60)."""
    import torch
    if rsr_loss_scale < 0:
        raise ValueError(f"rsr_loss_scale must be non-negative, got {rsr_loss_scale}")
    if max_replay_size is None:
        max_replay_size = num_timesteps
    if wrap_fn is None:
        from ..envs.airbot import wrap as wrap_fn_
        wrap_fn = lambda e, n, ep, rf: wrap_fn_(e, n, episode_length=ep, action_repeat=action_repeat, randomization_fn=rf)
    xt = time.time()
    env_steps_per_actor_step = action_repeat * num_envs
    num_prefill_actor_steps = -(-min_replay_size // num_envs)
    num_evals_after_init = max(num_evals - 1, 1)
    num_training_steps_per_epoch = -(-(num_timesteps - num_prefill_actor_steps * env_steps_per_actor_step) // (num_evals_after_init * env_steps_per_actor_step))
    num_training_steps_per_epoch = max(num_training_steps_per_epoch, 1)
    key = prng.PRNGKey(seed)
    global_key, local_key = prng.split(key, 2)
    local_key, rb_key, env_key, eval_key = prng.split(local_key, 4)
    torch.manual_seed(int(global_key[0]) << 32 | int(global_key[1]))
    env_key, key_rand = prng.split(env_key, 2)
    rand_for = lambda k, n: (None if randomization_fn is None else (lambda sys: randomization_fn(sys, prng.split(k, n))))
    env = wrap_fn(environment, num_envs, episode_length, rand_for(key_rand, num_envs))
    state = env.reset(prng.split(env_key, num_envs))
    device = state.obs.device
    obs_size, act_size = state.obs.shape[-1], env.action_size
    policy_net = _mlp([obs_size, *hidden_layer_sizes, 2 * act_size], device)
    q, target_q = TwinQ(obs_size, act_size, hidden_layer_sizes, device), TwinQ(obs_size, act_size, hidden_layer_sizes, device)
    for tp, p in zip(target_q.parameters(), q.parameters()):
        tp.data.copy_(p.data)
    log_alpha = torch.zeros((), device=device, requires_grad=True)
    normalizer = RunningStatistics(obs_size, device) if normalize_observations else None
    norm = (lambda o: normalizer.normalize(o)) if normalizer is not None else (lambda o: o)
    policy = lambda o: policy_net(norm(o))
    qf = lambda o, a: q(norm(o), a)
    tqf = lambda o, a: target_q(norm(o), a)
    alpha_opt = torch.optim.Adam([log_alpha], lr=3e-4)                                  # brax sac/train.py: alpha_optimizer
    policy_opt = torch.optim.Adam(policy_net.parameters(), lr=learning_rate)
    q_opt = torch.optim.Adam(q.parameters(), lr=learning_rate)
    alpha_loss, critic_loss, actor_loss = sac_losses.make_losses(policy, qf, reward_scaling, discounting, act_size, past_data=past_data,
                                                                 rsr_loss_scale=rsr_loss_scale)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    buffer = ReplayBuffer(max_replay_size, obs_size, act_size, device)

    def make_policy(_params=None, deterministic: bool = False):
        def act(obs, key):
            with torch.no_grad():
                logits = policy(obs)
                if deterministic:
                    return ppo_losses.tanh_normal_mode(logits), {}
                loc, scale = ppo_losses._split(logits)
                g = torch.Generator(device=obs.device)
                g.manual_seed(int(np.asarray(key, dtype=np.uint32)[0]) << 32 | int(np.asarray(key, dtype=np.uint32)[1]))
                return torch.tanh(loc + scale * torch.randn(loc.shape, generator=g, device=obs.device)), {}
        return act

    def actor_step(state, key):
        obs = state.obs.clone()
        action, _ = make_policy()(obs, key)
        state = env.step(state, action)
        ok = torch.isfinite(obs).all(-1) & torch.isfinite(state.obs).all(-1) & torch.isfinite(state.reward)
        nz = lambda x: torch.nan_to_num(x, 0.0, 0.0, 0.0)
        if normalizer is not None:
            normalizer.update(obs[ok])
        buffer.insert(nz(obs), nz(action), nz(state.reward), 1.0 - state.done, nz(state.obs), state.info["truncation"].to(torch.float32))
        return state

    for _ in range(num_prefill_actor_steps):                                             # brax: prefill_replay_buffer
        local_key, k = prng.split(local_key, 2)
        state = actor_step(state, k)
    evaluator = Evaluator(wrap_fn(environment, num_eval_envs, episode_length, rand_for(eval_key, num_eval_envs)),
                          lambda p: make_policy(p, deterministic=deterministic_eval), num_eval_envs, episode_length, action_repeat, eval_key)
    metrics: Dict[str, Any] = {}
    if num_evals > 1:
        metrics = evaluator.run_evaluation(None, training_metrics={})
        progress_fn(0, metrics)
    current_step = num_prefill_actor_steps * env_steps_per_actor_step
    walltime = 0.0
    for _ in range(num_evals_after_init):
        t0 = time.time()
        agg = {"alpha_loss": 0.0, "critic_loss": 0.0, "actor_loss": 0.0}
        for _s in range(num_training_steps_per_epoch):
            local_key, k = prng.split(local_key, 2)
            state = actor_step(state, k)
            for _u in range(grad_updates_per_step):
                tr = buffer.sample(batch_size, gen)
                noise = lambda: torch.randn((batch_size, act_size), generator=gen, device=device)
                la, lc, lp = sgd_step(log_alpha, q, target_q, policy_net, qf, tqf, (alpha_loss, critic_loss, actor_loss),
                                      (alpha_opt, q_opt, policy_opt), tr, (noise(), noise(), noise()), tau)
                agg["alpha_loss"] += float(la.detach()); agg["critic_loss"] += float(lc.detach()); agg["actor_loss"] += float(lp.detach())
            current_step += env_steps_per_actor_step
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dt = time.time() - t0
        walltime += dt
        nupd = num_training_steps_per_epoch * grad_updates_per_step
        tm = {"training/sps": num_training_steps_per_epoch * env_steps_per_actor_step / dt, "training/walltime": walltime,
              "training/alpha": float(torch.exp(log_alpha.detach())), "buffer_current_size": buffer.size, **{f"training/{k}": v / nupd for k, v in agg.items()}}
        metrics = evaluator.run_evaluation(None, tm)
        progress_fn(current_step, metrics)
    assert current_step >= num_timesteps
    metrics["walltime"] = time.time() - xt
    return make_policy, (normalizer, policy_net, q), metrics
