"""rsr_mjx_amd/learning (RSR distribution loss, GAE, PPO / SAC losses) against the numpy fp64 restatement in
oracle/losses_np.py and against closed-form properties."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import losses_np as O  # noqa: E402
from rsr_mjx_amd import prng  # noqa: E402
from rsr_mjx_amd.learning import ppo_losses as P, rsr_loss as R, sac_losses as S  # noqa: E402
from rsr_mjx_amd.rollout import Transition  # noqa: E402

T64 = lambda a: torch.as_tensor(a, dtype=torch.float64)


def test_kde_kl_wasserstein_match_the_restatement():
    rng = np.random.default_rng(0)
    data, grid = rng.normal(size=(50, 7)), rng.uniform(-3, 3, size=(12, 7))
    for h in (0.1, 0.5, 2.0):
        p = R.evaluate_kde(T64(data), T64(grid), h).numpy()
        np.testing.assert_allclose(p, O.evaluate_kde(data, grid, h), rtol=1e-9, atol=1e-300)
        assert abs(p.sum() - 1) < 1e-12 and (p >= 0).all()
    p, q = O.evaluate_kde(data, grid, 1.0), O.evaluate_kde(data * 0.5 + 0.3, grid, 1.0)
    assert R.kl_divergence(T64(p), T64(q)).item() == pytest.approx(O.kl_divergence(p, q), rel=1e-12) and O.kl_divergence(p, q) > 0
    assert R.kl_divergence(T64(p), T64(p)).item() == pytest.approx(0, abs=1e-12)
    assert R.wasserstein_distance(T64(p), T64(q)).item() == pytest.approx(O.wasserstein_distance(p, q), rel=1e-12)
    assert R.wasserstein_distance(T64(p), T64(p)).item() == 0
    # the grid is jax.random.uniform(PRNGKey(seed), (n, D), -3, 3) on the threefry restatement
    g = R.make_grid(10, 5, seed=0).numpy()
    np.testing.assert_array_equal(g, prng.uniform(prng.PRNGKey(0), (10, 5), -3.0, 3.0))
    assert g.min() >= -3 and g.max() < 3


def test_rsr_loss_value_and_policy_gradient():
    rng = np.random.default_rng(1)
    D_o, D_a, n = 4, 2, 40
    W = D_o + D_a + D_o
    real, prev, cur = (T64(rng.normal(size=(n, W)) * s + m) for s, m in ((1.0, 0.0), (1.2, 0.3), (1.1, 0.1)))
    data = R.build_rsr_data(real, prev, cur, num_samples=16, bandwidth=1.5, seed=3)
    grid = data.grid.double()
    data = data._replace(grid=grid, reference_density=R.evaluate_kde(cur, grid, 1.5),
                         divergence=R.kl_divergence(R.evaluate_kde(real, grid, 1.5), R.evaluate_kde(prev, grid, 1.5)))
    obs, nobs = T64(rng.normal(size=(3, 5, D_o))), T64(rng.normal(size=(3, 5, D_o)))
    act = T64(rng.uniform(-1, 1, size=(3, 5, D_a))).requires_grad_(True)
    loss, dist = R.compute_rsr_loss(obs, act, nobs, data, loss_scale=0.7)
    want, wdist = O.compute_rsr_loss(obs.numpy(), act.detach().numpy(), nobs.numpy(), data.divergence.item(), data.reference_density.numpy(),
                                     cur.numpy(), grid.numpy(), 1.5, 0.7)
    assert loss.item() == pytest.approx(want, rel=1e-9) and dist.item() == pytest.approx(wdist, rel=1e-9) and dist.item() > 0
    loss.backward()
    g = act.grad.numpy()
    assert np.abs(g).max() > 0                                      # the term carries a policy gradient (rsr_loss.py docstring)
    a0 = act.detach().numpy()
    for idx in ((0, 0, 0), (2, 4, 1)):
        e = np.zeros_like(a0); e[idx] = 1e-6
        fd = (O.compute_rsr_loss(obs.numpy(), a0 + e, nobs.numpy(), data.divergence.item(), data.reference_density.numpy(), cur.numpy(), grid.numpy(), 1.5, 0.7)[0]
              - O.compute_rsr_loss(obs.numpy(), a0 - e, nobs.numpy(), data.divergence.item(), data.reference_density.numpy(), cur.numpy(), grid.numpy(), 1.5, 0.7)[0]) / 2e-6
        assert g[idx] == pytest.approx(fd, rel=1e-4, abs=1e-10)
    z, zd = R.compute_rsr_loss(obs, act, nobs, None)
    assert z.item() == 0 and zd.item() == 0 and R.compute_rsr_loss(obs, act, nobs, data, loss_scale=0.0)[0].item() == 0
    legacy = (data.divergence, data.reference_density[:10], cur)      # legacy 3-tuple: grid of reference_density.shape[0] points, bandwidth 0.1
    assert R._as_rsr_data(legacy).grid.shape == (10, W) and R._as_rsr_data(legacy).bandwidth == 0.1
    with pytest.raises(ValueError):
        R.compute_rsr_loss(obs[..., :3], act, nobs, data)
    with pytest.raises(ValueError):
        R.build_rsr_data(real, prev[:-1], cur)


def test_gae_matches_restatement_and_closed_forms():
    rng = np.random.default_rng(2)
    T, B = 9, 5
    trunc = (rng.uniform(size=(T, B)) < 0.15).astype(np.float64)
    term = (rng.uniform(size=(T, B)) < 0.1).astype(np.float64) * (1 - trunc)
    rew, val, boot = rng.normal(size=(T, B)), rng.normal(size=(T, B)), rng.normal(size=B)
    vs, adv = P.compute_gae(T64(trunc), T64(term), T64(rew), T64(val), T64(boot), 0.95, 0.9)
    wvs, wadv = O.compute_gae(trunc, term, rew, val, boot, 0.95, 0.9)
    np.testing.assert_allclose(vs.numpy(), wvs, rtol=1e-12); np.testing.assert_allclose(adv.numpy(), wadv, rtol=1e-12)
    assert not vs.requires_grad and not adv.requires_grad
    # lambda = 1, no truncation / termination: vs = discounted return to go + bootstrap
    z = np.zeros((T, B))
    vs1, _ = P.compute_gae(T64(z), T64(z), T64(rew), T64(val), T64(boot), 1.0, 0.9)
    ret = boot.copy(); want = np.zeros((T, B))
    for t in reversed(range(T)):
        ret = rew[t] + 0.9 * ret; want[t] = ret
    np.testing.assert_allclose(vs1.numpy(), want, rtol=1e-12)
    # lambda = 0: one-step TD target
    vs0, adv0 = P.compute_gae(T64(z), T64(z), T64(rew), T64(val), T64(boot), 0.0, 0.9)
    vtp1 = np.concatenate([val[1:], boot[None]])
    np.testing.assert_allclose(vs0.numpy(), rew + 0.9 * vtp1, rtol=1e-12)


def _transition(rng, B, T, Do, A):
    f = lambda *s: T64(rng.normal(size=s))
    trunc = T64((rng.uniform(size=(B, T)) < 0.1).astype(np.float64))
    done = torch.maximum(trunc, T64((rng.uniform(size=(B, T)) < 0.1).astype(np.float64)))
    return Transition(observation=f(B, T, Do), action=torch.tanh(f(B, T, A)), reward=f(B, T), discount=1 - done, next_observation=f(B, T, Do),
                      extras={"state_extras": {"truncation": trunc}, "policy_extras": {"raw_action": f(B, T, A), "log_prob": f(B, T) * 0.1 - 2.0}})


def test_ppo_loss_matches_restatement():
    rng = np.random.default_rng(3)
    B, T, Do, A = 6, 8, 5, 3
    data = _transition(rng, B, T, Do, A)
    Wp, Wv = T64(rng.normal(size=(Do, 2 * A)) * 0.3).requires_grad_(True), T64(rng.normal(size=(Do,)) * 0.3).requires_grad_(True)
    policy, value = (lambda o: o @ Wp), (lambda o: o @ Wv)
    noise = T64(rng.normal(size=(T, B, A)))
    total, m = P.compute_ppo_loss(policy, value, data, noise, past_data=None)
    sw = lambda x: x.transpose(0, 1).detach().numpy()
    obs = sw(data.observation)
    want = O.ppo_loss(obs @ Wp.detach().numpy(), obs @ Wv.detach().numpy(), sw(data.next_observation)[-1] @ Wv.detach().numpy(), sw(data.reward),
                      sw(data.discount), sw(data.extras["state_extras"]["truncation"]), sw(data.extras["policy_extras"]["raw_action"]),
                      sw(data.extras["policy_extras"]["log_prob"]), noise.numpy(), 0.0)
    assert total.item() == pytest.approx(want[0], rel=1e-10) and m["policy_loss"].item() == pytest.approx(want[1], rel=1e-10)
    assert m["v_loss"].item() == pytest.approx(want[2], rel=1e-10) and m["entropy_loss"].item() == pytest.approx(want[3], rel=1e-10)
    assert m["sim2real_loss"].item() == 0
    total.backward()
    assert Wp.grad.abs().max() > 0 and Wv.grad.abs().max() > 0
    # with RSR data the sim2real term is added and differentiable through the policy's MODE action
    W = Do + A + Do
    ref = T64(rng.normal(size=(30, W)))
    rd = R.build_rsr_data(ref, ref * 1.3 + 0.2, ref * 0.9, num_samples=12, bandwidth=2.0)
    rd = rd._replace(grid=rd.grid.double(), reference_density=R.evaluate_kde(ref * 0.9, rd.grid.double(), 2.0), divergence=rd.divergence.double())
    Wp.grad = None
    total2, m2 = P.compute_ppo_loss(policy, value, data, noise, past_data=rd, rsr_loss_scale=2.0)
    assert m2["sim2real_loss"].item() > 0 and total2.item() == pytest.approx(m2["task_loss"].item() + m2["sim2real_loss"].item(), rel=1e-12)
    assert m2["task_loss"].item() == pytest.approx(total.item(), rel=1e-12)
    g = torch.autograd.grad(m2["sim2real_loss"], Wp)[0]
    assert g[:, :A].abs().max() > 0 and g[:, A:].abs().max() == 0     # mode = tanh(loc): no gradient into the scale half


def test_tanh_normal_distribution_helpers():
    rng = np.random.default_rng(4)
    logits, raw, noise = rng.normal(size=(7, 6)), rng.normal(size=(7, 3)), rng.normal(size=(7, 3))
    np.testing.assert_allclose(P.tanh_normal_log_prob(T64(logits), T64(raw)).numpy(), O.log_prob(logits, raw), rtol=1e-12)
    np.testing.assert_allclose(P.tanh_normal_entropy(T64(logits), T64(noise)).numpy(), O.entropy(logits, noise), rtol=1e-12)
    # log_prob integrates to one over the squashed action (1-D, numerical quadrature in raw space)
    lg = np.array([[0.3, -0.2]])
    xs = np.linspace(-12, 12, 200001)
    dens = np.exp(O.log_prob(np.repeat(lg, xs.size, 0), xs[:, None]) + O.log_det_tanh(xs))    # back to the raw-space density
    assert np.trapezoid(dens, xs) == pytest.approx(1.0, abs=1e-6)
    assert P.tanh_normal_mode(T64(lg)).item() == pytest.approx(np.tanh(0.3))


def test_sac_losses_shapes_and_rsr_term():
    rng = np.random.default_rng(5)
    B, Do, A = 16, 5, 2
    f = lambda *s: T64(rng.normal(size=s))
    tr = Transition(observation=f(B, Do), action=torch.tanh(f(B, A)), reward=f(B), discount=T64((rng.uniform(size=B) < 0.9).astype(np.float64)),
                    next_observation=f(B, Do), extras={"state_extras": {"truncation": T64(np.zeros(B))}})
    Wp = f(Do, 2 * A).requires_grad_(True); Wq = f(Do + A, 2).requires_grad_(True)
    policy = lambda o: o @ Wp
    q = lambda o, a: torch.cat([o, a], -1) @ Wq
    alpha_loss, critic_loss, actor_loss = S.make_losses(policy, q, reward_scaling=1.0, discounting=0.99, action_size=A)
    la = T64(0.0).requires_grad_(True)
    noise = f(B, A)
    al = alpha_loss(la, tr, noise); al.backward()
    # d/dlog_alpha [ exp(log_alpha) * c ] = mean(c) at log_alpha = 0
    logits = (tr.observation @ Wp).detach().numpy()
    loc, scale = O.tanh_normal(logits)
    lp = O.log_prob(logits, loc + scale * noise.numpy())
    assert la.grad.item() == pytest.approx(np.mean(-lp + 0.5 * A), rel=1e-10)
    cl = critic_loss(q, q, torch.tensor(0.2, dtype=torch.float64), tr, noise)
    assert cl.item() > 0 and torch.autograd.grad(cl, Wq)[0].abs().max() > 0
    a0 = actor_loss(q, torch.tensor(0.2, dtype=torch.float64), tr, noise)
    ref = T64(rng.normal(size=(20, Do + A + Do)))
    rd = R.build_rsr_data(ref, ref + 0.5, ref * 1.1, num_samples=8, bandwidth=2.0)
    rd = rd._replace(grid=rd.grid.double(), reference_density=R.evaluate_kde(ref * 1.1, rd.grid.double(), 2.0), divergence=rd.divergence.double())
    _, _, actor_rsr = S.make_losses(policy, q, 1.0, 0.99, A, past_data=rd, rsr_loss_scale=3.0)
    a1 = actor_rsr(q, torch.tensor(0.2, dtype=torch.float64), tr, noise)
    extra = R.compute_rsr_loss(tr.observation, torch.tanh(T64(loc + scale * noise.numpy())), tr.next_observation, rd, loss_scale=3.0)[0]
    assert a1.item() == pytest.approx(a0.item() + extra.item(), rel=1e-10) and extra.item() > 0


@pytest.mark.gpu
def test_losses_on_the_gpu_match_fp64_host():
    """The same PPO + RSR loss on torch-ROCm fp32 tensors (GEMM-form KDE on the device) and on fp64 host tensors."""
    rng = np.random.default_rng(6)
    B, T, Do, A = 64, 20, 23, 5
    data = _transition(rng, B, T, Do, A)
    Wp, Wv = T64(rng.normal(size=(Do, 2 * A)) * 0.2), T64(rng.normal(size=(Do,)) * 0.2)
    noise = T64(rng.normal(size=(T, B, A)))
    ref = T64(rng.normal(size=(256, Do + A + Do)))
    def run(dev, dt):
        c = lambda x: x.to(dev, dt)
        d = Transition(c(data.observation), c(data.action), c(data.reward), c(data.discount), c(data.next_observation),
                       {"state_extras": {k: c(v) for k, v in data.extras["state_extras"].items()},
                        "policy_extras": {k: c(v) for k, v in data.extras["policy_extras"].items()}})
        rd = R.build_rsr_data(c(ref), c(ref * 1.2 + 0.1), c(ref * 0.9), num_samples=10, bandwidth=3.0)
        wp, wv = c(Wp).detach().clone().requires_grad_(True), c(Wv).detach().clone()
        total, m = P.compute_ppo_loss(lambda o: o @ wp, lambda o: o @ wv, d, c(noise), past_data=rd, rsr_loss_scale=1.0)
        total.backward()
        return total.item(), m["sim2real_loss"].item(), wp.grad.double().cpu().numpy()
    t64, s64, g64 = run("cpu", torch.float64)
    t32, s32, g32 = run("cuda", torch.float32)
    assert t32 == pytest.approx(t64, rel=2e-4) and s32 == pytest.approx(s64, rel=2e-3, abs=1e-6) and s64 > 0
    np.testing.assert_allclose(g32, g64, rtol=5e-3, atol=5e-4 * np.abs(g64).max())


def test_running_statistics_match_numpy():
    from rsr_mjx_amd.learning.ppo_train import RunningStatistics
    rng = np.random.default_rng(7)
    rs = RunningStatistics(4)
    allx = []
    for n in (5, 1, 37):
        x = rng.normal(size=(n, 3, 4)) * np.array([1, 10, 0.1, 3]) + np.array([0, 5, -2, 100])
        rs.update(torch.as_tensor(x, dtype=torch.float32)); allx.append(x.reshape(-1, 4))
    X = np.concatenate(allx)
    np.testing.assert_allclose(rs.mean.numpy(), X.mean(0), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rs.std.numpy(), X.std(0), rtol=1e-4)
    assert rs.count.item() == X.shape[0]
    np.testing.assert_allclose(rs.normalize(torch.as_tensor(X, dtype=torch.float32)).numpy().std(0), 1.0, rtol=1e-3)


class _BanditEnv:
    """Reward = -(action - target(obs))^2 summed: one-step episodes; PPO must move the policy mode towards the target."""

    def __init__(self, n):
        from rsr_mjx_amd.envs.airbot import State
        self.n, self.action_size, self.State = n, 2, State
        self.gen = torch.Generator().manual_seed(0)
        self.obs = torch.zeros(n, 3); self.reward = torch.zeros(n); self.done = torch.zeros(n)
        self.trunc = torch.zeros(n); self.steps = torch.zeros(n)

    def _new_obs(self):
        self.obs.copy_(torch.rand(self.n, 3, generator=self.gen) * 2 - 1)

    def reset(self, keys):
        self._new_obs(); self.reward.zero_(); self.done.zero_()
        return self.State(None, self.obs, self.reward, self.done, {"dist": self.reward.clone()}, {"truncation": self.trunc, "steps": self.steps})

    def step(self, state, action):
        target = torch.stack([0.5 * self.obs[:, 0], -0.3 * torch.ones(self.n)], 1)
        self.reward.copy_(-((action - target) ** 2).sum(1))
        state.metrics["dist"] = -self.reward
        self.done.fill_(1.0); self.steps.fill_(1.0)
        self._new_obs()
        return state


def test_ppo_training_loop_improves_a_bandit():
    from rsr_mjx_amd.learning.ppo_train import train
    hist = []
    mk, (norm, nets), metrics = train(None, num_timesteps=40_000, episode_length=1, num_envs=64, num_eval_envs=64, learning_rate=3e-3,
                                      entropy_cost=1e-3, discounting=0.9, unroll_length=4, batch_size=32, num_minibatches=4,
                                      num_updates_per_batch=4, num_evals=4, normalize_observations=True, rsr_loss_scale=0.0,
                                      deterministic_eval=True, progress_fn=lambda s, m: hist.append((s, m["eval/episode_reward"])),
                                      wrap_fn=lambda e, n, ep, rf: _BanditEnv(n), policy_hidden_layer_sizes=(32, 32), value_hidden_layer_sizes=(64, 64))
    assert len(hist) == 4 and hist[0][0] == 0 and hist[-1][0] >= 40_000
    assert hist[-1][1] > hist[0][1] + 0.05 and hist[-1][1] > -0.25, hist
    for k in ("training/total_loss", "training/policy_loss", "training/v_loss", "training/entropy_loss", "training/sim2real_loss", "eval/avg_episode_length"):
        assert k in metrics and np.isfinite(metrics[k])
    a, _ = mk(None, deterministic=True)(torch.tensor([[0.8, 0.0, 0.0]]), prng.PRNGKey(0))
    assert abs(a[0, 0].item() - 0.4) < 0.25 and abs(a[0, 1].item() + 0.3) < 0.25


@pytest.mark.gpu
def test_ppo_training_runs_on_the_stepper():
    """A few PPO updates with the RSR term on the Airbot cube env: the loop runs end to end on the GPU and reports finite metrics."""
    from rsr_mjx_amd.envs.airbot import AirbotPlaySF, domain_randomize
    from rsr_mjx_amd.learning.ppo_train import train
    W = 23 + 5 + 23
    ref = torch.randn(128, W, generator=torch.Generator().manual_seed(0)).cuda()
    rd = R.build_rsr_data(ref, ref * 1.1 + 0.1, ref * 0.95, num_samples=10, bandwidth=3.0)
    seen = []
    mk, params, metrics = train(AirbotPlaySF(), num_timesteps=2 * 256 * 10 * 4, episode_length=200, past_data=rd, num_envs=256, num_eval_envs=128,
                                learning_rate=3e-4, unroll_length=10, batch_size=256, num_minibatches=4, num_updates_per_batch=2, num_evals=3,
                                normalize_observations=True, rsr_loss_scale=1.0, progress_fn=lambda s, m: seen.append(s),
                                randomization_fn=domain_randomize)
    assert len(seen) == 3 and np.isfinite(metrics["eval/episode_reward"]) and metrics["training/sps"] > 0
    assert np.isfinite(metrics["training/sim2real_loss"]) and metrics["training/rsr_distribution_distance"] >= 0


class _PointEnv:
    """x' = clip(x + 0.2 a); reward = -x'^2; episodes truncate after `horizon` steps and restart at a random x (auto-reset).
    Multi-step credit assignment: driving x to 0 early pays for the rest of the episode."""

    def __init__(self, n, horizon=10):
        from rsr_mjx_amd.envs.airbot import State
        self.n, self.h, self.action_size, self.State = n, horizon, 1, State
        self.gen = torch.Generator().manual_seed(1)
        self.obs = torch.zeros(n, 1); self.reward = torch.zeros(n); self.done = torch.zeros(n)
        self.trunc = torch.zeros(n); self.steps = torch.zeros(n)

    def reset(self, keys):
        self.obs.copy_(torch.rand(self.n, 1, generator=self.gen) * 2 - 1); self.steps.zero_(); self.done.zero_(); self.reward.zero_()
        return self.State(None, self.obs, self.reward, self.done, {}, {"truncation": self.trunc, "steps": self.steps})

    def step(self, state, action):
        x = (self.obs[:, 0] + 0.2 * action[:, 0]).clamp(-1, 1)
        self.reward.copy_(-x * x)
        t = self.steps + 1
        over = t >= self.h
        self.trunc.copy_(over.float()); self.done.copy_(over.float())
        fresh = torch.rand(self.n, generator=self.gen) * 2 - 1
        self.obs[:, 0] = torch.where(over, fresh, x)
        self.steps.copy_(torch.where(over, torch.zeros_like(t), t))
        return state


def test_ppo_training_loop_learns_a_multi_step_task():
    from rsr_mjx_amd.learning.ppo_train import train
    hist = []
    train(None, num_timesteps=120_000, episode_length=10, num_envs=128, num_eval_envs=128, learning_rate=1e-3, entropy_cost=1e-3,
          discounting=0.9, unroll_length=5, batch_size=64, num_minibatches=4, num_updates_per_batch=4, num_evals=4,
          normalize_observations=True, rsr_loss_scale=0.0, deterministic_eval=True,
          progress_fn=lambda s, m: hist.append(m["eval/episode_reward"]), wrap_fn=lambda e, n, ep, rf: _PointEnv(n, 10),
          policy_hidden_layer_sizes=(32, 32), value_hidden_layer_sizes=(64, 64))
    # untrained mode action ~ 0: about -1.4 per 10-step episode; driving x to zero: about -0.25
    assert hist[0] < -1.0 and hist[-1] > -0.6, hist


def test_sac_training_loop_learns_the_point_task():
    from rsr_mjx_amd.learning.sac_train import ReplayBuffer, train
    rb = ReplayBuffer(10, 2, 1, None)
    for k in range(3):
        rb.insert(torch.full((4, 2), float(k)), torch.zeros(4, 1), torch.full((4,), float(k)), torch.ones(4), torch.zeros(4, 2), torch.zeros(4))
    assert rb.size == 10 and rb.pos == 2 and set(rb.reward.tolist()) == {0.0, 1.0, 2.0} and (rb.reward[:2] == 2).all()   # ring overwrite
    s = rb.sample(64, torch.Generator().manual_seed(0))
    assert s.observation.shape == (64, 2) and s.extras["state_extras"]["truncation"].shape == (64,)
    hist = []
    train(None, num_timesteps=6_000, episode_length=10, num_envs=16, num_eval_envs=128, learning_rate=3e-3, discounting=0.9, batch_size=128,
          num_evals=4, normalize_observations=True, min_replay_size=256, max_replay_size=10_000, grad_updates_per_step=4, deterministic_eval=True,
          rsr_loss_scale=0.0, progress_fn=lambda s, m: hist.append(float(m["eval/episode_reward"])), wrap_fn=lambda e, n, ep, rf: _PointEnv(n, 10),
          hidden_layer_sizes=(64, 64))
    assert len(hist) == 4 and hist[-1] > hist[0] + 0.4 and hist[-1] > -0.8, hist
    with pytest.raises(ValueError):
        train(None, 10, 10, rsr_loss_scale=-1.0)


@pytest.mark.gpu
def test_sac_training_runs_on_the_stepper():
    from rsr_mjx_amd.envs.airbot import AirbotPlaySF, domain_randomize
    from rsr_mjx_amd.learning.sac_train import train
    ref = torch.randn(128, 23 + 5 + 23, generator=torch.Generator().manual_seed(0)).cuda()
    rd = R.build_rsr_data(ref, ref * 1.1 + 0.1, ref * 0.95, num_samples=10, bandwidth=3.0)
    seen = []
    _, _, m = train(AirbotPlaySF(), num_timesteps=256 * 60, episode_length=200, past_data=rd, num_envs=256, num_eval_envs=128, learning_rate=3e-4,
                    batch_size=256, num_evals=3, normalize_observations=True, min_replay_size=1024, max_replay_size=100_000, grad_updates_per_step=2,
                    rsr_loss_scale=1.0, randomization_fn=domain_randomize, progress_fn=lambda s, mm: seen.append(s))
    assert len(seen) == 3 and np.isfinite(m["eval/episode_reward"]) and np.isfinite(m["training/critic_loss"]) and m["buffer_current_size"] > 1024


def test_pipeline_builds_rsr_data_and_dispatches():
    from rsr_mjx_amd.learning import pipeline
    rng = np.random.default_rng(8)
    s, a = rng.normal(size=(30, 1)), rng.uniform(-1, 1, size=(30, 1))
    nr, ns, cs = s + 0.2 * a + 0.05, s + 0.2 * a, s + 0.2 * a + 0.01
    d = pipeline.build_policy_rsr_data(s, a, nr, ns, cs, num_samples=12, bandwidth=1.0, device="cpu")
    assert d.reference_data.shape == (30, 3) and d.grid.shape == (12, 3) and d.divergence.item() > 0
    want = O.kl_divergence(O.evaluate_kde(np.hstack([s, a, nr]), d.grid.double().numpy(), 1.0), O.evaluate_kde(np.hstack([s, a, ns]), d.grid.double().numpy(), 1.0))
    assert d.divergence.item() == pytest.approx(want, rel=2e-3)
    for bad in ((s[:-1], a, nr, ns, cs), (s, a, nr[:, :0], ns, cs), (s[:0], a[:0], nr[:0], ns[:0], cs[:0]), (s, a.ravel(), nr, ns, cs)):
        with pytest.raises(ValueError):
            pipeline.build_policy_rsr_data(*bad, device="cpu")
    with pytest.raises(ValueError):
        pipeline.policy_params_training(None, past_states=s, device="cpu")
    with pytest.raises(ValueError):
        pipeline.policy_params_training(None, past_states=s, past_actions=a, past_next_states_real=nr, past_next_states_sim=ns,
                                        current_next_states_sim=cs, algorithm="dqn", device="cpu")
    seen = []
    for algo, kw in (("ppo", dict(num_timesteps=2_000, num_envs=16, batch_size=8, num_minibatches=2, num_updates_per_batch=1, unroll_length=5)),
                     ("SAC ", dict(num_timesteps=400, num_envs=16, batch_size=32, min_replay_size=64))):
        mk, params = pipeline.policy_params_training(None, progress_fn=lambda st, m: seen.append((algo, st)), past_states=s, past_actions=a,
                                                     past_next_states_real=nr, past_next_states_sim=ns, current_next_states_sim=cs,
                                                     algorithm=algo, bandwidth=1.0, rsr_loss_scale=0.5, episode_length=10, num_evals=2,
                                                     num_eval_envs=16, wrap_fn=lambda e, n, ep, rf: _PointEnv(n, 10), device="cpu", **kw)
        act, _ = mk(None, deterministic=True)(torch.zeros(4, 1), prng.PRNGKey(0))
        assert act.shape == (4, 1) and torch.isfinite(act).all()
    assert {a for a, _ in seen} == {"ppo", "SAC "}


def test_checkpoint_roundtrip(tmp_path):
    from rsr_mjx_amd.learning.checkpoint import load_params, save_params
    from rsr_mjx_amd.learning.ppo_train import PPONetworks, RunningStatistics, make_inference_fn
    torch.manual_seed(0)
    a, b = PPONetworks(6, 2), PPONetworks(6, 2)
    na, nb = RunningStatistics(6), RunningStatistics(6)
    na.update(torch.randn(50, 6) * 3 + 1)
    obs = torch.randn(5, 6)
    pa = make_inference_fn(a, na)(None, deterministic=True)(obs, prng.PRNGKey(0))[0]
    assert not torch.allclose(pa, make_inference_fn(b, nb)(None, deterministic=True)(obs, prng.PRNGKey(0))[0])
    save_params(str(tmp_path / "ckpt.npz"), (na, a))
    load_params(str(tmp_path / "ckpt"), (nb, b))
    assert torch.equal(pa, make_inference_fn(b, nb)(None, deterministic=True)(obs, prng.PRNGKey(0))[0])
    assert torch.equal(a.value(obs), b.value(obs)) and nb.count.item() == 50
    with pytest.raises(Exception):
        load_params(str(tmp_path / "ckpt.npz"), (nb, PPONetworks(7, 2)))


@pytest.mark.gpu
def test_device_losses_match_the_numpy_restatement():
    """f2: the learner-side arithmetic AS IT RUNS ON THE GPU (fp32 tensors on cuda:0, GEMM-form KDE) against the independent numpy
    fp64 restatement of the reference's formulas (oracle/losses_np.py: explicit (M, N, D) differences, reverse-scan GAE) --
    an oracle, not the same torch code in another precision.  PPO total / policy / value / entropy terms, the RSR term, KDE,
    KL and Wasserstein."""
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(13)
    B, T, Do, A = 48, 16, 23, 5
    data = _transition(rng, B, T, Do, A)
    Wp, Wv = rng.normal(size=(Do, 2 * A)) * 0.2, rng.normal(size=(Do,)) * 0.2
    noise = rng.normal(size=(T, B, A))
    c = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32, device=dev)
    d = Transition(c(data.observation), c(data.action), c(data.reward), c(data.discount), c(data.next_observation),
                   {"state_extras": {k: c(v) for k, v in data.extras["state_extras"].items()},
                    "policy_extras": {k: c(v) for k, v in data.extras["policy_extras"].items()}})
    wp, wv = c(Wp), c(Wv)
    ref = rng.normal(size=(200, Do + A + Do))
    rd = R.build_rsr_data(c(ref), c(ref * 1.2 + 0.1), c(ref * 0.9), num_samples=10, bandwidth=3.0)
    total, m = P.compute_ppo_loss(lambda o: o @ wp, lambda o: o @ wv, d, c(noise), past_data=rd, rsr_loss_scale=1.5)
    # the restatement, fed the same numbers in float64
    sw = lambda x: np.asarray(x, dtype=np.float64).swapaxes(0, 1)
    obs, nobs = sw(data.observation), sw(data.next_observation)
    grid = rd.grid.double().cpu().numpy()
    dens = {k: O.evaluate_kde(v, grid, 3.0) for k, v in (("real", ref), ("prev", ref * 1.2 + 0.1), ("cur", ref * 0.9))}
    np.testing.assert_allclose(rd.reference_density.cpu().numpy(), dens["cur"], rtol=2e-4, atol=1e-7)
    kl = O.kl_divergence(dens["real"], dens["prev"])
    assert rd.divergence.item() == pytest.approx(kl, rel=2e-3)
    logits = obs @ Wp
    loc = np.tanh(np.split(logits, 2, axis=-1)[0])                          # the policy's mode action enters the RSR term
    rsr, dist = O.compute_rsr_loss(obs, loc, nobs, kl, dens["cur"], ref * 0.9, grid, 3.0, loss_scale=1.5)
    want = O.ppo_loss(logits, obs @ Wv, nobs[-1] @ Wv, sw(data.reward), sw(data.discount), sw(data.extras["state_extras"]["truncation"]),
                      sw(data.extras["policy_extras"]["raw_action"]), sw(data.extras["policy_extras"]["log_prob"]), noise, rsr)
    assert m["sim2real_loss"].item() == pytest.approx(rsr, rel=5e-3, abs=1e-7) and rsr > 0
    assert m["policy_loss"].item() == pytest.approx(want[1], rel=2e-3, abs=2e-5)
    assert m["v_loss"].item() == pytest.approx(want[2], rel=2e-4)
    assert m["entropy_loss"].item() == pytest.approx(want[3], rel=2e-4)
    assert total.item() == pytest.approx(want[0], rel=1e-3)
    p, q = rng.dirichlet(np.ones(40)), rng.dirichlet(np.ones(40))
    assert R.wasserstein_distance(c(p), c(q)).item() == pytest.approx(O.wasserstein_distance(p, q), rel=1e-4)
    assert R.kl_divergence(c(p), c(q)).item() == pytest.approx(O.kl_divergence(p, q), rel=1e-4)


def _sac_losses_case(dev):
    """f2, SAC leg: alpha, critic (with the truncation mask) and actor + RSR losses AS THEY RUN ON THE GPU (fp32 on cuda:0)
    against the independent numpy fp64 restatement of RSR/sac_losses.py:40-128 (oracle/losses_np.py: sac_*), same weights,
    same noise draws.  Networks: a linear policy and twin critics with a tanh layer, written once per side."""
    rng = np.random.default_rng(29)
    B, Do, A = 96, 23, 5
    obs, nobs = rng.normal(size=(B, Do)), rng.normal(size=(B, Do))
    act = np.tanh(rng.normal(size=(B, A)))
    reward, discount = rng.normal(size=B), (rng.uniform(size=B) > 0.1).astype(np.float64)
    trunc = (rng.uniform(size=B) > 0.8).astype(np.float64)
    Wp = rng.normal(size=(Do, 2 * A)) * 0.3
    Wq1, Wq2, Wt1, Wt2 = (rng.normal(size=(Do + A, 16)) * 0.3 for _ in range(4))
    vq1, vq2, vt1, vt2 = (rng.normal(size=16) * 0.5 for _ in range(4))
    n_alpha, n_critic, n_actor = rng.normal(size=(B, A)), rng.normal(size=(B, A)), rng.normal(size=(B, A))
    log_alpha, rs, disc = -0.7, 2.0, 0.97
    # numpy side
    pol = lambda o: o @ Wp
    twin = lambda W1, v1, W2, v2: (lambda o, a: np.stack([np.tanh(np.concatenate([o, a], -1) @ W1) @ v1, np.tanh(np.concatenate([o, a], -1) @ W2) @ v2], -1))
    q_np, tq_np = twin(Wq1, vq1, Wq2, vq2), twin(Wt1, vt1, Wt2, vt2)
    ref = rng.normal(size=(150, Do + A + Do))
    # torch side
    c = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32, device=dev)
    rd = R.build_rsr_data(c(ref), c(ref * 1.1 + 0.05), c(ref * 0.95), num_samples=10, bandwidth=3.0)
    wp = c(Wp)
    ttwin = lambda W1, v1, W2, v2: (lambda o, a: torch.stack([torch.tanh(torch.cat([o, a], -1) @ c(W1)) @ c(v1), torch.tanh(torch.cat([o, a], -1) @ c(W2)) @ c(v2)], -1))
    q_t, tq_t = ttwin(Wq1, vq1, Wq2, vq2), ttwin(Wt1, vt1, Wt2, vt2)
    tr = Transition(c(obs), c(act), c(reward), c(discount), c(nobs), {"state_extras": {"truncation": c(trunc)}})
    alpha_loss, critic_loss, actor_loss = S.make_losses(lambda o: o @ wp, q_t, rs, disc, A, past_data=rd, rsr_loss_scale=1.5)
    la = torch.tensor(log_alpha, device=dev)
    alpha = float(np.exp(log_alpha))
    got_alpha = alpha_loss(la, tr, c(n_alpha)).item()
    got_critic = critic_loss(q_t, tq_t, alpha, tr, c(n_critic)).item()
    got_actor = actor_loss(q_t, alpha, tr, c(n_actor)).item()
    # restatement
    want_alpha = O.sac_alpha_loss(log_alpha, pol, obs, n_alpha, A)
    want_critic = O.sac_critic_loss(q_np, tq_np, pol, alpha, obs, act, reward, discount, nobs, trunc, n_critic, rs, disc)
    grid = rd.grid.double().cpu().numpy()
    dens_real, dens_prev, dens_cur = (O.evaluate_kde(v, grid, 3.0) for v in (ref, ref * 1.1 + 0.05, ref * 0.95))
    kl = O.kl_divergence(dens_real, dens_prev)
    rsr_fn = lambda o, a, no: O.compute_rsr_loss(o, a, no, kl, dens_cur, ref * 0.95, grid, 3.0, loss_scale=1.5)[0]
    want_actor, want_base = O.sac_actor_loss(q_np, pol, alpha, obs, nobs, n_actor, rsr_fn)
    assert got_alpha == pytest.approx(want_alpha, rel=2e-4, abs=1e-6)
    assert got_critic == pytest.approx(want_critic, rel=5e-4)
    assert got_actor == pytest.approx(want_actor, rel=2e-3, abs=2e-5) and abs(want_actor - want_base) > 1e-4      # the RSR term is in there
    # the truncation mask matters: without it the critic loss is a different number
    assert abs(O.sac_critic_loss(q_np, tq_np, pol, alpha, obs, act, reward, discount, nobs, np.zeros(B), n_critic, rs, disc) - want_critic) > 1e-3 * abs(want_critic)


def test_sac_losses_match_the_numpy_restatement_on_the_host():
    _sac_losses_case(torch.device("cpu"))


@pytest.mark.gpu
def test_device_sac_losses_match_the_numpy_restatement():
    _sac_losses_case(torch.device("cuda:0"))


def test_sac_sgd_step_uses_the_entering_training_state():
    """brax 0.12.1 sac/train.py sgd_step: alpha, critic and actor losses are all evaluated at the training state as it entered
    the step (alpha before the alpha update, the actor against the critic before the critic update); the target critic then
    tracks the new critic.  One step against a by-hand replay of that order on clones, and against the sequential order
    (each loss after the previous update), which must come out different."""
    import copy
    from rsr_mjx_amd.learning.sac_train import TwinQ, _mlp, sgd_step
    torch.manual_seed(3)
    Do, A, B, tau = 6, 2, 32, 0.05
    rng = np.random.default_rng(3)
    f = lambda *s: torch.as_tensor(rng.normal(size=s), dtype=torch.float32)
    tr = Transition(f(B, Do), torch.tanh(f(B, A)), f(B), torch.ones(B), f(B, Do), {"state_extras": {"truncation": torch.zeros(B)}})
    noises = (f(B, A), f(B, A), f(B, A))

    def fresh():
        torch.manual_seed(11)
        pol = _mlp([Do, 16, 2 * A], None)
        q, tq = TwinQ(Do, A, (16,), None), TwinQ(Do, A, (16,), None)
        for a, b in zip(tq.parameters(), q.parameters()):
            a.data.copy_(b.data)
        la = torch.full((), 0.3, requires_grad=True)
        losses = S.make_losses(lambda o: pol(o), lambda o, a: q(o, a), 1.0, 0.95, A)
        opts = (torch.optim.Adam([la], lr=3e-2), torch.optim.Adam(q.parameters(), lr=3e-2), torch.optim.Adam(pol.parameters(), lr=3e-2))
        return pol, q, tq, la, losses, opts
    flat = lambda mod: torch.cat([p.detach().reshape(-1) for p in mod.parameters()])

    pol, q, tq, la, losses, opts = fresh()
    sgd_step(la, q, tq, pol, lambda o, a: q(o, a), lambda o, a: tq(o, a), losses, opts, tr, noises, tau)
    got = (la.detach().clone(), flat(q), flat(pol), flat(tq))

    # by hand, brax's order
    pol, q, tq, la, (alpha_loss, critic_loss, actor_loss), (oa, oq, op) = fresh()
    tq0 = flat(tq)
    alpha0 = torch.exp(la).detach()
    g_a = torch.autograd.grad(alpha_loss(la, tr, noises[0]), la)[0]
    g_q = torch.autograd.grad(critic_loss(lambda o, a: q(o, a), lambda o, a: tq(o, a), alpha0, tr, noises[1]), list(q.parameters()))
    g_p = torch.autograd.grad(actor_loss(lambda o, a: q(o, a), alpha0, tr, noises[2]), list(pol.parameters()))
    la.grad = g_a
    for p_, g in zip(q.parameters(), g_q): p_.grad = g
    for p_, g in zip(pol.parameters(), g_p): p_.grad = g
    oa.step(); oq.step(); op.step()
    want_tq = (1 - tau) * tq0 + tau * flat(q)
    for a, b in zip(got, (la.detach(), flat(q), flat(pol), want_tq)):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)

    # the sequential order (what this module did before) gives another actor and critic
    pol, q, tq, la, (alpha_loss, critic_loss, actor_loss), (oa, oq, op) = fresh()
    oa.zero_grad(); alpha_loss(la, tr, noises[0]).backward(); oa.step()
    alpha1 = torch.exp(la).detach()
    oq.zero_grad(); critic_loss(lambda o, a: q(o, a), lambda o, a: tq(o, a), alpha1, tr, noises[1]).backward(); oq.step()
    op.zero_grad(); actor_loss(lambda o, a: q(o, a), alpha1, tr, noises[2]).backward(inputs=list(pol.parameters())); op.step()
    assert (flat(pol) - got[2]).abs().max() > 1e-5 and (flat(q) - got[1]).abs().max() > 1e-6


def test_checkpoint_keeps_the_critics_own_normaliser(tmp_path):
    """The critic of the Go2 recipe reads `privileged_state` through its own running normaliser (ppo_train: value_obs_key); the
    reference checkpoints the statistics of every observation key.  Save / load must give the same value function."""
    from rsr_mjx_amd.learning.checkpoint import load_params, save_params
    from rsr_mjx_amd.learning.ppo_train import PPONetworks, RunningStatistics, make_mlp
    torch.manual_seed(1)
    def build():
        net = PPONetworks(6, 2)
        net.value = make_mlp([9, 16, 1], None)
        net.value_normalizer = RunningStatistics(9)
        return RunningStatistics(6), net
    na, a = build()
    nb, b = build()
    na.update(torch.randn(40, 6) * 2 + 1)
    a.value_normalizer.update(torch.randn(70, 9) * 5 - 3)
    vobs = torch.randn(8, 9) * 5 - 3
    va = a.value(a.value_normalizer.normalize(vobs))
    assert not torch.allclose(va, b.value(b.value_normalizer.normalize(vobs)))
    save_params(str(tmp_path / "c.npz"), (na, a))
    load_params(str(tmp_path / "c.npz"), (nb, b))
    assert torch.equal(va, b.value(b.value_normalizer.normalize(vobs))) and b.value_normalizer.count.item() == 70
    # a checkpoint without those statistics is refused rather than silently restarting the critic's normaliser
    a.value_normalizer = None
    save_params(str(tmp_path / "d.npz"), (na, a))
    with pytest.raises(KeyError):
        load_params(str(tmp_path / "d.npz"), (nb, b))


def test_rsr_dataset_tables(tmp_path):
    """learning.datasets.load_rsr_datasets: the five transition sets of reference test/rsr_policy_training.py:150-209, its
    truncation rule and its error cases (missing file, too few rows, wrong width, empty file)."""
    from rsr_mjx_amd.learning import datasets as D, pipeline
    rng = np.random.default_rng(0)
    T, od, ad = 12, 4, 2
    tabs = {"real_obs.txt": rng.normal(size=(T + 1, od)), "real_action.txt": rng.normal(size=(T, ad)),
            "past_sim_obs.txt": rng.normal(size=(T + 3, od)), "current_sim_obs.txt": rng.normal(size=(T + 1, od)),
            "obs.txt": rng.normal(size=(T + 1, od)), "actions.txt": rng.normal(size=(T, ad))}
    def write(d, t):
        d.mkdir(exist_ok=True)
        for k, v in t.items():
            np.savetxt(d / k, v, delimiter=",")
        return d
    d = write(tmp_path / "a", tabs)
    s, a, nr, nps, ncs = D.load_rsr_datasets(d, max_transitions=50)
    assert s.shape == (T, od) and a.shape == (T, ad)
    np.testing.assert_allclose(s, tabs["real_obs.txt"][:T]); np.testing.assert_allclose(nr, tabs["real_obs.txt"][1:T + 1])
    np.testing.assert_allclose(nps, tabs["past_sim_obs.txt"][1:T + 1]); np.testing.assert_allclose(ncs, tabs["current_sim_obs.txt"][1:T + 1])
    s5 = D.load_rsr_datasets(d, max_transitions=5)
    assert all(x.shape[0] == 5 for x in s5)
    np.testing.assert_allclose(s5[4], tabs["current_sim_obs.txt"][1:6])
    # what the loader returns is what the pipeline consumes
    data = pipeline.build_policy_rsr_data(*s5, num_samples=6, device="cpu")
    assert data.reference_data.shape == (5, 2 * od + ad) and data.grid.shape == (6, 2 * od + ad)
    # a one-line file is one row
    np.savetxt(tmp_path / "one.txt", np.arange(3.0)[None], delimiter=",")
    assert D.load_table(tmp_path / "one.txt").shape == (1, 3)
    # error cases
    (d / "obs.txt").unlink()
    with pytest.raises(FileNotFoundError, match="obs.txt"):
        D.load_rsr_datasets(d)
    short = dict(tabs); short["current_sim_obs.txt"] = tabs["current_sim_obs.txt"][:T]
    with pytest.raises(ValueError, match="current_sim_obs.txt: 12 rows, but 12 transitions take 13 observations"):
        D.load_rsr_datasets(write(tmp_path / "b", short))
    wide = dict(tabs); wide["actions.txt"] = rng.normal(size=(T, ad + 1))
    with pytest.raises(ValueError, match="actions.txt: 3 columns where the real data's actions have 2"):
        D.load_rsr_datasets(write(tmp_path / "c", wide))
    few = dict(tabs); few["real_obs.txt"] = tabs["real_obs.txt"][:1]
    with pytest.raises(ValueError, match="give no transition"):
        D.load_rsr_datasets(write(tmp_path / "e", few))
    e = write(tmp_path / "f", tabs); (e / "real_obs.txt").write_text("")
    with pytest.raises(ValueError, match="real_obs.txt: no rows of numbers"):
        D.load_rsr_datasets(e)
