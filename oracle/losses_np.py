"""numpy fp64 restatement of the reference's learner-side arithmetic.  TEST INFRASTRUCTURE ONLY (the checker of
rsr_mjx_amd/learning/): RSR/dataset_processor.py:17-43, RSR/rsr_loss.py:122-175, RSR/losses.py:39-205,
RSR/sac_losses.py:23-130, and Brax's NormalTanhDistribution.  Written the way the reference writes it (explicit (M, N, D)
differences, a reverse scan for GAE), not the way the product computes it.  Parity with JAX itself is unpinned (no JAX in
this pipeline); the formulas are short enough to check by reading.
"""
import numpy as np


def logsumexp(a, axis):
    m = a.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(a - m).sum(axis=axis, keepdims=True))).squeeze(axis)


def softmax(a):
    e = np.exp(a - a.max())
    return e / e.sum()


def evaluate_kde(data, grid, bandwidth=0.1):                       # dataset_processor.py:17-34
    diffs = grid[:, None, :] - data[None, :, :]
    log_kernel_vals = -np.sum(diffs ** 2, axis=-1) / (2 * bandwidth ** 2)
    log_pdf = logsumexp(log_kernel_vals, axis=-1) - np.log(data.shape[0])
    return softmax(log_pdf)


def kl_divergence(p, q):                                           # :36-38
    return np.sum(p * np.log((p + 1e-10) / (q + 1e-10)))


def wasserstein_distance(p, q):                                    # :40-42
    return np.sum(np.abs(np.cumsum(p) - np.cumsum(q)))


def compute_rsr_loss(obs, act, nobs, divergence, reference_density, reference_data, grid, bandwidth, loss_scale=1.0):   # rsr_loss.py:122-175
    cur = np.concatenate([obs.reshape(-1, obs.shape[-1]), act.reshape(-1, act.shape[-1]), nobs.reshape(-1, nobs.shape[-1])], axis=-1)
    aug = np.concatenate([reference_data, cur], axis=0)
    dist = wasserstein_distance(evaluate_kde(aug, grid, bandwidth), reference_density)
    return loss_scale * divergence * dist, dist


def compute_gae(truncation, termination, rewards, values, bootstrap_value, lambda_=1.0, discount=0.99):               # losses.py:39-95
    mask = 1 - truncation
    v_tp1 = np.concatenate([values[1:], bootstrap_value[None]], axis=0)
    deltas = (rewards + discount * (1 - termination) * v_tp1 - values) * mask
    acc = np.zeros_like(bootstrap_value)
    vs_minus = np.zeros_like(values)
    for t in reversed(range(values.shape[0])):
        acc = deltas[t] + discount * (1 - termination[t]) * mask[t] * lambda_ * acc
        vs_minus[t] = acc
    vs = vs_minus + values
    vs_tp1 = np.concatenate([vs[1:], bootstrap_value[None]], axis=0)
    adv = (rewards + discount * (1 - termination) * vs_tp1 - values) * mask
    return vs, adv


def softplus(x):
    return np.logaddexp(0.0, x)


def tanh_normal(logits):
    loc, raw = np.split(logits, 2, axis=-1)
    return loc, softplus(raw) + 0.001


def log_det_tanh(x):
    return 2.0 * (np.log(2.0) - x - softplus(-2.0 * x))


def log_prob(logits, raw_action):
    loc, scale = tanh_normal(logits)
    lp = -0.5 * ((raw_action - loc) / scale) ** 2 - np.log(scale) - 0.5 * np.log(2 * np.pi)
    return (lp - log_det_tanh(raw_action)).sum(-1)


def entropy(logits, noise):
    loc, scale = tanh_normal(logits)
    return (0.5 + 0.5 * np.log(2 * np.pi) + np.log(scale) + log_det_tanh(loc + scale * noise)).sum(-1)


def ppo_loss(logits, baseline, bootstrap, reward, discount, truncation, raw_action, behaviour_lp, noise, rsr_term, entropy_cost=1e-4,
             discounting=0.9, reward_scaling=1.0, gae_lambda=0.95, clipping_epsilon=0.3, normalize_advantage=True):       # losses.py:98-205, time-major inputs
    rewards = reward * reward_scaling
    termination = (1 - discount) * (1 - truncation)
    target_lp = log_prob(logits, raw_action)
    vs, adv = compute_gae(truncation, termination, rewards, baseline, bootstrap, gae_lambda, discounting)
    if normalize_advantage:
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    rho = np.exp(target_lp - behaviour_lp)
    policy_loss = -np.mean(np.minimum(rho * adv, np.clip(rho, 1 - clipping_epsilon, 1 + clipping_epsilon) * adv))
    v_loss = np.mean((vs - baseline) ** 2) * 0.5 * 0.5
    entropy_loss = entropy_cost * -np.mean(entropy(logits, noise))
    return policy_loss + v_loss + entropy_loss + rsr_term, policy_loss, v_loss, entropy_loss


# ---- SAC (RSR/sac_losses.py:23-130, after brax 0.12.1 agents/sac/losses.py) ----
# The networks enter as plain callables on numpy arrays (policy(obs) -> logits, q(obs, action) -> [..., 2], the twin critics);
# `noise` is the standard-normal draw the reference takes from `key` (sample_no_postprocessing = loc + scale * eps).
def sac_sample(logits, noise):
    loc, scale = tanh_normal(logits)
    return loc + scale * noise


def sac_alpha_loss(log_alpha, policy, obs, noise, action_size):                                          # :40-55
    logits = policy(obs)
    lp = log_prob(logits, sac_sample(logits, noise))
    return np.mean(np.exp(log_alpha) * (-lp - (-0.5 * action_size)))      # target_entropy = -0.5 * action_size (:33)


def sac_critic_loss(q, target_q, policy, alpha, obs, action, reward, discount, next_obs, truncation, noise,
                    reward_scaling, discounting):                                                     # :57-98
    old_q = q(obs, action)
    nlogits = policy(next_obs)
    nraw = sac_sample(nlogits, noise)
    next_q = target_q(next_obs, np.tanh(nraw))
    next_value = next_q.min(axis=-1) - alpha * log_prob(nlogits, nraw)
    target = reward * reward_scaling + discount * discounting * next_value
    q_error = (old_q - target[..., None]) * (1 - truncation)[..., None]
    return 0.5 * np.mean(q_error ** 2)


def sac_actor_loss(q, policy, alpha, obs, next_obs, noise, rsr_fn=None):                                  # :100-128
    """rsr_fn(obs, action, next_obs) -> the RSR penalty (compute_rsr_loss above with the reference data bound); None = 0."""
    logits = policy(obs)
    raw = sac_sample(logits, noise)
    action = np.tanh(raw)
    base = np.mean(alpha * log_prob(logits, raw) - q(obs, action).min(axis=-1))
    return base + (rsr_fn(obs, action, next_obs) if rsr_fn is not None else 0.0), base
