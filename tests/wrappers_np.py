"""brax's training wrappers restated in numpy around any object with a plain `step(state_dict, action)` (the oracle built from an
unwrapped env blob, or the plain HIP env behind tests/test_parity_gpu.py::_HipPlainEnv).  Reference: brax 0.12.1
envs/wrappers/training.py EpisodeWrapper.step (scan of env.step over action_repeat, rewards summed, steps += action_repeat,
truncation / done at episode_length, episode metrics) and AutoResetWrapper.step (steps <- 0 where done before the step, the cached first
state where done after it), as applied at RSR/train.py:224-229 and _src/wrapper.py:41-74.  Test infrastructure; pinned against the
oracle's own C wrappers at action_repeat = 1 by tests/test_oracle_env.py::test_numpy_wrappers_match_the_oracle_wrappers."""
import numpy as np


def np_repeat_step(orc_plain, st, act, repeat, L, pipeline, extra_restore=()):
    """brax EpisodeWrapper.step with action_repeat (scan of env.step, rewards summed, steps += repeat, done / truncation and the
    episode metrics once from the last state) inside AutoResetWrapper.step (steps <- 0 where done before; the cached first state
    where done after), restated in numpy around the PLAIN oracle env (wrapper flags 0)."""
    f32 = np.float32
    st["info_steps"][st["done"] != 0] = 0                                   # AutoResetWrapper.step, pre-step
    racc = np.zeros_like(st["reward"])
    for _ in range(repeat):
        orc_plain.step(st, act)
        racc = (racc + st["reward"]).astype(f32)
    st["reward"][...] = racc
    steps = st["info_steps"] + repeat
    over = steps >= L
    done = st["done"].copy()
    prev = st["info_episode_done"].copy()
    em = st["info_episode_metrics"]
    em[:, 0] = np.where(prev != 0, 0, (em[:, 0] + racc).astype(f32))
    em[:, 1] = np.where(prev != 0, 0, em[:, 1] + f32(repeat))
    em[:, 2:] = np.where(prev[:, None] != 0, 0, (em[:, 2:] + st["metrics"].reshape(len(prev), -1)).astype(f32))
    st["info_truncation"][...] = np.where(over, 1 - done, 0)
    done = np.where(over, 1, done).astype(done.dtype)
    st["info_steps"][...] = steps
    st["done"][...] = done
    st["info_episode_done"][...] = done
    sel = done != 0
    for k in pipeline:                                                      # AutoResetWrapper.step, post-step
        fk = "first_warmstart" if k == "qacc_warmstart" else "first_" + k
        st[k][sel] = st[fk][sel]
    for k, fk in extra_restore:
        st[k][sel] = st[fk][sel]
