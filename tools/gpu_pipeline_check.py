"""How much of the launch tail does stepping the batch as S independent sub-batches on S HIP streams hide?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rsr_mjx_amd import prng
from rsr_mjx_amd.envs.airbot import AirbotPlayBase, domain_randomize
N, K = 8192, 200
envdef = AirbotPlayBase()
keys = prng.split(prng.PRNGKey(0), N)
dr_all = domain_randomize(envdef.sys, prng.split(prng.PRNGKey(1), N))
acts = torch.clamp(torch.randn((64, N, 5), device="cuda"), -1, 1)
for S in (1, 2, 4, 8):
    n = N // S
    streams = [torch.cuda.Stream() for _ in range(S)]
    envs, states = [], []
    for k in range(S):
        dr = {f: v[k * n:(k + 1) * n] for f, v in dr_all.items()}
        e = envdef.batched(n, episode_length=1200, auto_reset=True, randomization=dr)
        envs.append(e); states.append(e.reset(keys[k * n:(k + 1) * n]))
    torch.cuda.synchronize()
    def run(steps):
        for i in range(steps):
            for k in range(S):
                with torch.cuda.stream(streams[k]):
                    envs[k].step(states[k], acts[i % 64, k * n:(k + 1) * n])
    run(30); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"streams {S}: {N * K / dt / 1e6:.3f} M env-steps/s  ({dt / K * 1e3:.3f} ms per step of {N} envs)")
    del envs, states
