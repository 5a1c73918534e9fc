"""Per-field parity envelopes from the measured statistics (tools/gpu_parity_stats.py --json): for every workload, phase
("reset" = the env-step straight after reset, "rollout" = later depths) and field
    max  = 5 x the largest scaled error seen between the HIP stepper and the fp32 oracle, and never below 1e-5: a field
           whose measured maximum is under 2e-6 is held to the north_star's own 1e-5, the others to 5 x their measurement
    p99  = 3 x the measured 99 % quantile                                                    (floor 3e-7)
    frac = 3 x the measured share of envs above 1e-5, plus 1 %
usage: python tools/make_parity_envelopes.py stats.json [more_stats.json ...] tests/golden/parity_envelopes.json"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16


def up(x, floor):
    x = max(x, floor)
    e = math.floor(math.log10(x)); m = math.ceil(x / 10 ** e * 10) / 10      # two significant digits, rounded up
    return float(f"{m * 10 ** e:.2g}")


src = json.load(open(sys.argv[1]))
for extra in sys.argv[2:-1]:
    more = json.load(open(extra))
    assert more["quantiles"] == src["quantiles"] and more["n"] == src["n"]
    src["stats"].update(more["stats"])
qs = src["quantiles"]
i99, imax = qs.index(0.99), qs.index(1.0)
out = {"_provenance": {"tool": "tools/gpu_parity_stats.py + tools/make_parity_envelopes.py", "envs_per_sample": src["n"], "csrc_sha16": csrc_sha16(),
                       "rule": "max = max(1e-5, 5 x measured max), p99 = 3 x measured p99, frac = 3 x measured share above 1e-5 + 0.01; reference side: fp32 oracle, MJX line-search rule",
                       "err": "|hip - oracle| / max(1, |oracle|_inf of that env's field), per env"}}
for kind, fields in src["stats"].items():
    out[kind] = {"reset": {}, "rollout": {}}
    for f, st in fields.items():
        for phase, sel in (("reset", lambda d: d == 0), ("rollout", lambda d: d > 0)):
            rows = [r for r in st["gpu_vs_f32"] if sel(r["depth"])]
            ref = [r for r in st["f32_vs_f64"] if sel(r["depth"])]
            m_max, m_p99, m_frac = max(r["q"][imax] for r in rows), max(r["q"][i99] for r in rows), max(r["frac_gt_1e-5"] for r in rows)
            out[kind][phase][f] = {"max": up(5 * m_max, 1e-5), "p99": up(3 * m_p99, 3e-7), "frac": round(3 * m_frac + 0.01, 4),
                                   "measured": {"max": m_max, "p99": m_p99, "frac": m_frac, "f32_vs_f64_max": max(r["q"][imax] for r in ref)}}
json.dump(out, open(sys.argv[-1], "w"), indent=1)
for kind in src["stats"]:
    for phase in ("reset", "rollout"):
        print(kind, phase, {f: (v["max"], v["p99"], v["frac"]) for f, v in out[kind][phase].items()})
