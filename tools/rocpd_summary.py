"""Summaries of rocprofv3's rocpd sqlite output (ROCm 7.2 default format) as small CSV files for profiles/.
usage: python tools/rocpd_summary.py kernels <results.db> <out.csv>
       python tools/rocpd_summary.py pmc <results.db> <out.csv>      (values summed over counter instances per dispatch,
                                                                      then averaged over the dispatches of a kernel)
Every row carries `csrc_sha16`, the hash of the kernel sources of the tree the summary was made in (bench.csrc_sha16): bench.py
quotes a PMC summary only when that hash is the running build's."""
import csv
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16  # noqa: E402

SHA = csrc_sha16()


def short(name: str) -> str:
    return name.replace("void ", "").split("(")[0][:160]


def kernels(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent", "csrc_sha16"])
        for n, k, t, a, lo, hi in rows:
            w.writerow([short(n), k, f"{t / 1e3:.1f}", f"{a / 1e3:.2f}", f"{lo / 1e3:.2f}", f"{hi / 1e3:.2f}", f"{100 * t / total:.3f}", SHA])


def pmc(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, dispatch_id, sum(value), max(grid_size) from counters_collection group by kernel_name, counter_name, dispatch_id").fetchall()
    agg = {}
    for n, cn, _, v, g in rows:
        a = agg.setdefault((n, cn), [0, 0.0, g])
        a[0] += 1; a[1] += v
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "avg_per_dispatch", "grid_size", "avg_per_workgroup_of_64", "csrc_sha16"])
        for (n, cn), (k, v, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if "rsr::" not in n:
                continue
            w.writerow([short(n), cn, k, f"{v / k:.1f}", g, f"{v / k / (g / 64):.2f}", SHA])


if __name__ == "__main__":
    {"kernels": kernels, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
