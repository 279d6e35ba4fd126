#!/bin/bash
# kernel trace of a 200 k-row rank (1/8 of Transport) driving the multi-rank path with one rank:
# where do the ~55 us of an iteration go -- kernel time or launch gaps?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
for mode in "--force-comm --transport auto" ""; do
  tag=$( [ -n "$mode" ] && echo p2p || echo single )
  rm -rf $OUT/small_$tag
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/small_$tag -o t --output-format csv -- \
      python $R/bench.py --rows 200264 $mode --steps 300 --warmup 30 --no-cpu-baseline --no-variants > $OUT/small_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for tag in ("p2p", "single"):
    f = glob.glob(f"{root}/small_{tag}/**/t_kernel_trace.csv", recursive=True)
    if not f: print(tag, "no trace"); continue
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the main timed run: take the longest run of kernels between 'FInit' markers
    names = [r["Kernel_Name"] for r in rows]
    starts = [i for i, n in enumerate(names) if "FInit" in n]
    seg = rows[starts[0]:starts[1]] if len(starts) > 1 else rows[starts[0]:]
    seg = seg[len(seg)//4:]                      # skip warm-up
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    per = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        k = r["Kernel_Name"].split("(")[0][:60]
        per[k][0] += 1; per[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print(f"== {tag}: {len(seg)} kernels over {(t1-t0)/1e3:.1f} us, busy {busy/1e3:.1f} us ({100*busy/(t1-t0):.0f} %)")
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"   {k:60s} n={c:5d} avg={t/c/1e3:6.2f} us")
PY
