#!/bin/bash
# round 5, first GPU call: parity of the plane-marching product, then its time against the slice-by-slice product (512^3, 256^3)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5c1
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_stencil.py -x -q 2>&1 | tail -25 > $OUT/pytest_stencil.txt
cat $OUT/pytest_stencil.txt
timeout 200 python tools/stencil_sweep.py 256 "" "BICG_STENCIL=0" "BICG_STENCIL_LINES=2 BICG_STENCIL_ZL=16" "BICG_STENCIL_LINES=4 BICG_STENCIL_ZL=8" "BICG_STENCIL_LINES=4 BICG_STENCIL_ZL=16" "BICG_STENCIL_LINES=2 BICG_STENCIL_ZL=32" "BICG_CA_FUSE=0" > $OUT/sweep256.txt 2>&1
cat $OUT/sweep256.txt
timeout 420 python tools/stencil_sweep.py 512 "" "BICG_STENCIL=0" "BICG_STENCIL_LINES=4 BICG_STENCIL_ZL=16" "BICG_STENCIL_LINES=4 BICG_STENCIL_ZL=64" "BICG_STENCIL_LINES=2 BICG_STENCIL_ZL=32" "BICG_CA_FUSE=0" "BICG_SELL_XCD=0" > $OUT/sweep512.txt 2>&1
cat $OUT/sweep512.txt
# counters of the product: L1 -> L2 read requests, bytes at the memory side
cd /tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -o p --output-format csv -- python $R/tools/lap512_spmv.py > $OUT/pmc$i.log 2>&1
done
python - > $OUT/pmc_summary.txt <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc*/**/p_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_spmv_stencil' in r['Kernel_Name'] or 'k_spmv_sell<' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in acc.items():
        v = v[3:] if len(v) > 6 else v
        print(f"  {k[0]:62s} {k[1]:36s} {sum(v)/len(v):18.1f}   ({len(v)} launches)")
for f in sorted(glob.glob("$OUT/pmc1/**/p_kernel_trace.csv", recursive=True)):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if 'k_spmv_stencil' in r['Kernel_Name']]
    if d: print(f"  kernel duration under counter pass 1: {sum(d[3:])/len(d[3:]):.1f} us ({len(d)} launches)")
PY
cat $OUT/pmc_summary.txt
