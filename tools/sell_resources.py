#!/usr/bin/env python3
"""VGPRs / scratch / occupancy / SGPR spills of the single-GPU sliced-ELL products (no offd, no in-kernel exchange, ticket mode):
   python tools/sell_resources.py [layouts, default 01]"""
import os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lay = sys.argv[1] if len(sys.argv) > 1 else "01"
kernels, cur = {}, None
for line in open(os.path.join(root, "mpi-bicgstab_amd", "build", "kernel_resources.txt")):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); kernels[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+):\s*(\d+)", line)
    if m and cur: kernels[cur][m.group(1).strip()] = int(m.group(2))
for k, r in kernels.items():
    m = re.search(r"k_spmv_sellILi(\d)ELb0ELb0ELi([%s])ELb0ELi0EEEv" % lay, k)
    if m: print("ndot %s layout %s: VGPR %d scratch %d occupancy %d SGPR spills %d" % (m.group(1), m.group(2), r["VGPRs"], r["ScratchSize [bytes/lane]"], r["Occupancy [waves/SIMD]"], r["SGPRs Spill"]))
