#!/usr/bin/env python3
"""Table of the compiler's per-kernel resource report (-Rpass-analysis=kernel-resource-usage output):
   python tools/kernel_resources.py mpi-bicgstab_amd/build/kernel_resources_persist.txt [name filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split()[0]
    g = lambda k: (re.search(k + r': (\d+)', b) or [None, '?'])[1]
    try:
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dn = name
    dn = dn.replace('bicg::(anonymous namespace)::', '').replace('(bicg::PersistArgs)', '').replace('void ', '')
    if flt in dn:
        print(f"{dn[:84]:84s} VGPR {g('    VGPRs'):>4s} AGPR {g('AGPRs'):>3s} spillV {g('VGPRs Spill'):>4s} spillS {g('SGPRs Spill'):>4s} "
              f"scratch {g('ScratchSize .bytes/lane.'):>4s} occ {g('Occupancy .waves/SIMD.')}")
