#!/usr/bin/env python3
"""STREAM rates of this GPU against the size of the arrays (libbicgstab_hip.so bicg_stream_bench): what copy / triad / read reach when
the arrays sit in the Infinity Cache -- the ceiling of the element-wise phases of a 1.6 M-row system (12.8 MB per vector)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpi_bicgstab_amd import hipsolver as H
H.lib().bicg_comm_init_single(0)
for mb in (12.8, 25.6, 51.2, 96, 256, 1024):
    n = int(mb * 1e6) // 8 * 8
    row = {k: max(H.stream_bench(k, n, 60 if mb < 200 else 10) for _ in range(3)) for k in ("copy", "triad", "read8", "read16")}
    print(f"{mb:7.1f} MB per array: " + ", ".join(f"{k} {v:6.0f} GB/s" for k, v in row.items()), flush=True)
