"""Host-only: seconds for bicg_mtx_load_block (rank 0 of 1) on the Transport-shaped Matrix-Market file (~840 MB, 23.9 M
entries) with 1..N tokeniser threads (BICG_MTX_THREADS). The file is written to /tmp on first use. No GPU needed."""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MTX = "/tmp/transport_like.mtx"

if not os.path.exists(MTX):
    import pandas as pd
    from mpi_bicgstab_amd import synth
    t = time.time()
    A = synth.transport_like(scale_decades=2.0)
    row, col, val = synth.colmajor_coo(A)
    with open(MTX, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{A.rows} {A.cols} {A.nnz}\n")
    pd.DataFrame({"i": row + 1, "j": col + 1, "v": val}).to_csv(MTX, sep=" ", header=False, index=False, mode="a", float_format="%.17g")
    print(f"wrote {MTX}: {os.path.getsize(MTX) / 1e6:.0f} MB, {A.nnz} entries, {time.time() - t:.0f} s", flush=True)

CHILD = r"""
import ctypes as C, sys, time
sys.path.insert(0, %r)
from mpi_bicgstab_amd import hipsolver as H
L = H.lib()
d, o, i = H.CSRMatrix(), H.CSRMatrix(), H.InfoMatrix()
L.bicg_mtx_load_block.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(H.CSRMatrix), C.POINTER(H.CSRMatrix), C.POINTER(H.InfoMatrix)]
t = time.time()
rc = L.bicg_mtx_load_block(%r.encode(), 0, 1, C.byref(d), C.byref(o), C.byref(i))
print(rc, time.time() - t)
""" % (ROOT, MTX)

for threads in sys.argv[1:] or ["1", "2", "4", "8"]:
    best = 1e9
    for _ in range(2):      # second pass: file in the page cache
        out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=dict(os.environ, BICG_MTX_THREADS=threads))
        rc, secs = out.stdout.split()[-2:]
        assert rc == "0", out.stderr
        best = min(best, float(secs))
    print(f"BICG_MTX_THREADS={threads}: {best:.2f} s", flush=True)
