import faulthandler, glob, os, sys, time
faulthandler.dump_traceback_later(25, exit=True)
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle_lib as O
from mpi_bicgstab_amd import hipsolver as H, synth
which = sys.argv[1]; name = sys.argv[2]
g = np.load([p for p in glob.glob('tests/golden/switching_*.npz') if name in p][0])
H.lib().bicg_comm_init_single(0)
A = synth.CSR(int(g["n"]), int(g["n"]), g["ptr"].astype(np.uint32), g["col"].astype(np.uint32), g["val"].astype(np.float64))
ctx = H.Context(H.single_rank_blocks(A))
print("context", ctx.flags(), flush=True)
t = time.time()
got = ctx.solve_shifted(g["b"], g["sigma"], int(g["seed"]), which=which, tol=1e-12, check_every=7)
print("gpu solve done k", got["k"], "in %.2f s" % (time.time() - t), flush=True)
n = int(g["n"]); row = np.repeat(np.arange(n, dtype=np.uint32), np.diff(g["ptr"].astype(np.int64)))
t = time.time()
orc = O.solve_switching(n, row, g["col"], g["val"], g["b"], g["sigma"], int(g["seed"]), which=which)
print("oracle done k", orc["k"], "in %.2f s" % (time.time() - t), flush=True)
