import faulthandler, glob, os, sys, time
faulthandler.dump_traceback_later(40, exit=True)
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle_lib as O
from mpi_bicgstab_amd import hipsolver as H, synth
for path in sorted(glob.glob('tests/golden/switching_*.npz')):
    for which in ("shifted_lopbicg", "shifted_lopbicg_switching"):
        g = np.load(path)
        H.lib().bicg_comm_init_single(0)
        A = synth.CSR(int(g["n"]), int(g["n"]), g["ptr"].astype(np.uint32), g["col"].astype(np.uint32), g["val"].astype(np.float64))
        ctx = H.Context(H.single_rank_blocks(A))
        print(os.path.basename(path), which, "context ok", flush=True)
        t = time.time()
        got = ctx.solve_shifted(g["b"], g["sigma"], int(g["seed"]), which=which, tol=1e-12, check_every=7)
        print("  gpu solve done k", got["k"], "in %.2f s" % (time.time() - t), flush=True)
        # the rest of the test body, step by step
        n = int(g["n"]); row = np.repeat(np.arange(n, dtype=np.uint32), np.diff(g["ptr"].astype(np.int64)))
        orc = O.solve_switching(n, row, g["col"], g["val"], g["b"], g["sigma"], int(g["seed"]), which=which)
        print("  oracle k", orc["k"], flush=True)
        for P in (2, 3, 4, 8):
            kk = O.solve_switching(n, row, g["col"], g["val"], g["b"], g["sigma"], int(g["seed"]), which=which, nranks=P)["k"]
            print("  oracle P", P, kk, flush=True)
        res = ctx.shifted_residuals(got["x"], g["b"], g["sigma"])
        print("  residuals", float(res.max()), flush=True)
        j = len(g["sigma"]) // 2
        y = ctx.spmv(got["x"][j])
        print("  spmv ok", flush=True)
        tr = ctx.trace(got["iterations"])
        print("  trace ok", flush=True)
        ctx.close()
        print("  closed", flush=True)
