"""GPU: the persistent pipelined iteration against the multi-launch path and the oracle on a small rank."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
from mpi_bicgstab_amd import hipsolver as H, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200264
METHOD = sys.argv[2] if len(sys.argv) > 2 else "pipe_bicgstab"
KRR = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # pipe_bicgstab_rr: replacement every KRR iterations, NRR times
NRR = int(sys.argv[4]) if len(sys.argv) > 4 else 0
KW = dict(krr=KRR, nrr=NRR) if KRR else {}
H.lib().bicg_comm_init_single(0)
A = synth.transport_like(n=n, scale_decades=2.0)
row, col, val = A.to_coo()
b = O.spmv(A.rows, row, col, val, np.ones(A.rows))
K = 12
orc = O.solve(METHOD, A.rows, row, col, val, b, tol=0.0, max_iter=K, **KW)
res = {}
for mode in ("0", "1"):
    os.environ["BICG_PERSIST"] = mode
    ctx = H.Context(H.single_rank_blocks(A))
    t0 = time.time()
    got = ctx.solve(METHOD, b, tol=0.0, max_iter=K, check_every=K, **KW)
    print("    flags", [k for k, v in ctx.flags().items() if v])
    tr = ctx.trace(K)
    res[mode] = (got, tr)
    print("persist", mode, "k", got["k"], "solve wall", round(time.time() - t0, 3), flush=True)
    for key in ("alpha", "omega", "beta", "dotr"):
        err = np.max(np.abs(tr[key] - orc[key]) / np.abs(orc[key]))
        print("   ", key, "max rel dev vs oracle %.2e" % err)
    print("    x dev vs oracle %.2e" % (np.abs(got["x"] - orc["x"]).max() / np.abs(orc["x"]).max()))
    # convergence run
    full = ctx.solve(METHOD, b, tol=1e-10, max_iter=2000, check_every=16, **KW)
    print("    full solve: k", full["k"], "relres %.3e" % np.sqrt(full["result"].dot_r / full["result"].dot_zero),
          "x err %.2e" % np.abs(full["x"] - 1.0).max(), flush=True)
    # timing
    ctx.load(np.zeros(A.rows), b)
    ctx.run_begin(METHOD, tol=0.0, max_iter=440, check_every=400, **KW)
    ctx.run_iterate(40); ctx.sync()
    t0 = time.perf_counter(); ctx.run_iterate(400); ctx.sync(); dt = time.perf_counter() - t0
    r = ctx.run_end()
    print("    %.2f us per iteration (400 its), k %d" % (1e6 * dt / 400, r.iterations), flush=True)
    ctx.close()
a, bb = res["0"], res["1"]
print("persist vs multi-launch: x max rel diff %.2e" % (np.abs(a[0]["x"] - bb[0]["x"]).max() / np.abs(a[0]["x"]).max()))
