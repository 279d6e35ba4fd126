"""GPU: plain BiCGStab with q / p formed in the SpMV windows (BICG_FUSE_PLAIN) against the five-launch iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpi_bicgstab_amd import hipsolver as H, synth
H.lib().bicg_comm_init_single(0)
WL = sys.argv[1] if len(sys.argv) > 1 else "transport"
if WL == "transport":
    A = synth.transport_like(scale_decades=2.0)
else:
    hb = int(WL)
    A = synth.banded(synth.banded_rows_for(24_000_000, hb), hb, scale_decades=2.0)
print("workload", WL, A.rows, A.nnz, flush=True)
res = {}
for mode in ("0", "1"):
    os.environ["BICG_FUSE_PLAIN"] = mode
    ctx = H.Context(H.single_rank_blocks(A))
    b = ctx.spmv(np.ones(A.rows))
    got = ctx.solve("bicgstab", b, tol=0.0, max_iter=40, check_every=7)
    tr = ctx.trace(40)
    res[mode] = (got, tr)
    best = 1e9
    for rep in range(3):
        ctx.load(np.zeros(A.rows), b); ctx.run_begin("bicgstab", tol=0.0, max_iter=230, check_every=200); ctx.run_iterate(20); ctx.sync()
        t = time.perf_counter(); ctx.run_iterate(200); ctx.sync(); best = min(best, (time.perf_counter() - t) / 200 * 1e6); ctx.run_end()
    print("fuse_plain", mode, "k", got["k"], "%.1f us per iteration" % best, flush=True)
    ctx.close()
a, b_ = res["0"], res["1"]
print("x identical:", np.array_equal(a[0]["x"], b_[0]["x"]), " r identical:", np.array_equal(a[0]["r"], b_[0]["r"]),
      " traces identical:", all(np.array_equal(a[1][k], b_[1][k]) for k in ("alpha", "omega", "beta", "dotr")))
print("max rel x diff %.2e" % (np.abs(a[0]["x"] - b_[0]["x"]).max() / np.abs(a[0]["x"]).max()))
