"""`-m gpu`: bench.py under torch.distributed.run exactly as the driver launches it for N > 1, with
the ranks sharing the one GPU of the box (transport host-p2p: gloo bootstrap + the peer-to-peer data
path, i.e. the code a real multi-GPU run takes with `--transport auto`). Validates the JSON line."""
import json
import math
import os
import subprocess
import sys

import pytest

from test_multirank import _free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(n, extra=()):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "20", "--warmup", "5",
           "--no-cpu-baseline", "--no-variants", *extra]
    env = dict(os.environ, BENCH_WATCHDOG_S="240", OMP_NUM_THREADS="2")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-4000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("n,transport", [(2, "host-p2p"), (4, "host-p2p"), (2, "host")])
def test_bench_under_torchrun(n, transport):
    d = run_bench(n, ("--transport", transport))
    assert d["n_gpus"] == n and d["steps"] == 20 and d["warmup"] == 5
    assert d["unit"] == "ms/iteration" and d["higher_is_better"] is False and d["scaling"] == "strong"
    assert isinstance(d["value"], float) and math.isfinite(d["value"]) and d["value"] > 0
    cfg = d["config"]
    assert cfg["iterations_genuine"] is True
    assert ("peer-to-peer" in cfg["transport"]) == (transport == "host-p2p"), cfg["transport"]
    assert math.isfinite(cfg["true_relres_after_timed_region"])
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0
