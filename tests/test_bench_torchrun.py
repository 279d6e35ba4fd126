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


def _roof_ok(r):
    """every leg names the resource that bounds it and claims a fraction of THAT bound only -- never above 1, and none at all
    for the latency-bound persistent iterations"""
    if r["bound"] == "latency":
        return r["frac"] is None
    if r["bound"] == "mall":
        return r["frac"] is None or 0 < r["frac"] <= 1.0
    return r["bound"] == "hbm" and 0 < r["frac"] < 1.0 and r["peak"] == 8000.0


def _walk_roofs(d, path=""):
    """(path, entry) of every dict below d that carries a `frac`"""
    if isinstance(d, dict):
        if "frac" in d and "bound" in d:
            yield path, d
        for k, v in d.items():
            yield from _walk_roofs(v, f"{path}/{k}")


def run_bench(n, extra=()):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "20", "--warmup", "5",
           "--no-cpu-baseline", "--no-variants", *extra]
    return _run(cmd, 400, dict(BENCH_WATCHDOG_S="240", OMP_NUM_THREADS="2"))


def _run(cmd, timeout, env_extra):
    """-> the COMPLETE record (bench_full.json); the one stdout line must be its compact form: < 8 KB, contract keys, same numbers"""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        full_path = os.path.join(tmp, "bench_full.json")
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                             env=dict(os.environ, BENCH_FULL_JSON=full_path, **env_extra))
        assert out.returncode == 0 and out.stdout.count("\n") == 1 and out.stdout.startswith("{"), out.stdout[-2000:] + out.stderr[-4000:]
        assert len(out.stdout) < 8192, len(out.stdout)
        assert len(out.stderr) < 12000, len(out.stderr)      # the driver keeps a bounded tail of both streams together
        line = json.loads(out.stdout)
        full = json.load(open(full_path))
    check_compact(line, full)
    return full


def check_compact(line, full):
    from test_bench_contract import COMPACT_KEYS, ROOFLINE_KEYS
    assert set(COMPACT_KEYS) <= set(line), set(COMPACT_KEYS) - set(line)
    assert set(ROOFLINE_KEYS) <= set(line["roofline"]), set(ROOFLINE_KEYS) - set(line["roofline"])
    for k in ("value", "ms_per_step"):
        assert abs(line[k] - full[k]) <= 1e-5 * full[k]
    for k in ("unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == full[k], k
    assert abs(line["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-5
    assert line["config"]["iterations_genuine"] == full["config"]["iterations_genuine"]
    assert line["full_record"] == "bench_full.json"


@pytest.mark.parametrize("n,transport,extras", [(2, "host-p2p", True), (4, "host-p2p", False), (2, "host", False)])
def test_bench_under_torchrun(n, transport, extras):
    d = run_bench(n, ("--transport", transport) + (() if extras else ("--no-extras",)))
    assert d["n_gpus"] == n and d["steps"] == 20 and d["warmup"] == 5
    assert d["unit"] == "ms/iteration" and d["higher_is_better"] is False and d["scaling"] == "strong"
    assert isinstance(d["value"], float) and math.isfinite(d["value"]) and d["value"] > 0
    cfg = d["config"]
    assert cfg["iterations_genuine"] is True
    assert ("peer-to-peer" in cfg["transport"]) == (transport == "host-p2p"), cfg["transport"]
    assert math.isfinite(cfg["true_relres_after_timed_region"])
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0
    # the communicator report: which data path produced `value`, what the self-test said, whether RCCL saw N ranks, and the
    # second leg over the RCCL collectives -- on this one-GPU box it cannot run (ranks share the device) but MUST say so
    cm = d["comm"]
    assert cm["world"] == n and cm["transport_used"] == cfg["transport"] and cm["fallback_reason"] is None
    assert (cm["p2p_selftest"] == "passed") == (transport == "host-p2p"), cm
    assert cm["rccl_leg"] is not None and ("unavailable" in cm["rccl_leg"] or cm["rccl_leg"]["ms_per_iteration"] > 0), cm
    if "unavailable" in cm["rccl_leg"]:
        # RCCL was asked all the same; its answer is in the record verbatim (ncclCommInitRank's result string + last error text)
        assert "share a device" in cm["rccl_leg"]["unavailable"] and len(cm["rccl_leg"]["attempt"]) > 10, cm["rccl_leg"]
    else:
        assert cm["rccl_leg"]["rccl_nranks"] == n and cm["rccl_leg"]["iterations_genuine"] is True, cm["rccl_leg"]
    st = d["roofline"]["stream_measured_gbps"]
    assert st and 3000 < st["copy"] < 8000 and 3000 < st["triad"] < 8000 and 3000 < st["read8"] < 8000, st
    if extras:      # north_star: "Transport.mtx and synthetic banded CSR reported at 1, 2, 4 and 8 GPUs"
        for hb in (8, 64, 512):
            e = d["extras"][f"banded_b{hb}"]
            assert e["nnz"] > 20_000_000
            for m in ("bicgstab", "pipe_bicgstab"):
                assert math.isfinite(e[m]["ms_per_iteration"]) and _roof_ok(e[m]), e[m]
        # ... and the unstructured matrix (RCM numbering) in the reference's partition at every rank count
        e = d["extras"]["mesh_rcm"]
        assert e["rows"] == 117 ** 3 and e["plan"]["halo"] > 0 and "jagged" in e["flags"]
        for m in ("bicgstab", "pipe_bicgstab"):
            assert e[m]["iterations_genuine"] is True and _roof_ok(e[m]), e[m]


def rfst(d):
    return d["roofline"]["stream_measured_gbps"]


def test_bench_single_gpu_line_has_every_leg():
    """the N = 1 line the driver records: headline unchanged, plus roofline fractions of every variant, the banded /
    FEM-like / 256^3-Laplacian legs and the HBM traffic measured in this run (rocprofv3 PMC passes)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"]
    d = _run(cmd, 900, dict(BENCH_WATCHDOG_S="800"))
    assert d["n_gpus"] == 1 and d["config"]["iterations_genuine"] is True and "configs[1]" in d["config"]["workload"]
    for m in ("bicgstab", "ca_bicgstab", "pipe_bicgstab", "pipe_bicgstab_rr", "shifted_lopbicgstab_16shifts", "shifted_pipe_lopbicgstab_16shifts"):
        r = d["variant_rooflines"][m]
        assert r["bound"] == "hbm" and 0.3 < r["frac"] < 1.0 and r["algorithmic_bytes"] > 8e8, (m, r)
    # no fraction of the HBM peak for legs that do not run out of HBM; nothing above 1 anywhere in the line
    roofs = list(_walk_roofs(d))
    assert len(roofs) >= 20 and all(_roof_ok(r) for _, r in roofs), [(p_, r) for p_, r in roofs if not _roof_ok(r)]
    assert d["extras"]["transport_rank_of_8"]["pipe_bicgstab"]["bound"] == "latency"
    # (banded b = 8 without column indices: 192 MB of values + six vectors sit at the edge of the 256 MiB Infinity Cache)
    assert d["extras"]["banded_b8"]["bicgstab"]["bound"] in ("hbm", "mall") and d["extras"]["laplace7_512_ca"]["ca_bicgstab"]["bound"] == "hbm"
    assert rfst(d)["mall_read8"] > rfst(d)["read8"]
    # the 8-GPU form with traffic in it: two ranks sharing this GPU, persistent launches, a real halo between them
    sh = d["extras"]["small_rank_with_halo"]
    assert "error" not in sh and sh["iterations_genuine"] is True and "persist" in sh["flags"] and sh["halo"] > 10000, sh
    assert "peer-to-peer" in sh["transport"] and 0 < sh["ms_per_iteration"] < 0.1
    # round 5: two processes on one GPU are not "over xGMI", and the persistent kernels report what the exchanges made them wait
    # (device clock, one sample per exchange): the all-reduce through the mailboxes and a boundary workgroup's hand-off
    assert "SHARING one device" in sh["transport"] and "xGMI link" in sh["transport"], sh["transport"]
    wu = sh["comm_wait_us"]
    assert wu and wu["allreduce_samples"] >= 100 and wu["handoff_samples"] >= 100, wu
    assert 0 < wu["allreduce_p50_us"] <= wu["allreduce_p99_us"] < 5000 and 0 < wu["handoff_p50_us"] <= wu["handoff_p99_us"] < 5000, wu
    # the headline is the median of three regions, each with the device's clock beside the host's
    hr = d["headline_regions"]
    assert len(hr) == 3 and all(0 < r["device_ms"] <= r["region_ms"] * 1.02 and r["enqueue_ms"] < r["region_ms"] for r in hr), hr
    assert abs(sorted(r["region_ms"] for r in hr)[1] / 20 - d["value"]) < 1e-3 * d["value"] + 1e-6
    # how much of the headline is the synthetic's structure: the same matrix without uniform slices is slower, and on record
    sd = d["roofline"]["structure_dependence"]
    assert sd and "uniform" not in sd["flags"] and sd["ms_per_iteration"] > d["value"] and 0.4 < sd["frac_of_format_bytes"] < 1.0, sd
    assert d["roofline"]["survey_8d_frac"] >= d["roofline"]["frac"]
    # the product of the UNSTRUCTURED matrix (mesh.py, RCM numbering) as a second headline: per-kernel time by events, both byte
    # bases, counters; the kernel it ran on is on record, and so are the generator order and the random permutation
    ru = d["roofline_unstructured"]
    assert ru and ru["numbering"] == "rcm" and ru["product_kernels"] == ["jagw_list"], ru
    assert 0.45 < ru["frac"] < 1.0 and ru["survey_8d_frac"] > ru["frac"] and ru["traffic"] is not None, ru
    assert 0.85 * ru["format_bytes_per_launch"] < ru["traffic"] < 1.6 * ru["format_bytes_per_launch"], ru
    # north_star's bar on the stand-in for Transport.mtx: the product at >= 60 % of this GPU's measured copy rate on SURVEY 8d bytes
    assert ru["survey_8d_frac_of_measured_copy"] >= 0.60, ru
    on = ru["other_numberings"]
    assert on["generator"]["product_kernels"] == ["jagw"] and on["generator"]["avg_launch_ms"] < ru["avg_launch_ms"] * 1.05, on
    assert on["random"]["product_kernels"] == ["jagd"] and on["random"]["avg_launch_ms"] > 3 * ru["avg_launch_ms"], on
    assert on["random"]["traffic"] > 3 * on["random"]["format_bytes_per_launch"] and "cache line" in on["random"]["bottleneck"], on
    # BASELINE.json configs[4] "batched SpMV": 16 vectors through the pipelined SpMM, well below 16 single products
    mm = d["extras"]["spmm_16_vectors"]
    assert mm["kernel"] == "pipelined" and mm["vectors"] == 16 and 0.25 < mm["frac"] < 1.0 and mm["spmv_equivalents"] < 8.5, mm
    # ... and on ragged rows (the mesh matrix, x windows by lists / by runs): the pipeline of csrc/bicg_spmm_jag.hip
    for key in ("mesh_rcm", "mesh_generator"):
        mj = d["extras"][key]["spmm_16_vectors"]
        assert mj["kernel"] == "pipelined" and mj["vectors"] == 16 and 0.15 < mj["frac"] < 1.0 and mj["spmv_equivalents"] < 9.0, (key, mj)
    assert d["extras"]["laplace7_512_ca"]["plane_marching_product"]["on"] == 1
    assert math.isfinite(d["extras"]["laplace7_512_ca"]["ca_bicgstab"]["true_relres_after_timed_region"])
    for key in ("banded_b8", "banded_b64", "banded_b512", "mesh_rcm", "mesh_generator", "mesh_random", "laplace7_256_ca", "laplace7_512_ca", "transport_rank_of_8"):
        assert key in d["extras"], key
        legs = [v for v in d["extras"][key].values() if isinstance(v, dict) and "ms_per_iteration" in v]
        assert legs and all(v["iterations_genuine"] is True for v in legs), (key, legs)   # no leg timed a converged solve
    assert d["extras"]["laplace7_256_ca"]["rows"] == 256 ** 3 and d["extras"]["laplace7_256_ca"]["ca_bicgstab"]["ms_per_iteration"] > 0
    # BASELINE.json configs[3] at its stated size on one GPU: generated and planned on the device in well under 3 s
    big = d["extras"]["laplace7_512_ca"]
    assert big["rows"] == 512 ** 3 and big["nnz"] == 7 * 512 ** 3 - 6 * 512 ** 2
    assert big["plan_seconds"] < 3.0 and big["generate_seconds"] < 3.0, big
    # (frac is claimed on the bytes the layout moves: with constant slices the interior of the Laplacian streams no matrix at all)
    assert 0.35 < big["ca_bicgstab"]["frac"] < 1.0 and "constant" in big["flags"] and big["ca_bicgstab"]["ms_per_iteration"] < 10 * d["extras"]["laplace7_256_ca"]["ca_bicgstab"]["ms_per_iteration"]
    # the rank one of 8 GPUs holds (configs[2]): the pipelined solver takes it as one persistent launch per chunk
    r8 = d["extras"]["transport_rank_of_8"]
    assert "persist" in r8["flags"] and r8["pipe_bicgstab"]["ms_per_iteration"] < 0.020, r8
    # ... one of 4 (two rows per thread in the persistent kernels of all three methods: the plain iteration below the five-launch
    # form's 46 us) and one of 2 (multi-launch), every leg a genuine iteration
    r4, r2 = d["extras"]["transport_rank_of_4"], d["extras"]["transport_rank_of_2"]
    assert "persist" in r4["flags"] and r4["bicgstab"]["ms_per_iteration"] < 0.043 and r4["ca_bicgstab"]["ms_per_iteration"] < 0.043, r4
    assert all(r[m]["iterations_genuine"] is True for r in (r4, r2) for m in ("bicgstab", "pipe_bicgstab", "ca_bicgstab")), (r4, r2)
    assert r4["bicgstab"]["ms_per_iteration"] < r2["bicgstab"]["ms_per_iteration"] < d["value"], (r4, r2, d["value"])
    assert "rowsplit" in d["extras"]["banded_b512"]["flags"] and d["extras"]["banded_b512"]["spmv_back_to_back"]["frac"] > 0.6
    rf = d["roofline"]
    st = rf["stream_measured_gbps"]
    assert st and 4000 < st["copy"] < 8000 and 4000 < st["triad"] < 8000 and 4000 < st["read8"] < 8000, st
    assert 0.5 < rf["frac_of_measured_stream"] < 1.1 and 0.5 < rf["frac_of_measured_read"] < 1.0, rf
    # fractions are claimed on the bytes the stored layout moves (uniform slices keep no column index: fewer than CSR's 12 per
    # non-zero); the CSR figure of SURVEY 8d stays beside it as `achieved` / `algorithmic_bytes`
    assert rf["algorithmic_bytes_per_launch"] == rf["format_bytes_per_launch"] <= rf["csr_bytes_per_launch"] and 0.4 < rf["frac"] < 1.0, rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["csr_equivalent_gbps"] >= rf["achieved"], rf
    fmt = [r for _, r in roofs if "format_bytes" in r]
    assert len(fmt) >= 20 and all(r["format_bytes"] <= r.get("algorithmic_bytes", r.get("algorithmic_bytes_rank0")) for r in fmt)
    assert d["comm"]["world"] == 1 and d["comm"]["rccl_leg"] is None
    # every timed region of the variant / extra legs is on record (the faster of two is what the leg reports)
    tr = d["timed_regions_ms"]
    assert len(tr) >= 15 and all(len(v) >= 1 and min(v) > 0 for v in tr.values()), tr
    assert rf["traffic"] is not None, "rocprofv3 counter passes did not deliver"
    assert 0.85 * rf["format_bytes_per_launch"] < rf["traffic"] < 1.5 * rf["format_bytes_per_launch"], rf
    assert 1.7 < rf["traffic_detail"]["fetch_factor_reproducing_k_vec_FPlainQ"] < 2.3
