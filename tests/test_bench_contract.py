"""bench.py / __graft_entry__.py driver contract, the parts that can be checked without a GPU."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_flags_and_loud_failure_without_gpu():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert f'"{flag}"' in src
    import torch
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0                       # no CPU fallback: the HIP path or nothing
    assert "needs a GPU" in (out.stderr + out.stdout)
    assert out.stdout.strip() == ""                  # and no fake result line on stdout


def test_bench_json_keys_are_the_contract():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert re.search(rf'"{key}"\s*:', src), key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_detail"):
        assert re.search(rf'"{key}"\s*:', src), key
    # round-2 additions: the other workloads north_star names and per-variant roofline fractions ride on the same line
    for key in ("variant_rooflines", "extras", "variants_ms_per_iteration", "cpu_baseline_multicore"):
        assert re.search(rf'"{key}"\s*:', src), key
    # round-3 additions: the box's own STREAM rates as roofline denominators (no hard-coded 6290 any more), and at N > 1
    # the communicator report + the second leg over RCCL collectives
    for key in ("stream_measured_gbps", "frac_of_measured_stream", "frac_of_measured_copy", "frac_of_measured_read", "comm"):
        assert re.search(rf'"{key}"\s*:', src), key
    for key in ("world", "p2p_selftest", "rccl_nranks", "transport_used", "fallback_reason"):
        assert f'"{key}"' in src, key
    assert "6290.0" not in src and "rccl_leg" in src
    for key in ("laplace7_512_ca", "transport_rank_of_8", "plan_seconds", "generate_seconds"):
        assert key in src, key
    for wl in ("banded", "fem_like", "laplace7", "mesh"):
        assert f'"{wl}"' in src
    assert "--half-bandwidth" in src and "--no-extras" in src and "--no-traffic" in src
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert re.search(rf'\b{key}=|"{key}"', open(os.path.join(ROOT, "tools", "cpu_baseline.py")).read()), key


def test_bench_names_the_bound_of_every_leg():
    """round 4: every variant / extra entry carries `bound` (hbm | mall | lds | latency) and `frac` against that bound only
    (bench.py roof()); the latency-bound persistent iterations claim no bandwidth fraction; the two-ranks-with-halo leg and the
    512^3 residual check are part of the default line"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('bound="latency"', 'bound="mall"', 'bound="hbm"', "mall_read8", "small_rank_with_halo", "true_relres_after_timed_region"):
        assert key in src, key
    assert "frac=None" in src
    # fractions on the bytes the stored layout moves; the CSR figure beside them
    for key in ("format_bytes", "format_gbps", "format_bytes_per_launch", "frac_of_format_bytes", "csr_bytes_per_launch", "csr_equivalent_gbps"):
        assert key in src, key
    # second half of round 4: every timed region of the variant / extra legs and the set-up seconds of every leg are in the line
    for key in ("timed_regions_ms", "setup_seconds", "setup_seconds_rank0"):
        assert key in src, key
    # round 5: the headline is the median of several regions, each on record with the device's clock beside the host's; the
    # SURVEY.md 8d basis of the roofline fraction is a named field again, the same matrix without its uniform slices and the
    # unstructured (FEM-like) product have blocks of their own; old keys keep their meaning (iteration_algorithmic_bytes = 8d figure)
    for key in ("headline_regions", "device_ms", "enqueue_ms", "--regions", "frac_basis", "survey_8d_frac", "structure_dependence",
                "roofline_unstructured", "iteration_format_bytes", "timed_region_clocks", "plane_marching_product", "wait_us", "comm_wait_us"):
        assert key in src, key
    assert re.search(r'"iteration_algorithmic_bytes":\s*iter_bytes,', src)
    assert "SHARING one device" in src             # two ranks on one GPU are not "over xGMI"


COMPACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "roofline_unstructured", "cpu_baseline", "cpu_baseline_multicore", "comm",
                "variants_ms_per_iteration", "extras", "full_record")
ROOFLINE_KEYS = ("kernel", "bound", "peak", "unit", "achieved", "frac", "frac_basis", "survey_8d_frac", "algorithmic_bytes_per_launch",
                 "avg_launch_ms", "traffic", "frac_of_measured_copy")


def test_the_stdout_line_is_compact():
    """Round 5's 27 KB line did not fit the driver's capture (BENCH_r05.json: parsed null). The line on stdout is now formed from
    the complete record by bench.compact(): < 8 KB, the contract's keys, roofline and cpu_baseline with the keys the review
    named, one {ms_per_iteration, bound, frac} per extra leg. Checked on the recorded complete lines of rounds 4-5 and on a record
    inflated with long texts."""
    import glob
    import json
    sys.path.insert(0, ROOT)
    import bench
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[45]", "bench_n1_driver_flags*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r06", "bench_full*.json")))
    assert recs
    for path in recs:
        full = json.load(open(path))
        text = bench.compact(full)
        assert len(text) < bench.COMPACT_LIMIT == 8192 and "\n" not in text, (path, len(text))
        line = json.loads(text)
        assert set(COMPACT_KEYS) <= set(line), (path, set(COMPACT_KEYS) - set(line))
        assert set(ROOFLINE_KEYS) <= set(line["roofline"])
        assert {"workload", "rows", "nnz", "method", "transport", "iterations_genuine"} <= set(line["config"])
        assert line["value"] == float(f"{full['value']:.6g}") and line["roofline"]["bound"] == "hbm"
        if full.get("cpu_baseline"):
            assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
        for name, e in line["extras"].items():
            for leg in ([e] if "ms_per_iteration" in e else [v for v in e.values() if isinstance(v, dict)]):
                assert set(leg) <= {"ms_per_iteration", "ms", "bound", "frac", "iterations_genuine", "error"}, (name, leg)
    # texts of any length cannot push the line over the limit
    full = json.load(open(recs[-1]))
    full["config"]["workload"] = "w" * 5000
    full["config"]["transport"] = "t" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    full["roofline"]["frac_basis"] = "f" * 5000
    full["cpu_baseline"] = dict(full.get("cpu_baseline") or {}, sample="s" * 5000, flags="g" * 5000)
    assert len(bench.compact(full)) < 8192
    full["extras"] = {f"leg{i}": {"bicgstab": {"ms_per_iteration": 1.0, "bound": "hbm", "frac": 0.5}} for i in range(400)}
    text = bench.compact(full)
    assert len(text) < 8192 and json.loads(text)["roofline"]["frac"] is not None


def test_bench_relaunches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus N` with no launcher around it re-executes under torch.distributed.run (one rank per GPU) instead of
    exiting 2; here, without a GPU, every rank then fails loudly"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "torch.distributed.run" in src and "os.execv(sys.executable" in src
    import torch
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert "re-executing under torch.distributed.run" in out.stderr and out.returncode != 0
    assert out.stderr.count("needs a GPU") >= 2 and out.stdout.strip() == ""


def test_graft_entry_has_build_and_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert callable(g.build) and callable(g.smoke)
