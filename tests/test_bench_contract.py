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
    for wl in ("banded", "fem_like", "laplace7"):
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


def test_graft_entry_has_build_and_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert callable(g.build) and callable(g.smoke)
