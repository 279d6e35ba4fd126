"""Seed-switching shifted solvers (reference src/shifted_switching_solver.c):
shifted_lopbicg (per-shift stop flags), shifted_lopbicg_switching and its _noovlp twin.
CPU: the oracle restatement is pinned bit-exactly to the real reference through the committed
fixtures (tests/golden/switching_*.npz, made by make_golden_switching.py from
oracle/_ref/libref_switching.so). GPU: the HIP path against oracle / fixtures."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "switching_*.npz")))


def _coo(g):
    n = int(g["n"])
    row = np.repeat(np.arange(n, dtype=np.uint32), np.diff(g["ptr"].astype(np.int64)))
    return n, row, g["col"], g["val"]


def test_fixtures_present():
    assert len(GOLDEN) >= 4
    assert all(bool(np.load(p)["noovlp_bit_identical"]) for p in GOLDEN)     # _noovlp is an arithmetic twin
    assert sum(int(np.load(p)["sw_k"]) - 1 != int(np.load(p)["flag_k"]) for p in GOLDEN) >= 2   # switches change the course


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_bitexact_vs_reference(path):
    g = np.load(path)
    n, row, col, val = _coo(g)
    flag = O.solve_switching(n, row, col, val, g["b"], g["sigma"], int(g["seed"]), which="shifted_lopbicg")
    assert flag["k"] == int(g["flag_k"])
    assert np.array_equal(flag["x"], g["flag_x"]) and np.array_equal(flag["r"], g["flag_r"])
    assert flag["stop"].all()
    sw = O.solve_switching(n, row, col, val, g["b"], g["sigma"], int(g["seed"]), which="shifted_lopbicg_switching")
    assert sw["k"] == int(g["sw_k"])
    assert np.array_equal(sw["x"], g["sw_x"]) and np.array_equal(sw["r"], g["sw_r"])
    assert sw["stop"].all()


@pytest.mark.parametrize("P", [2, 3])
def test_oracle_switching_virtual_ranks(P):
    """the distributed restatement (P virtual ranks) follows the same course as P = 1: same number of
    switches and final seed, iteration counts within 1, solutions to 1e-9"""
    g = np.load([p for p in GOLDEN if "lin8" in p][0])
    n, row, col, val = _coo(g)
    one = O.solve_switching(n, row, col, val, g["b"], g["sigma"], int(g["seed"]))
    many = O.solve_switching(n, row, col, val, g["b"], g["sigma"], int(g["seed"]), nranks=P)
    assert one["switches"] == many["switches"] >= 1 and one["final_seed"] == many["final_seed"]
    assert abs(one["k"] - many["k"]) <= 1
    assert np.abs(one["x"] - many["x"]).max() <= 1e-9 * np.abs(one["x"]).max()
