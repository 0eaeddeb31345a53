"""Randomised shapes through every product of the fixed-point family (scripts/fuzz_products.py): N and M around the tile and
block boundaries, missing genotypes, NA phenotypes, monomorphic / all-missing markers, operands over 60 decades, random work
decompositions.  Both resident layouts bit-identical everywhere; Ax / ATx / statistics against the oracle."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_both_layouts_bit_identical_and_vs_oracle(oracle):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_products.py")
    spec = importlib.util.spec_from_file_location("fuzz_products", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    bad = fz.main(120, 20261003)
    assert not bad, bad


def test_random_solver_runs_device_loop_host_loop_and_marker_shards():
    """scripts/fuzz_solvers.py: gv_cg_solve / gv_cg_solve2x / the XXT solvers on random shards -- device-resident loop against
    the host-driven one, and 2-4 in-process marker shards (empty ones included, exchange overlapped or not) against one shard"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_solvers.py")
    spec = importlib.util.spec_from_file_location("fuzz_solvers", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    bad = fz.main(100, 3)
    assert not bad, bad
    # seed 77: case 14 has an EMPTY marker shard under the pipelined XXT solver -- the rank without markers enqueues no pass, so
    # the N-space search direction the pass advances on its way in (CgHook::pn) has to be advanced by a launch of its own there
    bad = fz.main(40, 77)
    assert not bad, bad


def test_random_full_vamp_runs_vs_oracle(oracle):
    """scripts/fuzz_vamp.py: random shapes / priors / models (linear, XXT denoiser, probit) / kernel families / layouts /
    fuse levels / divide_work marker shards against the oracle's run of the same configuration"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_vamp.py")
    spec = importlib.util.spec_from_file_location("fuzz_vamp", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    bad = fz.main(40, 5)
    assert not bad, bad


def test_random_pvalue_runs_vs_oracle(oracle):
    """scripts/fuzz_pvals.py: LOO / LOCO p-values on random shards, both kernel families, both resident layouts"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_pvals.py")
    spec = importlib.util.spec_from_file_location("fuzz_pvals", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    bad = fz.main(80, 9)
    assert not bad, bad
