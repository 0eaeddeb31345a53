"""Seeded randomised differential cases (scripts/fuzz_parity.py): random shapes, missing rates, NA phenotypes, shard
offsets, kernel families, fuse levels, XXT and probit against the CPU oracle; and random marker-sharded runs."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "scripts", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_shapes_and_options_vs_oracle(seed):
    fz = _fuzz()
    rng = np.random.default_rng(seed)
    for i in range(25):
        fz.one_case(rng, i)


def test_random_sharded_runs_vs_oracle():
    fz = _fuzz()
    rng = np.random.default_rng(21)
    for i in range(8):
        fz.sharded_case(rng, 100 + i)
