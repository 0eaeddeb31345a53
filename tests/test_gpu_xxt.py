"""--use-XXT-denoiser 1 (denoiserXXT.cpp; BASELINE config 5, matrix-free form): people statistics, the N-space CG and a
full run of the product against the CPU oracle."""
import numpy as np
import pytest

from gvamp_amd import capi, hostapi, synth

pytestmark = pytest.mark.gpu
PROBS, VARS = [0.9, 0.1], [0, 0.01]


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def test_people_statistics_vs_oracle(oracle):
    N, M = 1003, 700
    rng = np.random.default_rng(1)
    bed = synth.synth_bed(N, M, seed=71, miss_ppm=15000)
    present = rng.random(N) >= 0.02
    m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
    for n in np.nonzero(present)[0]:
        m4[n >> 2] |= 1 << (n & 3)
    nonas = int(present.sum())
    o = oracle.people_stats(bed, N, M, mask4=m4, nonas=nonas)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_mask(m4, nonas)
        sh.compute_markers_statistics()
        g = sh.compute_people_statistics()
    for a, b in zip(g, o):
        assert np.allclose(a[:N], b[:N], rtol=1e-10, atol=1e-13)
    assert np.all(g[0][:N][~present] == 0) and np.all(g[1][:N][~present] == 0)


@pytest.mark.parametrize("mode", [0, 1])
def test_xxt_run_vs_oracle(oracle, mode):
    N, M = 600, 2000
    bed = synth.synth_bed(N, M, seed=72, miss_ppm=5000)
    beta, y = oracle.sim_phen(bed, N, M, 0.5, 60, 3)
    ref = oracle.infere(bed, N, M, y, PROBS, VARS, iterations=3, CG_max_iter=40, rho=0.5, seed=3, true_signal=beta,
                        use_XXT_denoiser=1)
    std = oracle.infere(bed, N, M, y, PROBS, VARS, iterations=3, CG_max_iter=40, rho=0.5, seed=3, true_signal=beta)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(mode)
        r = hostapi.infere_linear(sh, y, PROBS, VARS, iterations=3, CG_max_iter=40, rho=0.5, seed=3, true_signal=beta,
                                  use_XXT_denoiser=1)
    assert [t["cg_iters"] for t in r.trace] == [int(t["cg_iters"]) for t in ref.trace]
    assert rel(r.x_est, ref.x_est) < 1e-7
    assert rel(r.x2[2], ref.x2[2]) < 1e-7
    # Woodbury: the N-space solve gives the M-space LMMSE estimate up to the two CG tolerances (1e-4 / 1e-5)
    assert rel(ref.x_est, std.x_est) < 5e-3
