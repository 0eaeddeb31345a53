"""The product's MULTI-RANK branches over an asynchronous, in-stream exchange -- on one GPU.

`is_multi()` is false on a one-rank job, and every multi-rank test of this suite (in-process shards, gloo, shared memory) goes
through the host-synchronising `local` / `callback` transports: two hipStreamSynchronize per collective.  What an 8-GPU job
actually runs -- the N-vector all-reduce inside data::Ax (data.cpp:928/:995) and the packed scalar all-reduces
(utilities.cpp:203) enqueued IN STREAM between kernels that never wait for the host, k_finalize + all-reduce instead of the
one-rank shortcuts of the device-resident CG, the side-stream exchange of GV_OVERLAP with its two event edges -- would run for
the first time on first contact with RCCL.  gv_debug_force_multi makes a one-rank context take those branches with an exchange
that is in-stream as RCCL's: the loop-back transport moves every message through scratch, fills the buffer with NaNs meanwhile
and (delay_us) holds the stream before it brings the message back; transport 2 is ncclAllReduce on a 1-rank communicator.
A sum over one rank is the identity: every output must equal the plain one-rank run BIT FOR BIT, and a consumer that is not
ordered behind its exchange reads NaNs.
"""
import os

import numpy as np
import pytest

from gvamp_amd import capi, hostapi, synth

pytestmark = pytest.mark.gpu
PROBS, VARS = [0.90, 0.07, 0.03], [0, 0.001, 0.01]

# (transport, delay_us): loop-back; loop-back that holds the stream 40 us per message (widens every race window);
# 1-rank RCCL followed by the loop-back
TRANSPORTS = [(1, 0), (1, 40), (3, 0)]


class env:
    def __init__(self, **kw):
        self.kw = {k: v for k, v in kw.items() if v is not None}

    def __enter__(self):
        for k, v in self.kw.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k in self.kw:
            os.environ.pop(k, None)


def _same(a, b, what):
    if isinstance(a, dict):
        assert a.keys() == b.keys(), what
        for k in a:
            _same(a[k], b[k], "%s[%s]" % (what, k))
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), what
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, "%s[%d]" % (what, i))
    elif isinstance(a, np.ndarray):
        assert not np.isnan(b).any(), what + ": NaN (a consumer ran ahead of its exchange)"
        assert np.array_equal(a, b), "%s differs: max |d| = %g" % (what, float(np.max(np.abs(a - b))) if a.shape == b.shape else -1)
    else:
        assert a == b, (what, a, b)


def _shard(N, M, layout, seed=5, fna=0.0, miss_ppm=8000):
    sh = capi.Shard(N, M)
    sh.set_layout(False, layout)          # 1: two stripe sets, 2: one tile layout
    sh.upload_bed(synth.synth_bed(N, M, seed=seed, miss_ppm=miss_ppm))
    if fna > 0 or N % 4:
        rng = np.random.default_rng(N)
        present = rng.random(N) >= fna
        m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
        for n in np.nonzero(present)[0]:
            m4[n >> 2] |= 1 << (n & 3)
        sh.set_mask(m4, int(present.sum()))
    sh.compute_markers_statistics()
    return sh


def _configs():
    """(label, transport, delay, overlap tiles, GV_CG_DEVICE) of every forced run; the plain run of the same GV_CG_DEVICE is the
    reference"""
    out = []
    for cgdev in (1, 0):
        for ov in (0, 4):
            for tr, dl in (TRANSPORTS if cgdev == 1 else TRANSPORTS[:1]):
                out.append(("t%d d%d ov%d cgdev%d" % (tr, dl, ov, cgdev), tr, dl, ov, cgdev))
    return out


def _check_all(sh, run):
    """run() -> nested dict / list of arrays and scalars.  Plain per GV_CG_DEVICE, then every forced configuration."""
    base = {}
    for cgdev in (1, 0):
        with env(GV_CG_DEVICE=None if cgdev else 0):
            base[cgdev] = run()
    for label, tr, dl, ov, cgdev in _configs():
        sh.force_multi(tr, dl)
        sh.set_overlap(ov)
        try:
            with env(GV_CG_DEVICE=None if cgdev else 0):
                got = run()
        finally:
            sh.force_multi(0)
            sh.set_overlap(0)
        _same(base[cgdev], got, label)
    # ... and the hook leaves nothing behind
    _same(base[1], run(), "plain again")


@pytest.mark.parametrize("layout", [1, 2])
def test_matvecs_and_dots_forced_multi(layout):
    N, M = 3001, 2500
    with _shard(N, M, layout, fna=0.02) as sh:
        rng = np.random.default_rng(1)
        x, x2 = sh.vecM(rng.standard_normal(M)), sh.vecM(rng.standard_normal(M))

        def run():
            z, z2, w, w2 = sh.vecN(), sh.vecN(), sh.vecM(), sh.vecM()
            sh.ax_dev(x, z)
            sh.atx_dev(z, w)
            a = [z.download(), w.download()]
            sh.ax2_dev(x, x2, z, z2)
            sh.atx2_dev(z, z2, w, w2)
            a += [z.download(), z2.download(), w.download(), w2.download()]
            a.append(sh.Ax(x.download()))                    # host-pointer forms (gv_ax / gv_atx)
            a.append(sh.ATx(a[-1]))
            lm = sh.vecM()
            sh.lmmse_mult(x, 1.7, 0.3, lm)
            a.append(lm.download())
            a.append(np.array(sh.dots([(x, x2), (w, w2)], sync=1)))
            a.append(sh.allreduce_host(np.arange(5.0)))
            return a

        _check_all(sh, run)


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("warm,denoiser,max_iter", [(False, 1, 40), (True, 1, 40), (False, 0, 40), (True, 1, 3)])
def test_cg_solve_forced_multi(layout, warm, denoiser, max_iter):
    N, M = 2000, 3000
    with _shard(N, M, layout) as sh:
        rng = np.random.default_rng(M + denoiser)
        v = rng.standard_normal(M) * (np.sign(rng.standard_normal(M)) / np.sqrt(M) if denoiser == 0 else 1.0)
        dv = sh.vecM(v)
        mu0 = sh.vecM(rng.standard_normal(M) * 0.1) if warm else None

        def run():
            mu = sh.vecM()
            sh.counters(reset=True)
            st, rr = sh.cg_solve(dv, mu0, 2.0, 0.7, denoiser, max_iter, mu)
            c = sh.counters()
            return dict(st=(st.iters, st.converged, st.n_relres, st.n_ax, st.n_atx, st.rel_res, st.onsager), rr=np.asarray(rr),
                        mu=mu.download(), cnt=[c[k] for k in ("n_ax", "n_atx", "n_ax_pass", "n_atx_pass")])

        _check_all(sh, run)


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("warm,ride,max_iter", [(True, True, 40), (False, True, 40), (True, False, 2)])
def test_dual_solve_with_by_products_forced_multi(layout, warm, ride, max_iter):
    """gv_cg_solve2w as the VAMP loop calls it at --fuse-solves 2..4: lock-step LMMSE + Onsager solves, the rider, A mu_a and
    A^T A mu_b from the recurrences, the warm start's known products, the Onsager probe's kept product."""
    N, M = 2000, 3000
    with _shard(N, M, layout, seed=11) as sh:
        rng = np.random.default_rng(N + M)
        va = sh.vecM(rng.standard_normal(M))
        vb = sh.vecM(np.sign(rng.standard_normal(M)) / np.sqrt(M))
        mu0 = sh.vecM(rng.standard_normal(M) * 0.05) if warm else None
        rx = sh.vecM(rng.standard_normal(M)) if ride else None

        def run():
            mu_a, mu_b, ro, amu, ata = sh.vecM(), sh.vecM(), sh.vecN(), sh.vecN(), sh.vecM()
            sh.counters(reset=True)
            (sa, ra), (sb, rb) = sh.cg_solve2x(va, mu0, vb, 1.3, 0.9, max_iter, mu_a, mu_b, ride_x=rx, ride_out=ro if ride else None,
                                               a_mu_a=amu, ata_mu_b=ata)
            c = sh.counters()
            return dict(sa=(sa.iters, sa.converged, sa.n_relres), sb=(sb.iters, sb.converged, sb.n_relres, sb.onsager),
                        ra=np.asarray(ra), rb=np.asarray(rb), mu_a=mu_a.download(), mu_b=mu_b.download(), ro=ro.download(),
                        amu=amu.download(), ata=ata.download(), cnt=[c[k] for k in ("n_ax", "n_atx", "n_ax_pass", "n_atx_pass")])

        _check_all(sh, run)


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("warm", [False, True])
def test_xxt_joint_solver_forced_multi(layout, warm):
    """gv_cg_solve_aat2w in the form the VAMP loop uses: A^T mu accumulated, right-hand side completed inside, a rider; and the
    stand-alone N-space solve."""
    N, M = 3000, 9000
    with _shard(N, M, layout, seed=91, miss_ppm=5000) as sh:
        sh.compute_people_statistics()
        rng = np.random.default_rng(N + M)
        npad = 4 * ((N + 3) // 4)
        v = np.zeros(npad)
        v[:N] = rng.standard_normal(N)
        du = sh.vecM(np.where(rng.random(M) < 0.5, -1.0, 1.0) / np.sqrt(M))
        dx1, dr2 = sh.vecM(rng.standard_normal(M) * 0.1), sh.vecM(rng.standard_normal(M) * 0.1)
        mu0 = np.zeros(npad)
        mu0[:N] = 0.05 * rng.standard_normal(N)
        dm0 = sh.vecN(mu0) if warm else None

        def run():
            dv = sh.vecN(v)
            mu, at, mb, aat, ata, ro, po = sh.vecN(), sh.vecM(), sh.vecM(), sh.vecN(), sh.vecM(), sh.vecN(), sh.vecN()
            sh.counters(reset=True)
            (sa, ra), (sb, rb) = sh.cg_solve_aat2(dv, dm0, du, 1.5, 0.6, 12, mu, at, mb, aat_mu_a=aat, ata_mu_b=ata,
                                                  accumulate_at_mu_a=True, pre_x=dr2, pre_out=po, ride_x=dx1, ride_out=ro)
            c = sh.counters()
            out = dict(st=(sa.iters, sa.converged, sb.iters, sb.converged, sb.onsager), ra=np.asarray(ra), rb=np.asarray(rb),
                       vecs=[q.download() for q in (dv, mu, at, mb, aat, ata, ro, po)],
                       cnt=[c[k] for k in ("n_ax", "n_atx", "n_ax_pass", "n_atx_pass")])
            one = sh.vecN()
            s1, r1 = sh.cg_solve_aat(sh.vecN(v), dm0, 1.5, 0.6, 12, one)
            out["single"] = [np.asarray(r1), one.download(), np.array([s1.iters, s1.converged])]
            return out

        _check_all(sh, run)


def _trace(r):
    keys = ("cg_iters", "onsager_iters", "revar_rounds", "L_after", "n_ax", "n_atx", "n_ax_pass", "n_atx_pass", "gam1_denoise", "alpha1",
            "eta1", "gam2", "alpha2", "eta2", "gam1_next", "gamw", "probe_product")
    return dict(niter=r.niter, x_est=r.x_est, x1=list(r.x1), x2=list(r.x2), r1=list(r.r1), probs=r.probs, vars=r.vars,
                trace=[[float(t[k]) for k in keys] for t in r.trace])


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("fuse", [0, 4])
def test_infere_linear_forced_multi(layout, fuse):
    """vamp::infere (vamp.cpp:149-803) end to end: denoiser and EM sums (vamp.cpp:313,990,1012), both solves, gamw update."""
    N, M = 2000, 6000
    with _shard(N, M, layout, seed=2024, miss_ppm=5000) as sh:
        beta, y = hostapi.sim_phen(sh, 0.5, 300, 7)
        kw = dict(iterations=5, CG_max_iter=25, rho=0.5, seed=7, gam1=1e-8, gamw=2.0, true_signal=beta, fuse_solves=fuse)
        _check_all(sh, lambda: _trace(hostapi.infere_linear(sh, y, PROBS, VARS, **kw)))


def test_infere_probit_forced_multi():
    from scipy.stats import norm
    N, M = 1001, 1500
    with _shard(N, M, 1, seed=11, miss_ppm=5000) as sh:
        rng = np.random.default_rng(11)
        beta = rng.standard_normal(M) * (rng.random(M) < 0.05) * 0.2
        g = sh.Ax(beta * np.sqrt(N))[:N]
        y = (rng.random(N) < norm.cdf(3 * g)).astype(float)
        kw = dict(iterations=5, CG_max_iter=30, rho=0.5, seed=3, gam1=1e-8, gamw=1.0, model="bin_class")
        _check_all(sh, lambda: _trace(hostapi.infere_linear(sh, y, [0.9, 0.1], [0, 0.05], **kw)))


@pytest.mark.parametrize("fuse", [0, 2])
def test_infere_xxt_forced_multi(fuse):
    N, M = 500, 1800
    with _shard(N, M, 1, seed=74, miss_ppm=5000) as sh:
        beta, y = hostapi.sim_phen(sh, 0.5, 50, 4)
        kw = dict(iterations=3, CG_max_iter=40, rho=0.5, seed=4, use_XXT_denoiser=1, true_signal=beta, fuse_solves=fuse)
        _check_all(sh, lambda: _trace(hostapi.infere_linear(sh, y, PROBS, VARS, **kw)))


@pytest.mark.parametrize("layout", [1, 2])
def test_pvals_loco_forced_multi(layout):
    """data::pvals_calc_LOCO (data.cpp:1235-1353): one Ax with its cross-rank all-reduce per chromosome, the packed `present`
    exchange, then the marker pass; and the leave-one-out form."""
    N, M = 1203, 900
    with _shard(N, M, layout, seed=55, fna=0.02, miss_ppm=15000) as sh:
        rng = np.random.default_rng(21)
        x1 = rng.standard_normal(M) * (rng.random(M) < 0.05) * 3.0
        chrom = np.sort(rng.integers(1, 24, M)).astype(np.int32)
        chrom[chrom == 7] = 8
        dx = sh.vecM(x1)
        dz = sh.vecN()
        sh.ax_dev(dx, dz)
        z1 = dz.download()
        y = z1 + np.concatenate([rng.standard_normal(N), np.zeros(z1.size - N)]) * (z1 != 0)
        dy = sh.vecN(y)

        def run():
            loco, pred = sh.pvals_calc_loco_pred(dz, dy, dx, chrom)
            return [sh.pvals_calc(dz, dy, dx), sh.pvals_calc(dz, dy, dx, chrom=chrom), loco, pred]

        _check_all(sh, run)


def test_force_multi_is_refused_on_a_sharded_context():
    with capi.Shard(512, 256) as sh:
        sh.comm_init_local(9911, 2, 0)
        with pytest.raises(capi.GvError):
            sh.force_multi(1)


def test_the_harness_has_teeth_a_dropped_event_edge_is_seen():
    """Fault injection (transport bit 4): the overlapped exchange returns WITHOUT making the context's stream wait for the side
    stream -- the bug class this file exists for.  With the loop-back holding each message for a millisecond the consumer behind the
    pass reads the poison: the plain comparison above would fail, which is the point."""
    N, M = 40000, 2048
    with _shard(N, M, 1) as sh:
        x = sh.vecM(np.ones(M))
        z = sh.vecN()
        sh.ax_dev(x, z)
        good = z.download()
        sh.set_overlap(4)
        sh.force_multi(1, 1000)
        sh.ax_dev(x, z)
        assert np.array_equal(z.download(), good)            # edges in place: bit-identical, however long the message is away
        sh.force_multi(1 | 4, 1000)
        sh.ax_dev(x, z)
        bad = z.download()
        sh.synchronize()
        assert np.isnan(bad).any() or not np.array_equal(bad, good)
        sh.force_multi(0)
        sh.set_overlap(0)
        sh.ax_dev(x, z)
        assert np.array_equal(z.download(), good)


@pytest.mark.parametrize("driver", ["gvamp_sim", "gvamp_main_real"])
def test_drivers_under_GVAMP_FORCE_MULTI_write_the_same_files(tmp_path, driver):
    """The reference-style executables (sim.cpp / main_real.cpp command lines) with every context of the process forced
    (GVAMP_FORCE_MULTI=1:20 at gv_create; GV_OVERLAP=3): the .bin / .csv files must be the plain run's, byte for byte --
    p-values with their per-chromosome Ax exchanges included."""
    import filecmp
    import lzma
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    G = os.path.join(root, "tests", "golden", "survey_probe")
    bed = tmp_path / "toy.bed"
    bed.write_bytes(lzma.open(os.path.join(G, "toy.bed.xz")).read())
    with open(tmp_path / "toy.bim", "w") as f:
        for i in range(10000):
            f.write("%d\trs%d\t0\t%d\tA\tG\n" % (1 + i // 500, i, i + 1))
    common = ["--bed-file", str(bed), "--bim-file", str(tmp_path / "toy.bim"), "--N", "2000", "--Mt", "10000", "--iterations", "3",
              "--probs", "0.90,0.07,0.03", "--vars", "0,0.001,0.01", "--rho", "0.5", "--CG-max-iter", "20", "--model", "linear",
              "--seed", "7", "--h2", "0.5", "--store-pvals", "1"]
    if driver == "gvamp_sim":
        common += ["--num-mix-comp", "3", "--CV", "500"]
    else:
        common += ["--run-mode", "infere", "--phen-files", os.path.join(G, "toy.phen")]
    outs = {}
    for name, extra in (("plain", {}), ("forced", {"GVAMP_FORCE_MULTI": "1:20", "GV_OVERLAP": "3"})):
        out = str(tmp_path / name) + "/"
        res = subprocess.run([os.path.join(root, "gvamp_amd", driver)] + common + ["--out-dir", out, "--out-name", "t"],
                             capture_output=True, text=True, timeout=600, env=dict(os.environ, **extra))
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        outs[name] = out
    files = sorted(f for f in os.listdir(outs["plain"]) if f.endswith((".bin", ".csv")))
    assert len(files) >= 8 and sorted(f for f in os.listdir(outs["forced"]) if f.endswith((".bin", ".csv"))) == files
    for f in files:
        assert filecmp.cmp(outs["plain"] + f, outs["forced"] + f, shallow=False), f
