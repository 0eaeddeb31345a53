"""Randomised solver runs (development; run on a GPU box):   python scripts/fuzz_solvers.py [cases] [seed]
Per case a random shard (N, M around the tile / block boundaries, missing genotypes, NA phenotypes, either resident layout),
random tau / gam2 / iteration cap / warm start / rider, and
  A. gv_cg_solve and gv_cg_solve2x with the device-resident loop against the host-driven loop (GV_CG_DEVICE=0): same
     iteration counts, product counts and traces, iterates to 1e-12, the rider's product bit for bit;
  B. the same solves on 2-4 in-process marker shards (random cut points, empty shards allowed, exchange overlapped or not)
     against the single shard: same iteration counts, iterates to 1e-9 (the sharded sums add in another order), and the
     overlapped exchange bit-identical to the one-message form;
  C. (in every run of A and B) a second solve warm-started from the first with another (v, tau, gam2), its opening residual
     taken from the products the first solve left and solve b's first step from A^T A v_b (gv_cg_solve2w) against the explicit
     applications: same step counts, two Ax and two ATx fewer, iterates at the conditioning of the operator."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gvamp_amd import capi, synth

EDGE_N = [4, 5, 63, 64, 65, 255, 256, 257, 511, 513, 1023, 1024, 1025, 4097]
EDGE_M = [1, 2, 3, 5, 63, 64, 65, 127, 129, 255, 256, 257, 1023, 1025, 2049]
_group = [int(time.time()) % 100000 * 10]


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def close(a, b, tol):
    """rel. l2 distance below tol; where the reference side is not finite (a degenerate case: one marker gives every
    individual a 0/0 variance in compute_people_statistics, as in the reference) the two sides must be non-finite alike"""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if a.shape != b.shape:
        return False
    if not np.all(np.isfinite(b)):
        return bool(np.array_equal(np.isfinite(a), np.isfinite(b)))
    return bool(rel(a, b) < tol)


def trace_close(a, b, rtol):
    """residual traces of two runs that round differently: the first steps agree to rounding; later ones only while CG has
    not lost its orthogonality (a non-converging run on a tiny ill-conditioned system is chaotic in its late steps)"""
    n = min(len(a), len(b), 5)
    return bool(np.allclose(a[:n], b[:n], rtol=rtol, atol=1e-9, equal_nan=True))


def pick(rng, edges, hi):
    return int(rng.choice(edges)) if rng.random() < 0.5 else int(rng.integers(2, hi))


class host_loop:
    def __enter__(self):
        os.environ["GV_CG_DEVICE"] = "0"

    def __exit__(self, *a):
        os.environ.pop("GV_CG_DEVICE", None)


def range_close(a, b, mu, key, N, M, tol):
    """A^T mu_a (key atm) / A A^T mu_a (key aat) of two runs whose mu_a agree to `tol`: the difference on the scale of the operands,
    ||A||^k ||mu_a|| with ||A|| ~ 1 + sqrt(M / N), k = 1 / 2 -- and never further than 1e-2 apart on their own norm."""
    a, b, mu = np.asarray(a), np.asarray(b), np.asarray(mu)
    if a.shape != b.shape:
        return False
    if not (np.all(np.isfinite(b)) and np.all(np.isfinite(mu))):        # (degenerate people statistics: non-finite alike, see close)
        return bool(np.array_equal(np.isfinite(a), np.isfinite(b)))
    if not np.all(np.isfinite(a)):
        return False
    norm_a = 1.0 + np.sqrt(M / N)
    scale = norm_a ** (2 if key == "aat" else 1) * float(np.linalg.norm(mu))
    err = float(np.linalg.norm(a - b))
    if scale == 0.0:
        return err == 0.0
    own = float(np.linalg.norm(b))
    return err <= tol * scale and (own == 0.0 or err <= 1e-2 * own or err <= 1e-13 * scale)


def solves(sh, M, S, P):
    """the two solver entry points on one shard (or one rank of a group); returns everything comparable"""
    va, vb = sh.vecM(P["va"][S:S + M]), sh.vecM(P["vb"][S:S + M])
    mu0 = sh.vecM(P["mu0"][S:S + M]) if P["warm"] else None
    rx = sh.vecM(P["rx"][S:S + M]) if P["ride"] else None
    mu = sh.vecM()
    sh.counters(reset=True)
    st, rr = sh.cg_solve(va, mu0, P["tau"], P["gam2"], P["denoiser"], P["max_iter"], mu)
    c1 = sh.counters(reset=True)
    mu_a, mu_b, ro, amu, ata = sh.vecM(), sh.vecM(), sh.vecN(), sh.vecN(), sh.vecM()
    ata_a, ata_vb = sh.vecM(), sh.vecM()
    (sa, ra), (sb, rb) = sh.cg_solve2x(va, mu0, vb, P["tau"], P["gam2"], P["max_iter"], mu_a, mu_b, ride_x=rx,
                                       ride_out=ro if P["ride"] else None, a_mu_a=amu, ata_mu_b=ata, ata_mu_a=ata_a,
                                       ata_v_b=ata_vb)       # (captures A^T A vb from solve b's first application: output only)
    c2 = sh.counters(reset=True)
    # C. the next solve of a VAMP run: warm-started from mu_a with another (v, tau, gam2) -- the opening residual taken from the
    #    products the solve above left (gv_cg_solve2w) against the opening operator application; collectives included, this is
    #    what a sharded --fuse-solves 3 run does
    if True:      # (every rank of a group, an empty shard included: the solves are collective)
        va2 = sh.vecM(P["va2"][S:S + M])
        start = sh.vecM(mu_a.download())
        me, mbe, mw, mbw, amu_e = sh.vecM(), sh.vecM(), sh.vecM(), sh.vecM(), sh.vecN()
        t2, g2 = P["tau"] * 0.8, P["gam2"] * 1.7
        (se, re_), (sbe, _) = sh.cg_solve2x(va2, start, vb, t2, g2, P["max_iter"], me, mbe, a_mu_a=amu_e)
        ce = sh.counters(reset=True)
        known_b = P["max_iter"] > 0          # (captured above only if solve b applied the operator at all)
        (sw, rw), (sbw, _) = sh.cg_solve2x(va2, start, vb, t2, g2, P["max_iter"], mw, mbw, a_mu_a=amu, ata_mu_start_a=ata_a,
                                           a_mu_start_a=amu, ata_mu_a=ata_a, ata_v_b=ata_vb, have_ata_v_b=known_b)
        cw = sh.counters(reset=True)
        # (rounding differences between two correct CG runs grow with the conditioning of the operator; a run cut off far from
        # convergence amplifies them without bound: values are compared for converged runs, at a tolerance that follows kappa)
        kap = 1.0 + t2 / g2 * (1.0 + np.sqrt(sh.Mt / sh.N)) ** 2
        ctol = min(1e-5, 1e-12 * kap ** 2 + 1e-9)
        tame = P["max_iter"] >= 8 and se.converged and min(sh.N, sh.Mt) > 130
        assert (sw.iters, sbw.iters) == (se.iters, sbe.iters) or not tame, ("chained steps", sw.iters, sbw.iters, se.iters, sbe.iters)
        if tame:
            assert close(mw.download(), me.download(), ctol), ("chained mu", rel(mw.download(), me.download()), ctol)
            assert close(amu.download(), amu_e.download(), ctol), ("chained A mu", rel(amu.download(), amu_e.download()), ctol)
        less = 1 + (1 if known_b else 0)     # the warm start's application, and solve b's first one
        assert cw["n_ax"] == ce["n_ax"] - less and cw["n_atx"] == ce["n_atx"] - less or (sw.iters, sbw.iters) != (se.iters, sbe.iters), \
            ("chained products", cw["n_ax"], ce["n_ax"], less)
        if tame:
            assert close(mbw.download(), mbe.download(), ctol), ("chained mu_b", rel(mbw.download(), mbe.download()), ctol)
        for q in (va2, start, me, mbe, mw, mbw, amu_e):
            q.free()
    keys = ("n_ax", "n_atx", "n_ax_pass", "n_atx_pass")
    x = {}
    if P["xxt"]:       # the N-space solver of --use-XXT-denoiser 1, alone and sharing its passes with the Onsager solve
        sh.compute_people_statistics()
        vn = sh.vecN(P["vn"])
        mn0 = sh.vecN(P["mn0"]) if P["warm"] else None
        mn, mn2, atm, mb2, aat, ata2 = sh.vecN(), sh.vecN(), sh.vecM(), sh.vecM(), sh.vecN(), sh.vecM()
        s1, r1 = sh.cg_solve_aat(vn, mn0, P["tau"], P["gam2"], P["max_iter"], mn)
        (s2, r2), (s3, r3) = sh.cg_solve_aat2(vn, mn0, vb, P["tau"], P["gam2"], P["max_iter"], mn2, atm, mb2, aat_mu_a=aat,
                                              ata_mu_b=ata2)
        x = dict(xit=(s1.iters, s1.converged, s2.iters, s2.converged, s3.iters, s3.converged), xr1=r1, xr2=r2, xr3=r3,
                 mn=mn.download(), mn2=mn2.download(), atm=atm.download(), mb2=mb2.download(), aat=aat.download(),
                 ata2=ata2.download())
        # the form the VAMP loop uses at levels 3 / 4 (A^T mu_a accumulated inside the solve, a rider, the right-hand side completed
        # inside): its steady state is enqueued ahead of its statuses (gv_cg_solve_aat2w) -- against the same pipeline with every
        # fused piece as a launch of its own and an in-stream exchange between them (gv_debug_force_multi: what a sharded job runs):
        # every output and every counter bit for bit
        outs = []
        for forced in ((0, 1) if sh.L.gv_comm_size(sh.h) == 1 else (0, 0)):       # (a rank of a group is sharded already)
            if forced:
                sh.force_multi(forced)
            vq = sh.vecN(P["vn"])
            q_mn, q_at, q_mb, q_aat, q_ata, q_ro, q_po = sh.vecN(), sh.vecM(), sh.vecM(), sh.vecN(), sh.vecM(), sh.vecN(), sh.vecN()
            sh.counters(reset=True)
            (sa_, ra_), (sb_, rb_) = sh.cg_solve_aat2(vq, mn0, vb, P["tau"], P["gam2"], P["max_iter"], q_mn, q_at, q_mb, aat_mu_a=q_aat,
                                                      ata_mu_b=q_ata, accumulate_at_mu_a=True, pre_x=va, pre_out=q_po,
                                                      ride_x=mu, ride_out=q_ro)
            cq = sh.counters()
            outs.append(((sa_.iters, sa_.converged, sa_.n_relres, sb_.iters, sb_.converged, sb_.n_relres) + tuple(cq[k] for k in keys),
                         [ra_, rb_] + [q.download() for q in (vq, q_mn, q_at, q_mb, q_aat, q_ata, q_ro, q_po)]))
            for q in (vq, q_mn, q_at, q_mb, q_aat, q_ata, q_ro, q_po):
                q.free()
        if sh.L.gv_comm_size(sh.h) == 1:
            sh.force_multi(0)
        assert outs[0][0] == outs[1][0], ("pipelined XXT solver: counts", outs[0][0], outs[1][0])
        for a_, b_ in zip(outs[0][1], outs[1][1]):
            assert np.array_equal(a_, b_, equal_nan=True), "pipelined XXT solver: outputs differ from its unfused, exchanged form"
    return dict(**x, it=(st.iters, st.converged, st.n_relres, sa.iters, sa.converged, sa.n_relres, sb.iters, sb.converged, sb.n_relres),
                cnt=tuple(c1[k] for k in keys) + tuple(c2[k] for k in keys), rr=rr, ra=ra, rb=rb, ons=sb.onsager,
                mu=mu.download(), mu_a=mu_a.download(), mu_b=mu_b.download(), ro=ro.download(), amu=amu.download(),
                ata=ata.download())


def make_shard(N, M, Mt, S, bed_rows, layout, m4, nonas):
    sh = capi.Shard(N, M, Mt=Mt, S=S)
    sh.set_layout(False, layout)
    sh.set_kernel_mode(1)
    sh.upload_bed(bed_rows)
    if m4 is not None:
        sh.set_mask(m4, nonas)
    return sh


def run_group(N, Mt, bed, cuts, layout, m4, nonas, P, overlap):
    nr = len(cuts) - 1
    out, errors = [None] * nr, []
    _group[0] += 1
    group = _group[0]
    mb = (N + 3) // 4

    def work(rank):
        try:
            S, M = cuts[rank], cuts[rank + 1] - cuts[rank]
            with make_shard(N, M, Mt, S, bed[S * mb:(S + M) * mb], layout, m4, nonas) as sh:
                sh.comm_init_local(group, nr, rank)
                sh.set_overlap(overlap)
                sh.compute_markers_statistics()
                out[rank] = solves(sh, M, S, P)
        except Exception as e:   # noqa: BLE001
            errors.append((rank, repr(e)))

    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(nr)]
    for t in th:
        t.start()
    t_end = time.time() + 40
    for t in th:
        t.join(timeout=max(0.1, t_end - time.time()))
    if any(t.is_alive() for t in th):     # its peers left the sequence of collectives: nothing after this is trustworthy
        raise RuntimeError("a rank is stuck in a collective; errors so far: %r" % (errors,))
    assert not errors, ("rank failed", errors)
    return out


def run_case(seed0, k):
    rng = np.random.default_rng(seed0 * 100003 + k)
    N, M = pick(rng, EDGE_N, 3000), pick(rng, EDGE_M, 4000)
    miss = int(rng.choice([0, 2000, 50000]))
    fna = float(rng.choice([0.0, 0.0, 0.02]))
    layout = int(rng.integers(1, 3))
    bed = synth.synth_bed(N, M, seed=int(rng.integers(1 << 30)), miss_ppm=miss)
    present = rng.random(N) >= fna
    m4, nonas = None, N
    if fna > 0 or N % 4:
        m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
        for n in np.nonzero(present)[0]:
            m4[n >> 2] |= 1 << (n & 3)
        nonas = int(present.sum())
    P = dict(tau=float(10.0 ** rng.uniform(-2, 2)), gam2=float(10.0 ** rng.uniform(-3, 2)), denoiser=int(rng.integers(2)),
             max_iter=int(rng.choice([0, 1, 2, 5, 40])), warm=bool(rng.integers(2)), ride=bool(rng.integers(2)),
             va=rng.standard_normal(M), vb=np.sign(rng.standard_normal(M)) / np.sqrt(M), mu0=rng.standard_normal(M) * 0.1,
             rx=rng.standard_normal(M), xxt=bool(rng.random() < 0.4), va2=rng.standard_normal(M))
    n4 = 4 * ((N + 3) // 4)
    P["vn"], P["mn0"] = np.zeros(n4), np.zeros(n4)
    P["vn"][:N] = rng.standard_normal(N) * present
    P["mn0"][:N] = rng.standard_normal(N) * present * 0.1
    info = dict(N=N, M=M, miss=miss, fna=fna, layout=layout, **{q: P[q] for q in ("tau", "gam2", "denoiser", "max_iter", "warm", "ride", "xxt")})
    # ---- A: device-resident loop against the host-driven loop
    with make_shard(N, M, M, 0, bed, layout, m4, nonas) as sh:
        sh.compute_markers_statistics()
        d = solves(sh, M, 0, P)
        with host_loop():
            h = solves(sh, M, 0, P)
    assert d["it"] == h["it"], ("iteration counts", info, d["it"], h["it"])
    assert d["cnt"] == h["cnt"], ("product counts", info, d["cnt"], h["cnt"])
    for key in ("rr", "ra", "rb"):
        assert len(d[key]) == len(h[key]) and np.allclose(d[key], h[key], rtol=1e-9, atol=1e-13), (key, info)
    for key in ("mu", "mu_a", "mu_b", "amu", "ata"):
        assert rel(d[key], h[key]) < 1e-12, (key, info, rel(d[key], h[key]))
    if P["ride"]:
        assert np.array_equal(d["ro"], h["ro"]), ("rider", info)
    # rounding differences between two correct CG runs grow with the conditioning of tau A A^T + gam2 I (largest eigenvalue of
    # A A^T ~ (1 + sqrt(M/N))^2) and, once a run fails to converge within its cap, without bound: iterates are compared when the
    # solve converged or was capped within 5 steps, traces over their first 5 steps
    kappa = 1.0 + P["tau"] / P["gam2"] * (1.0 + np.sqrt(M / N)) ** 2
    short = P["max_iter"] <= 5
    # (40 steps on a system of fewer unknowns than that: CG has long lost orthogonality, two roundings drift apart further)
    loose = 1e3 if (not short and min(N, M) <= 130) else 1.0
    if P["xxt"]:
        # (the N-space solver's device scalars use fused multiply-adds the host-driven form does not: equal to rounding)
        for key in ("xr1", "xr2", "xr3"):
            assert trace_close(d[key], h[key], 1e-13 * kappa ** 2 + 1e-9), (key, info, d[key][:5], h[key][:5])
        okx = (short or (d["xit"][1] and h["xit"][1]), short or (d["xit"][3] and h["xit"][3] and d["xit"][5] and h["xit"][5]))
        # (seed 2718, case 318: N = 64, M = 63, 38 steps to the 1e-4 stopping rule -- both runs are 1.2e-4 from the exact solution
        # and 2.8e-6 from each other; seed 60606, case 2744: N = 440, M = 5, 19 steps on a rank-5 system -- A^T mu 1.2e-4 and 8.6e-5
        # from the exact value, 4.5e-5 from each other: in the lost-orthogonality regime two roundings differ by a fraction of the
        # stopping rule, and agree with the exact solution no better than that)
        tolx = loose * (1e-13 * kappa ** 2 + 1e-9)
        if loose > 1.0:
            # (seed 402, case 572, reproduced bit for bit on the round-3 tree: N = 2958, M = 3, 40 steps on a rank-3 system with
            # kappa = 2.9e3 -- A^T mu of the two loops 1.2e-3 apart: the 1e-4 stopping rule times sqrt(kappa) is the scale there)
            tolx = max(tolx, 2e-4, 1e-4 * np.sqrt(kappa))
        # A^T mu_a sees only the component of mu_a in the range of A, which is ~ 1 / kappa of mu_a when gam2 << tau (the rest sits in
        # the null space of A^T, divided by gam2 alone): two solutions that agree to tolx relative to ||mu_a|| agree to kappa * tolx
        # there (seed 512, case 111, reproduced bit for bit on the round-3 tree: N = 1025, M = 2, kappa = 1.8e4 -- mu_a within 3e-8, A^T
        # mu_a 3.5e-4 apart)
        # -- so these two are compared on the scale of what they are made from: ||A^T (mu_d - mu_h)|| <= ||A|| ||mu_d - mu_h||, i.e.
        # the error of A^T mu_a against ||A|| ||mu_a|| (||A|| ~ 1 + sqrt(M / N) for these matrices) at the tolerance of mu_a itself,
        # and A A^T mu_a against ||A||^2 ||mu_a||.  No factor kappa: where mu_a lies in the range of A this is kappa times tighter than
        # a relative comparison widened by kappa (a dropped or doubled `A mu += alpha A p` shows), and where it does not it is the
        # same bound.  On their own norm the two may still be off by kappa * tolx; that is capped at 1e-2 (worst case on record 3.6e-3).
        for key, ok in (("mn", okx[0]), ("mn2", okx[1]), ("atm", okx[1]), ("mb2", okx[1]), ("aat", okx[1]), ("ata2", okx[1])):
            if ok and key in ("atm", "aat"):
                assert range_close(d[key], h[key], d["mn2"], key, N, M, tolx), (key, info, rel(d[key], h[key]), kappa)
            elif ok:
                assert close(d[key], h[key], tolx), (key, info, rel(d[key], h[key]), kappa)
    # ---- B: marker shards in one process
    nr = int(rng.integers(2, 5))
    cuts = sorted(int(c) for c in rng.integers(0, M + 1, size=nr - 1))
    cuts = [0] + cuts + [M]
    info["cuts"] = cuts
    plain = run_group(N, M, bed, cuts, layout, m4, nonas, P, 0)
    ovl = run_group(N, M, bed, cuts, layout, m4, nonas, P, int(rng.integers(2, 6)))
    for r in range(nr):
        for key in plain[r]:
            a, b = plain[r][key], ovl[r][key]
            assert (a == b) if isinstance(a, tuple) else np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True), ("overlap", key, info)
        assert plain[r]["it"] == plain[0]["it"], ("ranks disagree on iteration counts", info, plain[r]["it"], plain[0]["it"])
    if short:
        assert plain[0]["it"] == d["it"], ("sharded iteration counts", info, plain[0]["it"], d["it"])
    cat = lambda key: np.concatenate([plain[r][key] for r in range(nr)])
    # a sharded run adds the same terms in another order: compare at the conditioning of the solve, not bit for bit
    tol_sh = loose * (1e-13 * kappa ** 2 + 1e-9)
    conv = lambda it, i: short or bool(it[i])
    if conv(d["it"], 1):
        assert rel(cat("mu"), d["mu"]) < tol_sh, ("sharded mu", info, rel(cat("mu"), d["mu"]), kappa)
    if conv(d["it"], 4) and conv(d["it"], 7):
        for key in ("mu_a", "mu_b"):
            assert rel(cat(key), d[key]) < tol_sh, ("sharded " + key, info, rel(cat(key), d[key]), kappa)
        # A^T A mu_b = (v_b - r_b - gam2 mu_b) / tau is a difference: its accuracy is that of its terms, not of the result (case 816 of
        # seed 909002: N = M = 3, gam2 = 62, tau = 21 -- the product itself is ~0)
        scale_ata = (np.linalg.norm(P["vb"]) + P["gam2"] * np.linalg.norm(d["mu_b"])) / P["tau"]
        e_ata = np.linalg.norm(cat("ata") - d["ata"]) / scale_ata
        assert e_ata < tol_sh, ("sharded ata", info, e_ata, kappa)
        assert rel(plain[0]["amu"], d["amu"]) < tol_sh, ("sharded amu", info, rel(plain[0]["amu"], d["amu"]), kappa)
    for key in ("rr", "ra", "rb"):
        assert trace_close(plain[0][key], d[key], tol_sh), ("sharded " + key, info, plain[0][key][:5], d[key][:5])
    if P["ride"]:
        # every rank quantises its own slice of the rider with its own exponent and the slices' products are added in another order:
        # equal at the scale of the operand, not of a result that may cancel (seed 402, case 312, also on the round-3 tree: N = 4, M = 3)
        e_ro = np.linalg.norm(plain[0]["ro"] - d["ro"])
        assert e_ro < 1e-12 * max(np.linalg.norm(d["ro"]), np.linalg.norm(P["rx"])), ("sharded rider", info, e_ro)
    if P["xxt"]:
        for key in ("xr1", "xr2", "xr3"):
            assert trace_close(plain[0][key], d[key], tol_sh), ("sharded " + key, info, plain[0][key][:5], d[key][:5])
        if short or all(d["xit"][1::2]):
            # (as in A: 40 steps on a system of a handful of unknowns end at the stopping rule times sqrt(kappa), whoever adds the sums
            # in which order -- seed 613, case 194: N = 827, M = 5, A A^T mu_a of four ranks 1.3e-4 from the single shard's)
            tol_shx = max(tol_sh, 2e-4, 1e-4 * np.sqrt(kappa)) if loose > 1.0 else tol_sh
            for key in ("atm", "mb2", "ata2"):       # (A^T mu_a: on the scale ||A|| ||mu_a||, as above; seed 512, case 896)
                assert (range_close(cat(key), d[key], d["mn2"], key, N, M, tol_shx) if key == "atm" else close(cat(key), d[key], tol_shx)), \
                    ("sharded " + key, info, rel(cat(key), d[key]), kappa)
            for key in ("mn", "mn2", "aat"):      # (aat: the range component again -- seed 721, case 1289: M = 2, kappa = 601, 3.6e-3)
                assert (range_close(plain[0][key], d[key], d["mn2"], key, N, M, tol_shx) if key == "aat" else close(plain[0][key], d[key], tol_shx)), \
                    ("sharded " + key, info, rel(plain[0][key], d[key]), kappa)
    return info


def main(ncases, seed):
    t0 = time.time()
    bad = []
    for k in range(ncases):
        try:
            info = run_case(seed, k)
        except AssertionError as e:
            bad.append((k, str(e)))
            print("CASE %d FAILED: %s" % (k, e), flush=True)
            continue
        if k % 10 == 0:
            print("case %d ok %s  (%.0f s)" % (k, info, time.time() - t0), flush=True)
    os.environ.pop("GV_CG_DEVICE", None)
    print("%d cases, %d failed, %.0f s" % (ncases, len(bad), time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
