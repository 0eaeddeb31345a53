#!/usr/bin/env python3
"""make_fixtures.py -- regenerates every reference-derived fixture under tests/golden/ from /root/reference.

Step 1 runs build_ref.sh (the real reference, real Boost, no stand-ins).  If that reports "reference unbuildable" this
script stops there and says "parity unpinned": nothing is written.  Otherwise it

  * writes the inputs (seeded numpy recipes): three tiny beds -- N%4=0 without missing genotypes, N%4=0
    with missing genotypes, N%4!=0 with NA phenotypes (scalar build only) -- and the N=2000 x Mt=10000 toy set;
  * runs oracle/_ref/harness on each bed  -> tests/golden/ref/<bed>_{mave,msig,x,Ax,p,ATx,g1_grid,prior_*,cg_*,pvals,...}
  * runs oracle/_ref/sim_scalar at np = 1, 2, 8 with the command line of tests/golden/survey_probe/README.md
                                         -> tests/golden/ref/sim_np{1,2,8}_*.bin, *_gam{1,2}s.csv, sim_np1_run.log
  * runs main_real (NA-bearing .phen; run modes infere / test), main_real with --use-XXT-denoiser 1, main_real_probit
    and the p-value modes                -> tests/golden/ref/{real,xxt,probit,pvals}_*.bin
  * stamps tests/golden/ref/PROVENANCE.json (reference commit date, compiler, Boost version, the commands).

tests/test_ref_fixtures.py compares oracle/ with every file found there; with an empty tests/golden/ref/ it skips and
states "parity unpinned".  Everything here is container-only test tooling: /root/reference does not exist on the GPU box.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFBIN = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(ROOT, "tests", "golden", "ref")
MPIEXEC = os.environ.get("MPIEXEC", "/opt/conda/bin/mpiexec")
SIM_ARGS = ["--N", "2000", "--Mt", "10000", "--iterations", "3", "--num-mix-comp", "3", "--probs", "0.90,0.07,0.03", "--vars",
            "0,0.001,0.01", "--CV", "500", "--h2", "0.5", "--rho", "0.5", "--CG-max-iter", "20", "--model", "linear", "--seed",
            "7", "--store-pvals", "0"]


def write_bed(path, N, M, seed, miss):
    """PLINK .bed: magic 6c 1b 01, SNP-major, individual 4j+k in bits 2k..2k+1 of byte j; 2 -> 00, 1 -> 10, 0 -> 11,
    missing -> 01, pad bits 00."""
    rng = np.random.default_rng(seed)
    maf = rng.uniform(0.05, 0.5, M)
    g = rng.binomial(2, maf[:, None], (M, N))
    code = np.choose(g, [3, 2, 0]).astype(np.uint8)
    code[rng.random((M, N)) < miss] = 1
    pad = (-N) % 4
    code = np.pad(code, ((0, 0), (0, pad)))
    b = code.reshape(M, -1, 4)
    packed = (b[:, :, 0] | (b[:, :, 1] << 2) | (b[:, :, 2] << 4) | (b[:, :, 3] << 6)).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        f.write(packed.tobytes())


def write_phen(path, N, seed, n_na):
    rng = np.random.default_rng(seed)
    y = rng.standard_normal(N) * 2.0 + 0.3
    na = set(rng.choice(N, n_na, replace=False).tolist())
    with open(path, "w") as f:
        for i in range(N):
            f.write("%d %d %s\n" % (i, i, "NA" if i in na else repr(float(y[i]))))


def run(cmd, log=None, np_=1, env=None):
    full = ([MPIEXEC, "-np", str(np_)] if np_ > 1 else []) + cmd
    e = dict(os.environ, OMP_NUM_THREADS="1" if np_ > 1 else os.environ.get("OMP_NUM_THREADS", "4"))
    e.update(env or {})
    r = subprocess.run(full, capture_output=True, text=True, env=e)
    if log:
        open(log, "w").write(r.stdout)
    if r.returncode != 0:
        sys.exit("FAILED: %s\n%s" % (" ".join(full), (r.stdout + r.stderr)[-3000:]))
    return r.stdout


def main():
    rc = subprocess.call([os.path.join(HERE, "build_ref.sh")])
    if rc == 3:
        print("make_fixtures: reference unbuildable here -> parity unpinned; no fixture written")
        return 3
    if rc != 0:
        return rc
    os.makedirs(OUT, exist_ok=True)
    work = os.path.join(REFBIN, "work")
    os.makedirs(work, exist_ok=True)
    cmds = []
    # ---- harness on the three tiny beds (+ a > 50k-marker one for the default prior)
    beds = [("t1", 400, 300, 0.0, None), ("t2", 400, 300, 0.02, None), ("t3", 403, 257, 0.01, 7), ("t4", 64, 50400, 0.005, None)]
    for name, N, M, miss, n_na in beds:
        bed = os.path.join(work, name + ".bed")
        write_bed(bed, N, M, 100 + len(name) + N, miss)
        phen = "-"
        if n_na:
            phen = os.path.join(work, name + ".phen")
            write_phen(phen, N, 5, n_na)
        cmd = [os.path.join(REFBIN, "harness"), bed, str(N), str(M), phen, os.path.join(OUT, name), "11"]
        out = run(cmd)
        trace = out.split("BEGIN_CG_TRACE")[1].split("END_CG_TRACE")[0] if "BEGIN_CG_TRACE" in out else ""
        open(os.path.join(OUT, name + "_cg_trace.txt"), "w").write(trace)
        for f in ([bed] + ([phen] if n_na else [])):
            subprocess.check_call(["xz", "-9", "-k", "-f", f])
            os.replace(f + ".xz", os.path.join(OUT, os.path.basename(f) + ".xz"))
        cmds.append(" ".join(cmd))
    # ---- full sim.cpp runs at np = 1, 2, 8 on the toy set
    toy = os.path.join(work, "toy.bed")
    write_bed(toy, 2000, 10000, 1, 0.01)
    subprocess.check_call(["xz", "-9", "-k", "-f", toy])
    os.replace(toy + ".xz", os.path.join(OUT, "toy.bed.xz"))
    for np_ in (1, 2, 8):
        od = os.path.join(work, "out%d" % np_) + "/"
        os.makedirs(od, exist_ok=True)
        cmd = [os.path.join(REFBIN, "sim_scalar"), "--bed-file", toy, "--out-dir", od, "--out-name", "toy"] + SIM_ARGS
        run(cmd, log=os.path.join(OUT, "sim_np%d_run.log" % np_), np_=np_)
        for k in ("it_1_x2_hat", "it_1", "it_2", "it_3", "it_2_x2_hat", "it_3_x2_hat", "r1_it_2", "r1_it_3"):
            os.replace(od + "toy_%s.bin" % k, os.path.join(OUT, "sim_np%d_%s.bin" % (np_, k)))
        for k in ("gam1s", "gam2s", "R2trains"):
            if os.path.exists(od + "toy_%s.csv" % k):
                os.replace(od + "toy_%s.csv" % k, os.path.join(OUT, "sim_np%d_%s.csv" % (np_, k)))
        if np_ == 1:
            os.replace(od + "toy_beta_true.bin", os.path.join(OUT, "sim_beta_true.bin"))
        cmds.append("np=%d: %s" % (np_, " ".join(cmd)))
    # ---- main_real: NA phenotypes; XXT denoiser; p-values; probit
    phen = os.path.join(work, "toy.phen")
    write_phen(phen, 2000, 9, 4)
    subprocess.check_call(["cp", phen, os.path.join(OUT, "toy.phen")])
    real = ["--model", "linear", "--bed-file", toy, "--phen-files", phen, "--N", "2000", "--Mt", "10000", "--iterations", "3",
            "--probs", "0.90,0.07,0.03", "--vars", "0,0.001,0.01", "--rho", "0.5", "--CG-max-iter", "20", "--seed", "7", "--h2",
            "0.5"]
    variants = {"real": [], "xxt": ["--use-XXT-denoiser", "1"], "pvals": ["--store-pvals", "1"]}
    for tag, extra in variants.items():
        od = os.path.join(work, "out_" + tag) + "/"
        os.makedirs(od, exist_ok=True)
        cmd = [os.path.join(REFBIN, "main_real"), "--run-mode", "infere", "--out-dir", od, "--out-name", "r"] + real + extra
        run(cmd, log=os.path.join(OUT, tag + "_run.log"))
        for f in sorted(os.listdir(od)):
            if f.endswith((".bin", ".csv")):
                os.replace(od + f, os.path.join(OUT, tag + "_" + f[2:]))
        cmds.append(" ".join(cmd))
    od = os.path.join(work, "out_probit") + "/"
    os.makedirs(od, exist_ok=True)
    yb = os.path.join(work, "toy01.phen")
    rng = np.random.default_rng(3)
    with open(yb, "w") as f:
        for i in range(2000):
            f.write("%d %d %d\n" % (i, i, int(rng.random() < 0.4)))
    subprocess.check_call(["cp", yb, os.path.join(OUT, "toy01.phen")])
    cmd = [os.path.join(REFBIN, "main_real_probit"), "--run-mode", "infere", "--model", "bin_class", "--bed-file", toy,
           "--phen-files", yb, "--N", "2000", "--Mt", "10000", "--iterations", "3", "--probs", "0.90,0.07,0.03", "--vars",
           "0,0.001,0.01", "--rho", "0.5", "--CG-max-iter", "20", "--seed", "7", "--out-dir", od, "--out-name", "p"]
    run(cmd, log=os.path.join(OUT, "probit_run.log"))
    for f in sorted(os.listdir(od)):
        if f.endswith((".bin", ".csv")):
            os.replace(od + f, os.path.join(OUT, "probit_" + f[2:]))
    cmds.append(" ".join(cmd))
    prov = {"reference": "/root/reference (medical-genomics-group/gVAMP)", "built": open(os.path.join(REFBIN, "BUILT")).read().strip(),
            "compiler": subprocess.check_output(["g++", "--version"], text=True).splitlines()[0], "commands": cmds}
    json.dump(prov, open(os.path.join(OUT, "PROVENANCE.json"), "w"), indent=1)
    print("make_fixtures: wrote", len(os.listdir(OUT)), "files to", OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
