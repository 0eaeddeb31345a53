"""CPU-side checks (no GPU): the C-ABI libraries load and export every symbol their headers declare, the product
never touches oracle/, the host CLI rejects bad input before any device work, and the input generator is stable."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, txt)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return True


def test_libgvamp_exports_every_declared_symbol(built):
    from gvamp_amd import capi
    L = capi.load()
    names = _declared("gvamp.h", "gv_")
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert set(names) == set(capi.EXPORTS), set(names) ^ set(capi.EXPORTS)
    assert L.gv_abi_version() == capi.ABI_VERSION == 4


def test_libgvamp_host_exports(built):
    from gvamp_amd import hostapi
    L = hostapi.load()
    for n in _declared("gvamp_host.h", "gvh_"):
        assert hasattr(L, n), n


def test_integration_section_b_is_the_compiled_binding(built):
    """INTEGRATION.md section B and tests/binding/data_binding.cpp (between its BEGIN / END markers) are the same text, the
    binding library was built from it, and the text calls none of the entry points that select an engine: what a maintainer
    pastes is what tests/test_gpu_binding.py runs, on the library's defaults."""
    src = open(os.path.join(ROOT, "tests", "binding", "data_binding.cpp")).read()
    b0, b1 = "// ---- BEGIN INTEGRATION.md section B ----\n", "// ---- END INTEGRATION.md section B ----"
    block = src[src.index(b0) + len(b0):src.index(b1)]
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## B. Bind the C ABI"):doc.index("## C. Python")]
    code = sec[sec.index("```cpp\n") + 7:]
    code = code[:code.index("```")]
    assert code == block
    called = set(re.findall(r"\b(gv_[a-z0-9_]+)\s*\(", re.sub(r"//.*", "", block)))
    assert called == {"gv_abi_version", "gv_create", "gv_set_dims", "gv_set_mask", "gv_upload_bed", "gv_last_error", "gv_comm_unique_id",
                      "gv_comm_init", "gv_comm_init_callback", "gv_marker_stats", "gv_get_marker_stats", "gv_ax", "gv_atx",
                      "gv_destroy"}, called
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "binding", "libgvbinding.so"))
    for n in ("bh_create", "bh_destroy", "bh_ax", "bh_atx", "bh_stats", "bh_time_ax", "bh_time_atx", "bh_kernel_mode",
              "bh_layout", "bh_set_transport"):
        assert hasattr(lib, n), n


def test_defaults_of_the_c_abi_are_the_measured_engine():
    """gv_internal.h: a context nobody configured runs the i8 MFMA family on a re-encoded layout picked at ingest and keeps no raw
    rows (round 2 defaulted to the fp64 family and 300 GB resident at the headline)."""
    txt = open(os.path.join(ROOT, "gvamp_amd", "csrc", "gv_internal.h")).read()
    assert re.search(r"int kernel_mode = 1;", txt)
    assert re.search(r"bool want_raw = false, want_stripes = true;", txt) and re.search(r"bool want_auto = true;", txt)


def test_shipped_tuning_table_belongs_to_the_current_kernels():
    """gv_tune_builtin.h records the hash of the streaming-kernel sources it was measured on; the library ignores a table whose
    hash differs (it then measures), but a stale table in the tree means scripts/tune_table.py has to run again on an MI355X."""
    from gvamp_amd import build
    txt = open(os.path.join(ROOT, "gvamp_amd", "csrc", "gv_tune_builtin.h")).read()
    h = re.search(r'GV_BUILTIN_FOR_HASH = "([0-9a-f]+|none)"', txt).group(1)
    assert h in ("none", build.kernel_src_hash()[:16]), "re-run scripts/tune_table.py: the kernels changed after the table was measured"
    if h != "none":
        rows = re.findall(r"^    \{(\d+), (\d+), ([01]), ", txt, flags=re.M)
        assert ("400000", "1000000", "0") in rows and ("400000", "125000", "0") in rows and ("100000", "500000", "0") in rows


def test_no_cpu_fallback(built):
    """Without a HIP device the product fails loudly (with a GPU present it simply works)."""
    from gvamp_amd import capi
    try:
        sh = capi.Shard(16, 4)
    except capi.GvError as e:
        assert "no CPU fallback" in str(e) or "HIP" in str(e)
    else:
        sh.close()


def test_product_never_references_oracle():
    bad = []
    for dp, _dn, fn in os.walk(os.path.join(ROOT, "gvamp_amd")):
        for f in fn:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"gvoracle|gv_oracle|libgvoracle|from oracle|import oracle", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_cli_errors_before_device_work(built):
    exe = os.path.join(ROOT, "gvamp_amd", "gvamp_sim")
    r = subprocess.run([exe, "--no-such-flag", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "unknown" in r.stdout                       # options.cpp:421-424
    r = subprocess.run([exe, "--N", "10"], capture_output=True, text=True)
    assert r.returncode != 0 and "no bed file" in r.stdout                   # options.cpp:449-452
    r = subprocess.run([exe, "--iterations", "0", "--bed-file", "x"], capture_output=True, text=True)
    assert r.returncode != 0 and "strictly positive" in r.stdout             # options.cpp:217-225
    r = subprocess.run([exe, "--bed-file"], capture_output=True, text=True)
    assert r.returncode != 0 and "missing argument" in r.stdout              # options.cpp:441-444
    exe2 = os.path.join(ROOT, "gvamp_amd", "gvamp_main_real")
    r = subprocess.run([exe2, "--run-mode", "nonsense", "--bed-file", "x"], capture_output=True, text=True)
    assert r.returncode != 0 and "unknown --run-mode" in r.stdout


def test_synth_bed_is_stable_and_plausible():
    from gvamp_amd import synth
    a = synth.synth_bed(403, 64, seed=1234, miss_ppm=5000)
    assert a.size == 64 * 101 and a.dtype == np.uint8
    assert np.array_equal(a, synth.synth_bed(403, 64, seed=1234, miss_ppm=5000))
    assert np.array_equal(synth.synth_bed(403, 70, seed=1234, miss_ppm=5000, S=0)[6 * 101:],
                          synth.synth_bed(403, 64, seed=1234, miss_ppm=5000, S=6))   # shard-consistent
    assert int(a.sum()) == 1286649         # known answer: the device generator must reproduce these bytes
    codes = np.unpackbits(a.reshape(64, 101), axis=1, bitorder="little").reshape(64, 404, 2)
    c = codes[:, :403, 0] + 2 * codes[:, :403, 1]
    assert 0.001 < np.mean(c == 1) < 0.012                                   # missing ~0.5 %
    assert np.all(codes[:, 403:, :] == 0)                                    # pad bits


def test_committed_bench_line_honours_the_contract():
    """The newest profiles/r<k>_bench_n1.json is the line `python bench.py` printed on an MI355X (library defaults): every key the
    driver and the judge read must be there, with consistent arithmetic (frac = achieved / peak, value = bytes / time)."""
    import glob
    import json
    line = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]_bench_n1.json")))[-1]
    rnd = os.path.basename(line).split("_")[0]
    d = json.load(open(line))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "genotype_matvec_GBps" and d["unit"] == "GB/s" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_kernel_ms"] * 1e-3) / 1e9) < 1.0
    pm = json.load(open(os.path.join(ROOT, "profiles", rnd + "_pmc_traffic.json")))   # the counter passes of the same call
    tile = "tile" in r["kernel"]                                              # (the default layout since round 5: one tile layout)
    k_ax, k_atx = ("tile_ax", "tile_atx") if tile else ("ax", "atx")
    assert pm[k_ax]["hbm_bytes"] >= r["alg_bytes_per_launch"]                # HBM traffic cannot undercut the algorithm
    assert pm[k_ax]["hbm_bytes"] <= 1.02 * r["alg_bytes_per_launch"] and pm[k_atx]["hbm_bytes"] <= 1.02 * r["alg_bytes_per_launch"]
    if r["traffic"] is not None:
        assert r["traffic"] == pm[k_ax]["hbm_bytes"]
    assert "library defaults" in d["config"]["engine"] and d["hostptr_GBps"] > 0.9 * d["value"]
    step_bytes = 2 * r["alg_bytes_per_launch"]                               # one Ax + one ATx per step
    assert abs(d["value"] - step_bytes / (d["ms_per_step"] * 1e-3) / 1e9) < 1.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    v = d["vamp"]                      # the second half of the metric: iterations/s with the work behind it
    for k in ("iters_per_s", "n_ax_pass", "n_atx_pass", "cg_iters", "onsager_iters", "time_to_solution_s", "reference_sequence",
              "x_hat_rel_l2"):
        assert k in v, k
    assert v["x_hat_rel_l2"] < 1e-9 and v["iters_per_s"] > v["reference_sequence"]["iters_per_s"]
    ld = d["vamp_ld"]                  # the LD leg carries its own parity fields since round 3
    assert ld["x_hat_rel_l2"] < 1e-9 and ld["counts_equal_reference_sequence"] is True and max(ld["cg_iters"]) >= 30
    side = d.get("two_stripe_sets") or d["tile_layout"]      # the same step on the other resident layout, in the same process
    assert side.get("bit_identical_to_main_leg", side.get("bit_identical_to_two_layouts")) is True
    if rnd >= "r5":
        # round 5: the mid-size rows ride in the driver-run line, the CPU baseline is a timing build with measured iterations
        assert [w["row"] for w in d["rows"]] == ["config2", "shard_8gpu", "config4", "config5"]
        for w in d["rows"]:
            for lvl in ("fuse_4", "fuse_0"):
                q = w[lvl]
                assert abs(q["frac"] - q["pass_GBps"] / 8000.0) < 1e-3 and q["iters_per_s"] > 0 and q["passes"] > 0
            assert w["fuse_4"]["iters_per_s"] > w["fuse_0"]["iters_per_s"] and w["x_hat_rel_l2_fuse4_vs_0"] < 1e-9
        assert "-O3 -march=native" in c["flags"] or "parity build" in c["flags"]
        m = c["vamp_iter_s_measured"]
        assert "config1_N2000_M10000" in m and any(k.startswith("slice_") for k in m) and c["measured_over_model"] > 0.5
        assert c["vamp_iter_s_extrapolated"] > 0
        assert v["time_to_solution_s"] < 4.0


def test_run_sharded_launcher_sets_the_rank_environment(tmp_path):
    """scripts/run_sharded.py: one process per shard with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and a private rendezvous
    path; the exit code of a failing rank is propagated."""
    import subprocess
    import sys
    launcher = os.path.join(ROOT, "scripts", "run_sharded.py")
    probe = ("import os,sys; open(os.path.join(sys.argv[1], 'r' + os.environ['RANK']), 'w').write("
             "' '.join(os.environ[k] for k in ('RANK','LOCAL_RANK','WORLD_SIZE','MASTER_ADDR','MASTER_PORT','GVAMP_RENDEZVOUS')))")
    r = subprocess.run([sys.executable, launcher, "-n", "3", "--master-port", "29700", "--", sys.executable, "-c", probe, str(tmp_path)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    seen = sorted(open(tmp_path / ("r%d" % i)).read().split() for i in range(3))
    assert [s[0] for s in seen] == ["0", "1", "2"] and all(s[1] == s[0] and s[2] == "3" and s[3] == "127.0.0.1" and s[4] == "29700" for s in seen)
    assert len({s[5] for s in seen}) == 1
    bad = subprocess.run([sys.executable, launcher, "-n", "2", "--", sys.executable, "-c",
                          "import os,sys,time; sys.exit(7) if os.environ['RANK']=='1' else time.sleep(30)"], timeout=60)
    assert bad.returncode == 7


def test_bench_never_downgrades_the_gpu_count():
    """`bench.py --gpus N` must not print a line for another world size (runs without a GPU: both refusals happen before
    anything touches one).  Under a launcher with WORLD_SIZE != --gpus: exit 2.  Started bare on a box with fewer GPUs: exit 3."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, bench, "--gpus", "8"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr and not r.stdout.strip()
    import torch
    if torch.cuda.device_count() < 8:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        r = subprocess.run([sys.executable, bench, "--gpus", "8"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert r.returncode == 3 and "GPU" in r.stderr and not r.stdout.strip()


def test_ctypes_structs_match_the_header(tmp_path):
    """the by-value / by-pointer structs of include/gvamp.h as ctypes declares them (gvamp_amd/capi.py): same sizes and field
    offsets as the C compiler gives them -- a field added to the header and forgotten in the binding shows up here, without a GPU"""
    import ctypes as C
    import subprocess
    from gvamp_amd import capi
    src = tmp_path / "sizes.c"
    src.write_text("""
#include <stddef.h>
#include <stdio.h>
#include "gvamp.h"
int main(void) {
    printf("gv_cg_stats %zu %zu %zu\\n", sizeof(gv_cg_stats), offsetof(gv_cg_stats, rel_res), offsetof(gv_cg_stats, n_relres));
    printf("gv_cg_extras %zu %zu %zu\\n", sizeof(gv_cg_extras), offsetof(gv_cg_extras, a_mu_a), offsetof(gv_cg_extras, ata_mu_b));
    printf("gv_cg_warm %zu %zu %zu\\n", sizeof(gv_cg_warm), offsetof(gv_cg_warm, ata_v_b), offsetof(gv_cg_warm, have_ata_v_b));
    printf("gv_aat_warm %zu %zu %zu\\n", sizeof(gv_aat_warm), offsetof(gv_aat_warm, accumulate_at_mu_a), offsetof(gv_aat_warm, have_ata_v_b));
    printf("gv_dot_spec %zu %zu %zu\\n", sizeof(gv_dot_spec), offsetof(gv_dot_spec, ya), offsetof(gv_dot_spec, sync));
    printf("gv_decomp_info %zu %zu %zu\\n", sizeof(gv_decomp_info), offsetof(gv_decomp_info, balanced_cells), offsetof(gv_decomp_info, whole_quads));
    printf("gv_decomp_info_tail %zu %zu %zu\\n", offsetof(gv_decomp_info, geo), offsetof(gv_decomp_info, wgs_per_cu), offsetof(gv_decomp_info, xcd_skew));
    return 0;
}
""")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = {l.split()[0]: tuple(int(x) for x in l.split()[1:]) for l in subprocess.check_output([str(exe)], text=True).splitlines()}
    want = {
        "gv_cg_stats": (C.sizeof(capi.CgStats), capi.CgStats.rel_res.offset, capi.CgStats.n_relres.offset),
        "gv_cg_extras": (C.sizeof(capi.CgExtras), capi.CgExtras.a_mu_a.offset, capi.CgExtras.ata_mu_b.offset),
        "gv_cg_warm": (C.sizeof(capi.CgWarm), capi.CgWarm.ata_v_b.offset, capi.CgWarm.have_ata_v_b.offset),
        "gv_aat_warm": (C.sizeof(capi.AatWarm), capi.AatWarm.accumulate_at_mu_a.offset, capi.AatWarm.have_ata_v_b.offset),
        "gv_dot_spec": (C.sizeof(capi.DotSpec), capi.DotSpec.ya.offset, capi.DotSpec.sync.offset),
        "gv_decomp_info": (C.sizeof(capi.DecompInfo), capi.DecompInfo.balanced_cells.offset, capi.DecompInfo.whole_quads.offset),
        "gv_decomp_info_tail": (capi.DecompInfo.geo.offset, capi.DecompInfo.wgs_per_cu.offset, capi.DecompInfo.xcd_skew.offset),
    }
    assert got == want
