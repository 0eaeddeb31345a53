"""Assembles profiles/r<N>_forced_multi_gaps.txt from the directory scripts/forced_multi_trace.sh leaves (development tool).
  python scripts/forced_multi_report.py gpurun_out/<tag> > profiles/r<N>_forced_multi_gaps.txt"""
import csv
import json
import os
import subprocess
import sys

d = sys.argv[1]
here = os.path.dirname(os.path.abspath(__file__))


def step(leg):
    out = subprocess.run([sys.executable, os.path.join(here, "trace_step.py"), os.path.join(d, leg + "_kernel_trace.csv")],
                         capture_output=True, text=True).stdout
    return out


def small(leg):
    # (launches, small-launch us) of the picked step
    lines = step(leg).splitlines()
    n = int(lines[0].split()[3])
    tail = lines[-1]
    us = float(tail.split("other launches")[1].split("us")[0])
    return n, us


def chunks(leg):
    t = [float(ln.split()[1]) for ln in step(leg).splitlines()[2:-1] if "k_mfma" in ln and "<1," in ln]
    return t


def bench(name):
    txt = open(os.path.join(d, name)).read()
    return json.loads([ln for ln in txt.splitlines() if ln.startswith("{")][-1])


n0, s0 = small("plain")
n1, s1 = small("forced")
n2, s2 = small("forced_ov4")
ch = chunks("forced_ov4")
whole = chunks("forced")
print("# the product's multi-rank branches on ONE GPU over an asynchronous in-stream exchange (gv_debug_force_multi / GVAMP_FORCE_MULTI)")
print("# shape: one shard of the 8-GPU headline job, N=400k x M=125k (12.5 GB), tile layout, --fuse-solves 4, device-resident CG")
print("# produced by scripts/forced_multi_trace.sh + scripts/forced_multi_report.py (rocprofv3 --kernel-trace; trace_gaps.py, trace_step.py)")
print("#")
print("# transport 3 = ncclAllReduce on a 1-rank RCCL communicator, then the loop-back kernels (k_loop_out: message -> scratch, buffer")
print("# poisoned with NaNs; k_loop_in: back).  A 1-rank in-place ncclAllReduce launches NO kernel (nothing shows in the trace), so at one")
print("# rank it is the loop-back pair that stands where the RCCL kernel of an 8-rank job will stand.")
print("#")
print("# A steady-state CG step (two systems on two-vector passes):")
print("#   plain one rank : %d launches --  prep_ax quant | Ax2 | fin_ax | prep_atx quant | ATx2 | fin_atx_dot | cgx_ab | cgx_decide" % n0)
print("#   forced multi   : %d launches --  + [exchange of w_n|w_n2] + k_scale + k_ride_copy behind Ax2, + 2 x k_finalize + [exchange of <d,p>]" % n1)
print("#                                    behind ATx2, + 2 x k_finalize + [exchange of <v,mu>, <r,z>, <r,r>] behind cgx_ab")
print("#                                    ([exchange] = k_loop_out + k_loop_in here, ONE ncclAllReduce kernel on an 8-rank job)")
print("#   small launches per step: %.1f us -> %.1f us, back to back in both" % (s0, s1))
print("#   forced + GV_OVERLAP=4: %d launches; the individual-chunks of Ax2 take %s = %.0f us against %s us undivided:" % (
    n2, " + ".join("%.0f" % t for t in ch), sum(ch), " / ".join("%.0f" % t for t in whole)))
print("#                                    each chunk runs a quarter of the row groups with the decomposition tuned for the whole; the")
print("#                                    exchange it hides is <= 0.1 ms -> GV_OVERLAP stays off.")
print("#")
print("# bench.py on the same shard (N=400k x Mt=125k, 1 GPU), step = lmmse_mult, and vamp it/s:")
print("#   GVAMP_FORCE_MULTI  GV_OVERLAP  ms_per_step  value GB/s  ms_allreduce_per_ax  vamp it/s")
for f, fm, ov in (("bench_fm0.json", 0, 0), ("bench_fm2.json", 2, 0), ("bench_fm3.json", 3, 0), ("bench_fm3_ov4.json", 3, 4)):
    b = bench(f)
    print("#   %17d  %10d  %11.4f  %10.1f  %19.4f  %9.2f" % (fm, ov, b["ms_per_step"], b["value"], b["forced_multi"]["ms_allreduce_per_ax"],
                                                          b["vamp"]["iters_per_s"]))
print("#   (ms_allreduce_per_ax: HIP events around exchange + k_scale; with GV_OVERLAP the exchange runs on the side stream, not bracketed)")
print()
for leg in ("plain", "forced", "forced_ov4"):
    txt = open(os.path.join(d, leg + "_gaps.txt")).read()
    head = txt.split("one CG step")[0].rstrip()
    print(head)
    print()
    print(step(leg))
