"""Launch one of the C++ drivers on the GPUs of one node, one process per GPU (the reference's `mpirun -np N`).

  python scripts/run_sharded.py -n 8 -- gvamp_amd/gvamp_main_real --run-mode infere --bed-file x.bed ...

Every process gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (what host/utilities.cpp and host/data.cpp read:
marker range by divide_work, GPU = LOCAL_RANK, RCCL unique id exchanged through $GVAMP_RENDEZVOUS).  Exit code = the first
non-zero exit code of a rank; the other ranks are terminated when one fails.

  --comm host   the sums travel through shared memory on the host instead of RCCL (GVAMP_COMM=host, host/shm_comm.cpp)
  --same-gpu    every rank uses GPU 0 (LOCAL_RANK = 0): with --comm host this runs the sharded drivers on a one-GPU box --
                same marker shards, same collective sequence, same files as one process per GPU
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", "--nproc", type=int, required=True, help="processes = GPUs = marker shards")
    ap.add_argument("--master-port", type=int, default=29611)
    ap.add_argument("--comm", choices=("rccl", "host"), default="rccl")
    ap.add_argument("--same-gpu", action="store_true", help="all ranks on GPU 0 (needs --comm host: RCCL wants one GPU per rank)")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    if not cmd:
        ap.error("no command given")
    if a.same_gpu and a.comm != "host":
        ap.error("--same-gpu needs --comm host (an RCCL communicator cannot hold two ranks on one GPU)")
    rdv = tempfile.mkdtemp(prefix="gvamp_rdv_")
    procs = []
    for r in range(a.nproc):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if a.same_gpu else str(r), WORLD_SIZE=str(a.nproc),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.master_port), GVAMP_RENDEZVOUS=os.path.join(rdv, "rccl_id"),
                   GVAMP_COMM=a.comm)
        env.setdefault("NCCL_SOCKET_IFNAME", "lo")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = set(range(a.nproc))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for q in live:
                    procs[q].terminate()
        time.sleep(0.05)
    sys.exit(rc)


if __name__ == "__main__":
    main()
