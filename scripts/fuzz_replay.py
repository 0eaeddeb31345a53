"""Development: replay single cases of scripts/fuzz_solvers.py.   python scripts/fuzz_replay.py <seed> <case> [<case> ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuzz_solvers

seed = int(sys.argv[1])
for k in sys.argv[2:]:
    try:
        print("case", k, "ok", fuzz_solvers.run_case(seed, int(k)))
    except AssertionError as e:
        print("case", k, "FAILED:", e)
