#!/bin/bash
# Builds and runs scripts/probes/fp4_mfma_probe.hip (exactness + operand layout of the fp4 x fp8 16x16x128 MFMA on this GPU) and counts
# the vector instructions its 2-bit -> fp4 expansion compiles to.  Output: stdout (kept as profiles/r6_fp4_probe.txt).
set -e
cd "$(dirname "$0")"
T=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w --save-temps=obj -o $T/probe fp4_mfma_probe.hip 2>/dev/null || hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -save-temps -o $T/probe fp4_mfma_probe.hip
S=$(ls $T/*gfx950*.s fp4_mfma_probe-hip-amdgcn-amd-amdhsa-gfx950.s 2>/dev/null | head -1)
echo "== (1) exactness and layout"
$T/probe || true
echo "== (2) the expansion: vector ALU instructions of k_expand_only's loop body (the compiler unrolled it twice: 4 code dwords = 64 codes per trip)"
python3 - "$S" <<'PY'
import re, sys, collections
src = open(sys.argv[1]).read()
body = src[src.index("_Z13k_expand_onlyPKjPji:"):]
body = body[:body.index("s_endpgm")]
# the hot loop = the basic block with the most v_lshl / v_and instructions
blocks = re.split(r"\n\.LBB\d+_\d+:", body)
hot = max(blocks, key=lambda b: len(re.findall(r"\bv_(lshl|and|or)", b)))
ops = collections.Counter(re.findall(r"^\s+(v_[a-z0-9_]+)", hot, re.M))
loads = len(re.findall(r"global_load_dword\b", hot))
expand = {k: v for k, v in ops.items() if re.match(r"v_(lshl|lshr|and|or|bfe|perm|lshl_or|and_or)", k)}
fold = {k: v for k, v in ops.items() if k not in expand}
n = sum(expand.values())
print("code dwords loaded per trip: %d" % loads)
print("expansion instructions (shift / and / or): %d  %s" % (n, dict(expand)))
print("other vector instructions of the trip (checksum fold, addresses, loop): %d" % sum(fold.values()))
print("=> %.1f vector instructions per code dword (16 genotypes) for TWO fp4 planes; the byte expansion of the shipped kernels: 11 (Ax), 13 (tile ATx)"
      % (n / max(loads, 1)))
PY
rm -rf $T fp4_mfma_probe-*gfx950* fp4_mfma_probe-host* fp4_mfma_probe.hip-hip* 2>/dev/null
