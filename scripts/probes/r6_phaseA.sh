# development: needs a temporary print around the denoiser section of host/vamp.cpp (fprintf of now_s() differences under GV_PHASE_TIMES, see profiles/r6_denoiser_section.txt);
O=gpurun_out/r6_phaseA; mkdir -p $O
export GV_PHASE_TIMES=1
timeout -k 10 200 python3 scripts/iter_time.py 50000 200000 12 4 1 > $O/cfg5.txt 2>&1 &&
timeout -k 10 200 python3 scripts/iter_time.py 400000 125000 10 4 0 > $O/shard.txt 2>&1 &&
timeout -k 10 200 python3 scripts/iter_time.py 100000 500000 10 4 0 > $O/cfg2.txt 2>&1
tail -n 14 $O/cfg5.txt; tail -n 12 $O/shard.txt; tail -n 12 $O/cfg2.txt
