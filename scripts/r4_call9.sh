# round 4, call 9: the two failing cases of the solver fuzz batch (seed 402) on this tree and on the round-3 tree, then the shipped table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O
python3 scripts/fuzz_replay.py 402 312 572 > $O/replay_r4.txt 2>&1; cat $O/replay_r4.txt | cut -c1-400
(cd gpurun_r3tree && python3 scripts/fuzz_replay.py 402 312 572) > $O/replay_r3.txt 2>&1; cat $O/replay_r3.txt | cut -c1-400
python3 scripts/tune_table.py --votes 5 --out $O/gv_tune_builtin.h > $O/tune_table.log 2>&1; tail -16 $O/tune_table.log
echo done
