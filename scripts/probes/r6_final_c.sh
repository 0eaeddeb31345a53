cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_final_c; rm -rf $O; mkdir -p $O
bash scripts/forced_multi_trace.sh r6_fm > $O/fm.log 2>&1
python3 scripts/forced_multi_report.py gpurun_out/r6_fm > $O/r6_forced_multi_gaps.txt 2>$O/fm_report.err
python3 scripts/drift_check.py 50000 200000 40 1 > $O/drift_xxt.json 2>$O/drift_xxt.err
python3 scripts/drift_check.py 100000 200000 40 0 > $O/drift_lin.json 2>$O/drift_lin.err
python3 scripts/leak_check.py > $O/leak.txt 2>&1
python3 scripts/stress_lifecycle.py > $O/stress.txt 2>&1
ls -la $O
