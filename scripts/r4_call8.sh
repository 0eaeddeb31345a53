# round 4, call 8: randomised batches on the round's tree (products incl. geometric splits, solvers, whole runs, p-values, mid sizes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
timeout -k 10 280 python3 scripts/fuzz_products.py 1200 401 > $O/fuzz_products.log 2>&1; echo "products rc $?"; tail -2 $O/fuzz_products.log
timeout -k 10 280 python3 scripts/fuzz_solvers.py 1200 402 > $O/fuzz_solvers.log 2>&1; echo "solvers rc $?"; tail -2 $O/fuzz_solvers.log
timeout -k 10 280 python3 scripts/fuzz_vamp.py 250 403 > $O/fuzz_vamp.log 2>&1; echo "vamp rc $?"; tail -2 $O/fuzz_vamp.log
timeout -k 10 120 python3 scripts/fuzz_pvals.py 500 404 > $O/fuzz_pvals.log 2>&1; echo "pvals rc $?"; tail -2 $O/fuzz_pvals.log
timeout -k 10 200 python3 scripts/fuzz_midsize.py 8 405 > $O/fuzz_midsize.log 2>&1; echo "midsize rc $?"; tail -2 $O/fuzz_midsize.log
echo done
