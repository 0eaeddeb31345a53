# round 4, call 5: the whole GPU suite on the round's tree; statistics kernel; file ingest with the allocation beside the reads
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
python3 scripts/stats_rate.py 400000 1000000 1 > $O/stats.txt 2>&1; python3 scripts/stats_rate.py 400000 1000000 2 >> $O/stats.txt 2>&1; cat $O/stats.txt
python3 scripts/ingest_rate.py > $O/ingest.json 2>$O/ingest.err; cat $O/ingest.json; tail -3 $O/ingest.err
echo done
