# round 3 exploration 8 (GPU box): streaming slabs with the per-variant autotune cache; autotune tests; bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "autotune or tunings or streaming or config5" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python tools/measure.py stream > $O/stream.log 2>&1
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
grep -v "^SUMMARY" $O/stream.log | cut -c1-260
