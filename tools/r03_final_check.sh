# round 3: the whole GPU suite, smoke(), the C++ and Python runBeamformerTests mirrors and a timed default bench, on the final build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest exit $?" >> $O/pytest_all.log; tail -3 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"
python tests/run_beamformer_tests.py > $O/run_beamformer_tests.log 2>&1; echo "run_beamformer_tests exit $?"
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; echo "bench exit $?"; cat $O/bench_time.txt
