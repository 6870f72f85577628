# round 3 exploration 6 (GPU box): whole GPU suite with the round's kernels; fp16 landscapes of both forms; bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1 || { tail -40 $O/pytest_all.log; exit 1; }
tail -2 $O/pytest_all.log
python tools/measure.py fp16 --modes 0 --form 3 --tpb 1 --wpc=4,5,6,-1 --cpb 16,24,32,48,64,96 > $O/fp16_mode0.log 2>&1
python tools/measure.py fp16 --modes 4 --form 3 --tpb 1 --wpc=4,5,6,7,-1 --cpb 16,20,24,28,32 > $O/fp16_mode4.log 2>&1
grep "^best\|^library" $O/fp16_mode0.log $O/fp16_mode4.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
