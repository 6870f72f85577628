# round 3 exploration 5 (GPU box): fp16 tests with the new sign logic; fine geometry landscape of the b16 form; clocks under its load
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -k "b16 or fp16 or beamform_accumulated or non_finite or streaming" > $O/pytest_fp16.log 2>&1 || { tail -30 $O/pytest_fp16.log; exit 1; }
tail -2 $O/pytest_fp16.log
python tools/measure.py fp16 --modes 4 --form 3 --tpb 1 --wpc=4,5,6,7,-1 --cpb 16,20,24,28,32,36,40,48 > $O/fp16_fine.log 2>&1
( python tools/measure.py fp16 --modes 4 --form 3 --cpb 24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24 --wpc=6 > $O/fp16_hold.log 2>&1 & )
for t in 1 2 3 4 5 6; do sleep 1.5; rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Power (W)" >> $O/smi_fp16.txt; done
sleep 3
grep "^best\|^library" $O/fp16_fine.log; cat $O/smi_fp16.txt
