# round 3 exploration 3 (GPU box): parity of the kChain beamformer, its rate against kSplit and the numbering; generator PMC + clocks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "beamform_accumulated or test_two_contexts" > $O/pytest_bfacc.log 2>&1 || { tail -30 $O/pytest_bfacc.log; exit 1; }
tail -2 $O/pytest_bfacc.log
for rep in 1 2; do
 for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  for shape in 256x64x1024x256 256x64x4096x256 256x16x4096x256 256x256x256x256 128x64x1024x256 192x64x1024x256 130x20x512x256; do
    echo -n "rep $rep order=$1 split=$2: " >> $O/bfacc_chain.log
    env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_ORDER=$1 DCS_BACC_UNSTAGED=$2 python tools/measure.py bfacc --modes 0 --shape $shape 2>&1 | grep "int8" >> $O/bfacc_chain.log
  done
 done
done
for order in 1 0; do for shape in 64x512x2048x256 64x128x4096x256; do
  echo -n "order=$order: " >> $O/bfacc_g8.log
  env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_ORDER=$order python tools/measure.py bfacc --modes 0 --shape $shape 2>&1 | grep "int8" >> $O/bfacc_g8.log
done; done
bash tools/pmc_kernels.sh $O/pmc_gen > $O/pmc_gen.txt 2>&1
# clocks and power under the fp32 and the fp16 generator (readings only)
( python tools/measure.py sustained --seconds 5 > $O/sustained_fp32.log 2>&1 & ) ; sleep 2.5; rocm-smi --showclocks --showpower > $O/smi_fp32.txt 2>&1; sleep 4
( python tools/measure.py fp16 --modes 4 --form 3 --cpb 32,32,32,32,32,32,32,32,32,32,32,32 --wpc=6 > $O/fp16_hold.log 2>&1 & ) ; sleep 4; rocm-smi --showclocks --showpower > $O/smi_fp16.txt 2>&1; sleep 5
grep -i "sclk\|power\|mclk\|fclk" $O/smi_fp32.txt $O/smi_fp16.txt | head -20
