# round 3 exploration 4 (GPU box): the beamformer tests with the launcher-chosen numbering; fp16 sign-logic A/B; clocks under fp16 load
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_host_abi.py -x -q -k "beamform_accumulated or test_two_contexts or scratch" > $O/pytest_bfacc.log 2>&1 || { tail -30 $O/pytest_bfacc.log; exit 1; }
tail -2 $O/pytest_bfacc.log
for rep in 1 2 3; do
  for lib in dc_sand_amd/csrc/libdcs_beamformer.so tools/variants/libdcs_sign64.so; do
    echo "== rep $rep $lib" >> $O/fp16_sign_ab.log
    env DCS_LIB_PATH=$lib python tools/measure.py fp16 --modes 4 --form 3 --cpb 24,32,48 --wpc=5,6,-1 2>&1 | grep "^fp16\|^best\|^library" >> $O/fp16_sign_ab.log
  done
done
( python tools/measure.py fp16 --modes 4 --form 3 --cpb 32,32,32,32,32,32,32,32,32,32,32,32,32,32,32,32,32,32,32,32,32,32,32,32 --wpc=6 > $O/fp16_hold.log 2>&1 & )
sleep 9; rocm-smi --showclocks --showpower > $O/smi_fp16_a.txt 2>&1; sleep 1; rocm-smi --showclocks --showpower > $O/smi_fp16_b.txt 2>&1; sleep 6
for shape in 256x64x1024x256 256x64x4096x256 64x256x4096x256 64x1024x2048x256 64x16x32768x256 64x64x4096x256; do
  python tools/measure.py bfacc --modes 0 --shape $shape 2>&1 | grep "int8" >> $O/bfacc_product.log
done
grep -i "sclk\|Power (W)" $O/smi_fp16_a.txt $O/smi_fp16_b.txt; tail -4 $O/fp16_sign_ab.log
