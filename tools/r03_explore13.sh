# round 3 exploration 13 (GPU box): kChain at 4 waves per SIMD (127 VGPRs: coefficients straight to LDS, results recombined one
# register at a time) against the same source allocated for 3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3r; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_host_abi.py -x -q -k "beamform_accumulated or scratch" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for rep in 1 2 3; do for lib in tools/variants/libdcs_chain3.so dc_sand_amd/csrc/libdcs_beamformer.so; do for shape in 256x64x1024x256 256x64x4096x256 128x64x1024x256 256x16x4096x256 192x256x512x256; do
  echo -n "rep $rep $(basename $lib): " >> $O/chain_waves.log
  env DCS_LIB_PATH=$lib python tools/measure.py bfacc --modes 0 --shape $shape --random 2>&1 | grep "int8" >> $O/chain_waves.log
done; done; done
sed 's/ -> .*T coefficient-products\/s,/ ->/; s/, [0-9.]* TFLOP.*//; s/\[int8 fixed point\]//' $O/chain_waves.log
