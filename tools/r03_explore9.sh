# round 3 exploration 9 (GPU box): kChain with the next step's loads issued before / after this step's wait
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "beamform_accumulated" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for rep in 1 2 3; do for lib in tools/variants/libdcs_chain_waitfirst.so dc_sand_amd/csrc/libdcs_beamformer.so; do for shape in 256x64x1024x256 256x64x4096x256 128x64x1024x256 256x16x4096x256; do
  echo -n "rep $rep $(basename $lib): " >> $O/chain_order.log
  env DCS_LIB_PATH=$lib python tools/measure.py bfacc --modes 0 --shape $shape --random 2>&1 | grep "int8" >> $O/chain_order.log
done; done; done
sed 's/ -> .*T coefficient-products\/s,/ ->/; s/, [0-9.]* TFLOP.*//; s/\[int8 fixed point\]//' $O/chain_order.log
