# round 3 exploration 11 (GPU box): eight beam tiles per workgroup of eight waves (staged form, > 64 beams)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "beamform_accumulated" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for rep in 1 2; do for nbt in 4 8; do for shape in 64x256x4096x256 64x128x4096x256 64x1024x2048x256 64x256x1024x256 32x256x4096x256; do
  echo -n "rep $rep nbt=$nbt: " >> $O/bfacc_nbt8.log
  env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_NBT=$nbt python tools/measure.py bfacc --modes 0 --shape $shape --random 2>&1 | grep "int8" >> $O/bfacc_nbt8.log
done; done; done
for order in 1 3; do for shape in 64x1024x2048x256 64x256x4096x256; do
  echo -n "nbt=8 order=$order: " >> $O/bfacc_nbt8.log
  env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_NBT=8 DCS_BACC_ORDER=$order python tools/measure.py bfacc --modes 0 --shape $shape --random 2>&1 | grep "int8" >> $O/bfacc_nbt8.log
done; done
sed 's/ -> .*T coefficient-products\/s,/ ->/; s/, [0-9.]* TFLOP.*//; s/\[int8 fixed point\]//' $O/bfacc_nbt8.log
