# round 3 exploration 10 (GPU box): beam tiles per workgroup of the staged beamformer at >= 64 beams (fewer stores per wave)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do for nbt in 4 2 1; do for shape in 64x256x4096x256 64x64x4096x256 64x1024x2048x256; do
  echo -n "rep $rep nbt=$nbt: " >> $O/bfacc_nbt.log
  env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_NBT=$nbt python tools/measure.py bfacc --modes 0 --shape $shape --random 2>&1 | grep "int8" >> $O/bfacc_nbt.log
done; done; done
for nbt in 2 1; do for order in 1 3; do
  echo -n "nbt=$nbt order=$order: " >> $O/bfacc_nbt.log
  env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_NBT=$nbt DCS_BACC_ORDER=$order python tools/measure.py bfacc --modes 0 --shape 64x256x4096x256 --random 2>&1 | grep "int8" >> $O/bfacc_nbt.log
done; done
sed 's/ -> .*T coefficient-products\/s,/ ->/; s/, [0-9.]* TFLOP.*//; s/\[int8 fixed point\]//' $O/bfacc_nbt.log
