# round 3 exploration 15 (GPU box): 16 / 12 / 8 sample blocks per workgroup of the staged beamformer at 64 - 1024 beams
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do for rounds in 0 12 8; do for shape in 64x256x4096x256 64x128x4096x256 64x1024x2048x256 64x64x16384x256 64x256x1024x256 64x64x4096x1024; do
  echo -n "rep $rep blocks/wg=$rounds: " >> $O/bfacc_rounds.log
  env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_ROUNDS=$rounds python tools/measure.py bfacc --modes 0 --shape $shape --random 2>&1 | grep "int8" >> $O/bfacc_rounds.log
done; done; done
sed 's/ -> .*T coefficient-products\/s,/ ->/; s/, [0-9.]* TFLOP.*//; s/\[int8 fixed point\]//' $O/bfacc_rounds.log
