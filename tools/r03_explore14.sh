# round 3 exploration 14 (GPU box): sample blocks per workgroup at 64 beams (one workgroup per channel: 4096 workgroups = 3.2 generations)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do for rounds in 0 8 4; do for shape in 64x64x4096x256 64x64x2048x256 64x64x8192x256 48x64x4096x256; do
  echo -n "rep $rep blocks/wg=$rounds: " >> $O/bfacc_rounds64.log
  env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_ROUNDS=$rounds python tools/measure.py bfacc --modes 0 --shape $shape --random 2>&1 | grep "int8" >> $O/bfacc_rounds64.log
done; done; done
sed 's/ -> .*T coefficient-products\/s,/ ->/; s/, [0-9.]* TFLOP.*//; s/\[int8 fixed point\]//' $O/bfacc_rounds64.log
