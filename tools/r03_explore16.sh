# round 3 exploration 16 (GPU box): does it matter WHICH XCD takes which channel?  The as-dispatched numbering rotated by 0..7
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3v; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do for rot in 0 1 2 3 4 5 6 7; do for shape in 64x64x4096x256 64x16x32768x256; do
  echo -n "rep $rep rotation $rot: " >> $O/bfacc_rot.log
  env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_ORDER=$((16+rot)) python tools/measure.py bfacc --modes 0 --shape $shape --random 2>&1 | grep "int8" >> $O/bfacc_rot.log
done; done; done
sed 's/ -> .*T coefficient-products\/s,/ ->/; s/, [0-9.]* TFLOP.*//; s/\[int8 fixed point\]//' $O/bfacc_rot.log
