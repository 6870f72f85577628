# round 3 exploration 12 (GPU box): 8- and 16-wave workgroups of the staged beamformer (coefficients of a tile shared by more waves,
# fewer blocks and stores per wave, a smaller window of rows held open)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
for w in 16 8; do
  env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_WAVES=$w timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "beamform_accumulated_on_the_matrix_cores or beamform_accumulated_seeded or beamform_accumulated_non_finite" > $O/pytest_w$w.log 2>&1 || { tail -30 $O/pytest_w$w.log; exit 1; }
  tail -1 $O/pytest_w$w.log
done
for rep in 1 2; do for cfg in "0 0" "4 16" "4 8" "8 16" "2 16"; do set -- $cfg; for shape in 64x256x4096x256 64x64x4096x256 64x1024x2048x256 64x128x4096x256; do
  echo -n "rep $rep nbt=$1 waves=$2: " >> $O/bfacc_waves.log
  env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_NBT=$1 DCS_BACC_WAVES=$2 python tools/measure.py bfacc --modes 0 --shape $shape --random 2>&1 | grep "int8" >> $O/bfacc_waves.log
done; done; done
for cfg in "0 0" "1 8" "1 16" "2 8" "2 16"; do set -- $cfg; for shape in 64x16x32768x256 64x32x16384x256; do
  echo -n "nbt=$1 waves=$2: " >> $O/bfacc_waves.log
  env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_NBT=$1 DCS_BACC_WAVES=$2 python tools/measure.py bfacc --modes 0 --shape $shape --random 2>&1 | grep "int8" >> $O/bfacc_waves.log
done; done
sed 's/ -> .*T coefficient-products\/s,/ ->/; s/, [0-9.]* TFLOP.*//; s/\[int8 fixed point\]//' $O/bfacc_waves.log
