# round 3 exploration 7 (GPU box): data dependence of the beamformer's rate; sample-block rounds at 16 beams; new fp16 defaults
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do for rnd in "" "--random"; do for shape in 256x64x4096x256 256x64x1024x256 64x16x32768x256 64x256x4096x256; do
  echo -n "rep $rep samples ${rnd:-constant}: " >> $O/bfacc_data.log
  python tools/measure.py bfacc --modes 0 --shape $shape $rnd 2>&1 | grep "int8" >> $O/bfacc_data.log
done; done; done
for rounds in 1 2 3 4; do
  echo -n "rounds $rounds: " >> $O/bfacc_rounds16.log
  env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_ROUNDS=$rounds python tools/measure.py bfacc --modes 0 --shape 64x16x32768x256 --random 2>&1 | grep "int8" >> $O/bfacc_rounds16.log
done
python tools/measure.py fp16 --modes 0,4 --form 0 --cpb 0 --wpc=0 > $O/fp16_defaults.log 2>&1
cat $O/bfacc_data.log | sed 's/ -> .*T coefficient-products\/s,/ ->/; s/, [0-9.]* TFLOP.*//'; cat $O/bfacc_rounds16.log | sed 's/ -> .*T coefficient-products\/s,/ ->/; s/, [0-9.]* TFLOP.*//'; grep "^fp16\|^library" $O/fp16_defaults.log
