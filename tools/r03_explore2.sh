# round 3 exploration 2 (GPU box): fp16 geometry landscape; repeated interleaved A/B of the beamformer numbering
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
python tools/measure.py fp16 --modes 4 --form 3 --tpb 1 --wpc=-1,5,6,7 --cpb 8,12,16,20,24,32,48,64 > $O/fp16_sweep_mode4.log 2>&1
python tools/measure.py fp16 --modes 4 --form 3 --tpb 2 --wpc=-1,3,4 --cpb 8,12,16,24,32 > $O/fp16_sweep_mode4_tpb2.log 2>&1
python tools/measure.py fp16 --modes 0 --form 3 --tpb 1 --wpc=-1,6,7 --cpb 16,24,32,48,64,96 > $O/fp16_sweep_mode0.log 2>&1
for rep in 1 2 3; do
  for order in 1 0; do
    for shape in 64x256x4096x256 64x64x4096x256 64x1024x2048x256; do
      echo -n "rep $rep order $order: " >> $O/bfacc_order_ab.log
      env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_ORDER=$order python tools/measure.py bfacc --modes 0 --shape $shape 2>&1 | grep "int8" >> $O/bfacc_order_ab.log
    done
  done
done
tail -4 $O/fp16_sweep_mode4.log
