# round 3 exploration 1 (GPU box): beamformer workgroup numbering A/B + PMC, K-split rounds, fp16 geometry landscape
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
P="DCS_LIB_PATH=probes/libdcs_probes.so"
for order in 1 2 0; do
  echo "== order $order (1 = as dispatched (round 2), 2 = contiguous eighth per XCD, 0 = sharers grouped per XCD (product))" >> $O/bfacc_order.log
  for shape in 64x16x32768x256 64x64x4096x256 64x256x1024x256 64x256x4096x256 256x64x1024x256 256x64x4096x256 64x1024x256x256 256x256x256x256; do
    env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_ORDER=$order python tools/measure.py bfacc --modes 0 --shape $shape 2>&1 | grep "int8" >> $O/bfacc_order.log
  done
done
for rounds in 2 4 8 16; do
  echo "== K-split rounds $rounds" >> $O/bfacc_rounds.log
  for shape in 256x64x1024x256 256x64x4096x256 256x16x4096x256; do
    env DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_ROUNDS=$rounds python tools/measure.py bfacc --modes 0 --shape $shape 2>&1 | grep "int8" >> $O/bfacc_rounds.log
  done
done
python tools/measure.py fp16 --modes 4 --form 3 --tpb 1 --wpc -1,5,6,7 --cpb 8,12,16,20,24,32,48,64 > $O/fp16_sweep_mode4.log 2>&1
python tools/measure.py fp16 --modes 0 --form 3 --tpb 1 --wpc -1,6,7 --cpb 16,24,32,48,64,96 > $O/fp16_sweep_mode0.log 2>&1
bash tools/pmc_bfacc.sh 64x256x1024x256 $O/pmc_64x256 > $O/pmc_64x256.txt 2>&1
bash tools/pmc_bfacc.sh 256x64x1024x256 $O/pmc_256x64 > $O/pmc_256x64.txt 2>&1
tail -3 $O/pmc_256x64.txt
