set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
export PYTHONPATH=$GRAFT_REPO_ROOT
for order in 1 0; do
  echo "== order $order (1 = round 2 numbering, 0 = XCD-contiguous)" >> gpurun_out/r3b/bfacc_order.log
  DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_ORDER=$order python tools/measure.py bfacc >> gpurun_out/r3b/bfacc_order.log 2>&1
  DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_ORDER=$order python tools/measure.py bfacc --shape 256x64x4096x256 >> gpurun_out/r3b/bfacc_order.log 2>&1
  DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_ORDER=$order python tools/measure.py bfacc --shape 64x256x4096x256 >> gpurun_out/r3b/bfacc_order.log 2>&1
  DCS_LIB_PATH=probes/libdcs_probes.so DCS_BACC_ORDER=$order python tools/measure.py bfacc --shape 64x16x32768x256 >> gpurun_out/r3b/bfacc_order.log 2>&1
done
bash tools/pmc_bfacc.sh 64x256x1024x256 gpurun_out/r3b/pmc_64x256 > gpurun_out/r3b/pmc_64x256.txt 2>&1
bash tools/pmc_bfacc.sh 256x64x1024x256 gpurun_out/r3b/pmc_256x64 > gpurun_out/r3b/pmc_256x64.txt 2>&1
cat gpurun_out/r3b/bfacc_order.log | grep -v "^/opt" | tail -40
