#!/bin/bash
# PMC passes over the coefficient-reuse beamformer (run on the GPU box from the repo root):
#   bash tools/pmc_bfacc.sh 64x256x1024x256 [out_dir]
# One rocprofv3 --pmc pass per counter group (separate passes, no tracing besides --kernel-trace), then the medians
# per kernel.  Units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; WRITE_SIZE / FETCH_SIZE KiB
# (FETCH_SIZE x 2 on gfx950: MI355X_MICROARCH.md).
set -e
SHAPE=${1:-64x256x1024x256}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${2:-$R/gpurun_out/pmc_bfacc_$SHAPE}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  PYTHONPATH=$R timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag -o out -- python3 $R/tools/measure.py bfacc --shape $SHAPE --modes 0 > $OUT/$tag.log 2>&1
done
cd $R
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
dur = []
for d in sorted(glob.glob(out + '/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if 'i8_kernel' in row['Kernel_Name']:
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
        for k, v in acc.items():
            v.sort()
            print(f"{k:28s} median {v[len(v)//2]:14.0f}   (n = {len(v)})")
    for f in glob.glob(d + '**/*kernel_trace.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if 'i8_kernel' in row['Kernel_Name']:
                dur.append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
dur.sort()
print(f"kernel duration under the profiler: median {dur[len(dur)//2]:.1f} us (n = {len(dur)})")
PY
