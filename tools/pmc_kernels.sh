#!/bin/bash
# PMC passes over the generator kernels (run on the GPU box from the repo root):
#   bash tools/pmc_kernels.sh [out_dir] [kernel-name regex]
# One rocprofv3 --pmc pass per counter group over `python3 tools/measure.py pmc` (six launches each of the fp32 generator, the
# fp16 generator in both arithmetic forms and the per-sample fused beamformer at config 3), separate passes, then medians per
# kernel.  Units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is the
# sum over the 8 XCDs.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${1:-$R/gpurun_out/pmc_kernels}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
RE=${2:-bf_tiled_kernel}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
            "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_EXP_GDS SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  PYTHONPATH=$R timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag -o out -- python3 $R/tools/measure.py pmc > $OUT/$tag.log 2>&1
done
cd $R
python3 - $OUT "$RE" <<'PY'
import csv, glob, collections, re, sys
out, rx = sys.argv[1], re.compile(sys.argv[2])
def short(n):
    m = re.search(r"bf_tiled_kernelIL(.*?)E+v", n)
    return ("tiled<" + m.group(1).replace("ELb", ",").replace("ELi", ",") + ">") if m else n[:60]
durs = collections.defaultdict(list)
for d in sorted(glob.glob(out + '/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if rx.search(row['Kernel_Name']):
                acc[(short(row['Kernel_Name']), row['Counter_Name'])].append(float(row['Counter_Value']))
        for (k, c), v in sorted(acc.items()):
            v.sort()
            print(f"{k:44s} {c:26s} median {v[len(v)//2]:16.0f}   (n = {len(v)})")
    for f in glob.glob(d + '**/*kernel_trace.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if rx.search(row['Kernel_Name']):
                durs[short(row['Kernel_Name'])].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
for k, v in durs.items():
    v.sort()
    print(f"{k:44s} duration under the profiler: median {v[len(v)//2]:.1f} us (n = {len(v)})")
PY
