set -e
mkdir -p gpurun_out/r2p/pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  PYTHONPATH=$R timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/r2p/pmc/$tag -o out -- python3 $R/tools/measure.py bfacc --shape 64x256x1024x256 --modes 0 > $R/gpurun_out/r2p/pmc/$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r2p/pmc/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if 'i8_kernel' in row['Kernel_Name']:
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
        for k, v in acc.items():
            v.sort()
            print(d.split('/')[-2], k, 'median', v[len(v)//2], 'n', len(v))
PY
