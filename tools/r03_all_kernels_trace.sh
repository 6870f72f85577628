# round 3: one rocprofv3 --kernel-trace --stats over every hot kernel of the library (generators fp32 / fp16 both forms, per-sample fused
# beamformer, coefficient-reuse beamformer staged and kChain)  ->  profiles/r03_all_kernels_stats.csv
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3trace; mkdir -p $O
cat > $O/run.py <<'PY'
import sys
sys.path.insert(0, sys.argv[1])
from tools import measure
import argparse
measure.cmd_pmc(argparse.Namespace())
for shape in ("64x16x32768x256", "64x256x4096x256", "256x64x4096x256"):
    measure.cmd_bfacc(argparse.Namespace(shape=shape, modes="0", random=True))
PY
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o run -- python3 $O/run.py $R > $O/run.log 2>&1
cd $R
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp $f $O/all_kernels_stats.csv; head -20 $O/all_kernels_stats.csv | cut -c1-200
