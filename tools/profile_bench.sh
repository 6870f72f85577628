#!/bin/bash
# The bench.py evidence set of profiles/rNN_bench_profile.md (run on the GPU box from the repo root):
#   bash tools/profile_bench.sh TAG        -> gpurun_out/TAG.json, TAG_kt*, TAG_pmc_w*, TAG_pmc_f*
# then, back in the container:  python tools/make_profile_summary.py TAG TAG.json rNN
# One un-profiled bench run, one rocprofv3 --kernel-trace --stats run and two --pmc passes (WRITE_SIZE, FETCH_SIZE:
# separate passes, nothing traced beside them).  The profiled runs skip the side measurements (--no-extras) so that the
# last 60 dispatches of the production kernel are the 10 warm-up + 50 timed steps.
set -e
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
python bench.py > $R/gpurun_out/$TAG.json 2> $R/gpurun_out/$TAG.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt -o runc -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_kt.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_w -o runc -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_pmc_w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_f -o runc -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_pmc_f.log 2>&1
cd $R
tail -c 400 gpurun_out/$TAG.json
