#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the graph-replay bench and three PMC passes (MFMA busy / LDS,
# FETCH_SIZE, WRITE_SIZE: separate passes, MI355X_MICROARCH.md "rocprofv3 PMC slots"), folded per kernel on the box.
#   bash tools/profile_all.sh <model> <tag>        -> gpurun_out/<tag>_<model>_{kernel_stats.csv,pmc.json,bench.json}
set -o pipefail
MODEL=${1:-ganomaly}
TAG=${2:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$MODEL && mkdir -p /tmp/prof_$MODEL
python3 $ROOT/bench.py --model $MODEL --no-cpu-baseline > $OUT/${TAG}_${MODEL}_bench.json 2> $OUT/${TAG}_${MODEL}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$MODEL/kt -- python3 $ROOT/bench.py --model $MODEL --no-cpu-baseline --no-kernel-timer --no-secondary --steps 10 > /tmp/prof_$MODEL/kt.log 2>&1 || { tail -5 /tmp/prof_$MODEL/kt.log; exit 2; }
cp $(find /tmp/prof_$MODEL/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${MODEL}_kernel_stats.csv
# the same with the filter gradients on the launch stream (VFD_SIDE_WGRAD=0): per-kernel durations without a neighbour on
# the CUs — what bench.py's roofline (timed in an eager, single-stream pass) must agree with; the PMC passes use it too
VFD_SIDE_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$MODEL/kx -- python3 $ROOT/bench.py --model $MODEL --no-cpu-baseline --no-kernel-timer --no-secondary --steps 10 > /tmp/prof_$MODEL/kx.log 2>&1 || { tail -5 /tmp/prof_$MODEL/kx.log; exit 2; }
cp $(find /tmp/prof_$MODEL/kx -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${MODEL}_kernel_stats_single_stream.csv
export VFD_SIDE_WGRAD=0
i=0
for CNT in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $CNT --output-format csv -d /tmp/prof_$MODEL/pmc$i -- python3 $ROOT/bench.py --model $MODEL --no-cpu-baseline --no-kernel-timer --no-secondary --steps 2 --warmup 1 > /tmp/prof_$MODEL/pmc$i.log 2>&1 || { tail -5 /tmp/prof_$MODEL/pmc$i.log; exit 3; }
  echo "pmc pass $i done"
done
python3 $ROOT/tools/pmc_fold.py $(find /tmp/prof_$MODEL/pmc1 /tmp/prof_$MODEL/pmc2 /tmp/prof_$MODEL/pmc3 -name "*counter_collection.csv") > $OUT/${TAG}_${MODEL}_pmc.json || exit 4
echo "profiled $MODEL"
