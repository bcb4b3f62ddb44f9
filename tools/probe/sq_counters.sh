#!/bin/bash
# SQ wait / issue counters of the conv kernels on single layers (rocprofv3 --pmc on tools/layer_bench.py): what share of a wave's
# cycles issues an instruction, waits to issue, or is parked at s_waitcnt / s_barrier.   bash tools/probe/sq_counters.sh > out.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for L in "enc.pyr 256" "enc.pyr 64" "dec.pyr 128" "dec.pyr 256"; do
  rm -rf /tmp/pmq
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS \
    --output-format csv -d /tmp/pmq -- python3 $ROOT/tools/layer_bench.py --only "$L" --iters 3 > /tmp/pmq.log 2>&1 || { tail -3 /tmp/pmq.log; exit 1; }
  python3 - "$L" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pmq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith(("conv_igemm", "conv_wgrad_kernel", "conv_halo", "conv_wgrad_halo")):
        acc[n.split("(")[0][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== layer %s" % sys.argv[1])
for n, c in acc.items():
    w = sum(c["SQ_WAVE_CYCLES"]) / len(c["SQ_WAVE_CYCLES"])
    f = lambda k: sum(c[k]) / len(c[k]) / w
    print("  %-72s issue %.2f  wait-to-issue %.2f  parked (waitcnt/barrier) %.2f  lds-issue-stall %.2f   launches %d"
          % (n, f("SQ_ACTIVE_INST_ANY"), f("SQ_WAIT_INST_ANY"), f("SQ_WAIT_ANY"), f("SQ_WAIT_INST_LDS"), len(c["SQ_WAVE_CYCLES"])))
PY
done
