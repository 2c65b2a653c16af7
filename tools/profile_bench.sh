#!/bin/bash
# Collects the rocprofv3 evidence for a workload on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_bench.sh r03_c2                       # bench.py's headline (BENCH_ARGS: extra bench.py arguments)
#   PROFILE_CMD="tools/bench_ssd_f32.py" bash tools/profile_bench.sh r03_ssd     # any other python program of the repo
# Pass 1: --kernel-trace --stats (per-kernel durations).  Further passes: PMC counters, each set in its own run with
# --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md, rocprofv3 PMC slots).  Results land in
# gpurun_out/<tag>/ and a summary JSON is printed by tools/summarize_profile.py; copy both into profiles/ with
# tools/copy_profile.py (delete the local gpurun_out/<tag>/ first: gpurun merges into it, and files of an earlier run would go
# stale there).
set -e
TAG=${1:-r03}
BENCH_ARGS=${BENCH_ARGS:-}
PROFILE_STEPS=${PROFILE_STEPS:-200}
PROFILE_CMD=${PROFILE_CMD:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -n "$PROFILE_CMD" ]; then
  LONG="$ROOT/$PROFILE_CMD"
  SHORT="$ROOT/$PROFILE_CMD"
else
  LONG="$ROOT/bench.py --steps $PROFILE_STEPS --warmup 20 --cpu-views 0 --batch-agents 0 --secondary 0 --agent-steps 0 $BENCH_ARGS"
  SHORT="$ROOT/bench.py --steps 20 --warmup 5 --cpu-views 0 --batch-agents 0 --secondary 0 --agent-steps 0 $BENCH_ARGS"
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $LONG > $OUT/bench_under_trace.json 2> $OUT/trace.err
if [ -z "$PROFILE_CMD" ]; then
  # the counter passes perturb the timing that picks the workgroup shape: pin the shape the trace pass used
  export DEJAVU_SHAPE=$(python3 -c "import json,sys; print(json.loads(open('$OUT/bench_under_trace.json').read().strip().splitlines()[-1])['config'].get('workgroup_shape', 0))")
fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $SHORT > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $SHORT > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $SHORT > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_inst -- python3 $SHORT > /dev/null 2> $OUT/pmc_inst.err
cd $ROOT && python3 tools/summarize_profile.py $OUT > $OUT/summary.json && python3 - $OUT/summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in d["kernels"][:8]:
    print("%-70s calls %5d avg %9.1f us min %9.1f" % (k["name"][:70], k["calls"], k["avg_us"], k["min_us"]))
for kern, c in d.items():
    if isinstance(c, dict) and "hbm_traffic_bytes_per_launch" in c:
        print(kern, c.get("kernel_name", "")[:60], "HBM bytes/launch %.4g" % c["hbm_traffic_bytes_per_launch"])
PY
