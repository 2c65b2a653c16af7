cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prep
timeout -k 5 600 python -m pytest tests -m gpu -q -x > gpurun_out/prep/pytest.log 2>&1
rc=$?
tail -4 gpurun_out/prep/pytest.log
[ $rc -ne 0 ] && exit $rc
source tools/runs/r3_ab_fn.sh
run c1_prep $C1
run c2_prep $C2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prep/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prep/trace -- python3 $R/bench.py --steps 30 --warmup 5 --cpu-views 0 --batch-agents 0 --secondary 0 --agent-steps 0 > $R/gpurun_out/prep/trace.json 2> $R/gpurun_out/prep/trace.err
f=$(ls -S $R/gpurun_out/prep/trace/*/*kernel_stats.csv | head -1)
grep "k_patch_prep\|k_fold\|k_sad_mfma" $f | cut -c1-60,200-400 | head
