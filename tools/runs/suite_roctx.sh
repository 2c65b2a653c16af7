cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/suite/pytest.log 2>&1
rc=$?
tail -25 gpurun_out/suite/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
# roctx ranges: a marker trace of a short agent + exchange run
cd /tmp && export TMPDIR=/tmp
DEJAVU_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/roctx -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --force-dist --steps 30 --warmup 5 --views 50000 --sensor 64 --headings 16 --cpu-views 0 > $GRAFT_REPO_ROOT/gpurun_out/suite/roctx_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/suite/roctx.err
echo roctx rc=$?
ls $GRAFT_REPO_ROOT/gpurun_out/roctx/*/ | head; tail -3 $GRAFT_REPO_ROOT/gpurun_out/suite/roctx.err
