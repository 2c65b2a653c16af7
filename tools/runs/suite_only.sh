cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout -k 10 1100 python -m pytest tests -m gpu -q -x "$@" > gpurun_out/suite/pytest.log 2>&1
rc=$?
tail -30 gpurun_out/suite/pytest.log
exit $rc
