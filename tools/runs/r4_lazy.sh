cd $GRAFT_REPO_ROOT
timeout -k 5 900 python -m pytest tests -q -x -m gpu -k "trajector or agent or pipelined or experiment or sensor or ensemble or ninety or plugin" > gpurun_out/r4_lazy_tests.log 2>&1; rc=$?; tail -8 gpurun_out/r4_lazy_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 5 300 python tools/exp/agent_track.py
DEJAVU_LAZY_SCENE=0 timeout -k 5 300 python tools/exp/agent_track.py
