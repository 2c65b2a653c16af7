cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests -q -x -m gpu -k "ensemble or experiment or off_the_landscape" > gpurun_out/r4_ensvec.log 2>&1; rc=$?; tail -12 gpurun_out/r4_ensvec.log
[ $rc -ne 0 ] && exit $rc
timeout -k 5 300 python tools/exp/ens_real.py
