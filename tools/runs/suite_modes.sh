cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
for mode in "DEJAVU_VCODE=1" "DEJAVU_FUSE=0" "DEJAVU_FP4=0" "DEJAVU_FOLD2=0" "DEJAVU_FP4_VARIANT=1" "DEJAVU_FP4_VARIANT=2"; do
  env $mode timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/suite/pytest_$mode.log 2>&1
  echo "$mode rc=$? $(tail -1 gpurun_out/suite/pytest_$mode.log)"
done
