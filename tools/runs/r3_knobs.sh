cd $GRAFT_REPO_ROOT
# The documented A/B knobs against the part of the suite that scores on the matrix cores.  Green under DEJAVU_VCODE=1, DEJAVU_LC=0/2,
# DEJAVU_HT=1; under DEJAVU_FUSE=0 / DEJAVU_MIXED=0 the only failures are the tests that assert the DEFAULT form of the step
# (scoring_form()["fused_finish"], library_info()["mixed_layout"]): every comparison with the oracle passes.
for cfg in "DEJAVU_VCODE=1" "DEJAVU_LC=0" "DEJAVU_LC=2" "DEJAVU_HT=1" "DEJAVU_FUSE=0" "DEJAVU_MIXED=0"; do
  echo "=== $cfg"
  env $cfg timeout -k 5 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fp4 or matrix_core or full_size or large_library or ragged or batched or ensemble or mixed or shipped" 2>&1 | grep -E "^FAILED|passed|failed" | head -12
done
