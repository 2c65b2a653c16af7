cd $GRAFT_REPO_ROOT
# The round's new knobs against the agent / ensemble / matrix-core part of the suite (every comparison with the oracle and the golden
# trajectories must pass whichever way the steps are driven).
for cfg in "DEJAVU_AGENT_PIPELINE=0" "DEJAVU_LAZY_SCENE=0" "DEJAVU_LC22=0" "DEJAVU_CHAINS=1" "DEJAVU_CHAIN_ORDER=2" "DEJAVU_LC=0" "DEJAVU_HT=1" "DEJAVU_VCODE=1"; do
  echo "=== $cfg"
  env $cfg timeout -k 5 500 python -m pytest tests -m gpu -q -k "agent or trajector or ensemble or batched or experiment or pipelined or worked_out or deferred or shared_accumulators or two_group or large_library or config_five" 2>&1 | grep -E "^FAILED|passed|failed" | head -8
done
