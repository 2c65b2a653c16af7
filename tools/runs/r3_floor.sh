cd $GRAFT_REPO_ROOT
# ring-loop time per workgroup when the library is small enough to sit in the caches: the loop's own floor
for cfg in "DEJAVU_LC=1" "DEJAVU_LC=1 DEJAVU_VCODE=1" "DEJAVU_LC=0" "DEJAVU_LC=2"; do
  for F in 8192 2048; do
    echo "=== $cfg F=$F"
    env $cfg DEJAVU_MFMA_CHUNK=1 DEJAVU_SHAPE=6 timeout -k 5 60 python tools/exp/stamps.py run $F 64 16 2>/dev/null | grep "phase 1->2\|workgroups"
  done
done
