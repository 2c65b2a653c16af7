cd $GRAFT_REPO_ROOT
# the 3-bit code tiles against the bit tiles: stamps of the first item, then the headline A/B
for cfg in "DEJAVU_VCODE=0" "DEJAVU_VCODE=1"; do
  for shape in "50000 64 16" "500000 128 32"; do
    echo "=== $cfg $shape"
    env $cfg timeout -k 5 90 python tools/exp/stamps.py run $shape 2>/dev/null | grep "phase 1->2\|^exit\|shader clock"
  done
done
source tools/runs/r3_ab_fn.sh
DEJAVU_VCODE=0 run c1_thermo $C1
DEJAVU_VCODE=1 run c1_code $C1
DEJAVU_VCODE=0 run c2_thermo $C2
DEJAVU_VCODE=1 run c2_code $C2
