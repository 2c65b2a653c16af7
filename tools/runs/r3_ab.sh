# A/B runs of bench.py's headline under env settings, one line per run (edit the `run` lines)
cd $GRAFT_REPO_ROOT
source tools/runs/r3_ab_fn.sh
DEJAVU_LC=0 run c1_lc0 $C1
DEJAVU_LC=1 run c1_lc1 $C1
DEJAVU_LC=0 run c2_lc0 $C2
DEJAVU_LC=1 run c2_lc1 $C2
DEJAVU_LC=1 DEJAVU_VCODE=1 run c2_lc1_code $C2
