cd $GRAFT_REPO_ROOT
for v in 1 2 3 4; do echo shape $v; DEJAVU_SHAPE=$v timeout -k 10 300 python tools/bench_ssd_f32.py; done
