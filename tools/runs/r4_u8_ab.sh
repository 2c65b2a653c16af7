# A/B of two builds of the library (tools/exp/ab/{old,new}.so, not tracked) on the ssd_u8 timing, interleaved.
cd $GRAFT_REPO_ROOT
L=navigation-by-deja-vu_amd/csrc/libdejavu_hip.so
for rep in 1 2 3; do
  for v in old new; do
    cp tools/exp/ab/$v.so $L
    for args in "50000 64 16" "200000 128 32"; do
      echo "$v $args: $(timeout -k 5 120 python tools/bench_ssd_u8.py $args | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["ms_per_step"])')"
    done
  done
done
cp tools/exp/ab/new.so $L
