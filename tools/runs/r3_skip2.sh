cd $GRAFT_REPO_ROOT
export DEJAVU_VCODE=1
for so in libdejavu_stamps.so libdejavu_stamps_s1.so libdejavu_stamps_s2.so libdejavu_stamps_s3.so; do
  for shape in "500000 128 32"; do
    echo "=== VCODE $so $shape"
    STAMPS_SO=$so timeout -k 5 90 python tools/exp/stamps.py run $shape 2>/dev/null | grep "phase 1->2\|^exit\|^loop"
  done
done
