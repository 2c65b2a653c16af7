cd $GRAFT_REPO_ROOT
for so in libdejavu_stamps.so libdejavu_stamps_s2.so libdejavu_stamps_s3.so; do
  for shape in "50000 64 16" "500000 128 32"; do
    echo "=== $so $shape"
    STAMPS_SO=$so timeout -k 5 90 python tools/exp/stamps.py run $shape 2>/dev/null | grep "phase 1->2\|^exit\|shader clock"
  done
done
