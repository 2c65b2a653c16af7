# Build first, in the repo (they travel with the snapshot):
#   for n in 1 2 3; do STAMPS_SO=libdejavu_stamps_s$n.so python3 tools/exp/stamps.py build -DDEJAVU_EXP_SKIP=$n; done
#   STAMPS_SO=libdejavu_stamps_s3m.so python3 tools/exp/stamps.py build -DDEJAVU_EXP_SKIP=3 -DDEJAVU_EXP_NOMASK      (…_s3l: -DDEJAVU_EXP_NOLDS)
# and run with DEJAVU_VCODE=1 in front for the code tiles.
cd $GRAFT_REPO_ROOT
# timing-only experiments (results wrong): the scoring kernel without its coefficient rows (s1), without its library rows (s2), without
# both (s3); s3 without the consumers' mask arithmetic (m) and without their LDS reads (l)
for so in ${SOS:-libdejavu_stamps.so libdejavu_stamps_s1.so libdejavu_stamps_s2.so libdejavu_stamps_s3.so}; do
  for shape in "50000 64 16" "500000 128 32"; do
    echo "=== $so $shape"
    STAMPS_SO=$so timeout -k 5 90 python tools/exp/stamps.py run $shape 2>/dev/null | grep "phase 1->2\|^exit\|^loop"
  done
done
