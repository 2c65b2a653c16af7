# Build first: STAMPS_SO=libdejavu_stamps_fin.so python3 tools/exp/stamps.py build -DDEJAVU_EXP_FIN
cd $GRAFT_REPO_ROOT
for shape in "50000 64 16" "500000 128 32" "100000 64 64"; do
  echo "=== $shape"
  STAMPS_FIN=1 STAMPS_SO=libdejavu_stamps_fin.so timeout -k 5 90 python tools/exp/stamps.py run $shape 2>/dev/null | grep "phase\|finishing\|^exit"
done
