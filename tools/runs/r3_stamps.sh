# in-kernel phase stamps of the scoring kernel (python tools/exp/stamps.py build first, here; the .so travels with the snapshot)
cd $GRAFT_REPO_ROOT
for cfg in "DEJAVU_LC=0" "DEJAVU_LC=1" "DEJAVU_LC=1 DEJAVU_VCODE=1"; do
  echo "=== $cfg"
  env $cfg python tools/exp/stamps.py run 50000 64 16 2>/dev/null | grep "phase\|workgroups\|exit"
done
