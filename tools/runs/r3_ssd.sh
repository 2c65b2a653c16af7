cd $GRAFT_REPO_ROOT
echo "=== default build"; timeout -k 5 120 python tools/bench_ssd_f32.py
echo "=== -fno-slp-vectorize"; DV_LIB=tools/exp/libdejavu_noslp.so timeout -k 5 120 python - <<'PY'
import os, sys, runpy
sys.path.insert(0, "navigation-by-deja-vu_amd")
from navsim_amd import _native
_native.LIB_PATH = os.path.abspath(os.environ["DV_LIB"])
runpy.run_path("tools/bench_ssd_f32.py", run_name="__main__")
PY
