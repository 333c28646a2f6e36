bash tools/ab_cmd_msm.sh
python - <<'PY'
import sys
sys.path.insert(0, '.')
from celo_bls_snark_rs_amd import ffi
import bench
ffi.init(0)
r = bench.pairing_leg(ffi, check_oracle=False)
r = bench.pairing_leg(ffi, check_oracle=False)
print("pairing", {k: round(r[k], 3) for k in ("miller_ms", "final_exp_ms")}, "%.3e" % r["value"])
PY
