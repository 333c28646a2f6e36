import sys, time, json
sys.path.insert(0, "/root/repo")
import numpy as np
from celo_bls_snark_rs_amd import ffi, synthetic as syn, codec
ffi.init(0)
out = {}
for ln in (16, 17, 18, 19, 20, 21):
    n = 1 << ln
    xy = syn.device_points("bls12_377_g1", n, 100).cpu().numpy().view(np.uint64).reshape(n, 12).copy()
    sc = syn.uniform_scalars("bls12_377_g1", n, 200)
    res = {}
    for name, kw in (("plain", {}), ("subgroup", {"subgroup": True})):
        r = ffi.msm("bls12_377_g1", xy, None, sc, **kw); ts = []
        for _ in range(8):
            t0 = time.perf_counter(); r = ffi.msm("bls12_377_g1", xy, None, sc, **kw); ts.append((time.perf_counter() - t0) * 1e3)
        res[name] = round(float(np.median(ts)), 3); res[name + "_pt"] = codec.jacobian_to_affine(r, codec.Q377, 1)
    out[ln] = {"plain_ms": res["plain"], "subgroup_ms": res["subgroup"], "equal": res["plain_pt"] == res["subgroup_pt"]}
print(json.dumps(out))
