for h in 2 3 4 6; do echo "SEG_HALVES=$h"; CELO_SEG_HALVES=$h python - <<'PY'
import os,sys,time
sys.path.insert(0,'.')
import numpy as np, torch
from celo_bls_snark_rs_amd import ffi, synthetic as syn
ffi.init(0)
for logn in (12,13,20):
    n=1<<logn
    b=syn.device_points("bls12_377_g1",n,5); sc=syn.uniform_scalars("bls12_377_g1",n,6); d=torch.from_numpy(sc.view(np.int64)).cuda()
    for sub in (False,True):
        ffi.msm_dev("bls12_377_g1",b.data_ptr(),0,d.data_ptr(),n,subgroup=sub)
        best=None
        for _ in range(5):
            t0=time.perf_counter(); ffi.msm_dev("bls12_377_g1",b.data_ptr(),0,d.data_ptr(),n,subgroup=sub); dt=(time.perf_counter()-t0)*1e3
            tm=ffi.msm_timings("bls12_377_g1")
            if best is None or tm["total_ms"]<best[0]["total_ms"]: best=(tm,dt)
        tm,dt=best
        print(logn,"sub" if sub else "plain","wall %.3f dev %.3f acc %.3f red %.3f c %d"%(dt,tm["total_ms"],tm["accumulate_ms"],tm["reduce_ms"],tm["window_bits"]))
PY
done
