"""ORACLE (test infrastructure only): ctypes binding of oracle/build/liboracle.so
plus Python-side conversions between Montgomery limb buffers and big ints.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

from .py.ecc import Q377, Q761, R377

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "build", "liboracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, "cpu", f) for f in ("capi.cpp", "field.hpp", "curve.hpp", "pairing.hpp")]
    if not force and os.path.exists(_LIB) and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.orc_time_msm_bls12_377_g1.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---- big-int <-> limb buffers -------------------------------------------------
R384 = 1 << 384
R768 = 1 << 768


def ints_to_limbs(vals, nlimbs):
    """list of python ints -> np.uint64 array [len, nlimbs] (little-endian limbs)."""
    buf = b"".join(int(v).to_bytes(8 * nlimbs, "little") for v in vals)
    return np.frombuffer(buf, dtype=np.uint64).reshape(len(vals), nlimbs).copy()


def limbs_to_ints(arr, nlimbs):
    a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, nlimbs)
    raw = a.tobytes()
    return [int.from_bytes(raw[i * 8 * nlimbs:(i + 1) * 8 * nlimbs], "little") for i in range(a.shape[0])]


def _mont_shape(p):
    """(u64 limbs, Montgomery radix) of arkworks' representation of the field of order p"""
    if p == R377:
        return 4, 1 << 256
    return (6, R384) if p == Q377 else (12, R768)


def to_mont(vals, p):
    n, Rm = _mont_shape(p)
    return ints_to_limbs([(v * Rm) % p for v in vals], n)


def from_mont(arr, p):
    n, Rm = _mont_shape(p)
    Rinv = pow(Rm, -1, p)
    return [(v * Rinv) % p for v in limbs_to_ints(arr, n)]


# ---- point packing: affine python points -> (xy limbs, inf bytes) ------------
def pack_g1_377(points):
    flat, inf = [], []
    for P in points:
        if P is None:
            flat += [0, 0]; inf.append(1)
        else:
            flat += [P[0], P[1]]; inf.append(0)
    return to_mont(flat, Q377).reshape(len(points), 12), np.array(inf, dtype=np.uint8)


def pack_g2_377(points):
    flat, inf = [], []
    for P in points:
        if P is None:
            flat += [0, 0, 0, 0]; inf.append(1)
        else:
            flat += [P[0][0], P[0][1], P[1][0], P[1][1]]; inf.append(0)
    return to_mont(flat, Q377).reshape(len(points), 24), np.array(inf, dtype=np.uint8)


def pack_761(points):
    flat, inf = [], []
    for P in points:
        if P is None:
            flat += [0, 0]; inf.append(1)
        else:
            flat += [P[0], P[1]]; inf.append(0)
    return to_mont(flat, Q761).reshape(len(points), 24), np.array(inf, dtype=np.uint8)


def jac_to_affine(arr, kind):
    """Jacobian Montgomery limbs (one point) -> python affine point (or None).
    kind: 'g1_377' | 'g2_377' | '761'."""
    if kind == "g1_377":
        X, Y, Z = from_mont(np.asarray(arr).reshape(3, 6), Q377)
        p = Q377
        if Z == 0:
            return None
        zi = pow(Z, -1, p)
        return (X * zi * zi % p, Y * zi * zi * zi % p)
    if kind == "761":
        X, Y, Z = from_mont(np.asarray(arr).reshape(3, 12), Q761)
        p = Q761
        if Z == 0:
            return None
        zi = pow(Z, -1, p)
        return (X * zi * zi % p, Y * zi * zi * zi % p)
    if kind == "g2_377":
        from .py.ecc import F2_377 as f2
        v = from_mont(np.asarray(arr).reshape(6, 6), Q377)
        X, Y, Z = (v[0], v[1]), (v[2], v[3]), (v[4], v[5])
        if Z == (0, 0):
            return None
        zi = f2.inv(Z)
        zi2 = f2.sqr(zi)
        return (f2.mul(X, zi2), f2.mul(Y, f2.mul(zi2, zi)))
    raise ValueError(kind)


# ---- thin call wrappers ---------------------------------------------------------
def msm(kind, xy, inf, scalars, threads=1, naive=False):
    """kind in {'bls12_377_g1','bls12_377_g2','bw6_761_g1','bw6_761_g2'}; returns Jacobian limbs."""
    n = xy.shape[0]
    out = np.zeros(18 if kind == "bls12_377_g1" else 36, dtype=np.uint64)
    xy = np.ascontiguousarray(xy, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    fn = getattr(lib(), "orc_msm_" + kind)
    rc = fn(_p(xy), _p(inf), _p(scalars), C.c_size_t(n), C.c_int(threads), C.c_int(1 if naive else 0), _p(out))
    assert rc == 0
    return out


def pairing_product_377(g1xy, inf1, g2xy, inf2):
    k = g1xy.shape[0]
    gt = np.zeros(72, dtype=np.uint64)
    one = C.c_int(0)
    rc = lib().orc_pairing_product_bls12_377(_p(g1xy), _p(inf1), _p(g2xy), _p(inf2), C.c_size_t(k), _p(gt), C.byref(one))
    assert rc == 0
    return gt, bool(one.value)


def miller_loop_377(g1xy, inf1, g2xy, inf2):
    k = g1xy.shape[0]
    gt = np.zeros(72, dtype=np.uint64)
    assert lib().orc_miller_loop_bls12_377(_p(g1xy), _p(inf1), _p(g2xy), _p(inf2), C.c_size_t(k), _p(gt)) == 0
    return gt


def final_exp_377(gt_in):
    gt_in = np.ascontiguousarray(gt_in, dtype=np.uint64)
    out = np.zeros(72, dtype=np.uint64)
    assert lib().orc_final_exp_bls12_377(_p(gt_in), _p(out)) == 0
    return out


def pairing_product_761(g1xy, inf1, g2xy, inf2):
    k = g1xy.shape[0]
    gt = np.zeros(72, dtype=np.uint64)
    one = C.c_int(0)
    rc = lib().orc_pairing_product_bw6_761(_p(g1xy), _p(inf1), _p(g2xy), _p(inf2), C.c_size_t(k), _p(gt), C.byref(one))
    assert rc == 0
    return gt, bool(one.value)


def gt377_to_flat(gt):
    """72 Montgomery limbs (tower order) -> flat Fq[w]/(w^12+5) coefficient list
    (oracle.py.pairing.tower_to_flat_377 convention)."""
    v = from_mont(np.asarray(gt).reshape(12, 6), Q377)
    # order: c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2, each (a0, a1)
    out = [0] * 12
    idx = 0
    for i in range(2):
        for j in range(3):
            for k in range(2):
                out[6 * k + 2 * j + i] = v[idx]
                idx += 1
    return out


def gt761_to_flat(gt):
    """tower (c0.c0,c0.c1,c0.c2,c1.c0,c1.c1,c1.c2), u = w^2, v = w -> flat Fq[w]/(w^6+4)."""
    v = from_mont(np.asarray(gt).reshape(6, 12), Q761)
    out = [0] * 6
    idx = 0
    for b in range(2):
        for a in range(3):
            out[2 * a + b] = v[idx]
            idx += 1
    return out


def ntt_fq377(data, log_n, omega, coset=None, coset_after=False, scale=None):
    """Radix-2 NTT over Fr(BW6-761) = Fq(BLS12-377) (orc_ntt_fq377).  data: (n, 6) uint64 arkworks Montgomery limbs
    (copied); omega / coset / scale: ints (canonical values).  Returns a new (n, 6) array."""
    out = np.ascontiguousarray(data, dtype=np.uint64).copy()
    w = to_mont([omega], Q377)
    g = None if coset is None else to_mont([coset], Q377)
    s = None if scale is None else to_mont([scale], Q377)
    assert lib().orc_ntt_fq377(_p(out), C.c_uint(log_n), _p(w), _p(g), C.c_int(1 if coset_after else 0), _p(s)) == 0
    return out


def ntt_fr253(data, log_n, omega, coset=None, coset_after=False, scale=None):
    """The same transform over Fr(BLS12-377) (orc_ntt_fr253).  data: (n, 4) uint64 arkworks Montgomery limbs (copied)."""
    out = np.ascontiguousarray(data, dtype=np.uint64).copy()
    w = to_mont([omega], R377)
    g = None if coset is None else to_mont([coset], R377)
    s = None if scale is None else to_mont([scale], R377)
    assert lib().orc_ntt_fr253(_p(out), C.c_uint(log_n), _p(w), _p(g), C.c_int(1 if coset_after else 0), _p(s)) == 0
    return out


def time_ntt_fq377(data, log_n, omega):
    out = np.ascontiguousarray(data, dtype=np.uint64).copy()
    lib().orc_time_ntt_fq377.restype = C.c_double
    return lib().orc_time_ntt_fq377(_p(out), C.c_uint(log_n), _p(to_mont([omega], Q377)))


def decompress(group, data, check_subgroup=True, threads=1):
    """arkworks GroupAffine::deserialize of n concatenated compressed BLS12-377 points (orc_decompress_bls12_377).
    group "g1" (48 B) / "g2" (96 B).  Returns (xy (n, 12|24) uint64 Montgomery limbs, status (n,) uint8: 0 ok, 1 infinity,
    2 invalid, 3 not in the subgroup)."""
    size, words, g2 = {"g1": (48, 12, 0), "g2": (96, 24, 1)}[group]
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    n = buf.size // size
    xy = np.zeros((n, words), dtype=np.uint64)
    st = np.zeros(n, dtype=np.uint8)
    if n:
        assert lib().orc_decompress_bls12_377(C.c_int(g2), _p(buf), C.c_size_t(n), C.c_int(1 if check_subgroup else 0), C.c_int(threads), _p(xy), _p(st)) == 0
    return xy, st


def time_decompress(group, data, check_subgroup=True, threads=1):
    size, words, g2 = {"g1": (48, 12, 0), "g2": (96, 24, 1)}[group]
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    n = buf.size // size
    xy = np.zeros((n, words), dtype=np.uint64)
    st = np.zeros(n, dtype=np.uint8)
    lib().orc_time_decompress_bls12_377.restype = C.c_double
    return lib().orc_time_decompress_bls12_377(C.c_int(g2), _p(buf), C.c_size_t(n), C.c_int(1 if check_subgroup else 0), C.c_int(threads), _p(xy), _p(st))


def normalize(kind, jac):
    """ProjectiveCurve::batch_normalization_into_affine restated (orc_normalize_*): jac (n, 18 | 36) uint64 Jacobian Montgomery
    limbs -> (xy (n, 12 | 24), inf (n,) uint8).  kind: 'g1_377' | 'g2_377'."""
    words, fn = {"g1_377": (6, "orc_normalize_bls12_377_g1"), "g2_377": (12, "orc_normalize_bls12_377_g2")}[kind]
    j = np.ascontiguousarray(jac, dtype=np.uint64).reshape(-1, 3 * words)
    n = j.shape[0]
    xy = np.zeros((n, 2 * words), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    if n:
        assert getattr(lib(), fn)(_p(j), C.c_size_t(n), _p(xy), _p(inf)) == 0
    xy[inf != 0] = 0
    return xy, inf
