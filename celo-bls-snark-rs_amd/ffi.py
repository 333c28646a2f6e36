"""ctypes binding of the gfx950 C-ABI library (include/celo_bls_amd.h).

There is NO CPU fallback: if build/libcelo_bls_amd.so is missing, import of `lib()` raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "build", "libcelo_bls_amd.so")

GROUP_ID = {"bls12_377_g1": 0, "bls12_377_g2": 1, "bw6_761_g1": 2, "bw6_761_g2": 2}
# (u64 limbs per affine point, u64 limbs per scalar, u64 limbs of the Jacobian result)
GROUP_SHAPE = {"bls12_377_g1": (12, 4, 18), "bls12_377_g2": (24, 4, 36), "bw6_761_g1": (24, 6, 36), "bw6_761_g2": (24, 6, 36)}

EXPORTS = [
    "celo_amd_init", "celo_amd_use_device", "celo_amd_device_count", "celo_amd_device_name",
    "msm_bls12_377_g1_multi", "msm_bls12_377_g2_multi", "msm_bw6_761_g1_multi", "msm_bw6_761_g2_multi",
    "msm_bls12_377_g1_multi_dev", "msm_bls12_377_g2_multi_dev", "msm_bw6_761_g1_multi_dev", "msm_bw6_761_g2_multi_dev",
    "msm_bls12_377_g1_multi_windows", "msm_bls12_377_g2_multi_windows", "msm_bw6_761_g1_multi_windows", "msm_bw6_761_g2_multi_windows",
    "msm_bls12_377_g1_subgroup_multi_windows", "msm_bls12_377_g2_subgroup_multi_windows",
    "msm_bls12_377_g1_multi_windows_dev", "msm_bls12_377_g2_multi_windows_dev", "msm_bw6_761_g1_multi_windows_dev", "msm_bw6_761_g2_multi_windows_dev",
    "msm_bls12_377_g1_subgroup_multi_windows_dev", "msm_bls12_377_g2_subgroup_multi_windows_dev",
    "msm_bls12_377_g1_window_shard_dev", "msm_bls12_377_g2_window_shard_dev", "msm_bw6_761_window_shard_dev",
    "msm_bls12_377_g1_join_windows", "msm_bls12_377_g2_join_windows", "msm_bw6_761_join_windows",
    "msm_bls12_377_g1_precompute", "msm_bls12_377_g2_precompute", "msm_bw6_761_g1_precompute", "msm_bw6_761_g2_precompute",
    "msm_bls12_377_g1_precompute_dev", "msm_bls12_377_g2_precompute_dev", "msm_bw6_761_g1_precompute_dev", "msm_bw6_761_g2_precompute_dev",
    "msm_bls12_377_g1_fixed", "msm_bls12_377_g2_fixed", "msm_bw6_761_g1_fixed", "msm_bw6_761_g2_fixed",
    "msm_bls12_377_g1_fixed_dev", "msm_bls12_377_g2_fixed_dev", "msm_bw6_761_g1_fixed_dev", "msm_bw6_761_g2_fixed_dev",
    "celo_amd_msm_fixed_release", "celo_amd_msm_fixed_info",
    "groth16_load_key_bw6_761", "groth16_load_key_bls12_377", "groth16_prove_with_key", "groth16_free_key",
    "msm_bls12_377_g1", "msm_bls12_377_g2", "msm_bw6_761_g1", "msm_bw6_761_g2",
    "msm_batch_bls12_377_g1", "msm_batch_bls12_377_g2", "msm_batch_bw6_761_g1", "msm_batch_bw6_761_g2",
    "msm_bls12_377_g1_dev", "msm_bls12_377_g2_dev", "msm_bw6_761_g1_dev", "msm_bw6_761_g2_dev",
    "pairing_product_is_one_bls12_377", "pairing_product_is_one_batch_bls12_377", "celo_amd_pairing_gt_bls12_377",
    "celo_amd_pairing_last_timings", "pairing_product_is_one_bw6_761", "celo_amd_pairing_gt_bw6_761",
    "celo_amd_sum_jacobian_bls12_377_g1", "celo_amd_sum_jacobian_bls12_377_g2", "celo_amd_sum_jacobian_bw6_761",
    "celo_amd_msm_last_timings", "celo_amd_msm_set_window_bits", "celo_amd_msm_set_host_chunks", "celo_amd_msm_set_batched_affine", "celo_amd_msm_host_chunk_plan", "celo_amd_host_alloc", "celo_amd_host_free", "celo_amd_ubench_fp", "celo_amd_selftest_accumulate",
    "celo_amd_gen_points_bls12_377_g1_dev", "celo_amd_gen_points_bls12_377_g2_dev", "celo_amd_gen_points_bw6_761_dev",
    "celo_amd_gen_points_grouped_bls12_377_g1_dev", "celo_amd_gen_points_grouped_bls12_377_g2_dev",
    "batch_verify_bls12_377", "batch_verify_bls12_377_dev", "celo_amd_draw_batch_exponents",
    "ntt_bw6_761_fr", "ntt_bw6_761_fr_dev",
    "groth16_witness_map_bw6_761", "groth16_witness_map_bw6_761_dev", "groth16_prove_bw6_761",
    "decompress_bls12_377_g1", "decompress_bls12_377_g2", "decompress_bls12_377_g1_dev", "decompress_bls12_377_g2_dev",
    "normalize_bls12_377_g1", "normalize_bls12_377_g2",
    "hash_to_g1_direct_bls12_377", "hash_to_g1_composite_bls12_377", "hash_to_g1_cip22_tail_bls12_377", "composite_crh_bls12_377",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with __graft_entry__.build() "
                "(make -C celo-bls-snark-rs_amd/csrc). There is no CPU fallback for the MSM/pairing path.")
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def init(device=0):
    rc = lib().celo_amd_init(C.c_int(device))
    if rc != 0:
        raise RuntimeError(f"celo_amd_init({device}) failed rc={rc} (no gfx950 device?)")


def use_device(device):
    """Binds the CALLING thread to a device (celo_amd_use_device)."""
    rc = lib().celo_amd_use_device(C.c_int(device))
    if rc != 0:
        raise RuntimeError(f"celo_amd_use_device({device}) failed rc={rc}")


def device_count():
    n = C.c_int(0)
    rc = lib().celo_amd_device_count(C.byref(n))
    if rc != 0:
        raise RuntimeError(f"celo_amd_device_count failed rc={rc} (no gfx950 device?)")
    return n.value


def msm_multi(group, devices, bases_xy, inf, scalars):
    """One MSM sharded by index range over `devices` (list of ordinals, repeats allowed) inside this process; host buffers."""
    A, S, O = GROUP_SHAPE[group]
    bases_xy = np.ascontiguousarray(bases_xy, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = bases_xy.size // A
    assert bases_xy.size == n * A and scalars.size == n * S, "bases / scalars length mismatch"
    devs = (C.c_int * len(devices))(*devices)
    out = np.zeros(O, dtype=np.uint64)
    rc = getattr(lib(), "msm_" + group + "_multi")(devs, C.c_int(len(devices)), _p(bases_xy), _p(inf), _p(scalars), C.c_size_t(n), _p(out))
    if rc != 0:
        raise RuntimeError(f"msm_{group}_multi failed rc={rc}")
    return out


def msm_multi_dev(group, devices, d_bases, d_infs, d_scalars, n_per):
    """One MSM over shards already resident on their devices: d_bases / d_scalars / d_infs (or None) are lists of integer
    device addresses, n_per the terms per shard."""
    O = GROUP_SHAPE[group][2]
    k = len(devices)
    devs = (C.c_int * k)(*devices)
    pb = (C.c_void_p * k)(*d_bases)
    ps = (C.c_void_p * k)(*d_scalars)
    pi = (C.c_void_p * k)(*[x or 0 for x in d_infs]) if d_infs is not None else None
    np_ = (C.c_size_t * k)(*n_per)
    out = np.zeros(O, dtype=np.uint64)
    rc = getattr(lib(), "msm_" + group + "_multi_dev")(devs, C.c_int(k), pb, pi, ps, np_, _p(out))
    if rc != 0:
        raise RuntimeError(f"msm_{group}_multi_dev failed rc={rc}")
    return out


def msm_multi_windows(group, devices, bases_xy, inf, scalars, subgroup=False):
    """One MSM partitioned by WINDOW over `devices` (repeats allowed): every shard stages all n terms and owns a range of the windows."""
    A, S, O = GROUP_SHAPE[group]
    bases_xy = np.ascontiguousarray(bases_xy, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = bases_xy.size // A
    assert bases_xy.size == n * A and scalars.size == n * S, "bases / scalars length mismatch"
    devs = (C.c_int * len(devices))(*devices)
    out = np.zeros(O, dtype=np.uint64)
    name = "msm_" + group + ("_subgroup" if subgroup else "") + "_multi_windows"
    rc = getattr(lib(), name)(devs, C.c_int(len(devices)), _p(bases_xy), _p(inf), _p(scalars), C.c_size_t(n), _p(out))
    if rc != 0:
        raise RuntimeError(f"{name} failed rc={rc}")
    return out


def msm_multi_windows_dev(group, devices, d_bases, d_infs, d_scalars, n, subgroup=False):
    """The window partition over replicas already resident on their devices (lists of integer device addresses, n terms each)."""
    O = GROUP_SHAPE[group][2]
    k = len(devices)
    devs = (C.c_int * k)(*devices)
    pb = (C.c_void_p * k)(*d_bases)
    ps = (C.c_void_p * k)(*d_scalars)
    pi = (C.c_void_p * k)(*[x or 0 for x in d_infs]) if d_infs is not None else None
    out = np.zeros(O, dtype=np.uint64)
    name = "msm_" + group + ("_subgroup" if subgroup else "") + "_multi_windows_dev"
    rc = getattr(lib(), name)(devs, C.c_int(k), pb, pi, ps, C.c_size_t(n), _p(out))
    if rc != 0:
        raise RuntimeError(f"{name} failed rc={rc}")
    return out


_SHARD_NAME = {"bls12_377_g1": "bls12_377_g1", "bls12_377_g2": "bls12_377_g2", "bw6_761_g1": "bw6_761", "bw6_761_g2": "bw6_761"}


def msm_window_shard_dev(group, d_bases, d_inf, d_scalars, n, shard, nshards, stream=0, subgroup=False):
    """One rank's share of a window-partitioned MSM (one process per GPU): returns (record, bit_lo); record = X || Y || ZZ || ZZZ in
    arkworks limbs (uint64[4 * A / 2]).  The ranks all-gather the records and join them with join_windows()."""
    A = GROUP_SHAPE[group][0]
    rec = np.zeros(2 * A, dtype=np.uint64)
    bit = C.c_int(0)
    fn = getattr(lib(), "msm_" + _SHARD_NAME[group] + "_window_shard_dev")
    if group.startswith("bw6"):
        rc = fn(C.c_void_p(d_bases), C.c_void_p(d_inf or 0), C.c_void_p(d_scalars), C.c_size_t(n), C.c_int(shard), C.c_int(nshards), _p(rec), C.byref(bit),
                C.c_void_p(stream or 0))
    else:
        rc = fn(C.c_void_p(d_bases), C.c_void_p(d_inf or 0), C.c_void_p(d_scalars), C.c_size_t(n), C.c_int(1 if subgroup else 0), C.c_int(shard), C.c_int(nshards),
                _p(rec), C.byref(bit), C.c_void_p(stream or 0))
    if rc != 0:
        raise RuntimeError(f"msm_{group}_window_shard_dev failed rc={rc}")
    return rec, bit.value


def join_windows(group, records, bit_lo):
    """total = sum_g 2^bit_lo[g] P_g over the shards' records (shard order)."""
    A, _, O = GROUP_SHAPE[group]
    records = np.ascontiguousarray(records, dtype=np.uint64).reshape(-1)
    k = len(bit_lo)
    assert records.size == k * 2 * A
    bits = (C.c_int * k)(*[int(b) for b in bit_lo])
    out = np.zeros(O, dtype=np.uint64)
    rc = getattr(lib(), "msm_" + _SHARD_NAME[group] + "_join_windows")(_p(records), bits, C.c_int(k), _p(out))
    if rc != 0:
        raise RuntimeError(f"msm_{group}_join_windows failed rc={rc}")
    return out


class FixedBase:
    """A key's fixed-base tables (msm_*_precompute): build once, then msm() / msm_dev() for every scalar vector.  window_bits 0 = automatic."""
    def __init__(self, group, bases_xy=None, inf=None, window_bits=0, d_bases=None, d_inf=0, n=None):
        A = GROUP_SHAPE[group][0]
        self.group, self.h = group, C.c_void_p()
        if d_bases is not None:
            rc = getattr(lib(), "msm_" + group + "_precompute_dev")(C.c_void_p(d_bases), C.c_void_p(d_inf or 0), C.c_size_t(n), C.c_int(window_bits), C.byref(self.h))
        else:
            bases_xy = np.ascontiguousarray(bases_xy, dtype=np.uint64)
            n = bases_xy.size // A
            rc = getattr(lib(), "msm_" + group + "_precompute")(_p(bases_xy), _p(inf), C.c_size_t(n), C.c_int(window_bits), C.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"msm_{group}_precompute failed rc={rc}")
        self.n = n

    def info(self):
        n, wb, w, by, ms = C.c_size_t(), C.c_int(), C.c_int(), C.c_size_t(), C.c_float()
        rc = lib().celo_amd_msm_fixed_info(self.h, C.byref(n), C.byref(wb), C.byref(w), C.byref(by), C.byref(ms))
        if rc != 0:
            raise RuntimeError(f"celo_amd_msm_fixed_info failed rc={rc}")
        return {"n": n.value, "window_bits": wb.value, "windows": w.value, "table_bytes": by.value, "build_ms": ms.value}

    def msm(self, scalars):
        S, O = GROUP_SHAPE[self.group][1:]
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        out = np.zeros(O, dtype=np.uint64)
        rc = getattr(lib(), "msm_" + self.group + "_fixed")(self.h, _p(scalars), C.c_size_t(scalars.size // S), _p(out))
        if rc != 0:
            raise RuntimeError(f"msm_{self.group}_fixed failed rc={rc}")
        return out

    def msm_dev(self, d_scalars, n_scalars, stream=0):
        out = np.zeros(GROUP_SHAPE[self.group][2], dtype=np.uint64)
        rc = getattr(lib(), "msm_" + self.group + "_fixed_dev")(self.h, C.c_void_p(d_scalars), C.c_size_t(n_scalars), _p(out), C.c_void_p(stream or 0))
        if rc != 0:
            raise RuntimeError(f"msm_{self.group}_fixed_dev failed rc={rc}")
        return out

    def release(self):
        if self.h:
            lib().celo_amd_msm_fixed_release(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def msm(group, bases_xy, inf, scalars, subgroup=False):
    """Host-buffer MSM. bases_xy: uint64 [n, A]; inf: uint8 [n] or None; scalars: uint64 [n, S]. Returns Jacobian limbs.
    subgroup=True (bls12_377_g1 / _g2): the bases are vouched to lie in the prime-order group (msm_bls12_377_g1_subgroup / _g2_subgroup: GLV split)."""
    A, S, O = GROUP_SHAPE[group]
    n = int(bases_xy.shape[0]) if bases_xy.ndim == 2 else int(bases_xy.size // A)
    bases_xy = np.ascontiguousarray(bases_xy, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    assert bases_xy.size == n * A and scalars.size == n * S, "bases / scalars length mismatch"
    out = np.zeros(O, dtype=np.uint64)
    rc = getattr(lib(), "msm_" + group + ("_subgroup" if subgroup else ""))(_p(bases_xy), _p(inf), _p(scalars), C.c_size_t(n), _p(out))
    if rc != 0:
        raise RuntimeError(f"msm_{group} failed rc={rc}")
    return out


def msm_dev(group, d_bases, d_inf, d_scalars, n, stream=0, subgroup=False):
    """Device-pointer MSM: d_* are integer device addresses (e.g. torch tensor .data_ptr()).  subgroup: as for msm()."""
    O = GROUP_SHAPE[group][2]
    out = np.zeros(O, dtype=np.uint64)
    rc = getattr(lib(), "msm_" + group + ("_subgroup_dev" if subgroup else "_dev"))(C.c_void_p(d_bases), C.c_void_p(d_inf or 0), C.c_void_p(d_scalars),
                                                 C.c_size_t(n), _p(out), C.c_void_p(stream or 0))
    if rc != 0:
        raise RuntimeError(f"msm_{group}_dev failed rc={rc}")
    return out


def msm_timings(group):
    ms = (C.c_float * 5)()
    cfg = (C.c_int * 3)()
    rc = lib().celo_amd_msm_last_timings(C.c_int(GROUP_ID[group]), ms, cfg)
    assert rc == 0
    return {"convert_ms": ms[0], "sort_ms": ms[1], "accumulate_ms": ms[2], "reduce_ms": ms[3], "total_ms": ms[4],
            "window_bits": cfg[0], "windows": cfg[1], "buckets": cfg[2]}


def selftest_accumulate(group, gen_xy, runs=16384, length=24, seed=1, check=2048, chunked=False):
    """The library's own k_accumulate<group> / k_accumulate_chunk<group> against the host replay of the same templates; returns the number of
    runs whose partial sum differs (celo_amd_selftest_accumulate)."""
    gen_xy = np.ascontiguousarray(gen_xy, dtype=np.uint64)
    d = C.c_uint32(0xFFFFFFFF)
    rc = lib().celo_amd_selftest_accumulate(C.c_int(GROUP_ID[group]), _p(gen_xy), C.c_uint32(runs), C.c_uint32(length), C.c_uint32(seed), C.c_uint32(check),
                                            C.c_int(1 if chunked else 0), C.byref(d))
    if rc != 0:
        raise RuntimeError(f"celo_amd_selftest_accumulate failed rc={rc}")
    return d.value


def ubench_fp():
    """The multiplier peaks of this device, measured now (celo_amd_ubench_fp): G products/s for the 377- and 761-bit fields + the loop's clock."""
    out = (C.c_float * 9)()
    rc = lib().celo_amd_ubench_fp(out)
    if rc != 0:
        raise RuntimeError(f"celo_amd_ubench_fp failed rc={rc}")
    return {"fq377_mul_G": out[0], "fq377_sqr_G": out[1], "fq761_mul_G": out[2], "fq761_sqr_G": out[3], "clock_mhz": out[4],
            "kernel_ms": [out[5], out[6], out[7], out[8]]}


def set_batched_affine(on):
    """BW6-761 resident MSMs: 1 = batched-affine tree levels before the XYZZ chain (csrc/msm_ba.h), 0 = the chain alone, -1 = the default.  Process-wide."""
    if lib().celo_amd_msm_set_batched_affine(C.c_int(on)) != 0:
        raise ValueError(f"batched affine {on} not supported")


def set_host_chunks(chunks, head_split=None, tail_split=None):
    """Host-pointer msm(): index chunks of the pipelined transfer (0 = unpipelined, 1 = one chunk, -1 = default) and, optionally, how often the first /
    the last chunk is cut in halves (default: the library's).  Process-wide."""
    if chunks >= 0:
        if head_split is not None:
            chunks |= (head_split + 1) << 8
        if tail_split is not None:
            chunks |= (tail_split + 1) << 12
    rc = lib().celo_amd_msm_set_host_chunks(C.c_int(chunks))
    if rc != 0:
        raise ValueError(f"host chunks {chunks} not supported")


class PinnedArray:
    """A numpy array over page-locked host memory from celo_amd_host_alloc (freed with the object): `.a` is the array."""

    def __init__(self, shape, dtype):
        self.a = None
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = C.c_void_p()
        rc = lib().celo_amd_host_alloc(C.c_size_t(nbytes), C.byref(self._p))
        if rc != 0 or not self._p.value:
            raise MemoryError(f"celo_amd_host_alloc({nbytes}) failed rc={rc}")
        buf = (C.c_uint8 * nbytes).from_address(self._p.value)
        self.a = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def close(self):
        if getattr(self, "_p", None) is not None and self._p.value:
            self.a = None
            lib().celo_amd_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:                        # noqa: BLE001 (interpreter shutdown)
            pass


def host_chunk_plan(n, chunks, head_split=1, tail_split=0):
    """(cm, [chunk lengths]) of the pipelined host-pointer entry for n terms; None where the entry would not pipeline.  No device call."""
    cm = C.c_uint32(0)
    lens = (C.c_uint32 * 80)()
    k = lib().celo_amd_msm_host_chunk_plan(C.c_uint64(n), C.c_int(chunks), C.c_int(head_split), C.c_int(tail_split), C.byref(cm), lens)
    return None if k < 0 else (cm.value, [lens[i] for i in range(k)])


def set_window_bits(group, c):
    rc = lib().celo_amd_msm_set_window_bits(C.c_int(GROUP_ID[group]), C.c_int(c))
    if rc != 0:
        raise ValueError(f"window bits {c} not supported")


def gen_points_dev(group, d_out, n, seed, gen_xy, stream=0):
    name = {"bls12_377_g1": "celo_amd_gen_points_bls12_377_g1_dev", "bls12_377_g2": "celo_amd_gen_points_bls12_377_g2_dev",
            "bw6_761_g1": "celo_amd_gen_points_bw6_761_dev", "bw6_761_g2": "celo_amd_gen_points_bw6_761_dev"}[group]
    gen_xy = np.ascontiguousarray(gen_xy, dtype=np.uint64)
    rc = getattr(lib(), name)(C.c_void_p(d_out), C.c_size_t(n), C.c_uint64(seed), _p(gen_xy), C.c_void_p(stream or 0))
    if rc != 0:
        raise RuntimeError(f"{name} failed rc={rc}")


def gen_points_grouped_dev(group, d_out, n, seed, gens_xy, per, stream=0):
    """P_i = k_i * gens[i // per] into device memory (celo_amd_gen_points_grouped_*): gens_xy (ngens, A) uint64 host limbs."""
    name = {"bls12_377_g1": "celo_amd_gen_points_grouped_bls12_377_g1_dev", "bls12_377_g2": "celo_amd_gen_points_grouped_bls12_377_g2_dev"}[group]
    A = GROUP_SHAPE[group][0]
    gens_xy = np.ascontiguousarray(gens_xy, dtype=np.uint64).reshape(-1, A)
    rc = getattr(lib(), name)(C.c_void_p(d_out), C.c_size_t(n), C.c_uint64(seed), _p(gens_xy), C.c_size_t(gens_xy.shape[0]), C.c_uint32(per),
                              C.c_void_p(stream or 0))
    if rc != 0:
        raise RuntimeError(f"{name} failed rc={rc}")


def batch_verify(pk_xy, sig_xy, exponents, offsets, hash_xy, neg_g2_xy):
    """Batch::verify for m batches, host buffers (batch_verify_bls12_377).  Returns uint8 [m] verdicts."""
    pk_xy = np.ascontiguousarray(pk_xy, dtype=np.uint64)
    sig_xy = np.ascontiguousarray(sig_xy, dtype=np.uint64)
    exponents = np.ascontiguousarray(exponents, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    hash_xy = np.ascontiguousarray(hash_xy, dtype=np.uint64)
    ng2 = np.ascontiguousarray(neg_g2_xy, dtype=np.uint64).reshape(24)
    m = offsets.size - 1
    tot = int(offsets[-1])
    assert pk_xy.size == tot * 24 and sig_xy.size == tot * 12 and exponents.size == tot * 4 and hash_xy.size == m * 12
    out = np.zeros(m, dtype=np.uint8)
    rc = lib().batch_verify_bls12_377(_p(pk_xy), None, _p(sig_xy), None, _p(exponents), _p(offsets), _p(hash_xy), None, _p(ng2), C.c_size_t(m), _p(out))
    if rc != 0:
        raise RuntimeError(f"batch_verify_bls12_377 failed rc={rc}")
    return out


def batch_verify_dev(d_pk, d_sig, d_exp, offsets, d_hash, neg_g2_xy):
    """The same on inputs resident in HBM (integer device addresses); offsets: host uint32 [m+1]."""
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    ng2 = np.ascontiguousarray(neg_g2_xy, dtype=np.uint64).reshape(24)
    m = offsets.size - 1
    out = np.zeros(m, dtype=np.uint8)
    rc = lib().batch_verify_bls12_377_dev(C.c_void_p(d_pk), None, C.c_void_p(d_sig), None, C.c_void_p(d_exp), _p(offsets), C.c_void_p(d_hash), None, _p(ng2),
                                          C.c_size_t(m), _p(out))
    if rc != 0:
        raise RuntimeError(f"batch_verify_bls12_377_dev failed rc={rc}")
    return out


def draw_batch_exponents(key, offsets):
    """The exponents batch_verify_strict's device kernel draws for one call under `key` (8 uint32): uint64 [offsets[-1], 4]."""
    key = np.ascontiguousarray(key, dtype=np.uint32).reshape(8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    m = offsets.size - 1
    out = np.zeros((int(offsets[-1]), 4), dtype=np.uint64)
    rc = lib().celo_amd_draw_batch_exponents(_p(key), _p(offsets), C.c_size_t(m), _p(out))
    if rc != 0:
        raise RuntimeError(f"celo_amd_draw_batch_exponents failed rc={rc}")
    return out


def sum_jacobian(group, jac):
    """Plain sum of k Jacobian points (uint64 [k, O]) -> Jacobian limbs."""
    O = GROUP_SHAPE[group][2]
    jac = np.ascontiguousarray(jac, dtype=np.uint64).reshape(-1, O)
    out = np.zeros(O, dtype=np.uint64)
    name = {"bls12_377_g1": "celo_amd_sum_jacobian_bls12_377_g1", "bls12_377_g2": "celo_amd_sum_jacobian_bls12_377_g2",
            "bw6_761_g1": "celo_amd_sum_jacobian_bw6_761", "bw6_761_g2": "celo_amd_sum_jacobian_bw6_761"}[group]
    rc = getattr(lib(), name)(_p(jac), C.c_size_t(jac.shape[0]), _p(out))
    if rc != 0:
        raise RuntimeError(f"{name} failed rc={rc}")
    return out


def pairing_product_is_one(g1_xy, inf1, g2_xy, inf2):
    """One product of k pairings == 1 ?  g1_xy uint64 [k,12], g2_xy uint64 [k,24]."""
    g1_xy = np.ascontiguousarray(g1_xy, dtype=np.uint64)
    g2_xy = np.ascontiguousarray(g2_xy, dtype=np.uint64)
    k = g1_xy.size // 12
    one = C.c_int(0)
    rc = lib().pairing_product_is_one_bls12_377(_p(g1_xy), _p(inf1), _p(g2_xy), _p(inf2), C.c_size_t(k), C.byref(one))
    if rc != 0:
        raise RuntimeError(f"pairing_product_is_one_bls12_377 failed rc={rc}")
    return bool(one.value)


def pairing_product_is_one_batch(g1_xy, inf1, g2_xy, inf2, offsets):
    """m independent products; offsets uint32 [m+1]. Returns uint8 [m]."""
    g1_xy = np.ascontiguousarray(g1_xy, dtype=np.uint64)
    g2_xy = np.ascontiguousarray(g2_xy, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    m = offsets.size - 1
    out = np.zeros(m, dtype=np.uint8)
    rc = lib().pairing_product_is_one_batch_bls12_377(_p(g1_xy), _p(inf1), _p(g2_xy), _p(inf2), _p(offsets), C.c_size_t(m), _p(out))
    if rc != 0:
        raise RuntimeError(f"pairing_product_is_one_batch_bls12_377 failed rc={rc}")
    return out


def pairing_gt(g1_xy, inf1, g2_xy, inf2, offsets, miller_only=False):
    g1_xy = np.ascontiguousarray(g1_xy, dtype=np.uint64)
    g2_xy = np.ascontiguousarray(g2_xy, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    m = offsets.size - 1
    out = np.zeros((m, 72), dtype=np.uint64)
    rc = lib().celo_amd_pairing_gt_bls12_377(_p(g1_xy), _p(inf1), _p(g2_xy), _p(inf2), _p(offsets), C.c_size_t(m),
                                             C.c_int(1 if miller_only else 0), _p(out))
    if rc != 0:
        raise RuntimeError(f"celo_amd_pairing_gt_bls12_377 failed rc={rc}")
    return out


def ntt(data, log_n, omega6, coset6=None, coset_after=False, scale6=None):
    """In-place NTT over Fr(BW6-761) on HOST data: (n, 6) uint64 arkworks Montgomery limbs; omega6 / coset6 / scale6: (6,) uint64
    Montgomery limbs.  Returns the transformed array (a copy)."""
    out = np.ascontiguousarray(data, dtype=np.uint64).copy()
    assert out.shape == (1 << log_n, 6)
    w = np.ascontiguousarray(omega6, dtype=np.uint64)
    g = None if coset6 is None else np.ascontiguousarray(coset6, dtype=np.uint64)
    sc = None if scale6 is None else np.ascontiguousarray(scale6, dtype=np.uint64)
    rc = lib().ntt_bw6_761_fr(_p(out), C.c_uint(log_n), _p(w), _p(g), C.c_int(1 if coset_after else 0), _p(sc))
    if rc != 0:
        raise RuntimeError("ntt_bw6_761_fr failed with code %d" % rc)
    return out


def ntt_dev(d_ptr, log_n, omega6, coset6=None, coset_after=False, scale6=None, stream=0):
    w = np.ascontiguousarray(omega6, dtype=np.uint64)
    g = None if coset6 is None else np.ascontiguousarray(coset6, dtype=np.uint64)
    sc = None if scale6 is None else np.ascontiguousarray(scale6, dtype=np.uint64)
    rc = lib().ntt_bw6_761_fr_dev(C.c_void_p(d_ptr), C.c_uint(log_n), _p(w), _p(g), C.c_int(1 if coset_after else 0), _p(sc), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError("ntt_bw6_761_fr_dev failed with code %d" % rc)


def witness_map(a, b, c, log_n, consts, canonical=False):
    """R1CStoQAP::witness_map from the QAP evaluations (groth16_witness_map_bw6_761).  a, b, c: (n, 6) uint64 arkworks Montgomery
    limbs (copied); consts: dict of (6,) uint64 Montgomery limbs: omega, omega_inv, coset, coset_inv, size_inv, vanishing_inv.
    Returns h (n, 6): field elements, or canonical integers with canonical=True."""
    bufs = [np.ascontiguousarray(x, dtype=np.uint64).copy() for x in (a, b, c)]
    assert all(x.shape == (1 << log_n, 6) for x in bufs)
    k = [np.ascontiguousarray(consts[n], dtype=np.uint64) for n in ("omega", "omega_inv", "coset", "coset_inv", "size_inv", "vanishing_inv")]
    rc = lib().groth16_witness_map_bw6_761(_p(bufs[0]), _p(bufs[1]), _p(bufs[2]), C.c_uint(log_n), *[_p(x) for x in k], C.c_int(1 if canonical else 0))
    if rc != 0:
        raise RuntimeError("groth16_witness_map_bw6_761 failed with code %d" % rc)
    return bufs[0]


def witness_map_dev(d_a, d_b, d_c, log_n, consts, canonical=False, stream=0):
    k = [np.ascontiguousarray(consts[n], dtype=np.uint64) for n in ("omega", "omega_inv", "coset", "coset_inv", "size_inv", "vanishing_inv")]
    rc = lib().groth16_witness_map_bw6_761_dev(C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_c), C.c_uint(log_n), *[_p(x) for x in k],
                                               C.c_int(1 if canonical else 0), C.c_void_p(stream or 0))
    if rc != 0:
        raise RuntimeError("groth16_witness_map_bw6_761_dev failed with code %d" % rc)


def groth16_prove(a_query, b_g2_query, h_query, l_query, alpha_g1, beta_g2, assignment, n_aux, h):
    """create_proof_no_zk's group arithmetic (groth16_prove_bw6_761).  queries: (k, 24) uint64; alpha / beta: (24,); assignment,
    h: (k, 6) uint64 canonical scalars.  Returns (A, B, C) Jacobian limbs (36 u64 each)."""
    arrs = [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 24) for x in (a_query, b_g2_query, h_query, l_query)]
    al, be = (np.ascontiguousarray(x, dtype=np.uint64).reshape(24) for x in (alpha_g1, beta_g2))
    asg = np.ascontiguousarray(assignment, dtype=np.uint64).reshape(-1, 6)
    hh = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 6)
    out = [np.zeros(36, dtype=np.uint64) for _ in range(3)]
    rc = lib().groth16_prove_bw6_761(_p(arrs[0]), C.c_size_t(arrs[0].shape[0]), _p(arrs[1]), C.c_size_t(arrs[1].shape[0]), _p(arrs[2]), C.c_size_t(arrs[2].shape[0]),
                                     _p(arrs[3]), C.c_size_t(arrs[3].shape[0]), _p(al), _p(be), _p(asg), C.c_size_t(asg.shape[0]), C.c_size_t(n_aux),
                                     _p(hh), C.c_size_t(hh.shape[0]), _p(out[0]), _p(out[1]), _p(out[2]))
    if rc != 0:
        raise RuntimeError("groth16_prove_bw6_761 failed with code %d" % rc)
    return out


class ProvingKey:
    """A loaded Groth16 proving key (groth16_load_key_bw6_761 / _bls12_377): the four queries' fixed-base tables, built once; prove() per
    assignment.  curve: "bw6_761" (points (k, 24), scalars (k, 6)) or "bls12_377" (G1 (k, 12), G2 (k, 24), scalars (k, 4))."""
    def __init__(self, curve, a_query, b_g2_query, h_query, l_query, alpha_g1, beta_g2, window_bits=0):
        self.curve = curve
        g1w = 24 if curve == "bw6_761" else 12
        self.sw = 6 if curve == "bw6_761" else 4
        self.ow = (36, 36, 36) if curve == "bw6_761" else (18, 36, 18)
        q = [np.ascontiguousarray(a_query, dtype=np.uint64).reshape(-1, g1w), np.ascontiguousarray(b_g2_query, dtype=np.uint64).reshape(-1, 24),
             np.ascontiguousarray(h_query, dtype=np.uint64).reshape(-1, g1w), np.ascontiguousarray(l_query, dtype=np.uint64).reshape(-1, g1w)]
        al = np.ascontiguousarray(alpha_g1, dtype=np.uint64).reshape(g1w)
        be = np.ascontiguousarray(beta_g2, dtype=np.uint64).reshape(24)
        self.h = C.c_void_p()
        fn = lib().groth16_load_key_bw6_761 if curve == "bw6_761" else lib().groth16_load_key_bls12_377
        rc = fn(_p(q[0]), C.c_size_t(q[0].shape[0]), _p(q[1]), C.c_size_t(q[1].shape[0]), _p(q[2]), C.c_size_t(q[2].shape[0]), _p(q[3]), C.c_size_t(q[3].shape[0]),
                _p(al), _p(be), C.c_int(window_bits), C.byref(self.h))
        if rc != 0:
            raise RuntimeError("groth16_load_key_%s failed with code %d" % (curve, rc))

    def prove(self, assignment, n_aux, h):
        asg = np.ascontiguousarray(assignment, dtype=np.uint64).reshape(-1, self.sw)
        hh = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, self.sw)
        out = [np.zeros(w, dtype=np.uint64) for w in self.ow]
        rc = lib().groth16_prove_with_key(self.h, _p(asg), C.c_size_t(asg.shape[0]), C.c_size_t(n_aux), _p(hh), C.c_size_t(hh.shape[0]), _p(out[0]), _p(out[1]), _p(out[2]))
        if rc != 0:
            raise RuntimeError("groth16_prove_with_key failed with code %d" % rc)
        return out

    def release(self):
        if self.h:
            lib().groth16_free_key(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


# ---- the hash-helper proof's field and curve: Fr(BLS12-377) elements are 4 u64, G1 points 12 u64 affine / 18 Jacobian, G2 24 / 36
def ntt_fr377(data, log_n, omega4, coset4=None, coset_after=False, scale4=None):
    """In-place NTT over Fr(BLS12-377) on HOST data (ntt_bls12_377_fr): (n, 4) uint64 arkworks Montgomery limbs.  Returns a copy."""
    out = np.ascontiguousarray(data, dtype=np.uint64).copy()
    assert out.shape == (1 << log_n, 4)
    w = np.ascontiguousarray(omega4, dtype=np.uint64)
    g = None if coset4 is None else np.ascontiguousarray(coset4, dtype=np.uint64)
    sc = None if scale4 is None else np.ascontiguousarray(scale4, dtype=np.uint64)
    rc = lib().ntt_bls12_377_fr(_p(out), C.c_uint(log_n), _p(w), _p(g), C.c_int(1 if coset_after else 0), _p(sc))
    if rc != 0:
        raise RuntimeError("ntt_bls12_377_fr failed with code %d" % rc)
    return out


def ntt_fr377_dev(d_ptr, log_n, omega4, coset4=None, coset_after=False, scale4=None, stream=0):
    w = np.ascontiguousarray(omega4, dtype=np.uint64)
    g = None if coset4 is None else np.ascontiguousarray(coset4, dtype=np.uint64)
    sc = None if scale4 is None else np.ascontiguousarray(scale4, dtype=np.uint64)
    rc = lib().ntt_bls12_377_fr_dev(C.c_void_p(d_ptr), C.c_uint(log_n), _p(w), _p(g), C.c_int(1 if coset_after else 0), _p(sc), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError("ntt_bls12_377_fr_dev failed with code %d" % rc)


def witness_map_fr377(a, b, c, log_n, consts, canonical=False):
    """groth16_witness_map_bls12_377: a, b, c (n, 4) uint64 Montgomery limbs (copied); consts as for witness_map, (4,) limbs each."""
    bufs = [np.ascontiguousarray(x, dtype=np.uint64).copy() for x in (a, b, c)]
    assert all(x.shape == (1 << log_n, 4) for x in bufs)
    k = [np.ascontiguousarray(consts[n], dtype=np.uint64) for n in ("omega", "omega_inv", "coset", "coset_inv", "size_inv", "vanishing_inv")]
    rc = lib().groth16_witness_map_bls12_377(_p(bufs[0]), _p(bufs[1]), _p(bufs[2]), C.c_uint(log_n), *[_p(x) for x in k], C.c_int(1 if canonical else 0))
    if rc != 0:
        raise RuntimeError("groth16_witness_map_bls12_377 failed with code %d" % rc)
    return bufs[0]


def witness_map_fr377_dev(d_a, d_b, d_c, log_n, consts, canonical=False, stream=0):
    k = [np.ascontiguousarray(consts[n], dtype=np.uint64) for n in ("omega", "omega_inv", "coset", "coset_inv", "size_inv", "vanishing_inv")]
    rc = lib().groth16_witness_map_bls12_377_dev(C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_c), C.c_uint(log_n), *[_p(x) for x in k],
                                                 C.c_int(1 if canonical else 0), C.c_void_p(stream or 0))
    if rc != 0:
        raise RuntimeError("groth16_witness_map_bls12_377_dev failed with code %d" % rc)


def groth16_prove_bls12_377(a_query, b_g2_query, h_query, l_query, alpha_g1, beta_g2, assignment, n_aux, h):
    """create_proof_no_zk's group arithmetic over BLS12-377 (groth16_prove_bls12_377).  a / h / l queries: (k, 12) uint64, b_g2_query:
    (k, 24); alpha (12,), beta (24,); assignment, h: (k, 4) canonical scalars.  Returns (A (18,), B (36,), C (18,)) Jacobian limbs."""
    aq, hq, lq = (np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 12) for x in (a_query, h_query, l_query))
    bq = np.ascontiguousarray(b_g2_query, dtype=np.uint64).reshape(-1, 24)
    al = np.ascontiguousarray(alpha_g1, dtype=np.uint64).reshape(12)
    be = np.ascontiguousarray(beta_g2, dtype=np.uint64).reshape(24)
    asg = np.ascontiguousarray(assignment, dtype=np.uint64).reshape(-1, 4)
    hh = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 4)
    out = [np.zeros(18, dtype=np.uint64), np.zeros(36, dtype=np.uint64), np.zeros(18, dtype=np.uint64)]
    rc = lib().groth16_prove_bls12_377(_p(aq), C.c_size_t(aq.shape[0]), _p(bq), C.c_size_t(bq.shape[0]), _p(hq), C.c_size_t(hq.shape[0]),
                                       _p(lq), C.c_size_t(lq.shape[0]), _p(al), _p(be), _p(asg), C.c_size_t(asg.shape[0]), C.c_size_t(n_aux),
                                       _p(hh), C.c_size_t(hh.shape[0]), _p(out[0]), _p(out[1]), _p(out[2]))
    if rc != 0:
        raise RuntimeError("groth16_prove_bls12_377 failed with code %d" % rc)
    return out


def ntt_timings():
    ms = (C.c_float * 4)()
    n = C.c_int(0)
    assert lib().celo_amd_ntt_last_timings(ms, C.byref(n)) == 0
    return {"load_ms": ms[0], "passes_ms": ms[1], "store_ms": ms[2], "total_ms": ms[3], "passes": n.value}


def pairing_timings():
    ms = (C.c_float * 4)()
    assert lib().celo_amd_pairing_last_timings(ms) == 0
    return {"miller_ms": ms[0], "product_ms": ms[1], "final_exp_ms": ms[2], "total_ms": ms[3]}


def msm_batch(group, bases_xy, inf, scalars, offsets, subgroup=False):
    """m independent MSMs in one call. offsets uint32 [m+1]. Returns uint64 [m, O] Jacobian results.
    subgroup=True (bls12_377_g2 only): the bases are vouched to lie in G2 (msm_batch_bls12_377_g2_subgroup: endomorphism split)."""
    A, S, O = GROUP_SHAPE[group]
    bases_xy = np.ascontiguousarray(bases_xy, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    m = offsets.size - 1
    assert bases_xy.size == int(offsets[-1]) * A and scalars.size == int(offsets[-1]) * S
    out = np.zeros((m, O), dtype=np.uint64)
    rc = getattr(lib(), "msm_batch_" + group + ("_subgroup" if subgroup else ""))(_p(bases_xy), _p(inf), _p(scalars), _p(offsets), C.c_size_t(m), _p(out))
    if rc != 0:
        raise RuntimeError(f"msm_batch_{group} failed rc={rc}")
    return out


def pairing_product_is_one_bw6(g1_xy, inf1, g2_xy, inf2):
    g1_xy = np.ascontiguousarray(g1_xy, dtype=np.uint64)
    g2_xy = np.ascontiguousarray(g2_xy, dtype=np.uint64)
    k = g1_xy.size // 24
    one = C.c_int(0)
    rc = lib().pairing_product_is_one_bw6_761(_p(g1_xy), _p(inf1), _p(g2_xy), _p(inf2), C.c_size_t(k), C.byref(one))
    if rc != 0:
        raise RuntimeError(f"pairing_product_is_one_bw6_761 failed rc={rc}")
    return bool(one.value)


def pairing_gt_bw6(g1_xy, inf1, g2_xy, inf2, offsets, miller_only=False):
    g1_xy = np.ascontiguousarray(g1_xy, dtype=np.uint64)
    g2_xy = np.ascontiguousarray(g2_xy, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    m = offsets.size - 1
    out = np.zeros((m, 72), dtype=np.uint64)
    rc = lib().celo_amd_pairing_gt_bw6_761(_p(g1_xy), _p(inf1), _p(g2_xy), _p(inf2), _p(offsets), C.c_size_t(m),
                                           C.c_int(1 if miller_only else 0), _p(out))
    if rc != 0:
        raise RuntimeError(f"celo_amd_pairing_gt_bw6_761 failed rc={rc}")
    return out


def decompress(group, data, check_subgroup=True):
    """Bulk decoding of compressed points (include/celo_bls_amd.h: decompress_bls12_377_g1/_g2).  group: "g1" (48 B each) or
    "g2" (96 B each); data: bytes of n concatenated encodings.  Returns (xy, status): xy (n, 12 | 24) uint64 affine arkworks
    Montgomery limbs (zero rows unless status == 0), status (n,) uint8 (0 ok, 1 infinity, 2 invalid, 3 not in subgroup)."""
    size, words, fn = {"g1": (48, 12, "decompress_bls12_377_g1"), "g2": (96, 24, "decompress_bls12_377_g2")}[group]
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    assert buf.size % size == 0
    n = buf.size // size
    xy = np.zeros((n, words), dtype=np.uint64)
    st = np.zeros(n, dtype=np.uint8)
    rc = getattr(lib(), fn)(_p(buf), C.c_size_t(n), C.c_int(1 if check_subgroup else 0), _p(xy), _p(st))
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (fn, rc))
    return xy, st


def decompress_dev(group, d_in, n, d_out, d_status, check_subgroup=True, stream=0):
    fn = {"g1": "decompress_bls12_377_g1_dev", "g2": "decompress_bls12_377_g2_dev"}[group]
    rc = getattr(lib(), fn)(C.c_void_p(d_in), C.c_size_t(n), C.c_int(1 if check_subgroup else 0), C.c_void_p(d_out), C.c_void_p(d_status), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (fn, rc))


def decompress_last_ms():
    ms = C.c_float(0)
    assert lib().celo_amd_decompress_last_ms(C.byref(ms)) == 0
    return ms.value


def hash_to_g1_composite(domain, messages, extras=None, cip22=False):
    """Batched hash-to-G1 with the composite hasher (include/celo_bls_amd.h: hash_to_g1_composite_bls12_377); same conventions
    as hash_to_g1_direct."""
    return hash_to_g1_direct(domain, messages, extras, composite=(2 if cip22 else 1))


def hash_to_g1_direct(domain, messages, extras=None, cip22_tail=False, composite=0):
    """Batched try-and-increment hash-to-G1 over the direct hasher (include/celo_bls_amd.h: hash_to_g1_direct_bls12_377), or,
    with cip22_tail, the CIP22 loop over precomputed inner CRHs (hash_to_g1_cip22_tail_bls12_377: `messages` are the inner
    hashes).  domain: 8 bytes; messages / extras: lists of bytes (extras None = no extra data).  Returns (xy (n, 12) uint64
    affine arkworks Montgomery limbs, attempts (n,) uint8; 255 = no point)."""
    n = len(messages)
    assert len(domain) == 8 and (extras is None or len(extras) == n)

    def pack(items):
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(b) for b in items], dtype=np.uint64) if n else []
        data = np.frombuffer(b"".join(items) or b"\0", dtype=np.uint8)
        return data, off
    mdat, moff = pack(messages)
    edat, eoff = pack(extras) if extras is not None else (None, None)
    dom = np.frombuffer(bytes(domain), dtype=np.uint8)
    xy = np.zeros((n, 12), dtype=np.uint64)
    att = np.zeros(n, dtype=np.uint8)
    if composite:
        rc = lib().hash_to_g1_composite_bls12_377(_p(dom), _p(mdat), _p(moff), _p(edat), _p(eoff), C.c_size_t(n), C.c_int(composite - 1), _p(xy), _p(att))
    else:
        fn = lib().hash_to_g1_cip22_tail_bls12_377 if cip22_tail else lib().hash_to_g1_direct_bls12_377
        rc = fn(_p(dom), _p(mdat), _p(moff), _p(edat), _p(eoff), C.c_size_t(n), _p(xy), _p(att))
    if rc != 0:
        raise RuntimeError("hash_to_g1 (%s) failed with code %d" % ("cip22 tail" if cip22_tail else "direct", rc))
    return xy, att


def hash_last_ms():
    ms = C.c_float(0)
    assert lib().celo_amd_hash_last_ms(C.byref(ms)) == 0
    return ms.value


def composite_crh(messages):
    """The composite hasher's Pedersen CRH of n messages in one GPU launch (include/celo_bls_amd.h: composite_crh_bls12_377).
    Returns a list of n 48-byte hashes."""
    n = len(messages)
    off = np.zeros(n + 1, dtype=np.uint64)
    if n:
        off[1:] = np.cumsum([len(b) for b in messages], dtype=np.uint64)
    data = np.frombuffer(b"".join(messages) or b"\0", dtype=np.uint8)
    out = np.zeros((n, 48), dtype=np.uint8)
    rc = lib().composite_crh_bls12_377(_p(data), _p(off), C.c_size_t(n), _p(out))
    if rc != 0:
        raise RuntimeError("composite_crh_bls12_377 failed with code %d" % rc)
    return [out[i].tobytes() for i in range(n)]


def normalize(group, jac):
    """Jacobian -> affine for n points in one GPU launch (include/celo_bls_amd.h: normalize_bls12_377_g1/_g2).  jac: (n, 18 | 36)
    uint64; returns (xy (n, 12 | 24) uint64, inf (n,) uint8)."""
    words, fn = {"g1": (6, "normalize_bls12_377_g1"), "g2": (12, "normalize_bls12_377_g2")}[group]
    j = np.ascontiguousarray(jac, dtype=np.uint64).reshape(-1, 3 * words)
    n = j.shape[0]
    xy = np.zeros((n, 2 * words), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    rc = getattr(lib(), fn)(_p(j), C.c_size_t(n), _p(xy), _p(inf))
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (fn, rc))
    return xy, inf
