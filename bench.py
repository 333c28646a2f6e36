#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's configurations, one JSON line per run (rank 0).

  --config 2 (default)  BLS12-377 G1 Pippenger MSM, 2^20 random bases/scalars            (the configuration the metric is quoted on)
  --config 3            Batch::verify: 4096 batches x 256 signers (G2 MSM + G1 MSM + 2-pair check per batch), chained on the device
  --config 4            BW6-761 G1 MSM (Groth16 prover shape): 2^24 bases over 8 GPUs = 2^21 per GPU
  --config 5            mixed: G1 MSM 2^22 + G2 MSM 2^22 + 2^14 Miller loops, the three legs issued concurrently
  --scaling weak        per-GPU work fixed as N grows (default; cfg4 weak = its per-GPU shard 2^21)
  --scaling strong      total work fixed (2^20 / 4096 batches / 2^24 / 2^22+2^22+2^14 split over the N ranks)
  --partition P         how ONE MSM is cut over N GPUs (configs 2 / 4): index = contiguous index ranges, each GPU holds 1/N of the terms
                        (the default for weak scaling and for the prover's 2^24 terms); windows = every GPU holds all
                        terms and owns 1/N of the Pippenger windows (the default for --scaling strong up to 2^21 terms: an index-range
                        shard of 2^20 / 8 terms is bound by the pipeline's fixed latencies, a window shard is not)

  --gpus N              N > 1 without a launcher around it: bench.py starts its own N ranks (torch.distributed.run, one process per GPU,
                        RCCL); under the driver's torch.distributed.run the ranks are already there.  A line whose n_gpus differs from
                        --gpus is never printed.
  --in-process          configs 2 / 4 with N > 1: ONE process drives the N devices through msm_*_multi_dev (the shape of the reference's
                        callers, crates/bls-snark-sys/src/signatures.rs:343, crates/epoch-snark/src/api/prover.rs:78): one host thread per
                        device inside the library, partial sums folded on the host, no collective

A "step" is one pass of the hot path over one batch of synthetic input that is resident in HBM before the timed region.
With N > 1 ranks (one process per GPU, launched by torch.distributed.run) an MSM is ONE job sharded by index range
(SURVEY.md section 8e): each rank computes the partial sum of its slice, the 144 / 288-byte Jacobian partials are exchanged with
one RCCL all_gather and every rank folds them; batches and pairing products shard with no exchange at all.

The line carries the driver's contract fields plus
  "roofline":     the dominant kernel - algorithmic bytes per launch / its HIP-event duration vs the HBM peak
  "cpu_baseline": the oracle's C++ restatement of the arkworks CPU path (kind "port"; the Rust reference cannot be built here)
                  timed on this box's host cores on a bounded sample, after the timed GPU result has been compared with it
                  (at N > 1 rank 0 gathers every rank's inputs once and checks the folded result at full size).
"""
import argparse
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
DTYPE = "u32 limbs (28-bit radix, 64-bit column accumulators)"
KIND = {"bls12_377_g1": "g1_377", "bls12_377_g2": "g2_377", "bw6_761_g1": "761", "bw6_761_g2": "761"}
ACC_KERNEL = {"bls12_377_g1": "k_accumulate<G1_377>", "bls12_377_g2": "k_accumulate_pair<G2_377, 1>", "bw6_761_g1": "k_accumulate<G_761>"}


class Ctx:
    pass


# launch shapes the committed PMC profiles were taken at, and the profile set (profiles/<tag>_traffic.json, written by
# tools/summarise_profile.py from the separate --pmc FETCH_SIZE / WRITE_SIZE passes of tools/archive/r5_profiles.sh) that holds each:
# kernel -> {shape: tag}.  Shapes: log2 n of an MSM, the number of products of a pairing launch, "cfg3" = 4096 batches x 256 signers.
PROFILED = {"k_accumulate<G1_377>": {20: "r6", 22: "r6_cfg5", "cfg3": "r6_cfg3"},
            "k_accumulate_pair<G2_377, 1>": {20: "r6_groups", 22: "r6_cfg5", "cfg3": "r6_cfg3"},
            "k_accumulate<G_761>": {20: "r6_groups", 21: "r6_cfg4"},
            "k_miller_product_slots<LPH377, 2>": {81920: "r6_pairing"}, "k_miller_prepared_slots<LPH377>": {81920: "r6"},
            "k_prepare_lines<LPH377>": {81920: "r6"}, "k_final_exp_slots<LPH377>": {81920: "r6"}}


_BUILD_SIG = None
_PEAKS = None


# Multiply-adds (v_mad_u64_u32 / v_mad_i64_i32 lane-operations) of the library's arithmetic, counted from csrc/fp.h, fp2.h, curve.h.
# 14-limb field (p = 1 mod 2^28: the quotient's p0 term is free): product sweep 14^2 = 196, symmetric (squaring) sweep 105, reduction sweep
# 14 * 13 = 182; 28-limb field: 784, 406, 784.
MADS = {
    "fq377_mul": 196 + 182, "fq377_sqr": 105 + 182, "fq761_mul": 784 + 784, "fq761_sqr": 406 + 784,
    # one XYZZ mixed addition (madd-2008-s, curve.h xyzz_madd): U2, S2, PPP, Q, ZZ3, ZZZ3 products, PP and R^2 squarings, Y3 = R t - Y1 PPP in one pass
    "madd": {"bls12_377_g1": 6 * (196 + 182) + 2 * (105 + 182) + (2 * 196 + 182),                                   # 3416
             # Fq2: a product = two passes of (2 sweeps + 1 reduction); a squaring = sqr2m5 (2 x 105 + 182) + one Fq product; Y3 = two mul4k passes (4 sweeps + 1 reduction)
             "bls12_377_g2": 6 * 2 * (2 * 196 + 182) + 2 * ((2 * 105 + 182) + (196 + 182)) + 2 * (4 * 196 + 182),     # 10360
             "bw6_761_g1": 6 * (784 + 784) + 2 * (406 + 784) + (2 * 784 + 784),                                      # 14140
             "bw6_761_g2": 6 * (784 + 784) + 2 * (406 + 784) + (2 * 784 + 784)},
    # one product round of the six-lane pairing backend: a half-Fq2 signed pass (2 sweeps + 1 reduction) on each of a group's six lanes
    "hex_round": 6 * (2 * 196 + 182),
    # what the kernel named in ACC_KERNEL EXECUTES per mixed addition where that differs from the formula's count above: the lane-pair G2 kernel
    # runs ten pair products (one two-product signed pass on each lane of the pair), ADVICE r5
    "madd_executed": {"bls12_377_g2": 10 * 2 * (2 * 196 + 182)},                                                     # 11480
}


_COPY_PEAK = None


def hbm_copy_peak():
    """"measured copy bandwidth as denominator too".  A device-to-device copy of 1 GiB (read + write = 2 GiB of HBM traffic)
    timed in THIS run on THIS device with events on torch's stream; best of 5.  GB/s."""
    global _COPY_PEAK
    if _COPY_PEAK is None:
        n = 1 << 30
        a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.uint8, device="cuda")
        a.zero_(); b.copy_(a)
        best = None
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); b.copy_(a); e1.record(); e1.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None or ms < best else best
        del a, b
        _COPY_PEAK = 2 * n / (best * 1e-3) / 1e9
    return _COPY_PEAK


def add_measured_hbm_peak(obj):
    """Every {"bound": "hbm"} block of the line also gets the copy bandwidth measured in this run and the fraction against it."""
    if isinstance(obj, dict):
        if obj.get("bound") == "hbm" and "achieved" in obj and "peak_measured" not in obj:
            obj["peak_measured"] = hbm_copy_peak()
            obj["frac_of_measured"] = obj["achieved"] / obj["peak_measured"]
            obj["peak_measured_note"] = "1 GiB device-to-device copy in this run (2 GiB of traffic), best of 5, GB/s"
        for v in list(obj.values()):
            add_measured_hbm_peak(v)
    elif isinstance(obj, list):
        for v in obj:
            add_measured_hbm_peak(v)


def valu_peaks():
    """The multiplier roofline of this run, in MULTIPLY-ADDS per second: the library's own field-product loops timed on THIS device in THIS
    process, right after the timed steps (celo_amd_ubench_fp, csrc/unit_ubench.hip; ~0.1 s) - products/s x multiply-adds per product (MADS).
    The mul and sqr loops of a field agree to 1 % in these units (14 limbs: 73.5 G x 378 = 27.8 T, 96.0 G x 287 = 27.6 T), which is what makes the
    unit the right one: kernels are priced by the multiply-adds their formulas execute, not by a nominal count of field operations (rounds 2-4
    priced an XYZZ mixed addition as 8 M + 2 S although its Y3 pass shares one reduction between two products, and an Fq2 product as four Fq
    products although it takes three: 0.96-0.98 and > 1 where the counts below give 0.91 and 0.8)."""
    global _PEAKS
    if _PEAKS is None:
        from celo_bls_snark_rs_amd import ffi
        u = ffi.ubench_fp()
        p377 = u["fq377_mul_G"] * MADS["fq377_mul"] / 1e3
        p761 = u["fq761_mul_G"] * MADS["fq761_mul"] / 1e3
        _PEAKS = {"measured": u, "tmads": {"bls12_377_g1": p377, "bls12_377_g2": p377, "bw6_761_g1": p761, "bw6_761_g2": p761},
                  "sqr_loops_tmads": {"fq377": u["fq377_sqr_G"] * MADS["fq377_sqr"] / 1e3, "fq761": u["fq761_sqr_G"] * MADS["fq761_sqr"] / 1e3},
                  "note": "peak measured in this run on this device by celo_amd_ubench_fp (register-resident loops of the library's own product bodies): "
                          "Fq377 mul %.1f / sqr %.1f, Fq761 mul %.1f / sqr %.1f G products/s = %.1f / %.1f / %.1f / %.1f T multiply-adds/s (378 / 287 / 1568 / 1190 per product), "
                          "shader clock during the first loop %.0f MHz"
                          % (u["fq377_mul_G"], u["fq377_sqr_G"], u["fq761_mul_G"], u["fq761_sqr_G"], p377, u["fq377_sqr_G"] * MADS["fq377_sqr"] / 1e3, p761,
                             u["fq761_sqr_G"] * MADS["fq761_sqr"] / 1e3, u["clock_mhz"])}
    return _PEAKS


def build_signature():
    """{kernel: {VGPRs, AGPRs, ScratchSize, TotalSGPRs}} of the library this process runs, from the compiler's remarks next to it
    (celo-bls-snark-rs_amd/build/unit_*.remarks.txt, written by the Makefile) - the same extraction tools/summarise_profile.py stores
    beside a profile's traffic figures."""
    global _BUILD_SIG
    if _BUILD_SIG is not None:
        return _BUILD_SIG
    import re
    import subprocess
    sig = {}
    for path in glob.glob(os.path.join(ROOT, "celo-bls-snark-rs_amd", "build", "unit_*.remarks.txt")):
        cur = None
        for ln in open(path, errors="replace"):
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                cur = m.group(1); sig.setdefault(cur, {}); continue
            m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|TotalSGPRs): (\d+)", ln)
            if m and cur:
                sig[cur][m.group(1).split(" ")[0]] = int(m.group(2))
    out = {}
    if sig:
        names = list(sig)
        try:
            dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, timeout=30).stdout.splitlines()
        except Exception:
            dem = []
        for n, d in zip(names, dem):
            out[d.replace("void celo::", "").replace("celo::", "").split("(")[0]] = sig[n]
    _BUILD_SIG = out
    return out


def committed_traffic(kernels, shape):
    """HBM bytes per launch (summed over `kernels`) from the committed PMC profile of this launch shape, or (None, why).  The source string
    says STALE when a kernel's registers / scratch in the build that runs differ from the profiled build's: the
    figure then describes other code."""
    kernels = [kernels] if isinstance(kernels, str) else list(kernels)
    total, srcs, stale = 0.0, [], []
    try:
        for k in kernels:
            tag = PROFILED.get(k, {}).get(shape)
            if tag is None:
                return None, "no committed PMC profile of this launch shape"
            doc = json.load(open(os.path.join(ROOT, "profiles", tag + "_traffic.json")))
            total += doc.get("kernels", {})[k]["hbm_bytes_per_launch"]
            if tag + "_traffic.json" not in srcs:
                srcs.append(tag + "_traffic.json")
            was, now = doc.get("build_signature", {}).get(k), build_signature().get(k)
            if was is None or now is None:
                stale.append("%s: no build signature to compare" % k)
            elif was != now:
                stale.append("%s: profiled build %s, this build %s" % (k, was, now))
    except Exception:
        return None, "no committed PMC profile holds these kernels"
    return total, " + ".join(srcs) + (" - STALE (%s)" % "; ".join(stale) if stale else " (build signature matches the profiled build)")


# ===================================================================================================== MSM configurations (2 and 4)
class MsmConfig:
    def __init__(self, cx, group, log_n_total, name):
        self.cx, self.group, self.name = cx, group, name
        a = cx.args
        log_n = a.log_n if a.log_n else log_n_total
        n_total = 1 << log_n
        # the window partition: every GPU holds ALL the terms of the ONE job and owns a range of the windows (strong scaling only)
        self.by_windows = cx.nshards > 1 and a.scaling == "strong" and (a.partition == "windows" or (a.partition == "auto" and n_total <= (1 << 21)))
        if a.partition == "windows" and cx.nshards > 1 and a.scaling != "strong":
            raise SystemExit("--partition windows cuts ONE job: use it with --scaling strong")
        self.n = n_total if self.by_windows else (n_total // cx.nshards if a.scaling == "strong" else n_total)
        if cx.cfg == 4 and a.scaling == "weak" and not a.log_n:
            self.n = 1 << 21                                   # cfg4's job is 2^24 over 8 GPUs: the per-GPU shard is the weak unit
        self.log_n = (self.n - 1).bit_length()
        self.fixed = None

    def setup(self):
        from celo_bls_snark_rs_amd import ffi, synthetic as syn
        cx = self.cx
        if cx.args.window_bits:
            ffi.set_window_bits(self.group, cx.args.window_bits)
        if cx.devices:
            return self.setup_in_process()
        sr = 0 if self.by_windows else cx.rank                 # window partition: every rank holds the SAME n terms
        self.bases = syn.device_points(self.group, self.n, 0x5EED0002 + 0x1000 * sr)
        sc = syn.witness_like_scalars(self.group, self.n, 0x5EED0001 + sr) if cx.args.witness_like else syn.uniform_scalars(self.group, self.n, 0x5EED0001 + sr)
        if cx.args.balanced and self.group == "bls12_377_g1":
            i = np.arange(self.n, dtype=np.uint64)
            sc = np.zeros((self.n, 4), dtype=np.uint64)
            for w in range(16):
                d = ((i * np.uint64(2 * w + 1) + np.uint64(977 * w)) % np.uint64(32768)) + np.uint64(1)
                sc[:, w // 4] |= d << np.uint64(16 * (w % 4))
        self.sc = sc
        self.d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        self.O = ffi.GROUP_SHAPE[self.group][2]
        self.fold = WindowJoiner(cx, self.group) if self.by_windows else Folder(cx, self.group, self.O)
        self.acc_ms, self.tot_ms = [], []
        self.fixed = None
        if cx.args.fixed_base:
            # the prover's shape: the key's tables are built ONCE (outside the timed region, reported in the line), every step is a new
            # scalar vector against them (msm_*_fixed_dev)
            self.fixed = ffi.FixedBase(self.group, d_bases=self.bases.data_ptr(), n=self.n, window_bits=cx.args.fixed_window_bits)
        torch.cuda.synchronize()

    def setup_in_process(self):
        """Shard d of the job resident on device cx.devices[d] (a device may be listed more than once: that many engines on it)."""
        from celo_bls_snark_rs_amd import ffi, synthetic as syn
        cx = self.cx
        self.sh_bases, self.sh_sc, self.sc_host = [], [], []
        for r, d in enumerate(cx.devices):
            torch.cuda.set_device(d)
            ffi.use_device(d)
            if self.by_windows:
                r = 0                                          # one replica of the same n terms per device
            self.sh_bases.append(syn.device_points(self.group, self.n, 0x5EED0002 + 0x1000 * r, device="cuda:%d" % d))
            sc = syn.witness_like_scalars(self.group, self.n, 0x5EED0001 + r) if cx.args.witness_like else syn.uniform_scalars(self.group, self.n, 0x5EED0001 + r)
            self.sc_host.append(sc)
            self.sh_sc.append(torch.from_numpy(sc.view(np.int64)).to("cuda:%d" % d))
            torch.cuda.synchronize(d)
        torch.cuda.set_device(cx.devices[0])
        ffi.use_device(cx.devices[0])
        self.bases, self.d_sc, self.sc = self.sh_bases[0], self.sh_sc[0], self.sc_host[0]
        self.O = ffi.GROUP_SHAPE[self.group][2]
        self.acc_ms, self.tot_ms = [], []

    def step(self):
        from celo_bls_snark_rs_amd import ffi
        cx = self.cx
        sub = bool(cx.args.subgroup_points)
        if self.fixed is not None:
            return self.fold(self.fixed.msm_dev(self.d_sc.data_ptr(), self.n, self.cx.stream))
        if self.by_windows and cx.devices:
            return ffi.msm_multi_windows_dev(self.group, cx.devices, [b.data_ptr() for b in self.sh_bases], None, [s_.data_ptr() for s_ in self.sh_sc], self.n, subgroup=sub)
        if self.by_windows:
            rec, bit = ffi.msm_window_shard_dev(self.group, self.bases.data_ptr(), 0, self.d_sc.data_ptr(), self.n, cx.rank, cx.world, self.cx.stream, subgroup=sub)
            return self.fold(rec, bit)
        if cx.args.subgroup_points and not cx.devices:
            out = ffi.msm_dev(self.group, self.bases.data_ptr(), 0, self.d_sc.data_ptr(), self.n, self.cx.stream, subgroup=True)
            return self.fold(out)
        if cx.devices:
            return ffi.msm_multi_dev(self.group, cx.devices, [b.data_ptr() for b in self.sh_bases], None, [s_.data_ptr() for s_ in self.sh_sc],
                                     [self.n] * len(cx.devices))
        out = ffi.msm_dev(self.group, self.bases.data_ptr(), 0, self.d_sc.data_ptr(), self.n, self.cx.stream)
        return self.fold(out)

    def after_step(self):
        from celo_bls_snark_rs_amd import ffi
        tm = ffi.msm_timings(self.group)
        self.acc_ms.append(tm["accumulate_ms"]); self.tot_ms.append(tm["total_ms"])

    def units_per_step(self):
        return self.n if self.by_windows else self.cx.nshards * self.n

    def report(self, line, result):
        from celo_bls_snark_rs_amd import ffi, synthetic as syn
        cx = self.cx
        tm = ffi.msm_timings(self.group)
        acc = float(np.median(self.acc_ms))
        alg = syn.ALG_BYTES[self.group]
        # algorithmic bytes of ONE launch of the dominant kernel: a window shard reads its n points once per window it owns, i.e. its
        # share windows_here / windows_of_the_job of the job's n * alg bytes
        nw_job = tm["windows"] * (cx.nshards if self.by_windows else 1)
        achieved = self.n * alg / (cx.nshards if self.by_windows else 1) / (acc * 1e-3) / 1e9
        traffic, src = committed_traffic(ACC_KERNEL[self.group], self.log_n) if not self.by_windows else (None, "no committed PMC profile of this launch shape")
        line["metric"] = "%s MSM scalar-muls/sec" % {"bls12_377_g1": "BLS12-377 G1", "bw6_761_g1": "BW6-761 G1"}[self.group]
        line["unit"] = "scalar-muls/s"
        line["config"] = {"workload": "%s, 2^%d random bases/scalars per GPU%s, inputs resident in HBM" % (self.name, self.log_n, " (witness-like scalar mix)" if cx.args.witness_like else ""),
                          "bases_per_gpu": self.n, "window_bits": tm["window_bits"], "windows": tm["windows"], "buckets": tm["buckets"],
                          "partition": ("windows" if self.by_windows else "index") if cx.nshards > 1 else None,
                          "sharding": ("window ranges over %d replicas of the job (%s), %d-B partial sums joined with the doublings between the ranges (no data-path collective)"
                                       % (cx.nshards, "one process: msm_%s_multi_windows_dev" % self.group if cx.devices else
                                          "one rank per GPU: msm_*_window_shard_dev + one all_gather of the records + msm_*_join_windows", self.O * 8 * 4 // 3)) if self.by_windows else
                                      ("index-range shards resident on %d devices of ONE process (msm_%s_multi_dev: a host thread per device, host fold of %d-B partial sums)"
                                       % (len(cx.devices), self.group, self.O * 8)) if cx.devices else
                                      "index-range shards + all_gather of %d-B partial sums" % (self.O * 8) if cx.world > 1 else "single GPU"}
        if self.by_windows:
            line["config"]["workload"] = "%s, ONE job of 2^%d random bases/scalars replicated on every GPU, windows partitioned, inputs resident in HBM" % (self.name, self.log_n)
            line["config"]["windows_of_the_job"] = nw_job
        line["roofline"] = {"bound": "hbm", "kernel": ACC_KERNEL[self.group], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": src,
                            "note": "integer-VALU bound, not HBM bound (SURVEY.md section 8d); algorithmic bytes = n*%d B per launch; kernel ms (median over the timed "
                                    "steps) from HIP events on the MSM stream: accumulate=%.3f of total=%.3f (convert=%.3f sort=%.3f reduce=%.3f)"
                                    % (alg, acc, float(np.median(self.tot_ms)), tm["convert_ms"], tm["sort_ms"], tm["reduce_ms"])}
        # the honest roofline: integer-VALU issue, in multiply-adds.  One XYZZ mixed addition per (scalar, window); peak = the chip-wide rate of the
        # library's own product bodies in a register-resident loop, measured in this run (valu_peaks)
        madd = MADS["madd"][self.group]
        fq_ops = self.n * tm["windows"] * madd
        pk = valu_peaks()
        valu_peak = pk["tmads"][self.group]
        line["valu_roofline"] = {"bound": "integer VALU (v_mad_u64_u32 issue)", "kernel": ACC_KERNEL[self.group], "achieved": fq_ops / (acc * 1e-3) / 1e12,
                                 "peak": valu_peak, "unit": "T multiply-adds/s", "frac": fq_ops / (acc * 1e-3) / 1e12 / valu_peak,
                                 "multiply_adds_per_mixed_addition": madd, "peak_measured_in_run": pk["measured"],
                                 "note": pk["note"] + "; achieved = n*windows mixed additions x %d multiply-adds (6 products, 2 squarings, the one-pass Y3: bench.py MADS) / "
                                                      "accumulate time" % madd}
        if self.group in MADS["madd_executed"]:      # the named kernel executes more than the formula's count: priced are the USEFUL multiply-adds
            ex = MADS["madd_executed"][self.group]
            line["valu_roofline"]["multiply_adds_executed_per_mixed_addition"] = ex
            line["valu_roofline"]["frac_executed"] = line["valu_roofline"]["frac"] * ex / madd
            line["valu_roofline"]["note"] += "; %s executes %d per mixed addition (ten pair products): `frac` prices the useful %d of the one-lane formula, `frac_executed` the issued ones" % (ACC_KERNEL[self.group], ex, madd)
        if self.fixed is not None:
            fi = self.fixed.info()
            line["config"]["entry_point"] = "msm_%s_fixed_dev: per-key tables T[j][i] = 2^(c j) P_i built once by msm_%s_precompute_dev (the prover's queries stay, the assignment changes)" % (self.group, self.group)
            line["config"]["fixed_base"] = {"window_bits": fi["window_bits"], "digits_per_scalar": fi["windows"], "virtual_windows": tm["windows"],
                                            "table_bytes": fi["table_bytes"], "table_build_ms": fi["build_ms"],
                                            "note": "the table build is outside the timed region: once per proving key (crates/epoch-snark/src/api/setup.rs:63-105)"}
            fq_ops = self.n * fi["windows"] * madd
            line["valu_roofline"]["achieved"] = fq_ops / (acc * 1e-3) / 1e12
            line["valu_roofline"]["frac"] = line["valu_roofline"]["achieved"] / valu_peak
            line["valu_roofline"]["note"] += "; fixed base: n * digits_per_scalar mixed adds"
            line["roofline"]["traffic"], line["roofline"]["traffic_source"] = None, "no committed PMC profile of this launch shape"
        if cx.args.subgroup_points:
            line["config"]["entry_point"] = "msm_bls12_377_g1_subgroup_dev: bases vouched to lie in G1 (what Signature::batch hands over), GLV split"
        if cx.world > 1 and not cx.devices and self.fixed is None and not self.by_windows and cx.args.scaling == "weak" and not cx.args.subgroup_points and not cx.args.witness_like and not cx.args.balanced and self.n <= (1 << 21):
            line["strong"] = self.strong_block()
        if cx.world == 1 and not cx.devices and self.fixed is None and not cx.args.subgroup_points:
            line["host_pointer"] = self.host_pointer(result, line["ms_per_step"])
        if cx.world == 1 and not cx.devices and self.fixed is None:
            line["two_callers"] = self.two_callers()
            if self.group == "bls12_377_g1" and not cx.args.subgroup_points:
                line["subgroup_entry"] = self.subgroup_entry(result)
        if not cx.args.no_cpu_baseline:
            line["cpu_baseline"] = self.cpu_baseline(result)
            pw = (line["cpu_baseline"] or {}).get("parity_with_gpu")
            line["parity"] = {"checked": pw is True, "against": "the CPU port (oracle/cpu) on the whole job, affine results compared: " + str((line["cpu_baseline"] or {}).get("sample"))
                              if pw is True else str(pw)}
        else:
            line["parity"] = {"checked": False, "against": "--no-cpu-baseline"}

    def two_callers(self):
        """Secondary number (not `value`): the same MSM issued from TWO host threads at once (the library has no global lock: each call
        leases its own engine and stream), so one call's latency-bound sort / bucket-reduction / host epilogue overlaps the other's
        bucket accumulation.  Both results are checked against the sequential one.
        The pool hands a serial caller the same warm engine every time, so the second engine only comes into being when two calls
        overlap: the warm-up pass below therefore runs the two threads CONCURRENTLY (untimed) - in round 2 it ran them one after the
        other, the timed pass then paid for the second engine's arena (about 1 GB of hipMalloc) and the number depended on which
        thread drew the cold engine (2.15e8 on one box, 3.4e8 on another)."""
        from celo_bls_snark_rs_amd import ffi
        reps = max(4, self.cx.args.steps)
        ref = ffi.msm_dev(self.group, self.bases.data_ptr(), 0, self.d_sc.data_ptr(), self.n)
        outs = [None, None]
        lat = [[], []]

        def run(i, count, record):
            for _ in range(count):
                t0 = time.perf_counter()
                outs[i] = ffi.msm_dev(self.group, self.bases.data_ptr(), 0, self.d_sc.data_ptr(), self.n)
                if record:
                    lat[i].append((time.perf_counter() - t0) * 1e3)

        def both(count, record):
            th = [threading.Thread(target=run, args=(i, count, record)) for i in range(2)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            return time.perf_counter() - t0
        both(3, False)                                        # warm-up: two engines, their arenas and streams
        dt = both(reps, True)
        from celo_bls_snark_rs_amd import codec
        p = codec.Q377 if self.group.startswith("bls12_377") else codec.Q761
        ext = 2 if self.group == "bls12_377_g2" else 1
        aff = [codec.jacobian_to_affine(o, p, ext) for o in (ref, outs[0], outs[1])]   # the Jacobian representative depends on the schedule
        if not (aff[0] == aff[1] == aff[2]):
            raise SystemExit("PARITY FAILURE: concurrent MSM calls returned a different point")
        return {"value": 2 * reps * self.n / dt, "unit": "scalar-muls/s", "ms_per_msm": dt * 1e3 / (2 * reps),
                "per_thread_median_call_ms": [float(np.median(x)) for x in lat], "per_thread_max_call_ms": [float(np.max(x)) for x in lat],
                "note": "two host threads, %d MSMs each after a concurrent 3-call warm-up, engines and streams from the pool; results identical to the sequential call" % reps}

    def strong_block(self):
        """N > 1, weak line: the STRONG-scaling number beside it (one SCALE run yields both curves).  ONE job of this
        config's size - rank 0's terms, replicated on every rank - cut by the window partition: every rank runs the windows it owns
        (msm_*_window_shard_dev), one all_gather of the fixed-size records, the join on every rank (msm_*_join_windows).  Timed like the
        headline: barrier + synchronize on both sides, max over ranks; the joined point is compared with rank 0's own full MSM of the job."""
        from celo_bls_snark_rs_amd import ffi, synthetic as syn, codec
        cx = self.cx
        steps, warm = max(3, cx.args.steps), 2
        if cx.rank == 0:
            bases, d_sc = self.bases, self.d_sc
        else:
            bases = syn.device_points(self.group, self.n, 0x5EED0002)
            d_sc = torch.from_numpy(syn.uniform_scalars(self.group, self.n, 0x5EED0001).view(np.int64)).cuda()
        join = WindowJoiner(cx, self.group)

        def one():
            rec, bit = ffi.msm_window_shard_dev(self.group, bases.data_ptr(), 0, d_sc.data_ptr(), self.n, cx.rank, cx.world, cx.stream)
            return join(rec, bit)

        def barrier():
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        for _ in range(warm):
            out = one()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = one()
        barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cx.xdev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        ms = float(el.item()) * 1e3 / steps
        tm = ffi.msm_timings(self.group)
        ok = True
        if cx.rank == 0:
            p = codec.Q377 if self.group.startswith("bls12_377") else codec.Q761
            whole = ffi.msm_dev(self.group, self.bases.data_ptr(), 0, self.d_sc.data_ptr(), self.n)
            ok = codec.jacobian_to_affine(out, p, 1) == codec.jacobian_to_affine(whole, p, 1)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=cx.xdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            raise SystemExit("PARITY FAILURE: the window-partitioned job != the whole MSM on rank 0")
        return {"value": self.n / (ms * 1e-3), "unit": "scalar-muls/s", "ms_per_step": ms, "steps": steps, "scaling": "strong", "partition": "windows",
                "job": "ONE %s MSM of 2^%d terms (rank 0's), replicated on the %d ranks, windows partitioned" % (self.group, self.log_n, cx.world),
                "windows_here": tm["windows"], "shard_kernel_ms_rank0": {k: tm[k] for k in ("convert_ms", "sort_ms", "accumulate_ms", "reduce_ms", "total_ms")},
                "parity": {"checked": True, "against": "rank 0's msm_%s_dev of the whole job (same affine point)" % self.group}}

    def host_pointer(self, resident_result, resident_ms):
        """Secondary number (not `value`; SURVEY.md section 8d "report also with H2D included"): the SAME job through the host-pointer entry
        point msm_<group> - the call INTEGRATION.md's Rust wrapper makes with the slices bls-crypto hands to multi_scalar_mul
        (crates/bls-crypto/src/bls/signature.rs:82-85, public.rs:58-61) - on PAGEABLE numpy buffers: scalars and bases cross PCIe in index
        chunks, each sorted beside and accumulated behind the one before while the next is in flight (csrc/msm.h HostIn).  Reported beside the unpipelined
        form (three transfers, then the resident pipeline) and the bare transfer time of the same bytes."""
        from celo_bls_snark_rs_amd import ffi, codec
        reps = max(4, min(self.cx.args.steps, 20))
        A = ffi.GROUP_SHAPE[self.group][0]
        h_bases = self.bases.cpu().numpy().view(np.uint64).reshape(self.n, A).copy()      # pageable host memory, as a caller's Vec is
        h_sc = np.ascontiguousarray(self.sc).copy()
        p = codec.Q377 if self.group.startswith("bls12_377") else codec.Q761
        ext = 2 if self.group == "bls12_377_g2" else 1
        want = codec.jacobian_to_affine(resident_result, p, ext)

        def timed(chunks):
            ffi.set_host_chunks(chunks)
            try:
                t0 = time.perf_counter()
                out = ffi.msm(self.group, h_bases, None, h_sc)
                first = (time.perf_counter() - t0) * 1e3
                ffi.msm(self.group, h_bases, None, h_sc)
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    out = ffi.msm(self.group, h_bases, None, h_sc)
                    ts.append((time.perf_counter() - t0) * 1e3)
                tm = ffi.msm_timings(self.group)
            finally:
                ffi.set_host_chunks(-1)
            if codec.jacobian_to_affine(out, p, ext) != want:
                raise SystemExit("PARITY FAILURE: msm_%s (host pointers, chunks=%d) != the resident entry point" % (self.group, chunks))
            return first, float(np.median(ts)), tm
        first_p, ms_p, tm_p = timed(-1)
        _, ms_u, _ = timed(0)
        # ... and on buffers the HIP runtime has not seen before (a caller that builds new Vecs for every MSM): the runtime pins a pageable range
        # the first time it copies from it, so a never-seen buffer costs more than a reused one; the copies are made outside the timing
        fr = []
        for _ in range(min(reps, 6)):
            fb, fs = h_bases.copy(), h_sc.copy()
            t0 = time.perf_counter()
            out = ffi.msm(self.group, fb, None, fs)
            fr.append((time.perf_counter() - t0) * 1e3)
            del fb, fs
        if codec.jacobian_to_affine(out, p, ext) != want:
            raise SystemExit("PARITY FAILURE: msm_%s (host pointers, fresh buffers) != the resident entry point" % self.group)
        ms_f = float(np.median(fr))
        # ... and from page-locked buffers of the library's own allocator (celo_amd_host_alloc: what a wrapper that builds the limb arrays anyway
        # would write them into): no pinning by the runtime, transfers that do not hold the calling thread
        pb, ps = ffi.PinnedArray(h_bases.shape, np.uint64), ffi.PinnedArray(h_sc.shape, np.uint64)
        try:
            pb.a[...] = h_bases; ps.a[...] = h_sc
            ffi.msm(self.group, pb.a, None, ps.a)
            pn = []
            for _ in range(reps):
                t0 = time.perf_counter()
                out = ffi.msm(self.group, pb.a, None, ps.a)
                pn.append((time.perf_counter() - t0) * 1e3)
        finally:
            pb.close(); ps.close()
        if codec.jacobian_to_affine(out, p, ext) != want:
            raise SystemExit("PARITY FAILURE: msm_%s (host pointers, page-locked buffers) != the resident entry point" % self.group)
        ms_pin = float(np.median(pn))
        # the bare transfer of the same bytes from the same pageable buffers
        d_b = torch.empty_like(self.bases); d_s = torch.empty_like(self.d_sc)
        tb, tsc = torch.from_numpy(h_bases.view(np.int64).reshape(-1)), torch.from_numpy(h_sc.view(np.int64))
        h2d = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d_s.copy_(tsc.reshape(d_s.shape)); d_b.copy_(tb.reshape(d_b.shape))
            torch.cuda.synchronize()
            h2d.append((time.perf_counter() - t0) * 1e3)
        h2d_ms = float(np.median(h2d[1:]))
        nbytes = h_bases.nbytes + h_sc.nbytes
        return {"value": self.n / (ms_p * 1e-3), "unit": "scalar-muls/s", "wall_ms": ms_p, "first_call_ms": first_p,
                "resident_ms": resident_ms, "ratio_to_resident": ms_p / resident_ms,
                "fresh_buffers_wall_ms": ms_f, "fresh_buffers_ratio_to_resident": ms_f / resident_ms,
                "pinned_buffers_wall_ms": ms_pin, "pinned_buffers_ratio_to_resident": ms_pin / resident_ms,
                "unpipelined_wall_ms": ms_u, "h2d_only_ms": h2d_ms, "h2d_GBps": nbytes / (h2d_ms * 1e-3) / 1e9, "h2d_share_of_wall": h2d_ms / ms_p,
                "bytes": nbytes, "chunks": "default (CELO_HOST_CHUNKS, else n / 2^18 within [4, 16]; BW6-761 [8, 16]; G2 n / 2^19 within [4, 8]), the first halved once", "parity_with_resident": True,
                "kernel_ms": {k: tm_p[k] for k in ("convert_ms", "sort_ms", "accumulate_ms", "reduce_ms", "total_ms")},
                "note": "entry point msm_%s on pageable numpy buffers, wall clock per call (median of %d after 2 warm calls; first_call_ms = the first call, "
                        "which also sizes the engine's staging buffers and takes the driver's first-touch of the pages; fresh_buffers_wall_ms = every call on newly "
                        "allocated copies of the inputs, wall_ms = the same buffers call after call, pinned_buffers_wall_ms = buffers from celo_amd_host_alloc); kernel_ms.convert = the first chunk's scalars, "
                        "digits and sort, .accumulate = from there to the last chunk's end (the transfers hide here)" % (self.group, reps)}

    def subgroup_entry(self, plain_result):
        """Secondary number (not `value`): the same job through msm_bls12_377_g1_subgroup_dev - the entry point for bases that are elements
        of the prime-order group G1 (every Signature of the reference is one: crates/bls-crypto/src/bls/signature.rs:31-57, 70-89), where
        the library may split the scalars with the GLV endomorphism.  Same affine result as the plain entry point."""
        from celo_bls_snark_rs_amd import ffi, codec
        reps = max(4, self.cx.args.steps)
        out = None
        for _ in range(2):
            out = ffi.msm_dev(self.group, self.bases.data_ptr(), 0, self.d_sc.data_ptr(), self.n, self.cx.stream, subgroup=True)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = ffi.msm_dev(self.group, self.bases.data_ptr(), 0, self.d_sc.data_ptr(), self.n, self.cx.stream, subgroup=True)
            ts.append(time.perf_counter() - t0)
        tm = ffi.msm_timings(self.group)
        if codec.jacobian_to_affine(out, codec.Q377, 1) != codec.jacobian_to_affine(plain_result, codec.Q377, 1):
            raise SystemExit("PARITY FAILURE: msm_bls12_377_g1_subgroup_dev != msm_bls12_377_g1_dev")
        ms = float(np.median(ts)) * 1e3
        return {"value": self.n / (ms * 1e-3), "unit": "scalar-muls/s", "ms_per_msm": ms, "windows": tm["windows"], "window_bits": tm["window_bits"],
                "kernel_ms": {k: tm[k] for k in ("convert_ms", "sort_ms", "accumulate_ms", "reduce_ms", "total_ms")},
                "note": "GLV split k = k0 + k1 x^2, [x^2]P = (beta x, -y): 2n terms of 127 bits, half the windows; same affine result as `value`'s entry point"}

    def cpu_baseline(self, gpu_result):
        """Full-size parity of the timed result (every rank's inputs gathered on rank 0), then bounded timings of the port."""
        from oracle import cpu_oracle as co
        cx = self.cx
        A = self.bases.numel() // self.n
        if self.by_windows:                                    # one job, replicated: rank 0's copy IS the job
            h_b, h_s = self.bases.view(self.n, A).cpu().numpy().view(np.uint64), self.d_sc.view(self.n, -1).cpu().numpy().view(np.uint64)
        elif cx.devices:
            h_b = np.concatenate([b.view(self.n, A).cpu().numpy().view(np.uint64) for b in self.sh_bases])
            h_s = np.concatenate([s_.view(self.n, -1).cpu().numpy().view(np.uint64) for s_ in self.sh_sc])
        else:
            n_all = cx.world * self.n
            if cx.world > 1 and self.fixed is None and not (n_all <= (1 << 21) if self.group == "bls12_377_g1" else n_all <= (1 << 19)):
                return self.cpu_baseline_distributed(gpu_result)
            h_b, h_s = gather_to_rank0(cx, self.bases.view(self.n, A)), gather_to_rank0(cx, self.d_sc.view(self.n, -1))
        if cx.rank != 0:
            return None
        hw = co.lib().orc_hardware_threads()
        n_all = h_b.shape[0]
        bits = 253 if self.group == "bls12_377_g1" else 377
        lg = (n_all - 1).bit_length()
        c = 3 if n_all < 32 else (lg * 69) // 100 + 2
        windows = (bits + c - 1) // c
        T = max(1, min(hw, windows))
        full = n_all <= (1 << 23) if self.group == "bls12_377_g1" else n_all <= (1 << 19)
        res = {}
        if full:
            t0 = time.perf_counter()
            out = co.msm(self.group, h_b, None, h_s, threads=T)
            secs = time.perf_counter() - t0
            ok = co.jac_to_affine(out, KIND[self.group]) == co.jac_to_affine(gpu_result, KIND[self.group])
            if not ok:
                raise SystemExit("PARITY FAILURE: GPU MSM result != CPU oracle result at full size")
            res = {"value": n_all / secs, "seconds": secs, "parity_with_gpu": True,
                   "sample": "the full %d-term job once (all %d shard(s)), arkworks windowing c=%d (%d windows), one thread per window like rayon" % (n_all, cx.nshards, c, windows)}
        elif (cx.world == 1 or self.by_windows) and self.fixed is None and n_all >= (1 << 20):
            # one GPU, a job too large for ONE oracle call in bounded time (cfg4 as named: 2^24 BW6-761 terms): the port over index ranges side by
            # side on the host's cores, the partial points added with the big-integer group law - full-size parity in n / (cores' rate) seconds
            from oracle.py import ecc
            chunks = max(1, min(16, hw // T))
            parts = [None] * chunks
            def run_part(i):
                lo, hi = n_all * i // chunks, n_all * (i + 1) // chunks
                parts[i] = co.jac_to_affine(co.msm(self.group, h_b[lo:hi], None, h_s[lo:hi], threads=T), KIND[self.group])
            th = [threading.Thread(target=run_part, args=(i,)) for i in range(chunks)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            secs = time.perf_counter() - t0
            E = ecc.E1_377 if self.group == "bls12_377_g1" else ecc.E1_761
            tot = None
            for P in parts:
                tot = E.add(tot, P)
            if tot != co.jac_to_affine(gpu_result, KIND[self.group]):
                raise SystemExit("PARITY FAILURE: GPU MSM result != CPU oracle result at full size (sum over %d index ranges)" % chunks)
            res = {"value": n_all / secs, "seconds": secs, "parity_with_gpu": True,
                   "sample": "the full %d-term job cut into %d index ranges run side by side (%d threads each, arkworks windowing c=%d), partial points added with the "
                             "big-integer group law" % (n_all, chunks, T, c)}
            T = chunks * T
        else:                                                   # bounded: time a 2^18 sample; parity by linearity on the sample (a fresh GPU call)
            from celo_bls_snark_rs_amd import ffi
            k = 1 << 18
            t0 = time.perf_counter()
            out = co.msm(self.group, h_b[:k], None, h_s[:k], threads=T)
            secs = time.perf_counter() - t0
            got = (self.fixed.msm_dev(self.d_sc.data_ptr(), k, cx.stream) if self.fixed is not None else
                   ffi.msm_dev(self.group, self.bases.data_ptr(), 0, self.d_sc.data_ptr(), k, cx.stream))
            if self.fixed is not None:       # and the timed full-size result against the variable-base entry point on the same inputs
                vb = ffi.msm_dev(self.group, self.bases.data_ptr(), 0, self.d_sc.data_ptr(), self.n, cx.stream)
                if cx.world == 1 and co.jac_to_affine(vb, KIND[self.group]) != co.jac_to_affine(gpu_result, KIND[self.group]):
                    raise SystemExit("PARITY FAILURE: msm_*_fixed_dev != msm_*_dev at full size")
            if co.jac_to_affine(out, KIND[self.group]) != co.jac_to_affine(got, KIND[self.group]):
                raise SystemExit("PARITY FAILURE: GPU MSM result != CPU oracle result on the 2^18 sample")
            res = {"value": k / secs, "seconds": secs, "parity_with_gpu": "2^18 sample of rank 0's shard (full size: tests/test_configs_gpu.py)",
                   "sample": "first 2^18 terms of rank 0's shard, arkworks windowing (%d threads, one per window)" % T}
        res.update({"unit": "scalar-muls/s", "cores": T, "kind": "port", "hardware_threads": hw,
                    "note": "C++ restatement of ark-ec VariableBaseMSM (not the Rust binary: no Rust toolchain); rayon parallelism in arkworks is per window, so "
                            "threads beyond the window count do not help ONE msm"})
        k1 = 1 << 15                                           # one core
        t0 = time.perf_counter()
        co.msm(self.group, h_b[:k1], None, h_s[:k1], threads=1)
        res["one_thread"] = {"value": k1 / (time.perf_counter() - t0), "sample": "2^15 terms, 1 thread"}
        chunks = max(1, hw // T)                               # every core: the job cut into hw/windows chunks, each with one thread per window
        if chunks > 1 and full:
            parts = [None] * chunks
            def run(i):
                lo, hi = n_all * i // chunks, n_all * (i + 1) // chunks
                parts[i] = co.msm(self.group, h_b[lo:hi], None, h_s[lo:hi], threads=T)
            th = [threading.Thread(target=run, args=(i,)) for i in range(chunks)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            res["all_cores"] = {"value": n_all / (time.perf_counter() - t0), "threads": chunks * T,
                                "sample": "the same job cut into %d index ranges run side by side (not how the reference calls arkworks)" % chunks}
        return res


def _cpu_baseline_distributed(self, gpu_result):
    """N > 1 ranks, index-range shards too large for one oracle run on rank 0: FULL-SIZE parity with the ranks' host cores side by side.
    Every rank runs the CPU port on its OWN shard (hardware threads / ranks each) and compares it with its own GPU partial sum; the CPU
    partials are all-gathered and rank 0 adds them with the oracle's big-integer group law and compares the sum with the timed, folded
    GPU result.  Any mismatch on any rank ends every rank."""
    from oracle import cpu_oracle as co
    from oracle.py import ecc
    from celo_bls_snark_rs_amd import ffi
    cx = self.cx
    A = self.bases.numel() // self.n
    hw = co.lib().orc_hardware_threads()
    bits = 253 if self.group == "bls12_377_g1" else 377
    lg = (self.n - 1).bit_length()
    c = 3 if self.n < 32 else (lg * 69) // 100 + 2
    windows = (bits + c - 1) // c
    T = max(1, min(windows, hw // cx.world))
    h_b = self.bases.view(self.n, A).cpu().numpy().view(np.uint64)
    h_s = self.d_sc.view(self.n, -1).cpu().numpy().view(np.uint64)
    dist.barrier()
    t0 = time.perf_counter()
    mine = co.msm(self.group, h_b, None, h_s, threads=T)
    secs = time.perf_counter() - t0
    own = ffi.msm_dev(self.group, self.bases.data_ptr(), 0, self.d_sc.data_ptr(), self.n, cx.stream)
    kind = KIND[self.group]
    ok = co.jac_to_affine(mine, kind) == co.jac_to_affine(own, kind)
    t_all = torch.tensor([secs], dtype=torch.float64, device=cx.xdev)
    dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
    part = torch.from_numpy(np.ascontiguousarray(mine).view(np.int64).reshape(-1)).to(cx.xdev)
    allp = torch.empty(cx.world * part.numel(), dtype=torch.int64, device=cx.xdev)
    dist.all_gather_into_tensor(allp, part)
    if cx.rank == 0 and ok:
        E = ecc.E1_377 if self.group == "bls12_377_g1" else ecc.E1_761
        tot = None
        for r in range(cx.world):
            tot = E.add(tot, co.jac_to_affine(allp.cpu().numpy().view(np.uint64).reshape(cx.world, -1)[r], kind))
        ok = tot == co.jac_to_affine(gpu_result, kind)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=cx.xdev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        raise SystemExit("PARITY FAILURE: a rank's GPU partial sum != the CPU port on its shard, or the folded GPU result != the sum of the CPU partials")
    if cx.rank != 0:
        return None
    secs = float(t_all.item())
    res = {"value": cx.world * self.n / secs, "seconds": secs, "parity_with_gpu": True, "unit": "scalar-muls/s", "cores": T * cx.world, "kind": "port", "hardware_threads": hw,
           "sample": "the full job: every rank ran the port on its own %d-term shard with %d threads at the same time (arkworks windowing c=%d, %d windows); each shard's "
                     "partial sum compared with that rank's GPU partial, their big-integer sum with the folded result" % (self.n, T, c, windows),
           "note": "C++ restatement of ark-ec VariableBaseMSM (not the Rust binary: no Rust toolchain); %d x %d threads is not how ONE arkworks call parallelises "
                   "(rayon: one task per window) - it is the box's cores applied to the sharded job" % (cx.world, T)}
    k1 = 1 << 15
    t0 = time.perf_counter()
    co.msm(self.group, h_b[:k1], None, h_s[:k1], threads=1)
    res["one_thread"] = {"value": k1 / (time.perf_counter() - t0), "sample": "2^15 terms, 1 thread"}
    return res


MsmConfig.cpu_baseline_distributed = _cpu_baseline_distributed


class Folder:
    """all_gather of the per-rank Jacobian partial sums + local fold (EC addition is not an RCCL reduction op)."""
    def __init__(self, cx, group, words):
        self.cx, self.group = cx, group
        if cx.world > 1:
            self.mine = torch.empty(words, dtype=torch.int64, device=cx.xdev)
            self.all = torch.empty(cx.world * words, dtype=torch.int64, device=cx.xdev)
            self.host = torch.empty(cx.world * words, dtype=torch.int64).pin_memory() if cx.xdev == "cuda" else None

    def __call__(self, out):
        from celo_bls_snark_rs_amd import ffi
        cx = self.cx
        if cx.world == 1:
            return out
        self.mine.copy_(torch.from_numpy(out.view(np.int64)), non_blocking=True)
        dist.all_gather_into_tensor(self.all, self.mine)
        if self.host is not None:
            self.host.copy_(self.all, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            parts = self.host.numpy().view(np.uint64)
        else:
            parts = self.all.numpy().view(np.uint64)
        return ffi.sum_jacobian(self.group, parts.reshape(cx.world, -1))


class WindowJoiner:
    """Window partition, one rank per GPU: all_gather of the ranks' fixed-size records (X || Y || ZZ || ZZZ + the range's first bit) and the
    join total = sum_g 2^bit_g P_g on every rank (msm_*_join_windows: host arithmetic, the doublings between the ranges)."""
    def __init__(self, cx, group):
        from celo_bls_snark_rs_amd import ffi
        self.cx, self.group = cx, group
        self.words = 2 * ffi.GROUP_SHAPE[group][0] + 1         # the record + bit_lo
        self.mine = torch.empty(self.words, dtype=torch.int64, device=cx.xdev)
        self.all = torch.empty(cx.world * self.words, dtype=torch.int64, device=cx.xdev)
        self.host = torch.empty(cx.world * self.words, dtype=torch.int64).pin_memory() if cx.xdev == "cuda" else None
        self.stage = torch.empty(self.words, dtype=torch.int64).pin_memory() if cx.xdev == "cuda" else torch.empty(self.words, dtype=torch.int64)

    def __call__(self, rec, bit):
        from celo_bls_snark_rs_amd import ffi
        cx = self.cx
        st = self.stage.numpy()
        st[:-1] = rec.view(np.int64); st[-1] = bit
        self.mine.copy_(self.stage, non_blocking=True)
        dist.all_gather_into_tensor(self.all, self.mine)
        if self.host is not None:
            self.host.copy_(self.all, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            parts = self.host.numpy()
        else:
            parts = self.all.numpy()
        parts = parts.reshape(cx.world, self.words)
        return ffi.join_windows(self.group, np.ascontiguousarray(parts[:, :-1]).view(np.uint64), [int(b) for b in parts[:, -1]])


def gather_to_rank0(cx, t):
    """Rows of every rank's tensor concatenated on rank 0 as a numpy uint64 array (one-off, outside the timed region)."""
    if cx.world == 1:
        return t.cpu().numpy().view(np.uint64)
    t = t.contiguous()
    if cx.xdev == "cpu":
        t = t.cpu()
    bufs = [torch.empty_like(t) for _ in range(cx.world)] if cx.rank == 0 else None
    dist.gather(t, bufs, dst=0)
    if cx.rank != 0:
        return None
    return np.concatenate([b.cpu().numpy().view(np.uint64) for b in bufs])


# ===================================================================================================== config 3
class BatchVerifyConfig:
    def __init__(self, cx):
        self.cx = cx
        m_total = cx.args.batches
        self.m = m_total // cx.world if cx.args.scaling == "strong" else m_total
        self.n = cx.args.signers

    def setup(self):
        from celo_bls_snark_rs_amd import synthetic as syn
        cx = self.cx
        rng = np.random.default_rng(0x5EED0030 + cx.rank)
        self.corrupt = np.unique(rng.choice(self.m, size=max(1, self.m // 100), replace=False))     # 1 % of the batches
        self.w = syn.valid_batches(self.m, self.n, 0x5EED0300 + 0x100 * cx.rank, self.corrupt)
        self.ex = syn.batch_exponents(self.m * self.n, 0x5EED0301 + cx.rank)
        self.d_ex = torch.from_numpy(self.ex.view(np.int64)).cuda()
        self.ng2 = syn.neg_g2_limbs()
        self.dev_ms = []
        torch.cuda.synchronize()

    def step(self):
        from celo_bls_snark_rs_amd import ffi
        w = self.w
        return ffi.batch_verify_dev(w["pk"].data_ptr(), w["sig"].data_ptr(), self.d_ex.data_ptr(), w["offsets"], w["hash"].data_ptr(), self.ng2)

    def after_step(self):
        from celo_bls_snark_rs_amd import ffi
        g2, g1, pr = ffi.msm_timings("bls12_377_g2"), ffi.msm_timings("bls12_377_g1"), ffi.pairing_timings()
        self.dev_ms.append((g2["total_ms"], g2["accumulate_ms"], g1["total_ms"], pr["total_ms"], pr["miller_ms"], pr["final_exp_ms"]))

    def units_per_step(self):
        return self.cx.world * self.m

    def report(self, line, result):
        from celo_bls_snark_rs_amd import synthetic as syn
        cx = self.cx
        if result.tolist() != self.w["expect"].tolist():
            raise SystemExit("PARITY FAILURE: accept vector of the timed step differs from the constructed one")
        d = np.median(np.array(self.dev_ms), axis=0)
        tot = self.m * self.n
        achieved = tot * syn.ALG_BYTES["bls12_377_g2"] / (d[1] * 1e-3) / 1e9
        line["metric"] = "Batch::verify batches/sec (BLS12-377: G2 MSM + G1 MSM + 2-pair product per batch)"
        line["unit"] = "batches/s"
        line["config"] = {"workload": "batched aggregated-signature verify: %d batches x %d signers per GPU, 136-bit exponents, 1 %% of the batches corrupted, inputs resident in HBM"
                                      % (self.m, self.n), "batches_per_gpu": self.m, "signers_per_batch": self.n,
                          "signatures_per_s": None, "miller_loops_per_step": 2 * self.m, "final_exps_per_step": self.m,
                          "sharding": "batches sharded by rank, no exchange" if cx.world > 1 else "single GPU"}
        traffic, src = committed_traffic("k_accumulate_pair<G2_377, 1>", "cfg3") if (self.m, self.n) == (4096, 256) else (None, "no committed PMC profile of this launch shape")
        line["roofline"] = {"bound": "hbm", "kernel": "k_accumulate_pair<G2_377, 1> (batched path)", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": src,
                            "note": "integer-VALU bound; algorithmic bytes = signers*224 B per launch; median HIP-event ms: G2 MSM %.2f (accumulate %.2f), G1 MSM %.2f, "
                                    "pairings %.2f (Miller %.2f, final exp %.2f) - the two MSMs overlap on the GPU, so their event times include each other's work"
                                    % (d[0], d[1], d[2], d[3], d[4], d[5])}
        from celo_bls_snark_rs_amd import ffi
        g2t = ffi.msm_timings("bls12_377_g2")
        pk = valu_peaks()
        # the batched G2 accumulation: every (term, window) of the (GLS-expanded) instances is one mixed addition over Fq2
        terms = tot * (3 if g2t["windows"] * g2t["window_bits"] < 100 else 1)     # psi-split (csrc/msm.h gls_digits): three 64-bit digits per 136-bit exponent
        fq2_ops = terms * g2t["windows"] * MADS["madd"]["bls12_377_g2"]
        line["valu_roofline"] = {"bound": "integer VALU (v_mad_u64_u32 issue)", "kernel": "k_accumulate_pair<G2_377, 1> (batched path)", "achieved": fq2_ops / (d[1] * 1e-3) / 1e12,
                                 "peak": pk["tmads"]["bls12_377_g2"], "unit": "T multiply-adds/s", "frac": fq2_ops / (d[1] * 1e-3) / 1e12 / pk["tmads"]["bls12_377_g2"],
                                 "multiply_adds_per_mixed_addition": MADS["madd"]["bls12_377_g2"],
                                 "peak_measured_in_run": pk["measured"], "windows": g2t["windows"], "window_bits": g2t["window_bits"], "expanded_terms": terms,
                                 "note": pk["note"] + "; achieved = expanded terms x windows mixed additions over Fq2 x %d multiply-adds / accumulate ms - an upper count (a zero digit "
                                         "adds nothing: 1/64 of them at 6-bit windows) over a time that also holds part of the overlapping G1 leg's work; the count is curve.h's one-lane "
                                         "formula (6 products, 2 squarings, the fused Y3) - the lane-pair kernel that runs spends 11 480 (ten pair products): its useful share is what is priced" % MADS["madd"]["bls12_377_g2"]}
        line["config"]["signatures_per_s"] = line["value"] * self.n
        line["parity"] = {"checked": True, "against": "this rank's accept vector of the timed step == the one built into the workload (1 % of the batches corrupted)"}
        if not cx.args.no_cpu_baseline and cx.rank == 0:
            line["cpu_baseline"] = self.cpu_baseline(result)
            line["parity"]["against"] += "; a sample of batches incl. rejected ones against the CPU port"

    def cpu_baseline(self, result):
        from oracle import cpu_oracle as co
        w, n = self.w, self.n
        k = min(16, self.m)
        idx = sorted(set(list(range(k - 2)) + [int(c) for c in self.corrupt[:2]]))   # a sample that holds rejected batches
        pk = w["pk"].view(self.m * n, 24).cpu().numpy().view(np.uint64)
        sg = w["sig"].view(self.m * n, 12).cpu().numpy().view(np.uint64)
        hh = w["hash"].view(self.m, 12).cpu().numpy().view(np.uint64)
        t0 = time.perf_counter()
        want = []
        for b in idx:
            sl = slice(b * n, (b + 1) * n)
            P = co.msm("bls12_377_g2", pk[sl], None, self.ex[sl], threads=1)
            S = co.msm("bls12_377_g1", sg[sl], None, self.ex[sl], threads=1)
            Pa, ip = co.normalize("g2_377", P.reshape(1, 36))
            Sa, isg = co.normalize("g1_377", S.reshape(1, 18))
            g1 = np.stack([Sa[0], hh[b]]); g2 = np.stack([self.ng2, Pa[0]])
            want.append(int(co.pairing_product_377(g1, np.array([isg[0], 0], dtype=np.uint8), g2, np.array([0, ip[0]], dtype=np.uint8))[1]))
        secs = time.perf_counter() - t0
        if [int(result[b]) for b in idx] != want:
            raise SystemExit("PARITY FAILURE: GPU verdicts != oracle verdicts on the sample")
        return {"value": len(idx) / secs, "unit": "batches/s", "cores": 1, "kind": "port", "parity_with_gpu": True,
                "sample": "%d of the batches (two of them corrupted), Batch::verify restated on one core: two %d-term MSMs with arkworks windowing + one 2-pair product" % (len(idx), n)}


def verify_shaped_products(mprod, seed):
    """mprod DISTINCT two-pair products e(sig_b, -g2) * e(H_b, pk_b), every 97th with a foreign signature (synthetic.verify_products; rounds 2-5
    tiled eight triples - VERDICT r5 item 4).  Returns (g1 (2 mprod, 12), g2 (2 mprod, 24), offsets, expected accept list)."""
    from celo_bls_snark_rs_amd import synthetic as syn
    return syn.verify_products(mprod, seed)


# ===================================================================================================== config 5
class MixedConfig:
    def __init__(self, cx):
        self.cx = cx
        a = cx.args
        div = cx.world if a.scaling == "strong" else 1
        self.n = (1 << (a.log_n or 22)) // div
        self.loops = (1 << 14) // div

    def setup(self):
        from celo_bls_snark_rs_amd import ffi, synthetic as syn
        cx = self.cx
        r = cx.rank
        self.b1 = syn.device_points("bls12_377_g1", self.n, 0x5EED0500 + 0x10 * r)
        self.b2 = syn.device_points("bls12_377_g2", self.n, 0x5EED0501 + 0x10 * r)
        self.s1 = syn.uniform_scalars("bls12_377_g1", self.n, 0x5EED0502 + r)
        self.s2 = syn.uniform_scalars("bls12_377_g2", self.n, 0x5EED0503 + r)
        self.d1 = torch.from_numpy(self.s1.view(np.int64)).cuda()
        self.d2 = torch.from_numpy(self.s2.view(np.int64)).cuda()
        self.mprod = self.loops // 2
        self.g1, self.g2, self.offs, self.expect = verify_shaped_products(self.mprod, 0x5EED0504)
        self.fold1, self.fold2 = Folder(cx, "bls12_377_g1", 18), Folder(cx, "bls12_377_g2", 36)
        self.ms = []
        torch.cuda.synchronize()

    def _legs(self):
        from celo_bls_snark_rs_amd import ffi
        return [lambda: ffi.msm_dev("bls12_377_g1", self.b1.data_ptr(), 0, self.d1.data_ptr(), self.n),
                lambda: ffi.msm_dev("bls12_377_g2", self.b2.data_ptr(), 0, self.d2.data_ptr(), self.n),
                lambda: ffi.pairing_product_is_one_batch(self.g1, None, self.g2, None, self.offs)]

    def step(self, concurrent=True):
        legs = self._legs()
        res = [None] * 3
        if concurrent:
            def run(i):
                res[i] = legs[i]()
            th = [threading.Thread(target=run, args=(i,)) for i in range(3)]
            for t in th: t.start()
            for t in th: t.join()
        else:
            res = [f() for f in legs]
        return [self.fold1(res[0]), self.fold2(res[1]), res[2]]

    def after_step(self):
        from celo_bls_snark_rs_amd import ffi
        self.ms.append((ffi.msm_timings("bls12_377_g1")["total_ms"], ffi.msm_timings("bls12_377_g2")["total_ms"], ffi.pairing_timings()["total_ms"]))

    def units_per_step(self):
        return self.cx.world * 2 * self.n

    def report(self, line, result):
        from celo_bls_snark_rs_amd import synthetic as syn
        cx = self.cx
        if result[2].tolist() != self.expect:
            raise SystemExit("PARITY FAILURE: pairing accept vector differs from the constructed one")
        # overlap gain: the same three legs back to back
        t0 = time.perf_counter()
        for _ in range(3):
            seq = self.step(concurrent=False)
        t_seq = (time.perf_counter() - t0) / 3 * 1e3
        d = np.median(np.array(self.ms), axis=0)
        bytes_step = self.n * (syn.ALG_BYTES["bls12_377_g1"] + syn.ALG_BYTES["bls12_377_g2"]) + self.loops * syn.ALG_BYTES_PER_MILLER_LOOP
        achieved = bytes_step / (line["ms_per_step"] * 1e-3) / 1e9
        line["metric"] = "BLS12-377 mixed G1+G2 MSM scalar-muls/sec with concurrent Miller loops"
        line["unit"] = "scalar-muls/s"
        line["config"] = {"workload": "mixed: G1 MSM 2^%d + G2 MSM 2^%d + %d Miller loops (%d two-pair products) per GPU, issued concurrently from three host threads, "
                                      "MSM inputs resident in HBM" % ((self.n - 1).bit_length(), (self.n - 1).bit_length(), self.loops, self.mprod),
                          "terms_per_group_per_gpu": self.n, "miller_loops_per_gpu": self.loops, "miller_loops_per_s": cx.world * self.loops / (line["ms_per_step"] * 1e-3),
                          "sequential_ms_per_step": t_seq, "overlap_gain": t_seq / line["ms_per_step"]}
        traffic, src = committed_traffic(["k_accumulate<G1_377>", "k_accumulate_pair<G2_377, 1>"], (self.n - 1).bit_length())
        line["roofline"] = {"bound": "hbm", "kernel": "whole step (three concurrent legs)", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": src,
                            "traffic_kernels": "k_accumulate<G1_377> + k_accumulate_pair<G2_377, 1> (the two dominant kernels of the step)",
                            "note": "BASELINE config 5 asks for the HBM-roofline fraction of the mixed job: algorithmic bytes (128 B per G1 term, 224 B per G2 term, 288 B per "
                                    "Miller loop) / step wall time; integer-VALU bound.  Median HIP-event ms of the legs while overlapped: G1 MSM %.2f, G2 MSM %.2f, pairings %.2f"
                                    % (d[0], d[1], d[2])}
        # the multiplier roofline of the mixed step: useful multiply-adds of the three legs / step wall time / the in-run peak
        from celo_bls_snark_rs_amd import ffi
        t1, t2 = ffi.msm_timings("bls12_377_g1"), ffi.msm_timings("bls12_377_g2")
        pk = valu_peaks()
        legs_mads = {"g1_msm": self.n * t1["windows"] * MADS["madd"]["bls12_377_g1"], "g2_msm": self.n * t2["windows"] * MADS["madd"]["bls12_377_g2"],
                     "pairings": self.mprod * (1157 + 900) * MADS["hex_round"]}
        tot_mads = float(sum(legs_mads.values()))
        peak = pk["tmads"]["bls12_377_g1"]
        line["valu_roofline"] = {"bound": "integer VALU (v_mad_u64_u32 / v_mad_i64_i32 issue)", "kernel": "whole step (three concurrent legs)", "unit": "T multiply-adds/s",
                                 "achieved": tot_mads / (line["ms_per_step"] * 1e-3) / 1e12, "peak": peak, "frac": tot_mads / (line["ms_per_step"] * 1e-3) / 1e12 / peak,
                                 "multiply_adds_per_step": legs_mads, "peak_measured_in_run": pk["measured"],
                                 "note": pk["note"] + "; achieved = (n x windows mixed additions x 3416 [G1] + n x windows x 10360 [G2, the one-lane formula's useful count] + "
                                         "products x (1157 + 900) rounds x 3444 [pairings]) / step wall time: sort, reduction, host epilogues and the non-multiply share of the pairing "
                                         "rounds are what the fraction below 1 is made of"}
        if cx.world == 1 and not cx.devices and self.n >= (1 << 20):
            line["host_pointer_g2"] = self.host_pointer_g2()
        if not cx.args.no_cpu_baseline:
            cb = self.cpu_baseline(result, seq)
            if cx.rank == 0:
                line["cpu_baseline"] = cb

    def host_pointer_g2(self):
        """Secondary number: msm_bls12_377_g2 at 2^20 terms through the host-pointer entry on pageable buffers (the first
        2^20 of this configuration's G2 bases and scalars), beside the resident entry on the same terms - MsmConfig.host_pointer's block."""
        from celo_bls_snark_rs_amd import ffi
        n = 1 << 20

        class Shim:
            pass
        sh = Shim()
        sh.cx, sh.group, sh.n = self.cx, "bls12_377_g2", n
        sh.bases = self.b2[:n * 24]
        sh.sc = self.s2[:n]
        sh.d_sc = self.d2[:n]
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            ref = ffi.msm_dev("bls12_377_g2", sh.bases.data_ptr(), 0, sh.d_sc.data_ptr(), n)
            ts.append((time.perf_counter() - t0) * 1e3)
        return MsmConfig.host_pointer(sh, ref, float(np.median(ts[1:])))

    def cpu_baseline(self, result, seq):
        from oracle import cpu_oracle as co
        from celo_bls_snark_rs_amd import ffi
        cx = self.cx
        if co.jac_to_affine(result[0], "g1_377") != co.jac_to_affine(seq[0], "g1_377") or co.jac_to_affine(result[1], "g2_377") != co.jac_to_affine(seq[1], "g2_377"):
            raise SystemExit("PARITY FAILURE: concurrent legs != sequential legs")
        if cx.rank != 0:
            return None
        hw = co.lib().orc_hardware_threads()
        # the WHOLE job of this rank on the CPU port (not a 2^17 sample): 2^22 G1 + 2^22 G2 terms take ~15-25 s on 17 threads
        k = self.n
        T = max(1, min(hw, 17))
        h1 = self.b1.view(self.n, 12).cpu().numpy().view(np.uint64); h2 = self.b2.view(self.n, 24).cpu().numpy().view(np.uint64)
        t0 = time.perf_counter()
        o1 = co.msm("bls12_377_g1", h1, None, self.s1, threads=T)
        o2 = co.msm("bls12_377_g2", h2, None, self.s2, threads=T)
        secs = time.perf_counter() - t0
        if co.jac_to_affine(o1, "g1_377") != co.jac_to_affine(seq[0] if cx.world == 1 else ffi.msm_dev("bls12_377_g1", self.b1.data_ptr(), 0, self.d1.data_ptr(), k), "g1_377") or \
           co.jac_to_affine(o2, "g2_377") != co.jac_to_affine(seq[1] if cx.world == 1 else ffi.msm_dev("bls12_377_g2", self.b2.data_ptr(), 0, self.d2.data_ptr(), k), "g2_377"):
            raise SystemExit("PARITY FAILURE: GPU MSM != oracle on the whole job")
        t0 = time.perf_counter()
        npair = 16
        acc = [int(co.pairing_product_377(self.g1[2 * i: 2 * i + 2], None, self.g2[2 * i: 2 * i + 2], None)[1]) for i in range(npair)]
        psecs = time.perf_counter() - t0
        if acc != self.expect[:npair]:
            raise SystemExit("PARITY FAILURE: oracle pairing verdicts != constructed")
        return {"value": 2 * k / secs, "seconds": secs, "unit": "scalar-muls/s", "cores": T, "kind": "port", "hardware_threads": hw, "parity_with_gpu": True,
                "miller_loops_per_s_1core": 2 * npair / psecs,
                "sample": "the whole job of rank 0: 2^%d-term G1 and G2 MSMs (arkworks windowing, one thread per window) back to back, results compared with the GPU's; + %d "
                          "distinct two-pair products on one core (accept vector of all %d products against construction)" % ((k - 1).bit_length(), npair, self.mprod)}


# ===================================================================================================== secondary legs (cfg2, N = 1)
def pairing_leg(ffi, check_oracle=True):
    """"+ pairings/sec" of BASELINE.json's metric: m independent 2-pair checks e(sig,-g2)*e(H,pk) == 1 (PublicKey::verify /
    Batch::verify's final check) in one launch; Miller loops/s with one final exponentiation per 2 loops."""
    m = 81920                                 # one 6-lane group per product, 10 groups per wave: 8192 waves = four full rounds of 2 waves/SIMD
    g1, g2, offs, expect = verify_shaped_products(m, 0x5EED0005)
    ffi.pairing_product_is_one_batch(g1, None, g2, None, offs)          # warm-up
    best, got = None, None
    for _ in range(3):
        t0 = time.perf_counter()
        got = ffi.pairing_product_is_one_batch(g1, None, g2, None, offs)
        dt = time.perf_counter() - t0
        tm = dict(ffi.pairing_timings(), wall_ms=dt * 1e3)
        if best is None or tm["total_ms"] < best["total_ms"]:
            best = tm
    ok = got.tolist() == expect
    cpu_rate = None
    if check_oracle:
        from oracle import cpu_oracle as co
        t0 = time.perf_counter()
        cpu_ok = [co.pairing_product_377(g1[2 * i:2 * i + 2], None, g2[2 * i:2 * i + 2], None)[1] for i in range(16)]
        cpu_rate = 2 / ((time.perf_counter() - t0) / 16)
        ok = ok and [int(x) for x in cpu_ok] == expect[:16]
    if not ok:
        raise SystemExit("PARITY FAILURE: GPU pairing accept vector != expected / oracle")
    secs = best["total_ms"] * 1e-3
    gbps = 2 * m * 288 / secs / 1e9
    # every product's first pair is (sig, -g2): the engine evaluates prepared line coefficients for it (pairing.h run_staged)
    kernels = ["k_prepare_lines<LPH377>", "k_miller_prepared_slots<LPH377>", "k_final_exp_slots<LPH377>"]
    traffic, src = committed_traffic(kernels, m)
    # multiplier roofline of the leg: the six-lane kernels work in PRODUCT ROUNDS - one half-Fq2 signed pass (2 limb-product sweeps + 1 reduction
    # = 574 multiply-adds) on each of a group's six lanes = 3444 multiply-adds; DESIGN.md section 5 counts
    # 63 x 17.0 + 6 x 14.3 = 1157 rounds in the prepared-line Miller loop of a two-pair product and ~900 in its final exponentiation (315
    # cyclotomic squarings of 2, ~45 Fq12 products of 6, the easy part)
    pk = valu_peaks()
    rounds = {"miller": 1157, "final_exp": 900}
    peak = pk["tmads"]["bls12_377_g1"]
    hr = MADS["hex_round"]
    vr = {"bound": "integer VALU (v_mad_u64_u32 / v_mad_i64_i32 issue)", "unit": "T multiply-adds/s", "peak": peak, "peak_measured_in_run": pk["measured"],
          "rounds_per_product": rounds, "multiply_adds_per_round": hr}
    for k_, ms_ in (("miller", best["miller_ms"]), ("final_exp", best["final_exp_ms"])):
        vr[k_] = {"achieved": m * rounds[k_] * hr / (ms_ * 1e-3) / 1e12, "frac": m * rounds[k_] * hr / (ms_ * 1e-3) / 1e12 / peak}
    vr["achieved"] = m * (rounds["miller"] + rounds["final_exp"]) * hr / ((best["miller_ms"] + best["final_exp_ms"]) * 1e-3) / 1e12
    vr["frac"] = vr["achieved"] / peak
    vr["note"] = pk["note"] + "; achieved = products x product rounds x 3444 multiply-adds / kernel ms (HIP events); the rounds exclude the tower's additions, carries, " \
                              "selects and lane exchanges, which is what the fraction below 1 is made of"
    return {"metric": "BLS12-377 Miller loops/s (2-pair products, 1 final exponentiation per product)", "value": 2 * m / secs, "valu_roofline": vr,
            "products": m, "device_ms": best["total_ms"], "wall_ms_incl_pcie": best["wall_ms"], "miller_ms": best["miller_ms"], "final_exp_ms": best["final_exp_ms"],
            "roofline": {"bound": "hbm", "kernel": " + ".join(kernels), "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": gbps / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": src, "note": "algorithmic bytes = 288 B per Miller loop (SURVEY.md section 8d); integer-VALU bound"},
            "cpu_port_miller_loops_per_s_1core": cpu_rate, "accept_vector_matches_oracle": ok if check_oracle else None}


def ntt_leg(ffi, check_oracle=True):
    """SURVEY.md section 8f row f3 (the prover's witness-map FFTs): one 2^20-point NTT over Fr(BW6-761), data resident in HBM."""
    from oracle import cpu_oracle as co
    from oracle.py import ntt as ontt, ecc
    log_n = 20
    n = 1 << log_n
    w = co.to_mont([ontt.root_of_unity(log_n)], ecc.Q377)[0]
    x = np.random.default_rng(0x5EED0006).integers(0, 1 << 62, size=(n, 6), dtype=np.int64)
    x[:, 5] &= (1 << 56) - 1
    d = torch.from_numpy(x).cuda()
    ffi.ntt_dev(d.data_ptr(), log_n, w)               # warm-up: builds the twiddle table
    best = None
    for _ in range(5):
        ffi.ntt_dev(d.data_ptr(), log_n, w)
        tm = ffi.ntt_timings()
        if best is None or tm["total_ms"] < best["total_ms"]:
            best = tm
    secs = best["total_ms"] * 1e-3
    gbps = n * 96 / secs / 1e9
    res = {"metric": "Fr(BW6-761) NTT elements/s (2^20 points, forward, in place)", "value": n / secs, "device_ms": best["total_ms"], "butterfly_passes": best["passes"],
           "roofline": {"bound": "hbm", "kernel": "k_ntt_tile4 (x%d launches)" % best["passes"], "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                        "traffic": None, "note": "algorithmic bytes = one 48-B read + one 48-B write per element; work bound: (n/2) log2 n field products = %.3g/s = %.2f of the multiplier peak"
                        % ((n // 2) * log_n / secs, (n // 2) * log_n / secs / 78e9)}}
    if check_oracle:
        m = 1 << 16
        wm = ontt.root_of_unity(16)
        xs = np.ascontiguousarray(x[:m]).view(np.uint64)
        cpu_secs = co.time_ntt_fq377(xs, 16, wm)
        ok = np.array_equal(ffi.ntt(xs, 16, co.to_mont([wm], ecc.Q377)[0]), co.ntt_fq377(xs, 16, wm))
        if not ok:
            raise SystemExit("PARITY FAILURE: GPU NTT != oracle NTT")
        res.update({"cpu_port_elements_per_s_1core": m / cpu_secs, "parity_2p16_vs_oracle": ok})
    return res


def wire_leg(ffi, check_oracle=True):
    """2^16 compressed G2 keys decoded with the subgroup check, 2^16 messages hashed to G1."""
    from oracle import cpu_oracle as co
    from oracle.py import ecc
    n = 1 << 16
    P, enc = ecc.G2_377, []
    for i in range(64):
        P = ecc.E2_377.add(ecc.E2_377.add(P, P), ecc.G2_377)
        enc.append(ecc.ser_point(ecc.E2_377, P if i % 2 else ecc.E2_377.neg(P)))
    host = np.tile(np.frombuffer(b"".join(enc), dtype=np.uint8), n // 64)
    d_in = torch.from_numpy(host.copy()).cuda()
    d_out = torch.zeros((n, 24), dtype=torch.int64, device="cuda")
    d_st = torch.zeros(n, dtype=torch.uint8, device="cuda")
    best = None
    for _ in range(3):
        ffi.decompress_dev("g2", d_in.data_ptr(), n, d_out.data_ptr(), d_st.data_ptr(), True)
        ms = ffi.decompress_last_ms()
        best = ms if best is None or ms < best else best
    if d_st.any().item():
        raise SystemExit("PARITY FAILURE: a valid G2 encoding was rejected")
    gbps = n * 288 / (best * 1e-3) / 1e9
    res = {"decompress_g2_checked_points_per_s": n / (best * 1e-3), "decompress_ms": best,
           "roofline": {"bound": "hbm", "kernel": "k_decompress<true>", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "traffic": None,
                        "note": "algorithmic bytes = 96 B in + 192 B out per key; integer-VALU bound"}}
    raw = np.random.default_rng(0x5EED0007).integers(0, 256, size=(n, 32), dtype=np.uint8)
    msgs = [raw[i].tobytes() for i in range(n)]
    best = None
    for _ in range(3):
        xy, att = ffi.hash_to_g1_direct(b"ULforxof", msgs, [b"\x01\x02"] * n)
        ms = ffi.hash_last_ms()
        best = ms if best is None or ms < best else best
    res.update({"hash_to_g1_direct_hashes_per_s": n / (best * 1e-3), "hash_ms": best, "hash_mean_attempts": float(att.mean()) + 1.0})
    if check_oracle:
        m = 128
        T = co.lib().orc_hardware_threads()
        secs = co.time_decompress("g2", host[: m * 96].tobytes(), True, 1)
        wxy, wst = co.decompress("g2", host[: m * 96].tobytes(), True, min(T, 8))
        ok = np.array_equal(d_out[:m].cpu().numpy().view(np.uint64), wxy) and not wst.any()
        if not ok:
            raise SystemExit("PARITY FAILURE: GPU decompression != oracle")
        res.update({"cpu_port_decompress_points_per_s_1core": m / secs, "decompress_parity_vs_oracle": ok})
    return res


# ===================================================================================================== driver
def launch_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: re-run this command line as N ranks (one process per GPU) under
    torch.distributed.run on 127.0.0.1; rank 0's JSON line passes through on stdout.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    env["CELO_BENCH_LAUNCHER"] = "bench.py --gpus %d (self-launched torch.distributed.run)" % n
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configuration (default 2: the one the metric is quoted on)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--partition", default="auto", choices=["auto", "index", "windows"], help="configs 2 / 4, N > 1: how one MSM is cut over the GPUs (see the module docstring)")
    ap.add_argument("--log-n", type=int, default=0, help="override log2 of the MSM size (per GPU when weak, total when strong)")
    ap.add_argument("--batches", type=int, default=4096, help="config 3: batches")
    ap.add_argument("--signers", type=int, default=256, help="config 3: signers per batch")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--witness-like", action="store_true", help="configs 2/4: about 60 %% of the scalars are 0 or 1 (a Groth16 witness)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--all-configs", action="store_true", help="N = 1: after this configuration's line, one more line for each of the other BASELINE configurations (3, 4, 5 / 2)")
    ap.add_argument("--no-pairing", action="store_true", help="config 2, N = 1: skip the secondary pairing / NTT / wire legs")
    ap.add_argument("--balanced", action="store_true", help="diagnostic: scalars whose digits fill every bucket equally (not the headline workload)")
    ap.add_argument("--subgroup-points", action="store_true", help="config 2: time msm_bls12_377_g1_subgroup_dev (bases vouched to lie in G1: GLV split) instead of the plain entry point")
    ap.add_argument("--fixed-base", action="store_true", help="configs 2 / 4: time msm_*_fixed_dev against per-key tables built once (msm_*_precompute_dev): the Groth16 prover's shape")
    ap.add_argument("--fixed-window-bits", type=int, default=0, help="--fixed-base: table window size c (16..22; 0 = automatic)")
    ap.add_argument("--in-process", action="store_true", help="configs 2 / 4, N > 1: one process drives the N devices through msm_*_multi_dev (no ranks, no collective)")
    ap.add_argument("--devices", default="", help="--in-process: comma-separated device ordinals (default 0..N-1; repeats allowed, e.g. 0,0 on a 1-GPU box)")
    args = ap.parse_args()

    world_env = int(os.environ.get("WORLD_SIZE", "0"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.fixed_base and (args.config not in (2, 4) or args.in_process or args.subgroup_points or (args.gpus > 1 and args.scaling == "strong" and args.partition != "index")):
        raise SystemExit("--fixed-base: configs 2 and 4, one table per rank over its index-range shard (weak scaling, or strong with --partition index)")
    if args.in_process and args.config not in (2, 4):
        raise SystemExit("--in-process drives msm_*_multi_dev: configs 2 and 4 only")
    if args.gpus > 1 and not args.in_process and world_env == 0:
        sys.exit(launch_ranks(args.gpus))            # plain `python bench.py --gpus N`: start the N ranks ourselves
    if world_env and (args.in_process or world_env != args.gpus):
        raise SystemExit("bench.py: --gpus %d%s but the launcher started WORLD_SIZE=%d ranks: refusing to print a line for a different job"
                         % (args.gpus, " --in-process" if args.in_process else "", world_env))

    cx = Ctx()
    cx.args, cx.cfg = args, args.config
    cx.world = max(1, world_env)
    cx.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cx.devices = None
    if args.in_process and args.gpus > 1:
        cx.devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
        if len(cx.devices) != args.gpus:
            raise SystemExit("--devices must list --gpus ordinals")
        if max(cx.devices) >= torch.cuda.device_count():
            raise SystemExit("--in-process --gpus %d: only %d device(s) visible (list repeats with --devices to share one)" % (args.gpus, torch.cuda.device_count()))
        local_rank = cx.devices[0]
    cx.nshards = len(cx.devices) if cx.devices else cx.world
    # CELO_BENCH_BACKEND=gloo + CELO_BENCH_DEVICE=0 lets the N>1 code path be smoke-tested on a 1-GPU box (both ranks on
    # one device, host-staged exchange); the driver's multi-GPU runs use the defaults: RCCL, one GPU per rank.
    backend = os.environ.get("CELO_BENCH_BACKEND", "nccl")
    if "CELO_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["CELO_BENCH_DEVICE"])
    elif cx.world > 1 and local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d device(s) visible (CELO_BENCH_BACKEND=gloo CELO_BENCH_DEVICE=0 shares one GPU for a smoke run)"
                         % (cx.rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)      # before the process group: RCCL binds its communicator to the current device
    if cx.world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend=backend, rank=cx.rank, world_size=cx.world)
    cx.xdev = "cuda" if backend == "nccl" else "cpu"
    cx.stream = torch.cuda.current_stream().cuda_stream

    from celo_bls_snark_rs_amd import ffi
    ffi.init(local_rank)
    if cx.cfg == 2:
        job = MsmConfig(cx, "bls12_377_g1", 20, "BLS12-377 G1 Pippenger MSM")
    elif cx.cfg == 4:
        job = MsmConfig(cx, "bw6_761_g1", 24, "epoch-snark Groth16 prover MSM over BW6-761 G1 (2^24 bases over 8 GPUs)")
    elif cx.cfg == 3:
        job = BatchVerifyConfig(cx)
    else:
        job = MixedConfig(cx)
    job.setup()

    def barrier():
        if cx.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    result = None
    for _ in range(args.warmup):
        result = job.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = job.step()
        job.after_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if cx.world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cx.xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    line = {"metric": None, "value": job.units_per_step() * args.steps / elapsed, "unit": None, "n_gpus": cx.nshards, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": DTYPE, "data": "synthetic", "baseline_config": cx.cfg}
    line["launch"] = ({"mode": "in-process", "devices": cx.devices, "ranks": 1, "rccl_world": None, "backend": None,
                       "note": "one process, msm_*_multi_dev: a host thread and an engine per listed device, host fold; no collective"} if cx.devices else
                      {"mode": "one process per GPU", "ranks": cx.world, "rccl_world": dist.get_world_size() if cx.world > 1 else 1,
                       "backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend) if cx.world > 1 else None,
                       "launcher": os.environ.get("CELO_BENCH_LAUNCHER", "external (torch.distributed.run)" if cx.world > 1 else "none"),
                       "device_of_rank0": local_rank})
    job.report(line, result)                                  # every rank takes part (gathers for the full-size parity check)
    if "parity" not in line:
        pw = (line.get("cpu_baseline") or {}).get("parity_with_gpu")
        line["parity"] = {"checked": pw is True, "against": "the CPU port (oracle/cpu), see cpu_baseline" if pw is True else str(pw)}
    if line["n_gpus"] != args.gpus:
        raise SystemExit("bench.py: the job ran on %d GPU(s) but --gpus %d was asked for: no line" % (line["n_gpus"], args.gpus))
    if cx.rank == 0:
        if cx.cfg == 2 and cx.world == 1 and not args.no_pairing:
            line["pairing"] = pairing_leg(ffi, check_oracle=not args.no_cpu_baseline)
            line["ntt"] = ntt_leg(ffi, check_oracle=not args.no_cpu_baseline)
            line["wire"] = wire_leg(ffi, check_oracle=not args.no_cpu_baseline)
        add_measured_hbm_peak(line)
        print(json.dumps(line), flush=True)
    if args.all_configs and cx.world == 1 and not cx.devices:
        # one more line per remaining BASELINE configuration (a single run times them all); each in a process of its own so that
        # its arenas and streams start fresh - the first line printed stays the headline's
        import subprocess
        del job, result
        torch.cuda.empty_cache()
        for c in (3, 4, 5):
            if c == cx.cfg:
                continue
            cmd = [sys.executable, os.path.abspath(__file__), "--config", str(c), "--steps", str(args.steps), "--warmup", str(args.warmup)]
            if args.no_cpu_baseline:
                cmd.append("--no-cpu-baseline")
            r = subprocess.run(cmd, capture_output=True, text=True)
            out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            print(out[-1] if out and r.returncode == 0 else json.dumps({"baseline_config": c, "error": (r.stderr or r.stdout)[-400:], "rc": r.returncode}), flush=True)
    if cx.world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
