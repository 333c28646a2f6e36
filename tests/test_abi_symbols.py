"""CPU (-m "not gpu"): the C-ABI library loads and exports every symbol include/celo_bls_amd.h declares.
No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "celo_bls_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(\w+)\s*\(", txt)))


def test_header_declares_expected_entry_points():
    from celo_bls_snark_rs_amd import ffi
    decl = declared_symbols()
    for name in ffi.EXPORTS:
        assert name in decl, name


def test_library_exports_every_declared_symbol():
    from celo_bls_snark_rs_amd import ffi
    if not os.path.exists(ffi.LIB_PATH):
        pytest.fail(f"{ffi.LIB_PATH} missing — run __graft_entry__.build()")
    lib = C.CDLL(ffi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"missing export {name}"


def test_no_device_fails_loudly():
    """Without a GPU the product must refuse, not fall back to a CPU path."""
    import numpy as np
    import torch
    from celo_bls_snark_rs_amd import ffi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    xy = np.zeros((2, 12), dtype=np.uint64)
    sc = np.ones((2, 4), dtype=np.uint64)
    with pytest.raises(RuntimeError):
        ffi.msm("bls12_377_g1", xy, None, sc)
    # the rows either side of the path refuse as well: decoding, normalisation, hashing, the NTT
    with pytest.raises(RuntimeError):
        ffi.decompress("g1", bytes(48))
    with pytest.raises(RuntimeError):
        ffi.normalize("g1", np.zeros((1, 18), dtype=np.uint64))
    with pytest.raises(RuntimeError):
        ffi.hash_to_g1_direct(b"ULforxof", [b"m"])
    with pytest.raises(RuntimeError):
        ffi.hash_to_g1_composite(b"ULforxof", [b"m"], cip22=True)
    with pytest.raises(RuntimeError):
        ffi.composite_crh([b"m"])
    with pytest.raises(RuntimeError):
        ffi.ntt(np.zeros((2, 6), dtype=np.uint64), 1, np.zeros(6, dtype=np.uint64))
    # round 2 entry points: multi-device MSM, chained Batch::verify, the prover row, device binding
    with pytest.raises(RuntimeError):
        ffi.msm_multi("bls12_377_g1", [0, 0], xy, None, sc)
    with pytest.raises(RuntimeError):
        ffi.batch_verify(np.zeros((1, 24), dtype=np.uint64), np.zeros((1, 12), dtype=np.uint64), np.ones((1, 4), dtype=np.uint64),
                         np.array([0, 1], dtype=np.uint32), np.zeros((1, 12), dtype=np.uint64), np.zeros(24, dtype=np.uint64))
    z6 = np.zeros(6, dtype=np.uint64)
    with pytest.raises(RuntimeError):
        ffi.witness_map(np.zeros((2, 6), dtype=np.uint64), np.zeros((2, 6), dtype=np.uint64), np.zeros((2, 6), dtype=np.uint64), 1,
                        {k: z6 for k in ("omega", "omega_inv", "coset", "coset_inv", "size_inv", "vanishing_inv")})
    with pytest.raises(RuntimeError):
        ffi.groth16_prove(np.zeros((2, 24), dtype=np.uint64), np.zeros((2, 24), dtype=np.uint64), np.zeros((1, 24), dtype=np.uint64),
                          np.zeros((1, 24), dtype=np.uint64), np.zeros(24, dtype=np.uint64), np.zeros(24, dtype=np.uint64),
                          np.ones((1, 6), dtype=np.uint64), 1, np.ones((1, 6), dtype=np.uint64))
    # round 4: the exponent generator of batch_verify_strict is a device kernel
    with pytest.raises(RuntimeError):
        ffi.draw_batch_exponents(np.arange(8, dtype=np.uint32), np.array([0, 3], dtype=np.uint32))
    with pytest.raises(RuntimeError):
        ffi.use_device(0)
    with pytest.raises(RuntimeError):
        ffi.device_count()


def test_no_stray_c_symbols_exported():
    """the library once exported unprefixed globals (q_mu, q_cv, q_pending, q_leader, run_products) next to the
    reference-mandated `init` / `verify` - a collision waiting in a cgo host process.  Every unmangled dynamic symbol must be declared in
    include/*.h or carry the celo_ prefix."""
    import subprocess
    from celo_bls_snark_rs_amd import ffi
    out = subprocess.run(["nm", "-D", "--defined-only", ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    plain = [n for n in names if not n.startswith("_Z") and not n.startswith("__hip") and not n.startswith("_")]
    hdr = ""
    for h in ("celo_bls_amd.h", "celo_bls_snark_sys.h"):
        hdr += open(os.path.join(ROOT, "include", h)).read()
    stray = [n for n in plain if not n.startswith("celo_") and not re.search(r"\b%s\s*\(" % re.escape(n), hdr)]
    assert stray == [], stray


def test_accumulate_kernels_are_out_of_the_sgpr_spill_regime():
    """rounds 2-4 shipped k_accumulate with 107-211 spilled SGPRs (the K p tables of the cold canonical reduction, hoisted
    into the prologue and parked in VGPR lanes) - the register shape next to which the round-3 signed-pass instantiation miscompiled.  Round 5
    made those tables immediates (csrc/fp.h reduce / cond_sub_k): what the compiler still parks are the exec masks of nested divergent
    regions, a handful.  The build's own resource remarks (-Rpass-analysis=kernel-resource-usage, written by the Makefile next to every
    object) are the evidence: every instantiation of k_accumulate / k_accumulate_chunk reports at most 8 spilled SGPRs and no scratch.
    Round 6: the 28-limb field (the prover's kernels) is no longer exempt - its tables are immediates too (227 / 221 -> 4 / 2 spilled SGPRs);
    what kept it on pointer tables was the return-address hazard of test_no_far_branch_runs_on_the_return_address, not the tables."""
    import glob
    import re
    build = os.path.join(ROOT, "celo-bls-snark-rs_amd", "build")
    seen = 0
    for f in glob.glob(os.path.join(build, "unit_*.remarks.txt")):
        txt = open(f).read()
        for m in re.finditer(r"Function Name: (\S*k_accumulate\S*).*?ScratchSize \[bytes/lane\]: (\d+).*?SGPRs Spill: (\d+)", txt, re.S):
            seen += 1
            assert int(m.group(3)) <= 8 and int(m.group(2)) == 0, (os.path.basename(f), m.group(1), m.group(2), m.group(3))
    assert seen >= 6, "no resource remarks found: build with make -C celo-bls-snark-rs_amd/csrc"


def test_horner_kernels_of_the_28_limb_field_left_the_spill_regime():
    """k_batch_horner / k_batch_horner_lanes <G_761> parked 585 / 595 SGPRs in VGPR lanes while the K p tables sat behind
    pointers; with immediates what remains are exec masks of nested divergent regions."""
    import glob
    build = os.path.join(ROOT, "celo-bls-snark-rs_amd", "build")
    seen = 0
    for f in glob.glob(os.path.join(build, "unit_*.remarks.txt")):
        for m in re.finditer(r"Function Name: (\S*k_batch_horner\S*G_761\S*).*?SGPRs Spill: (\d+)", open(f).read(), re.S):
            seen += 1
            assert int(m.group(2)) <= 64, (os.path.basename(f), m.group(1), m.group(2))
    assert seen >= 1, "no resource remarks found: build with make -C celo-bls-snark-rs_amd/csrc"


def test_no_far_branch_runs_on_the_return_address():
    """Round 5's hang of k_combine_big<G_761> (tools/repro_combine/REPORT.md): in an out-of-line device function longer
    than the reach of s_cbranch the compiler's long-branch reserved register can be s[30:31], the function's own return address.  The library is
    built with -mllvm -amdgpu-long-branch-factor=0 (no reservation); tools/scan_long_branch.py disassembles every code object of the BUILT
    library and fails if any non-kernel function still has the shape."""
    import subprocess
    import sys
    from celo_bls_snark_rs_amd import ffi
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_long_branch.py"), ffi.LIB_PATH], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"(\d+) non-kernel device functions", r.stdout)
    assert m and int(m.group(1)) >= 10, r.stdout      # the scan did look at the out-of-line routines
