#!/usr/bin/env python3
"""Build-time guard against the LLVM AMDGPU long-branch / return-address hazard (tools/repro_combine/REPORT.md).

In a NON-KERNEL device function whose body exceeds the +-2^17-byte reach of s_cbranch, branch relaxation expands far branches into
    s_getpc_b64 s[N:N+1]; s_add_u32; s_addc_u32; s_setpc_b64 s[N:N+1]
With this compiler (ROCm 7.2, clang 22) the pair is the "long-branch reserved register": the highest free SGPR pair before register allocation,
moved to the LOWEST pair the function does not use after it.  In a leaf function that needs few SGPRs that lowest unused pair is s[30:31] - the
function's own return address, whose only reader is the final s_setpc_b64: the first far branch taken destroys it and the function returns into
its own body; the wave never returns.  That is what made k_combine_big<G_761> hang in round 5 once the K p tables became immediates (the
out-of-line addition it calls stopped needing ~100 SGPRs for them).  Any OTHER pair is safe: device functions are internalised, so callers take
their clobber masks from the callee's actual register use (IPRA), far-branch pair included - checked in the ISA of the callers.

-mllvm -amdgpu-long-branch-factor=0 (csrc/Makefile) switches the reservation off; the pair is then scavenged from registers proved dead.

This script disassembles every gfx950 code object of a shared library / object file and reports each non-kernel function that contains
`s_getpc_b64 s[30:31]` without having saved s30 in its prologue (v_writelane_b32 vN, s30, lane).  Exit status 1 if any is found.
usage: scan_long_branch.py <lib.so | unit.o> [...]"""
import os, re, subprocess, sys, tempfile, shutil

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def code_objects(path, tmp):
    dst = os.path.join(tmp, os.path.basename(path))
    shutil.copy(path, dst)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", dst], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if f.startswith(os.path.basename(path) + ".") and "amdgcn" in f)


def scan(co):
    syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", co], capture_output=True, text=True).stdout
    kernels = set(m.group(1)[:-3] for m in re.finditer(r"\s(\S+\.kd)\s*$", syms, re.M))
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    bad, nfun, nfar = [], 0, 0
    for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^[0-9a-f]+ <[^>]+>:\n|\Z)", dis, re.M | re.S):
        name, body = m.group(1), m.group(2)
        if name in kernels or name.startswith("L") or name.startswith("$"):
            continue
        nfun += 1
        far = re.findall(r"s_getpc_b64 s\[(\d+):\d+\]", body)       # (also counts pc-relative address computations; only the pair matters here)
        nfar += len(far)
        saved = re.search(r"v_writelane_b32 v\d+, s30,", body) is not None
        if "30" in far and not saved:
            bad.append((name, len(far), far.count("30")))
    return nfun, nfar, bad


def main(paths):
    rc = 0
    for p in paths:
        tmp = tempfile.mkdtemp(prefix="scanlb_")
        try:
            tot_f = tot_far = 0
            file_bad = False
            for co in code_objects(p, tmp):
                nfun, nfar, bad = scan(co)
                tot_f += nfun; tot_far += nfar
                for name, n, k in bad:
                    rc = 1
                    file_bad = True
                    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                    print(f"HAZARD {os.path.basename(p)}: {dn[:140]}: {k} of {n} s_getpc sequences run on s[30:31], the function's unsaved return address")
            print(f"{os.path.basename(p)}: {tot_f} non-kernel device functions, {tot_far} s_getpc sequences in them, {'see above' if file_bad else 'none on an unsaved return address'}")
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
