#!/usr/bin/env python3
"""Per-function code size / VGPR / AGPR / scratch table from a `hipcc --cuda-device-only -S` listing.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o x.s unit.hip && python tools/isa_funcs.py x.s"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
names = re.findall(r"^(\S+):\s+; @\1$", s, re.M)
infos = re.findall(r"; Function info:\n; codeLenInByte = (\d+)\n; TotalNumSgprs: (\S+)\n; NumVgprs: (\S+)\n; NumAgprs: (\S+)\n; TotalNumVgprs: (\S+)\n; ScratchSize: (\S+)", s)
kinfos = re.findall(r"; Kernel info:\n; codeLenInByte = (\d+)\n; TotalNumSgprs: (\S+)\n; NumVgprs: (\S+)\n; NumAgprs: (\S+)\n; TotalNumVgprs: (\S+)\n; ScratchSize: (\S+)", s)
blocks = re.split(r"^\S+:\s+; @\S+$", s, flags=re.M)[1:]
for n, b in zip(names, blocks):
    m = re.search(r"; (?:Function|Kernel) info:\n; codeLenInByte = (\d+)\n; TotalNumSgprs: (\S+)\n; NumVgprs: (\S+)\n; NumAgprs: (\S+)\n; TotalNumVgprs: (\S+)\n; ScratchSize: (\S+)", b)
    if not m:
        continue
    dn = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    calls = len(re.findall(r"s_swappc_b64", b))
    sl = len(re.findall(r"scratch_load", b)); ss = len(re.findall(r"scratch_store", b))
    print(f"{int(m.group(1)):>7} B v{m.group(3):>4} a{m.group(4):>3} scr{m.group(6):>6} calls{calls:>4} sld{sl:>5} sst{ss:>5}  {dn[:100]}")
