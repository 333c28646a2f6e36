#!/usr/bin/env python3
"""Summarises gpurun_out/prof_<tag>/ (rocprofv3 CSVs from tools/profile_bench.sh) into profiles/<tag>_summary.md."""
import csv, sys, os, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
out = os.path.join(root, "profiles", f"{tag}_summary.md")
os.makedirs(os.path.dirname(out), exist_ok=True)

def short(n):
    n = n.replace("void celo::", "").replace("celo::", "")
    return n.split("(")[0]

cmd = "python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
if os.path.exists(os.path.join(src, "command.txt")):
    cmd = open(os.path.join(src, "command.txt")).read().strip().replace(root + "/", "")
lines = [f"# rocprofv3 summary — {tag}", "", f"Command: `{cmd}`, MI355X gfx950.", "",
         "## kernel trace (`rocprofv3 --kernel-trace --stats`)", "", "| kernel | calls | avg µs | total ms | % |", "|---|---|---|---|---|"]
with open(os.path.join(src, "trace", "trace_kernel_stats.csv")) as f:
    for r in csv.DictReader(f):
        lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |")

def pmc(dirname):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    p = os.path.join(src, dirname, "pmc_counter_collection.csv")
    if not os.path.exists(p):
        return agg, meta
    with open(p) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"], r["Grid_Size"], r["Workgroup_Size"])
    return agg, meta

lines += ["", "## PMC passes (separate runs; per-launch averages)", "",
          "FETCH_SIZE / WRITE_SIZE are in KiB as reported; per MI355X_MICROARCH.md §HBM FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 — the `HBM read (corrected)` column doubles it.", ""]
fa, _ = pmc("pmc_fetch")
wa, _ = pmc("pmc_write")
sa, meta = pmc("pmc_sq")
lines += ["| kernel | FETCH_SIZE KiB | HBM read MB (x2 corrected) | WRITE_SIZE KiB | VGPR | AGPR | SGPR | scratch | grid | wg |", "|---|---|---|---|---|---|---|---|---|---|"]
for k in sorted(set(fa) | set(wa)):
    f = sum(fa[k].get("FETCH_SIZE", [0])) / max(1, len(fa[k].get("FETCH_SIZE", [0])))
    w = sum(wa[k].get("WRITE_SIZE", [0])) / max(1, len(wa[k].get("WRITE_SIZE", [0])))
    m = meta.get(k, ("?",) * 7)
    lines.append(f"| `{k}` | {f:.0f} | {2*f*1024/1e6:.1f} | {w:.0f} | {m[0]} | {m[1]} | {m[2]} | {m[3]} | {m[5]} | {m[6]} |")
lines += ["", "| kernel | SQ_WAVES | SQ_INSTS_VALU | SQ_WAVE_CYCLES | SQ_BUSY_CYCLES | SQ_WAIT_INST_ANY | SQ_ACTIVE_INST_VALU | SQ_INSTS_VMEM_RD | VALU insts/wave |", "|---|---|---|---|---|---|---|---|---|"]
for k in sorted(sa):
    def a(c):
        v = sa[k].get(c, [0]); return sum(v) / max(1, len(v))
    waves = a("SQ_WAVES")
    lines.append(f"| `{k}` | {waves:.0f} | {a('SQ_INSTS_VALU'):.3g} | {a('SQ_WAVE_CYCLES'):.3g} | {a('SQ_BUSY_CYCLES'):.3g} | {a('SQ_WAIT_INST_ANY'):.3g} | {a('SQ_ACTIVE_INST_VALU'):.3g} | {a('SQ_INSTS_VMEM_RD'):.3g} | {a('SQ_INSTS_VALU')/max(1,waves):.0f} |")
# effective clock of each kernel: GRBM_GUI_ACTIVE cycles of a dispatch / its duration (pmc_clock pass)
clk = collections.defaultdict(list)
pc = os.path.join(src, "pmc_clock", "pmc_counter_collection.csv")
if os.path.exists(pc):
    with open(pc) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            try:
                dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            except (KeyError, ValueError):
                continue
            if dur > 2e5:                       # dispatches of 0.2 ms and more: shorter ones are dominated by the counter's start / stop
                clk[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) / 8.0 / dur)     # the counter is summed over the 8 XCDs
if clk:
    lines += ["", "## effective clock (GRBM_GUI_ACTIVE cycles per dispatch, summed over the 8 XCDs, / 8 / dispatch duration; dispatches >= 0.2 ms)", "", "| kernel | dispatches | GHz (mean) | GHz (min) | GHz (max) |", "|---|---|---|---|---|"]
    for k in sorted(clk):
        v = clk[k]
        lines.append(f"| `{k}` | {len(v)} | {sum(v)/len(v):.3f} | {min(v):.3f} | {max(v):.3f} |")
import json, re, subprocess, glob
# the build that was profiled: registers / scratch of every kernel from the compiler's remarks (build/*.remarks.txt), keyed like the
# traffic table - bench.py compares it with the build it runs and marks roofline.traffic stale on a mismatch
def build_signature():
    sig = {}
    for path in glob.glob(os.path.join(root, "celo-bls-snark-rs_amd", "build", "unit_*.remarks.txt")):
        cur = None
        for ln in open(path, errors="replace"):
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                cur = m.group(1); sig.setdefault(cur, {}); continue
            m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|TotalSGPRs): (\d+)", ln)
            if m and cur:
                sig[cur][m.group(1).split(" ")[0]] = int(m.group(2))
    if not sig:
        return {}
    names = list(sig)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return {short(d): sig[n] for n, d in zip(names, dem)}
traffic = {}
for k in sorted(set(fa) | set(wa)):
    f = sum(fa[k].get("FETCH_SIZE", [0])) / max(1, len(fa[k].get("FETCH_SIZE", [0])))
    w = sum(wa[k].get("WRITE_SIZE", [0])) / max(1, len(wa[k].get("WRITE_SIZE", [0])))
    traffic[k] = {"fetch_bytes_x2_corrected": 2 * f * 1024, "write_bytes": w * 1024, "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024}
bs = build_signature()
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tag {tag}; FETCH_SIZE doubled per MI355X_MICROARCH.md",
           "kernels": traffic, "build_signature": {k: bs[k] for k in traffic if k in bs},
           "effective_clock_ghz": {k: sum(v) / len(v) for k, v in clk.items()}}, open(os.path.join(root, "profiles", f"{tag}_traffic.json"), "w"), indent=1)
open(out, "w").write("\n".join(lines) + "\n")
print(open(out).read())
