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
import json
traffic = {}
for k in sorted(set(fa) | set(wa)):
    f = sum(fa[k].get("FETCH_SIZE", [0])) / max(1, len(fa[k].get("FETCH_SIZE", [0])))
    w = sum(wa[k].get("WRITE_SIZE", [0])) / max(1, len(wa[k].get("WRITE_SIZE", [0])))
    traffic[k] = {"fetch_bytes_x2_corrected": 2 * f * 1024, "write_bytes": w * 1024, "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024}
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tag {tag}; FETCH_SIZE doubled per MI355X_MICROARCH.md",
           "kernels": traffic}, open(os.path.join(root, "profiles", f"{tag}_traffic.json"), "w"), indent=1)
open(out, "w").write("\n".join(lines) + "\n")
print(open(out).read())
