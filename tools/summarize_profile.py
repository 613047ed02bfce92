#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (rocprofv3 csv output) into profiles/<tag>_*.{csv,md}."""
import collections, csv, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
extra = sys.argv[2] if len(sys.argv) > 2 else ""  # extra bench.py arguments the profile was taken with
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
for b in (f"bench_{tag}.json", f"bench_gpu_{tag}.json"):
    if os.path.exists(os.path.join(ROOT, "gpurun_out", b)):
        shutil.copy(os.path.join(ROOT, "gpurun_out", b), os.path.join(dst, b))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); dur = collections.defaultdict(float)
for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_lds"):
    f = os.path.join(src, sub, "bench_counter_collection.csv")
    if not os.path.exists(f):
        continue
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if sub == "pmc_sq" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); calls[k] += 1
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w") as out:
    names = sorted({c for v in agg.values() for c in v})
    out.write("kernel,dispatches,ms_under_pmc," + ",".join(names) + "\n")
    for k, v in agg.items():
        out.write(f"\"{k}\",{calls[k]},{dur[k]:.3f}," + ",".join(f"{v.get(n, 0):.6g}" for n in names) + "\n")
lines = [f"# rocprofv3 summary {tag}", "", "Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra" + (" " + extra if extra else "") + "` and",
         "separate `--pmc` passes (`bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra" + (" " + extra if extra else "") + "`), MI355X, 800x800, 64+128 samples.", "", "## kernel-trace stats", "", "```"]
lines += open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read().strip().split("\n") + ["```", "", "## derived (PMC passes; all figures PER DISPATCH = totals of the pass / dispatches -- a pass may hold more than one frame)", ""]
for k, v in agg.items():
    if not any(n in k for n in ("nerf_mlp_kernel", "nerf_trunk_seq_kernel", "nerf_colour_kernel")) or not v.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        continue
    t = dur[k] * 1e-3
    nd = max(calls[k], 1)
    clk = v["GRBM_GUI_ACTIVE"] / 8 / t / 1e9
    busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (v["GRBM_GUI_ACTIVE"] / 8)
    wave = v["SQ_WAVE_CYCLES"]
    lines.append(f"* `{k}`: {nd} dispatches, {dur[k] / nd:.1f} ms each under PMC; effective clock {clk:.2f} GHz (GRBM_GUI_ACTIVE/8/t); MFMA pipe busy "
                 f"{100 * busy:.1f} % of cycles (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs; one f32 32x32x2 MFMA = 64 busy cycles, one bf16 32x32x16 = 32); of the wave cycles "
                 f"{100 * v['SQ_WAIT_ANY'] / wave:.1f} % waitcnt/barrier (SQ_WAIT_ANY), {100 * v['SQ_WAIT_INST_ANY'] / wave:.1f} % issue stall "
                 f"(SQ_WAIT_INST_ANY, i.e. waiting for the matrix pipe), {100 * v['SQ_ACTIVE_INST_ANY'] / wave:.1f} % issuing; "
                 f"HBM traffic per dispatch: FETCH_SIZE x2 (gfx950 correction) = {2 * v.get('FETCH_SIZE', 0) / 1024 / nd:.1f} MB, WRITE_SIZE = {v.get('WRITE_SIZE', 0) / 1024 / nd:.1f} MB "
                 f"(algorithmic: weights 2.3 MB + inputs/outputs; the kernel is MFMA-bound, HBM is idle)" +
                 (f"; LDS: {100 * v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:.2f} % of the LDS-array cycles are bank-conflict cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE), "
                  f"unaligned-access stalls {v.get('SQ_LDS_UNALIGNED_STALL', 0):.3g}, address conflicts {v.get('SQ_LDS_ADDR_CONFLICT', 0):.3g}" if v.get("SQ_LDS_IDX_ACTIVE") else ""))
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-4:]))
