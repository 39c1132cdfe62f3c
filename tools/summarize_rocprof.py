#!/usr/bin/env python
"""Condense rocprofv3 output (kernel-trace --stats CSV, --pmc counter CSVs) into small JSON summaries.
usage: summarize_rocprof.py <dir with prof_stats/ pmc_fetch/ pmc_write/> <tag>
Writes <dir>/<tag>_rocprof_kernel_stats.json and <dir>/<tag>_pmc_traffic.json.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE passes (they need 3 + 2 of the 4 TCC slots), both are reported in KiB, and on
gfx950 FETCH_SIZE counts 64 B per 128-B request for wide streaming reads, so it is DOUBLED.  WRITE_SIZE
was calibrated in this repo against a kernel with a known write volume (dw forward: 172 MB counted vs
176 MB written) and is used as is."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from change3d_amd._lib import csrc_digest  # noqa: E402

ENTRY = [  # kernel-name prefix -> C-ABI entry point (bench.py's kernel table key)
    ("pw_gemm_kernel", "c3d_pw_gemm"), ("pw_cfwd_kernel", "c3d_pw_gemm"), ("pw_cdg_a_kernel", "c3d_pw_gemm"), ("pw_cdg_c_kernel", "c3d_pw_gemm"),
    ("pw_wgrad_kernel", "c3d_pw_wgrad"), ("pw_wgrad_v2", "c3d_pw_wgrad"), ("pw_wgrad_reduce", "c3d_pw_wgrad"),
    ("dw_fwd", "c3d_dw333_fwd"), ("dw_bwd_fused", "c3d_dw333_bwd_fused"), ("dw_bwd_ring", "c3d_dw333_bwd_fused"),
    ("block_out_fwd", "c3d_block_out_fwd"), ("block_out_bwd", "c3d_block_out_bwd"),
    ("se_bn_bwd_coef", "c3d_se_bn_bwd_coef"), ("bn_se_finalize", "c3d_bn_se_finalize"),
    ("bn_finalize", "c3d_bn_finalize"), ("bn_bwd_coef", "c3d_bn_bwd_coef"), ("stem_", "c3d_stem_*"),
    ("convT", "c3d_convT4s2_*"), ("convt_", "c3d_convT4s2_*"), ("head3x3", "c3d_head3x3_*"), ("adam", "c3d_adam_step"),
    ("enhance_", "c3d_enhance_*"), ("frame_", "c3d_frame_*"), ("bce_dice", "c3d_bce_dice_*"), ("confusion", "c3d_confusion2"),
    ("col_sum", "c3d_col_sum"), ("fold_bn", "c3d_stage_fold_bn"), ("bcd_preprocess", "c3d_bcd_preprocess"),
]


def entry_of(kname):
    base = re.sub(r"^void\s+", "", kname)
    base = re.sub(r"^\(anonymous namespace\)::", "", base)
    for pre, e in ENTRY:
        if base.startswith(pre) or ("::" + pre) in base or pre in base.split("<")[0]:
            return e
    return "other:" + base.split("<")[0].split("(")[0][:40]


def find(d, pat):
    r = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)
    return r[-1] if r else None


def main():
    root, tag = sys.argv[1], sys.argv[2]
    # ---- kernel stats (default run: side stream on; "_serial": side stream off)
    for sub, suffix, cmd in (("prof_stats", "", "python bench.py --no-cpu-baseline"),
                             ("prof_stats_serial", "_serial", "C3D_WGRAD_SIDE=0 python bench.py --no-cpu-baseline")):
      f = find(os.path.join(root, sub), "*kernel_stats.csv")
      if f:
          rows = list(csv.DictReader(open(f)))
          agg = defaultdict(lambda: [0, 0.0])
          for r in rows:
              e = entry_of(r["Name"])
              agg[e][0] += int(r["Calls"])
              agg[e][1] += float(r["TotalDurationNs"])
          out = [{"entry": e, "calls": c, "total_ms": round(t / 1e6, 3), "avg_us": round(t / c / 1e3, 2)}
                 for e, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])]
          json.dump({"source": "rocprofv3 --kernel-trace --stats -- " + cmd,
                     "note": "all launches of the run (python bench.py --no-also: 40 settling + 5 warm-up + 20 timed steps + 2 set-up / per-kernel profile steps through the stage driver = 67 steps)",
                     "by_entry": out,
                     "top_kernels": [{"name": r["Name"][:160], "calls": int(r["Calls"]),
                                      "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                                      "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3),
                                      "pct": float(r["Percentage"])} for r in rows[:40]]},
                    open(os.path.join(root, f"{tag}_rocprof_kernel_stats{suffix}.json"), "w"), indent=1)
          print("kernel stats:", f)
    # ---- PMC traffic
    traffic = {}
    for key, sub, col, corr in (("fetch", "pmc_fetch", "FETCH_SIZE", 2.0), ("write", "pmc_write", "WRITE_SIZE", 1.0)):
        f = find(os.path.join(root, sub), "*counter_collection.csv")
        if not f:
            continue
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != col:
                continue
            e = entry_of(r["Kernel_Name"])
            agg[e][0] += 1
            agg[e][1] += float(r["Counter_Value"]) * 1024.0 * corr
        traffic[key] = {e: {"dispatches": c, "bytes_total": v} for e, (c, v) in agg.items()}
        print("pmc", key, f)
    if traffic:
        ents = sorted(set(traffic.get("fetch", {})) | set(traffic.get("write", {})))
        steps = max(1, (traffic.get("fetch") or traffic.get("write")).get("c3d_adam_step", {"dispatches": 1})["dispatches"])
        total = sum(v["bytes_total"] for k in ("fetch", "write") for v in traffic.get(k, {}).values())
        out = {}
        for e in ents:
            fd, wd = traffic.get("fetch", {}).get(e), traffic.get("write", {}).get(e)
            n = (fd or wd)["dispatches"]
            fb, wb = (fd or {"bytes_total": 0})["bytes_total"], (wd or {"bytes_total": 0})["bytes_total"]
            out[e] = {"dispatches": n, "fetch_bytes_per_launch": round(fb / n), "write_bytes_per_launch": round(wb / n),
                      "hbm_bytes_per_launch": round((fb + wb) / n)}
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace (separate passes) -- python bench.py "
                             "--no-cpu-baseline --no-graph --no-kernel-profile --steps 1 --warmup 1",
                   "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request); WRITE_SIZE x1",
                   "note": f"{steps} steps of B=32 bf16 (one Adam launch per step); per-launch averages over all dispatches of an entry",
                   "csrc_sha16": csrc_digest(),   # the kernel sources these counters were taken at (bench.py prints traffic: null when they differ)
                   "steps": steps, "hbm_bytes_per_step": round(total / steps),
                   "hbm_bytes_per_sample": round(total / steps / 32),
                   "by_entry_per_step": {e: round(sum(traffic.get(k, {}).get(e, {"bytes_total": 0})["bytes_total"] for k in ("fetch", "write")) / steps) for e in ents},
                   "by_entry": out}, open(os.path.join(root, f"{tag}_pmc_traffic.json"), "w"), indent=1)

    # ---- MFMA utilisation (own --pmc pass; utilisation = MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), the
    #      normalisation profiles/r01_pmc_mfma.json introduced)
    f = find(os.path.join(root, "pmc_mfma"), "*counter_collection.csv")
    if f:
        agg = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        for r in csv.DictReader(open(f)):
            e = entry_of(r["Kernel_Name"])
            agg[e][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[e].add(r["Dispatch_Id"])
        kernels = {}
        for e, c in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
            gui = c.get("GRBM_GUI_ACTIVE", 0.0)
            kernels[e] = {"dispatches": len(disp[e]), **{k: v for k, v in c.items()},
                          "mfma_utilisation": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui / 8.0), 4) if gui else None}
        json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace (own pass, side "
                             "stream off) -- python bench.py --no-cpu-baseline --no-graph --no-kernel-profile --steps 1 --warmup 1",
                   "note": "B=32 bf16; sums over all dispatches of an entry point's kernels",
                   "kernels": kernels}, open(os.path.join(root, f"{tag}_pmc_mfma.json"), "w"), indent=1)
        print("pmc mfma", f)


if __name__ == "__main__":
    main()
