#!/usr/bin/env python
"""Summarise the passes of tools/pmc_collect.sh into one tracked JSON (profiles/r03_pmc.json): per dominant kernel and launch
shape, L2-miss (fabric-side) traffic per launch, L2 hit rate, and matrix-pipe utilisation.

  * FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B for wide coalesced reads -- MI355X_MICROARCH.md section
    HBM) and converted KiB -> bytes; WRITE_SIZE is uncalibrated and reported as it is.  Infinity-Cache hits are counted,
    so `traffic` is what leaves the XCD's L2, an upper bound on HBM bytes.
  * mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed
    over the chip's SIMDs (32 per v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (5.99 M for a
    0.37 ms launch = 8 x 2.0 GHz), hence the / 8.  The JSON also carries counter / (algorithmic MFMA count x 32) as a
    calibration of the numerator (1.014 measured: the counter is trustworthy).

    python tools/pmc_summary.py gpurun_out/pmc_r02 profiles/r02_pmc.json
"""
import collections
import csv
import glob
import json
import os
import sys

SIMDS = 256 * 4
XCDS = 8


def load(root, name):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not k.startswith("pe::"):
                continue
            key = (k, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and "End_Timestamp" in r and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                agg[key]["_duration_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return agg


def mean(v):
    return sum(v) / len(v) if v else None


# algorithmic FLOPs of the block GEMMs at the bench geometry, by (epilogue, work-groups): mean over T_pos = 512 / T_neg = 272
def gemm_flops(epi, wgs, sched=15):
    S = 8192 + (512 + 272) / 2
    if sched in (17, 21) and wgs == 256:        # persistent grid (one work-group per CU): the shape is told by the epilogue
        return {1: 2 * S * 12288 * 3072, 4: 2 * S * 9216 * 3072}.get(epi)
    return {(1, 1632): 2 * S * 12288 * 3072, (4, 1224): 2 * S * 9216 * 3072, (3, 408): 2 * S * 3072 * (3072 + 12288) / 2}.get((epi, wgs))


def main(root, out):
    cmd = open(os.path.join(root, "command.txt")).read().strip()
    fetch, write, l2, sq, grbm = (load(root, n) for n in ("fetch", "write", "l2", "sq", "grbm"))
    rows = []
    for key in sorted(fetch, key=lambda k: -sum(fetch[k]["FETCH_SIZE"])):
        name, wgs = key
        f, w = mean(fetch[key]["FETCH_SIZE"]), mean(write.get(key, {}).get("WRITE_SIZE", []))
        h, m = sum(l2.get(key, {}).get("TCC_HIT_sum", [])), sum(l2.get(key, {}).get("TCC_MISS_sum", []))
        busy, gui = mean(sq.get(key, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", [])), mean(grbm.get(key, {}).get("GRBM_GUI_ACTIVE", []))
        row = {"kernel": name, "workgroups": wgs, "launches": len(fetch[key]["FETCH_SIZE"]),
               "fetch_bytes_per_launch_corrected": 2 * f * 1024, "write_bytes_per_launch": None if w is None else w * 1024,
               "traffic_bytes_per_launch": (2 * f + (w or 0)) * 1024, "l2_hit_rate": h / (h + m) if h + m else None,
               "sq_valu_mfma_busy_cycles": busy, "grbm_gui_active": gui,
               "mfma_busy": busy / (gui / XCDS * SIMDS) if busy and gui else None}
        dur = mean(grbm.get(key, {}).get("_duration_ns", []))
        if dur and gui:
            row["duration_us_while_profiled"] = dur / 1e3
            row["effective_clock_ghz"] = gui / XCDS / dur          # GRBM_GUI_ACTIVE (per XCD) / wall time of the launch
        for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            row[c.lower()] = mean(sq.get(key, {}).get(c, []))
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS"):
            row[c.lower()] = mean(grbm.get(key, {}).get(c, []))
        if "gemm_bf16_kernel" in name and "<" in name:
            targs = name.split("<")[1].split(">")[0].split(",")
            epi, sched = int(targs[0]), int(targs[1]) if len(targs) > 1 else 15
            fl = gemm_flops(epi, wgs, sched)
            if fl and busy:
                row["algorithmic_gflop_per_launch"] = fl / 1e9
                row["mfma_busy_counter_over_algorithmic_mfma_cycles"] = busy / (fl / (2 * 32 * 32 * 16) * 32)      # pipe cycles: 32 per 32 KiFLOP in either MFMA shape
        rows.append(row)
    json.dump({"command": cmd, "config": {"layers": 60, "height": 1024, "width": 1024, "t_pos": 512, "t_neg": 272, "fp8": "--fp8" in cmd,
                                           "dual_stream": "--dual-stream" in cmd},
               "source": "rocprofv3 --kernel-trace --pmc, one pass per counter group (tools/pmc_collect.sh)",
               "corrections": "FETCH_SIZE x2 and KiB -> bytes (gfx950); WRITE_SIZE uncalibrated; Infinity-Cache hits are inside `traffic`",
               "kernels": rows}, open(out, "w"), indent=1)
    for r in rows[:10]:
        print(f"{r['kernel'][:42]:42s} wg={r['workgroups']:5d} n={r['launches']:4d} traffic {r['traffic_bytes_per_launch']/1e6:8.1f} MB "
              f"L2 hit {r['l2_hit_rate'] and round(r['l2_hit_rate'], 3)}  mfma_busy {r['mfma_busy'] and round(r['mfma_busy'], 3)} "
              f"calib {r.get('mfma_busy_counter_over_algorithmic_mfma_cycles') and round(r['mfma_busy_counter_over_algorithmic_mfma_cycles'], 3)}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
