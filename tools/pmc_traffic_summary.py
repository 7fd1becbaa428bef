#!/usr/bin/env python
"""Per-launch HBM traffic of the hot kernels from rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in
separate runs, as MI355X_MICROARCH.md section HBM prescribes).  gfx950 correction applied: FETCH_SIZE
counts 128-B requests at 64 B for wide coalesced reads -> doubled; units are KiB.  WRITE_SIZE is
uncalibrated (guide) and reported as-is.

    python tools/pmc_traffic_summary.py gpurun_out profiles/r01_pmc_traffic.json
"""
import collections
import csv
import json
import sys


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[(name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))].append(float(r["Counter_Value"]))
    return agg


def main(root, out):
    fetch = load(f"{root}/pmc_fetch/p_counter_collection.csv", "FETCH_SIZE")
    write = load(f"{root}/pmc_write/p_counter_collection.csv", "WRITE_SIZE")
    hit = load(f"{root}/pmc_l2/p_counter_collection.csv", "TCC_HIT_sum")
    miss = load(f"{root}/pmc_l2/p_counter_collection.csv", "TCC_MISS_sum")
    rows = []
    for key in sorted(fetch, key=lambda k: -sum(fetch[k])):
        name, wgs = key
        if not name.startswith("pe::"):
            continue
        f = sum(fetch[key]) / len(fetch[key])
        w = sum(write.get(key, [0])) / max(len(write.get(key, [0])), 1)
        h = sum(hit.get(key, [0])); m = sum(miss.get(key, [0]))
        rows.append({"kernel": name, "workgroups": wgs, "launches": len(fetch[key]),
                     "fetch_bytes_per_launch_corrected": 2 * f * 1024, "write_bytes_per_launch": w * 1024,
                     "hbm_bytes_per_launch": (2 * f + w) * 1024,
                     "l2_hit_rate": h / (h + m) if h + m else None})
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum (separate passes) on "
                         "`bench.py --layers 3 --inference-steps 3` (same kernels and shapes as the full bench)",
               "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), KiB -> bytes; WRITE_SIZE uncalibrated",
               "kernels": rows}, open(out, "w"), indent=1)
    for r in rows[:14]:
        print(f"{r['kernel'][:44]:44s} wg={r['workgroups']:5d} n={r['launches']:4d} fetch {r['fetch_bytes_per_launch_corrected']/1e6:9.1f} MB "
              f"write {r['write_bytes_per_launch']/1e6:8.1f} MB  L2 hit {r['l2_hit_rate'] if r['l2_hit_rate'] is None else round(r['l2_hit_rate'],3)}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
