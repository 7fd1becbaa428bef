#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a small committed table.

    python tools/rocprof_summary.py gpurun_out/prof_r01/bench_results.db profiles/r01_kernel_stats.md
"""
import sqlite3
import sys


def main(dbpath, outpath):
    db = sqlite3.connect(dbpath)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    span = cur.execute("select min(start), max(end) from kernels").fetchone()
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % of GPU kernel time |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) < 110 else name[:107] + "..."
        lines.append(f"| `{short}` | {n} | {tot/1e6:.2f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*tot/total:.2f} |")
    # the dominant GEMM broken down by grid size (= by problem shape)
    g = cur.execute(
        "select name, grid_x/workgroup_x, count(*), avg(duration) from kernels where name like '%gemm_bf16_kernel%' "
        "group by name, grid_x order by sum(duration) desc limit 12").fetchall()
    with open(outpath, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace summary\n\nsource: `{dbpath}`  \n"
                f"sum of kernel durations {total/1e9:.3f} s; first-start to last-end {(span[1]-span[0])/1e9:.3f} s\n\n")
        f.write("\n".join(lines) + "\n\n## gemm_bf16_kernel by launch shape (work-groups per launch)\n\n"
                "| kernel | work-groups | calls | avg us |\n|---|---|---|---|\n")
        for name, wg, n, avg in g:
            ep = name.split("<")[1].split(">")[0] if "<" in name else (name.split("ILi")[1].split("E")[0] if "ILi" in name else "?")
            f.write(f"| gemm_bf16_kernel<{ep}> (EPI, schedule, e4m3, 16 x 16 MFMA shape) | {wg} | {n} | {avg/1e3:.1f} |\n")
    print(open(outpath).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
