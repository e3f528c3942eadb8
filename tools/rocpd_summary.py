#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats ... writes *_results.db on ROCm 7.2)
into the per-kernel table the judge reads: calls, total/avg/min/max duration, share, registers, grid.
Launches of one kernel are grouped by grid (x = items of the pass, y = number of jobs per launch), so the launches of the
timed statement (2^19-item passes) are not averaged with the input generator's (2^16) or the host-pointer pass's (2^17 slices).
Usage: tools/rocpd_summary.py gpurun_out/prof1/c2_results.db > profiles/rNN_name.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tables if "kernel_dispatch" in t)
    ks = next(t for t in tables if "kernel_symbol" in t)
    rows = db.execute(f"""
        select s.display_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start),
               max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size),
               max(d.grid_size_x), max(d.grid_size_y), max(d.workgroup_size_x)
        from {kd} d join {ks} s on d.kernel_id = s.id group by s.display_name, d.grid_size_x, d.grid_size_y order by 3 desc""").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("%-34s %6s %12s %10s %10s %10s %6s %5s %5s %5s %7s %8s %14s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "pct",
                                                                    "vgpr", "agpr", "sgpr", "lds_B", "scratch", "grid(x,y)/wg"))
    for r in rows:
        name = (r[0] or "?").split("(")[0]
        print("%-34s %6d %12.3f %10.4f %10.4f %10.4f %6.2f %5s %5s %5s %7s %8s %14s" % (
            name[:34], r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6, 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10],
            "%d,%d/%d" % (r[11], r[12], r[13])))


if __name__ == "__main__":
    main(sys.argv[1])
