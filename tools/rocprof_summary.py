"""Prints a text summary of rocprofv3 rocpd databases (the default output of
`rocprofv3 --kernel-trace --stats` and `rocprofv3 --pmc ...` on ROCm 7.2):
per-kernel call counts and durations, and per-dispatch counter values.

    python tools/rocprof_summary.py gpurun_out/prof/stats/c4_results.db [more.db ...]
"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print("== %s" % path)
        rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        if rows:
            print("%-90s %6s %14s %14s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
            for name, calls, total, avg, pct in rows:
                print("%-90s %6d %14.3f %14.3f %7.3f" % (name[:90], calls, total, avg, pct))
        rows = list(cur.execute(
            "select dispatch_id, kernel_name, counter_name, value, duration, vgpr_count, sgpr_count, lds_block_size, "
            "grid_size, workgroup_size from counters_collection order by dispatch_id, counter_name"))
        last = None
        for did, kname, cname, value, dur, vgpr, sgpr, lds, grid, wg in rows:
            if did != last:
                print("dispatch %d  %s  duration_ns=%d vgpr=%d sgpr=%d lds=%d grid=%d wg=%d"
                      % (did, kname[:70], dur, vgpr, sgpr, lds, grid, wg))
                last = did
            print("    %-24s %.6g" % (cname, value))
        db.close()


if __name__ == "__main__":
    main()
