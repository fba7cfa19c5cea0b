"""Compact table of rocprofv3 PMC results: one row per dispatch of kernels whose name
contains the given substring, one column per counter.

    python tools/pmc_table.py <substring> a_results.db [b_results.db ...]
"""
import sqlite3
import sys


def main():
    pat = sys.argv[1]
    for path in sys.argv[2:]:
        db = sqlite3.connect(path)
        rows = list(db.execute(
            "select dispatch_id, kernel_name, counter_name, value, duration, grid_size, lds_block_size, vgpr_count, "
            "accum_vgpr_count from counters_collection where kernel_name like ? order by dispatch_id, counter_name",
            ("%" + pat + "%",)))
        by = {}
        for did, kn, cn, val, dur, grid, lds, vg, ag in rows:
            by.setdefault(did, dict(dur=dur, grid=grid, lds=lds, vg=vg, ag=ag, name=kn[:40]))[cn] = val
        print("== " + path)
        for did, d in by.items():
            head = "disp %d %s grid=%d lds=%d vgpr=%d+%d dur_us=%.0f" % (did, d["name"], d["grid"], d["lds"], d["vg"], d["ag"], d["dur"] / 1e3)
            vals = " ".join("%s=%.4g" % (k, v) for k, v in d.items() if k not in ("dur", "grid", "lds", "vg", "ag", "name"))
            print(head + " | " + vals)


if __name__ == "__main__":
    main()
