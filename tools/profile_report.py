"""Writes the text summary of one tools/profile_round.sh run that goes under profiles/:
the command, the bench line, the kernel table of the --kernel-trace --stats run and one row
per dispatch of the dominant kernel for every --pmc pass.

    python tools/profile_report.py <tag> <kernel substring> <bench args...>  > profiles/<name>.txt
"""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_table  # noqa: E402
import rocprof_summary  # noqa: E402


def capture(mod, argv):
    old = sys.argv
    sys.argv = argv
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            mod.main()
    finally:
        sys.argv = old
    return buf.getvalue()


def main():
    tag, kernel = sys.argv[1], sys.argv[2]
    args = " ".join(sys.argv[3:])
    prof = os.path.join("gpurun_out", "prof")
    print("rocprofv3 --kernel-trace --stats and separate --pmc passes (instruction mix; FETCH_SIZE; WRITE_SIZE; matrix pipe) of: "
          "python3 bench.py %s   (one MI355X; tools/profile_round.sh, summarised by tools/profile_report.py)" % args)
    with open(os.path.join(ROOT, prof, tag + "stats.log")) as f:
        for line in f:
            if line.startswith('{"metric"'):
                print(line.strip())
    sys.stdout.write(capture(rocprof_summary, ["x", os.path.join(prof, tag + "stats", "r_results.db")]).split("dispatch ")[0])
    for p in ("pmc1", "pmc2", "pmc3", "pmc4"):
        path = os.path.join(prof, tag + p, "r_results.db")
        if os.path.exists(os.path.join(ROOT, path)):
            sys.stdout.write(capture(pmc_table, ["x", kernel, path]))


if __name__ == "__main__":
    main()
