"""python3 scripts/trace_last.py <kernel_trace.csv> <launches per repetition>: the kernels of the last repetition with durations."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
per = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
tot = 0.0
for r in rows[-per:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print("%8.1f us grid %8s  %s" % (d, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r["Kernel_Name"][:100]))
print("total %.1f us over %d kernels" % (tot, per))
