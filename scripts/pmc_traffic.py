"""HBM-side bytes per launch per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; both in KiB).
gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes (MI355X_MICROARCH.md, HBM) -> doubled here.
python scripts/pmc_traffic.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> [table.json batch size "command"]
With the last four arguments also writes the table bench.py reads (profiles/traffic_table.json), stamped with the hash of
the kernel sources it was measured on."""
import collections
import csv
import glob
import json
import os
import sys


def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print("# per-launch averages; fetch_MB = 2 x FETCH_SIZE KiB (gfx950 correction), write_MB = WRITE_SIZE KiB")
print("%-88s %7s %10s %10s %10s" % ("kernel", "calls", "fetch_MB", "write_MB", "total_MB"))
rows = []
for k in fetch:
    f = 2.0 * sum(fetch[k]) / len(fetch[k]) * 1024 / 1e6
    w = sum(write.get(k, [0.0])) / max(1, len(write.get(k, [0.0]))) * 1024 / 1e6
    rows.append((f + w, k, len(fetch[k]), f, w))
for tot, k, n, f, w in sorted(rows, reverse=True):
    print("%-88s %7d %10.2f %10.2f %10.2f" % (k[:88], n, f, w, tot))
if len(sys.argv) > 6:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    table = {"sources_sha": bench.kernel_sources_sha(), "batch": int(sys.argv[4]), "size": int(sys.argv[5]), "command": sys.argv[6],
             "kernels": {k: {"calls": n, "fetch_bytes": f * 1e6, "write_bytes": w * 1e6, "total_bytes": tot * 1e6}
                         for tot, k, n, f, w in rows if "yolo::" in k}}
    json.dump(table, open(sys.argv[3], "w"), indent=1, sort_keys=True)
