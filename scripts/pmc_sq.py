"""Where the wave cycles of each kernel go: two rocprofv3 --pmc passes over the SQ block (<= 8 counters each).
WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~= WAVE_CYCLES, all in quad-cycles
(MI355X_MICROARCH.md, counters); SQ_VALU_MFMA_BUSY_CYCLES is in cycles, SQ_BUSY_CU_CYCLES too.
python scripts/pmc_sq.py <dir pass 1> <dir pass 2>"""
import collections
import csv
import glob
import sys


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


a, b = load(sys.argv[1]), load(sys.argv[2])
print("# per-launch averages; shares are of SQ_WAVE_CYCLES (quad-cycles summed over waves); mfma = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)")
print("%-74s %5s %9s %6s %6s %6s %6s %6s %6s %6s %8s %8s %8s %7s" % (
    "kernel", "calls", "waves", "park", "stall", "activ", "valu", "lds", "vmem", "mfma", "valu/wv", "lds/wv", "mfma/wv", "ldsconf"))
rows = []
for k in a:
    if "yolo::" not in k:
        continue
    av = lambda d, n: (sum(d[k][n]) / len(d[k][n])) if d[k].get(n) else float("nan")
    wc = av(a, "SQ_WAVE_CYCLES")
    waves = av(b, "SQ_WAVES")
    rows.append((wc, k, len(a[k]["SQ_WAVE_CYCLES"]), waves, av(a, "SQ_WAIT_ANY") / wc, av(a, "SQ_WAIT_INST_ANY") / wc,
                 av(a, "SQ_ACTIVE_INST_ANY") / wc, av(a, "SQ_ACTIVE_INST_VALU") / wc, av(a, "SQ_ACTIVE_INST_LDS") / wc,
                 av(a, "SQ_ACTIVE_INST_VMEM") / wc, av(a, "SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * av(b, "SQ_BUSY_CU_CYCLES")),
                 av(b, "SQ_INSTS_VALU") / waves, av(b, "SQ_INSTS_LDS") / waves, av(b, "SQ_INSTS_MFMA") / waves,
                 av(b, "SQ_LDS_BANK_CONFLICT") / max(1.0, av(b, "SQ_LDS_IDX_ACTIVE"))))
for r in sorted(rows, reverse=True):
    print("%-74s %5d %9.0f %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f %8.0f %8.0f %8.0f %7.3f" % ((r[1][:74],) + r[2:]))
