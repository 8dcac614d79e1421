"""GPU box helper: per-dispatch SQ_VALU_MFMA_BUSY_CYCLES of the dominant launch (conv_sp_kernel, 500 workgroups) from a
`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass — histogram of the issued GFLOP per dispatch (busy cycles / 16 MFMA
cycles x 16,384 FLOP), to see whether every dispatch closes at the algorithmic 151.0.  usage: mfma_dom_rows.py <pmcg dir>"""
import collections
import csv
import sys

per = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1] + "/run_counter_collection.csv")):
    if "conv_sp_kernel" not in r["Kernel_Name"] or int(r.get("Grid_Size", 0) or 0) != 500 * 512:
        continue
    per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
h = collections.Counter()
for d, k in sorted(per.items(), key=lambda kv: int(kv[0])):
    gf = k.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 16 * 16384 / 1e9
    h[round(gf, 0)] += 1
print("dispatches %d; issued GFLOP per dispatch -> count: %s" % (len(per), sorted(h.items())))
