"""Per-kernel means of every counter of a `rocprofv3 --pmc ... --kernel-trace` pass (tools/prof_pmc_generic.sh):
   python tools/pmc_kernel_table.py gpurun_out/pmcg_<tag> [name filter]"""
import collections
import csv
import sys


def main():
    d = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(d + "/run_counter_collection.csv")):
        k = per[r["Dispatch_Id"]]
        k["name"] = r["Kernel_Name"]
        k[r["Counter_Name"]] = float(r["Counter_Value"])
        k["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for k in per.values():
        n = k["name"].replace("(anonymous namespace)::", "").split("(")[0][-48:]
        if flt and flt not in k["name"]:
            continue
        agg[n]["n"] += 1
        for c, v in k.items():
            if c != "name":
                agg[n][c] += v
    for n, a in sorted(agg.items(), key=lambda t: -t[1]["us"]):
        cnt = a.pop("n")
        print("%-50s x%d  " % (n, cnt) + "  ".join("%s=%.4g" % (c, v / cnt) for c, v in sorted(a.items())))


if __name__ == "__main__":
    main()
