#!/bin/bash
# usage (GPU box): tools/pmc_traffic_by_kernel.sh <tag>  — per-kernel HBM bytes of the training step: separate rocprofv3 --pmc FETCH_SIZE /
# WRITE_SIZE passes of `bench.py --steps 4 --warmup 2` with one tuner cache, per-kernel means joined -> gpurun_out/<tag>_pmc_traffic_by_kernel.txt
TAG=${1:-r5}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export OSD_TUNER_CACHE=$O/osd_tuner_${TAG}_bk.json
rm -f $OSD_TUNER_CACHE; rm -rf $O/pmcg_fetch $O/pmcg_write
python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-conv-timing > /dev/null 2>&1      # writes the tuner cache
bash $R/tools/prof_pmc_generic.sh fetch "FETCH_SIZE" bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-conv-timing > /dev/null 2>&1
bash $R/tools/prof_pmc_generic.sh write "WRITE_SIZE" bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-conv-timing > /dev/null 2>&1
python3 - <<PY > $O/${TAG}_pmc_traffic_by_kernel.txt
import collections, csv
each = collections.defaultdict(list)      # the weight-gradient dispatches one by one (the tower, stage and query launches share kernels)
def load(d, ctr):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open("$O/pmcg_%s/run_counter_collection.csv" % d)):
        if r["Counter_Name"] != ctr: continue
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]
        a = agg[n]; a[0] += 1; a[1] += float(r["Counter_Value"]) * 1024.0; a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if "wgrad" in r["Kernel_Name"]: each[(d, n)].append((float(r["Counter_Value"]) * 1024.0, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)))
    return agg
f, w = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
rows = []
for n in f:
    cnt, fb, us = f[n]
    wb = w.get(n, [1, 0.0, 0.0])[1] / max(w.get(n, [1])[0], 1)
    rows.append((us, n, cnt, 2.0 * fb / cnt / 1e6, wb / 1e6, us / cnt))
print("per-kernel means over the whole process (tuning step + 2 warm-up + 4 timed steps); read = 2 x FETCH_SIZE (gfx950), MB per dispatch")
print("%-62s %6s %10s %10s %9s %8s" % ("kernel", "n", "read MB", "write MB", "us", "TB/s"))
for us, n, cnt, rd, wr, u in sorted(rows, reverse=True)[:45]:
    print("%-62s %6d %10.1f %10.1f %9.1f %8.2f" % (n, cnt, rd, wr, u, (rd + wr) / max(u, 1e-9)))
print()
print("weight-gradient dispatches of the LAST step, one by one (read MB = 2 x FETCH_SIZE | write MB | us | workgroups):")
for (d, n) in sorted(k for k in each if k[0] == "fetch"):
    fr, wr_ = each[("fetch", n)], each.get(("write", n), [])
    per_step = max(1, len(fr) // 7)
    for i in range(len(fr) - per_step, len(fr)):
        w_ = wr_[i][0] / 1e6 if i < len(wr_) else float("nan")
        print("  %-58s %9.1f %9.1f %9.1f %6d" % (n[-58:], 2.0 * fr[i][0] / 1e6, w_, fr[i][1], fr[i][2]))
PY
rm -rf $O/pmcg_fetch $O/pmcg_write
cat $O/${TAG}_pmc_traffic_by_kernel.txt
