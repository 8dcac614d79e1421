#!/bin/bash
# usage (GPU box): tools/rounds_table.sh <tag> — "a launch is rounds of workgroups": rocprofv3 kernel trace of the training step ->
# per kernel and grid size: workgroups, workgroups per CU the kernel's registers / LDS allow, rounds = workgroups / (256 CUs x that),
# launches per step, mean duration -> gpurun_out/<tag>_rounds.txt.  A fractional round far from 1.0 on a long launch is a candidate for
# grouping launches or another tile height.
TAG=${1:-r5}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; D=$O/rounds_trace
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
export OSD_TUNER_CACHE=$O/osd_tuner_${TAG}_rounds.json; rm -f $OSD_TUNER_CACHE
python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-conv-timing > /dev/null 2>&1      # writes the tuner cache
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -o run -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-conv-timing > $D/run.log 2>&1
python3 - <<PY > $O/${TAG}_rounds.txt
import collections, csv, glob
f = glob.glob("$D/**/run_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# the timed steps: between the fcos_loss_finalize launches, take the last 4 steps' worth
idx = [i for i, r in enumerate(rows) if "fcos_loss_finalize" in r["Kernel_Name"]]
lo, hi = idx[-5], idx[-1]
agg = collections.defaultdict(lambda: [0, 0.0])
meta = {}
for r in rows[lo:hi]:
    wg = int(r["Workgroup_Size"]) if "Workgroup_Size" in r else int(r["Workgroup_Size_X"])
    grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))
    nwg = grid // max(wg, 1)
    lds = int(r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", 0)) or 0)
    vg = int(r.get("VGPR_Count", r.get("Arch_VGPR_Count", 0)) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0)
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-58:]
    key = (name, nwg, wg)
    a = agg[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    waves = max(wg // 64, 1)
    by_lds = (160 * 1024) // lds if lds > 0 else 99
    alloc = max(((vg + 7) // 8) * 8, 8)
    per_simd = min(8, 512 // alloc)
    by_reg = (per_simd * 4) // waves if waves <= per_simd * 4 else 0
    meta[key] = (lds, vg, max(1, min(by_lds, by_reg, 32)))
print("timed steps: 4; CUs 256.  rounds = workgroups / (256 x workgroups per CU by STATIC LDS and registers).  The trace does not report dynamic LDS:")
print("conv_sp_kernel (151 KB), conv_wgrad_sk_kernel (131 KB) and the 128 x 256 / 256 x 256 conv_dma tiles (98 - 147 KB) run ONE workgroup per CU: halve their /CU, double their rounds.")
print("%-58s %7s %5s %7s %5s %4s %7s %6s %9s %9s" % ("kernel", "wgs", "thr", "LDS", "VGPR", "/CU", "rounds", "n/step", "us", "ms/step"))
out = []
for key, (n, us) in agg.items():
    lds, vg, occ = meta[key]
    out.append((us / 4.0, key, n / 4.0, us / n, lds, vg, occ))
for tot, (name, nwg, wg), n, us, lds, vg, occ in sorted(out, reverse=True)[:70]:
    print("%-58s %7d %5d %7d %5d %4d %7.2f %6.1f %9.1f %9.3f" % (name, nwg, wg, lds, vg, occ, nwg / (256.0 * occ), n, us, tot / 1e3))
PY
head -3 $D/run.log | cut -c1-150
rm -rf $D
cat $O/${TAG}_rounds.txt
