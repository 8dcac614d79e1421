"""GPU box: the step's critical chain = the main stream (its hardware queue runs the target backbone, the cls tower, the loss and
their backward).  From a rocprofv3 kernel trace of bench.py: per timed step, how long the main queue was busy, how long it sat
between two of its kernels (waiting for another stream or for the host), and per kernel family the time inside the multi-stream
step against the same launches in the single-stream roofline pass that follows the timed region.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline
       python tools/main_chain.py DIR/.../run_kernel_trace.csv [timed steps] [roofline-pass steps]"""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-zA-Z_0-9]+?)I(DF16b|f)(.*)", n)
    if m:
        t = re.findall(r"Li(\d+)E", m.group(3))
        return m.group(1) + "<" + ",".join(t) + ">"
    return re.sub(r"<.*", "", n.split("(")[0])[:40]


rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n_timed = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n_roof = int(sys.argv[3]) if len(sys.argv) > 3 else 5
marks = [i for i, r in enumerate(rows) if "fcos_loss_finalize" in r["Kernel_Name"]]
main_q = rows[marks[-1]]["Queue_Id"]
# steps are cut at the loss: [loss of step k, loss of step k + 1) is one step's worth of the steady-state pipeline
roof = marks[-n_roof:]
timed = marks[-n_roof - n_timed:-n_roof]


def span(lo, hi):
    sel = [r for r in rows[lo:hi] if r["Queue_Id"] == main_q]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
    wall = int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
    gaps = collections.Counter()
    gap_total = 0
    for a, b in zip(sel, sel[1:]):
        g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        if g > 3000:          # more than a launch-to-launch cadence: the queue waited for something
            gaps[short(b["Kernel_Name"])] += g
            gap_total += g
    fam = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        f = fam[short(r["Kernel_Name"])]
        f[0] += 1
        f[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return wall, busy, gap_total, gaps, fam, len(sel)


def avg(idx):
    out = []
    for a, b in zip(idx, idx[1:]):
        out.append(span(a, b))
    return out


t, rf = avg(timed), avg(roof)
n = len(t)
print("main stream = queue %s; %d timed steps, %d roofline-pass steps (single stream: every kernel of the step is on this queue)" % (main_q, n, len(rf)))
print("timed steps: wall %.2f ms per step; main queue busy %.2f ms in %d launches, waiting (gaps > 3 us) %.2f ms" %
      (sum(x[0] for x in t) / n / 1e6, sum(x[1] for x in t) / n / 1e6, sum(x[5] for x in t) // n, sum(x[2] for x in t) / n / 1e6))
gaps = collections.Counter()
for x in t:
    gaps.update(x[3])
print("the main queue waited in front of (ms per step): " + ", ".join("%s %.2f" % (k, v / n / 1e6) for k, v in gaps.most_common(8)))
fam_t, fam_r = collections.defaultdict(lambda: [0, 0]), collections.defaultdict(lambda: [0, 0])
for x in t:
    for k, v in x[4].items():
        fam_t[k][0] += v[0]; fam_t[k][1] += v[1]
for x in rf:
    for k, v in x[4].items():
        fam_r[k][0] += v[0]; fam_r[k][1] += v[1]
print("\n| kernel on the main queue | launches/step | ms/step in the step | avg us | avg us of the same kernel in the single-stream pass (all streams' launches) |")
print("|---|---|---|---|---|")
for k, v in sorted(fam_t.items(), key=lambda kv: -kv[1][1])[:28]:
    r = fam_r.get(k)
    print("| `%s` | %.1f | %.3f | %.1f | %s |" % (k, v[0] / n, v[1] / n / 1e6, v[1] / v[0] / 1e3, "%.1f" % (r[1] / r[0] / 1e3) if r and r[0] else "-"))
