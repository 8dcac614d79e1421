"""GPU box: kernel trace of a short bench run, printed as a per-stream sequence around one kernel (default: the loss finalize of the
LAST timed step) — name, queue, start offset, duration.  usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o run -- python3 bench.py ...;
python tools/trace_window.py DIR/…/run_kernel_trace.csv [anchor substring] [kernels after] [occurrence from the end]"""
import csv
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-zA-Z_0-9]+?)I(DF16b|f)(.*)", n)
    if m:
        t = re.findall(r"Li(\d+)E", m.group(3))
        return m.group(1) + "<" + ",".join(t) + ">"
    return n.split("(")[0][:48]


rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2] if len(sys.argv) > 2 else "fcos_loss_finalize"
after = int(sys.argv[3]) if len(sys.argv) > 3 else 60
back = int(sys.argv[4]) if len(sys.argv) > 4 else 7
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0 = idx[-back]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[max(0, i0 - 6):i0 + after]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +%7.1f us  q%-3s grid %-8s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r.get("Grid_Size", "?"), short(r["Kernel_Name"])))
