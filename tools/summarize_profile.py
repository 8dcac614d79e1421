"""Summarise a `rocprofv3 --kernel-trace --stats --output-format csv` run of bench.py into a small markdown file.

  python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<name>.md [steps]

Reports (a) the per-kernel totals of the whole process and (b) the conv_igemm dispatches of bench.py's ROOFLINE PASS
only (the last steps*172 conv dispatches: eager, single stream, the ones bench.py brackets with HIP events), whose
average duration is the number to compare with roofline.avg_launch_us in the bench JSON line.
"""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oneshotdet_amd import workload  # noqa: E402


def summarize_train(d, out, line, stats, convs):
    r = line["roofline"]
    per = r["launches_per_step"]
    nst = int(re.search(r"(\d+) eager steps", r["measured"]).group(1))
    last = convs[-nst * per:]
    dur = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in last]
    flops = r["gflop_per_step"] * 1e9 * nst
    kinds = {}
    for x, du in zip(last, dur):
        k = "conv_wgrad_kernel" if "wgrad" in x["Kernel_Name"] else "conv_dma/xr/igemm (forward + data gradient)"
        kinds.setdefault(k, [0, 0.0])
        kinds[k][0] += 1
        kinds[k][1] += du
    with open(out, "w") as f:
        f.write("# rocprofv3 summary: `bench.py` (train mode, %s, steps %d, warmup %d)\n\n" % (line["dtype"], line["steps"], line["warmup"]))
        f.write("bench line: value %.2f %s, %.3f ms/step (%s)\n\n" % (line["value"], line["unit"], line["ms_per_step"], line["config"]["launch"]))
        f.write("## conv family (forward + data gradient + weight gradient), roofline pass (last %d x %d dispatches, eager, one stream)\n\n" % (nst, per))
        f.write("| source | launches | avg duration us | total ms/step | TFLOP/s |\n|---|---|---|---|---|\n")
        f.write("| rocprofv3 kernel trace | %d | %.2f | %.3f | %.2f |\n" % (len(dur), sum(dur) / len(dur), sum(dur) / nst / 1e3, flops / (sum(dur) * 1e-6) / 1e12))
        f.write("| bench.py HIP events (same process, under the profiler) | %d | %.2f | %.3f | %.2f |\n\n" % (per * nst, r["avg_launch_us"], r["conv_ms_per_step"], r["achieved"]))
        for k, (c, t) in kinds.items():
            f.write("* %s: %d launches/step, %.3f ms/step\n" % (k, c // nst, t / nst / 1e3))
        f.write("\n## per-kernel totals, whole process (tuning + warm-up + %d timed steps + roofline pass)\n\n" % line["steps"])
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for s_ in stats[:40]:
            name = re.sub(r"\(anonymous namespace\)::", "", s_["Name"])
            name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:100]
            f.write("| `%s` | %s | %.2f | %.1f | %s |\n" % (name, s_["Calls"], float(s_["TotalDurationNs"]) / 1e6, float(s_["AverageNs"]) / 1e3, s_["Percentage"]))
    print(open(out).read()[:1800])


def main():
    d, out = sys.argv[1], sys.argv[2]
    log = open(os.path.join(d, "bench.log")).read()
    line = json.loads([l for l in log.splitlines() if l.startswith('{"metric"')][-1])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else line["steps"]
    stats = list(csv.DictReader(open(os.path.join(d, "run_kernel_stats.csv"))))
    trace = [r for r in csv.DictReader(open(os.path.join(d, "run_kernel_trace.csv")))]
    is_train = "configs[2]" in line["config"]["workload"]
    convs = [r for r in trace if any(t in r["Kernel_Name"] for t in ("conv_igemm", "conv_dma", "conv_xr_kernel", "conv_p8_kernel")) or
             (is_train and "conv_wgrad_kernel" in r["Kernel_Name"])]
    launches = workload.conv_launches(8, 800, 1024, 8, 127, 127)
    per = (line.get("roofline") or {}).get("launches_per_step") or len(launches)   # the tuner may split grouped launches
    if is_train:
        return summarize_train(d, out, line, stats, convs)
    last = convs[-steps * per:]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
    flops = sum(2.0 * m * n * k for _, m, n, k in launches) * steps
    with open(out, "w") as f:
        f.write("# rocprofv3 summary: `bench.py %s`\n\n" % " ".join(
            "--%s %s" % (k, v) for k, v in (("dtype", line["dtype"]), ("steps", line["steps"]), ("warmup", line["warmup"]))))
        f.write("bench line: value %.2f %s, %.3f ms/step (%s)\n\n" % (line["value"], line["unit"], line["ms_per_step"],
                                                                     line["config"].get("launch", "")))
        f.write("## conv_igemm, roofline pass (last %d x %d dispatches, eager single stream)\n\n" % (steps, per))
        f.write("| source | launches | avg duration us | total ms/step | TFLOP/s |\n|---|---|---|---|---|\n")
        f.write("| rocprofv3 kernel trace | %d | %.2f | %.3f | %.2f |\n" % (
            len(dur), sum(dur) / len(dur), sum(dur) / steps / 1e3, flops / (sum(dur) * 1e-6) / 1e12))
        r = line.get("roofline") or {}
        f.write("| bench.py HIP events | %d | %.2f | %.3f | %.2f |\n\n" % (
            r.get("launches_per_step", 0) * steps, r.get("avg_launch_us", 0), r.get("conv_ms_per_step", 0),
            r.get("achieved", 0)))
        f.write("## per-kernel totals, whole process (warm-up + graph capture + %d replayed steps + roofline pass)\n\n"
                % line["steps"])
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for s in stats[:22]:
            name = re.sub(r"\(anonymous namespace\)::", "", s["Name"])
            name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:100]
            f.write("| `%s` | %s | %.2f | %.1f | %s |\n" % (name, s["Calls"], float(s["TotalDurationNs"]) / 1e6,
                                                            float(s["AverageNs"]) / 1e3, s["Percentage"]))
        f.write("\n## conv_igemm per layer (one roofline-pass step)\n\n| layer | M | N | K | tile | us | TFLOP/s |\n"
                "|---|---|---|---|---|---|---|\n")
        if per != len(launches):
            f.write("| (per-layer listing skipped: %d dispatches per step vs %d enumerated layers — the tuner split grouped launches) | | | | | | |\n" % (per, len(launches)))
            launches = []
        for (name, m, n, k), r in zip(launches, last[-per:]):
            du = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            mt = re.search(r"Li(\d+)ELi(\d+)ELi(\d+)E", r["Kernel_Name"]) or \
                re.search(r"<\w+, (\d+), (\d+), (\d+),", r["Kernel_Name"])
            tile = "x".join(mt.groups()) if mt else "?"
            f.write("| %s | %d | %d | %d | %s | %.1f | %.1f |\n" % (name, m, n, k, tile, du, 2.0 * m * n * k / du / 1e6))
    print(open(out).read()[:1500])


if __name__ == "__main__":
    main()
