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


def short_name(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:110]


def phases_of(trace, line):
    """Split the kernel trace of a train-mode bench.py run into its phases.  Every step launches osd_fcos_loss_finalize
    exactly once, so the n-th such dispatch closes step n: step 0 = the tuning pass (each conv shape timed over all candidate
    algorithms), then `warmup` steps, `steps` timed steps, then the roofline pass (eager, ONE stream, every conv / correlate
    launch bracketed by HIP events) — the launches bench.py's roofline figures are measured on."""
    trace = sorted(trace, key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(trace) if "fcos_loss_finalize" in r["Kernel_Name"]]
    w, k = line["warmup"], line["steps"]
    nst = int(re.search(r"(\d+) eager steps", line["roofline"]["measured"]).group(1))
    assert len(marks) >= 1 + w + k + nst, (len(marks), w, k, nst)
    cut = lambda j: marks[j] + 1            # noqa: E731   first dispatch after step j's loss
    # the roofline pass: bench.py parks the stream behind a 150 M-cycle spin kernel before each of its nst steps — the pass is
    # everything from the first of the last nst long spin kernels on.  (Until round 5 it was cut at the previous step's loss
    # launch and so held the BACKWARD half of the last timed step too: ~5.5 steps of kernels divided by 5.)
    spins = [i for i, r in enumerate(trace) if "spin_kernel" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 30e6]
    roof0 = spins[-nst] if len(spins) >= nst else cut(len(marks) - nst - 1)
    ph = {"tuning": trace[:cut(0)], "warm-up": trace[cut(0):cut(w)], "timed steps": trace[cut(w):cut(w + k)],
          "roofline pass": [r for r in trace[roof0:] if "spin_kernel" not in r["Kernel_Name"]]}
    return ph, nst


def summarize_train(d, out, line, stats, trace):
    r = line["roofline"]
    ph, nst = phases_of(trace, line)
    dur = lambda x: (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3      # noqa: E731
    is_conv = lambda n: any(t in n for t in ("conv_igemm", "conv_dma", "conv_sp_kernel", "conv_pred_kernel", "conv_pw_kernel", "conv_px_kernel", "conv_wgrad_kernel",   # noqa: E731
                                             "conv_wgrad_sk_kernel", "conv_wgrad_xr_kernel"))
    is_corr = lambda n: "correlate_levels_kernel" in n or ("correlate_kernel" in n)      # noqa: E731
    per = {}
    for name, rows in ph.items():
        for x in rows:
            e = per.setdefault(short_name(x["Kernel_Name"]), {}).setdefault(name, [0, 0.0])
            e[0] += 1
            e[1] += dur(x)
    roof = [x for x in ph["roofline pass"] if is_conv(x["Kernel_Name"])]
    rdur = [dur(x) for x in roof]
    flops = r["gflop_per_step"] * 1e9 * nst
    corr_r = [dur(x) for x in ph["roofline pass"] if is_corr(x["Kernel_Name"])]
    corr_t = [dur(x) for x in ph["timed steps"] if is_corr(x["Kernel_Name"])]
    rc = line.get("roofline_correlation") or {}
    with open(out, "w") as f:
        f.write("# rocprofv3 summary: `bench.py` (train mode, %s, steps %d, warmup %d)\n\n" % (line["dtype"], line["steps"], line["warmup"]))
        f.write("bench line (under the profiler): value %.2f %s, %.3f ms/step (%s)\n\n" % (line["value"], line["unit"], line["ms_per_step"], line["config"]["launch"]))
        f.write("Phases are cut at the one `fcos_loss_finalize` launch per step: tuning (1 step, every conv shape timed over all "
                "candidate algorithms), warm-up (%d), timed steps (%d, multi-stream: kernels of different streams overlap and stretch "
                "each other), roofline pass (%d steps, eager on ONE stream = the launches bench.py brackets with HIP events).\n\n"
                % (line["warmup"], line["steps"], nst))
        f.write("## conv family (forward + data gradient + weight gradient), roofline pass\n\n")
        f.write("| source | launches | avg duration us | total ms/step | TFLOP/s |\n|---|---|---|---|---|\n")
        f.write("| rocprofv3 kernel trace, roofline-pass dispatches | %d | %.2f | %.3f | %.2f |\n" % (len(rdur), sum(rdur) / max(len(rdur), 1), sum(rdur) / nst / 1e3, flops / (sum(rdur) * 1e-6) / 1e12))
        f.write("| bench.py HIP events (same process) | %d | %.2f | %.3f | %.2f |\n\n" % (r["launches_per_step"] * nst, r["avg_launch_us"], r["conv_ms_per_step"], r["achieved"]))
        if corr_r:
            bpl = rc.get("bytes_per_launch", 0)
            f.write("## correlation (`correlate_levels_kernel`: all five FPN levels per launch; %.1f MB algorithmic bytes per launch)\n\n" % (bpl / 1e6))
            f.write("| where | launches | avg duration us | GB/s | fraction of 8 TB/s |\n|---|---|---|---|---|\n")
            for tag, ds in (("roofline pass (one stream; rocprofv3)", corr_r), ("timed steps (in the multi-stream step; rocprofv3)", corr_t)):
                if ds:
                    avg = sum(ds) / len(ds)
                    f.write("| %s | %d | %.2f | %.0f | %.3f |\n" % (tag, len(ds), avg, bpl / avg / 1e3, bpl / avg / 1e3 / 8000.0))
            f.write("| bench.py HIP events, roofline pass | %d | %.2f | %.0f | %.3f |\n\n" % (rc.get("launches", 0), rc.get("avg_launch_us", 0), rc.get("achieved", 0), rc.get("frac", 0)))
        f.write("## per kernel and phase: launches, average duration (us)\n\n")
        f.write("| kernel | timed steps: n | avg us | ms/step | roofline pass: n | avg us | tuning: n | total ms |\n|---|---|---|---|---|---|---|---|\n")
        order = sorted(per.items(), key=lambda kv: -kv[1].get("timed steps", [0, 0.0])[1])
        for name, e in order[:45]:
            t, rp, tu = e.get("timed steps", [0, 0.0]), e.get("roofline pass", [0, 0.0]), e.get("tuning", [0, 0.0])
            f.write("| `%s` | %d | %.1f | %.3f | %d | %.1f | %d | %.1f |\n" % (
                name, t[0], t[1] / max(t[0], 1), t[1] / line["steps"] / 1e3, rp[0], rp[1] / max(rp[0], 1), tu[0], tu[1] / 1e3))
    print(open(out).read()[:2500])


def main():
    d, out = sys.argv[1], sys.argv[2]
    log = open(os.path.join(d, "bench.log")).read()
    line = json.loads([l for l in log.splitlines() if l.startswith('{"metric"')][-1])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else line["steps"]
    stats = list(csv.DictReader(open(os.path.join(d, "run_kernel_stats.csv"))))
    trace = [r for r in csv.DictReader(open(os.path.join(d, "run_kernel_trace.csv")))]
    is_train = "configs[2]" in line["config"]["workload"]
    convs = [r for r in trace if any(t in r["Kernel_Name"] for t in ("conv_igemm", "conv_dma", "conv_sp_kernel", "conv_pred_kernel", "conv_pw_kernel", "conv_px_kernel")) or
             (is_train and any(t in r["Kernel_Name"] for t in ("conv_wgrad_kernel", "conv_wgrad_sk_kernel", "conv_wgrad_xr_kernel")))]
    launches = workload.conv_launches(8, 800, 1024, 8, 127, 127)
    per = (line.get("roofline") or {}).get("launches_per_step") or len(launches)   # the tuner may split grouped launches
    if is_train:
        return summarize_train(d, out, line, stats, trace)
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
