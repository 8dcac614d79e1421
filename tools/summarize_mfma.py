"""Summarise a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace` pass of bench.py
(tools/prof_pmc_generic.sh mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" bench.py ...) into profiles/<name>.md:
per conv kernel the share of SIMD-cycles in which the matrix pipe was busy.
  SQ_VALU_MFMA_BUSY_CYCLES  cycles, summed over all SIMDs (16 per v_mfma_f32_16x16x32_bf16: MI355X_MICROARCH.md, cycle constants)
  GRBM_GUI_ACTIVE           active cycles summed over the 8 XCDs -> / 8 = the dispatch's cycles at the clock it ran at
  MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
Round 4: durations are NOT taken from the counter-collection pass (its timestamps are not the kernel trace's: round 3's table
implied 1.9 PFLOP/s for the dominant launch).  With a third argument — the run_kernel_trace.csv of a plain --kernel-trace run of
the SAME command with the SAME tuner choices (OSD_TUNER_CACHE, so both runs launch the same kernels in the same order) — the
k-th dispatch of a kernel in the counter pass is joined with the k-th dispatch of that kernel in the trace: busy share and
cycles from the counters, duration from the trace, and the dominant launch's row closes arithmetically (MFMA instructions =
busy cycles / 16; x 16,384 FLOP = the launch's algorithmic FLOP; cycles / trace duration = the clock it ran at).
usage: python tools/summarize_mfma.py gpurun_out/pmcg_mfma profiles/<name>.md [gpurun_out/prof_<tag>/run_kernel_trace.csv]"""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-zA-Z_0-9]+?)I(DF16b|f)(.*)", n)
    if m:
        t = re.findall(r"Li(\d+)E", m.group(3))
        return m.group(1) + "<" + ("bf16" if m.group(2) == "DF16b" else "f32") + ("," + ",".join(t) if t else "") + ">"
    return n.split("(")[0][:60] or n[:60]


def main():
    d, out = sys.argv[1], sys.argv[2]
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(d + "/run_counter_collection.csv")):
        k = per[r["Dispatch_Id"]]
        k["name"] = r["Kernel_Name"]
        k[r["Counter_Name"]] = float(r["Counter_Value"])
        k["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k["grid"] = int(r.get("Grid_Size", 0) or 0)
    # kernel trace of a separate, counter-free run: durations by (kernel, occurrence index)
    trace = collections.defaultdict(list)
    if len(sys.argv) > 3:
        for r in sorted(csv.DictReader(open(sys.argv[3])), key=lambda r: int(r["Dispatch_Id"])):
            trace[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    occ = collections.defaultdict(int)
    joined = 0
    for did in sorted(per, key=int):
        k = per[did]
        i = occ[k["name"]]
        occ[k["name"]] += 1
        k["ns_pmc"] = k["ns"]
        if i < len(trace.get(k["name"], ())):
            k["ns"] = trace[k["name"]][i]
            joined += 1
    dom = []
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for k in per.values():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in k or "GRBM_GUI_ACTIVE" not in k:
            continue
        n = short(k["name"])
        if not any(t in n for t in ("conv_", "mfma")):
            continue
        names = [n]
        # the step's dominant launch on its own row: the tower 3x3 conv over P3 + P4 of one tower at bs 8 = 500 tiles of 512 threads
        # (the prediction convs' data gradient over P3 + P4 has the same grid with K = 576 instead of 2,304: told apart by the
        # MFMAs issued, 38 instead of 151 GFLOP)
        if "conv_sp_kernel" in n and k["grid"] == 500 * 512 and k["SQ_VALU_MFMA_BUSY_CYCLES"] / 16 * 16384 > 100e9:
            names.append(n + " — 500 tiles: the dominant launch (tower conv over P3 + P4)")
            dom.append(k)
        for nm in names:
            a = agg[nm]
            a[0] += 1
            a[1] += k["SQ_VALU_MFMA_BUSY_CYCLES"]
            a[2] += k["GRBM_GUI_ACTIVE"]
            a[3] += k["ns"]
    rows = sorted(agg.items(), key=lambda kv: -kv[1][3])
    tot = [sum(v[i] for n_, v in rows if "dominant launch" not in n_) for i in range(4)]
    with open(out, "w") as f:
        f.write("# MFMA-busy counters of the conv kernels, `bench.py` training step (bf16), one rocprofv3 --pmc pass\n\n")
        f.write("busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); effective clock = GRBM_GUI_ACTIVE / 8 / "
                "duration (reads high on dispatches shorter than ~0.3 ms).  Whole process: tuning + warm-up + timed steps.\n\n")
        f.write("| kernel | dispatches | total ms | MFMA busy share | eff. clock GHz |\n|---|---|---|---|---|\n")
        # the 24 largest by total time, plus — always — the 3x3 forward / data-gradient kernels (the step's dominant launch
        # runs on one of them; in a whole-process table the weight-gradient tuning sweeps outweigh it)
        top = rows[:24] + [r for r in rows[24:] if any(t in r[0] for t in ("conv_sp_kernel", "conv_px_kernel", "conv_wgrad_sk_kernel"))]
        for n, (c, busy, gui, ns) in top:
            f.write("| `%s` | %d | %.2f | %.3f | %.2f |\n" % (n, c, ns / 1e6, busy / (gui / 8 * 1024), gui / 8 / ns))
        f.write("| **all conv kernels** | %d | %.2f | **%.3f** | %.2f |\n" % (tot[0], tot[3] / 1e6, tot[1] / (tot[2] / 8 * 1024), tot[2] / 8 / tot[3]))
        if trace:
            f.write("\nDurations: kernel trace of a separate counter-free run with the same tuner choices, joined by (kernel, occurrence "
                    "index): %d of %d dispatches joined.\n" % (joined, len(per)))
        if dom:
            n = len(dom)
            busy = sum(k["SQ_VALU_MFMA_BUSY_CYCLES"] for k in dom) / n
            cyc = sum(k["GRBM_GUI_ACTIVE"] for k in dom) / n / 8
            ns, ns_pmc = sum(k["ns"] for k in dom) / n, sum(k["ns_pmc"] for k in dom) / n
            gflop = busy / 16 * 16384 / 1e9
            f.write("\n## The dominant launch, closed arithmetically (tower 3x3 conv over P3 + P4 of one tower: 500 tiles, 151.0 GFLOP algorithmic)\n\n"
                    "| dispatches | SQ_VALU_MFMA_BUSY_CYCLES per dispatch | = MFMA instructions x 16 -> GFLOP issued | cycles per dispatch (GRBM_GUI_ACTIVE / 8) | busy share | "
                    "duration us: kernel trace | duration us: counter pass | clock GHz = cycles / trace duration | TFLOP/s = issued GFLOP / trace duration |\n|---|---|---|---|---|---|---|---|---|\n")
            f.write("| %d | %.4g | %.1f | %.4g | %.3f | %.1f | %.1f | %.2f | %.0f |\n"
                    % (n, busy, gflop, cyc, busy / (cyc * 1024), ns / 1e3, ns_pmc / 1e3, cyc / ns, gflop / (ns * 1e-9) / 1e3))
            f.write("\nbusy share x 1,024 SIMDs x 1,024 FLOP per SIMD-cycle x clock = the TFLOP/s column; the issued GFLOP exceed the "
                    "algorithmic 151.0 by the tiles' padding (500 tiles x 256 pixels = 128,000: none here) only.\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
