"""Summarise tools/prof_pmc.sh output (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the training bench)
into profiles/<name>.json + .md: HBM bytes per launch of the conv family in ONE training step.

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE counts a wide coalesced
(16 B/lane) streaming read at exactly 1/2 of its bytes -> doubled; WRITE_SIZE is exact for 16-B/lane stores and float
atomics.  Calibration on correlate_kernel (known bytes: read = write = elements * 2 B) is printed beside it."""
import csv
import json
import os
import sys


def last_step(rows):
    rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
    # exactly one fcos_loss_finalize launch per step: the dispatches between two of them are one step's worth of every
    # kernel (second half of step k, first half of step k+1 in steady state)
    idx = [i for i, r in enumerate(rows) if "fcos_loss_finalize" in r["Kernel_Name"]]
    return rows[idx[-2] + 1:idx[-1] + 1]


def main():
    d, out = sys.argv[1], sys.argv[2]
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = last_step(list(csv.DictReader(open("%s/%s/run_counter_collection.csv" % (d, c)))))
        for r in rows:
            n = r["Kernel_Name"]
            # (round 5: conv_sp_kernel — the dominant launch — conv_pw / conv_px and conv_wgrad_sk_kernel were missing from these lists
            # since round 3: the rounds' "conv family" traffic figures counted the LDS-DMA tile kernels only, which is also why the
            # launch count never matched the run's)
            k = "conv_fwd_dgrad" if any(t in n for t in ("conv_dma", "conv_igemm", "conv_sp_kernel", "conv_pred_kernel", "conv_pw_kernel", "conv_px_kernel")) else (
                "conv_wgrad" if ("conv_wgrad_kernel" in n or "conv_wgrad_sk_kernel" in n or "conv_wgrad_xr_kernel" in n) else (
                    "correlate" if ("correlate_kernel" in n or "correlate_levels_kernel" in n) else None))
            if k is None:
                continue
            e = res.setdefault(k, {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
            e[c] += float(r["Counter_Value"]) * 1024.0
            if c == "FETCH_SIZE":
                e["launches"] += 1
    for k, e in res.items():
        e["hbm_read_bytes"] = 2.0 * e["FETCH_SIZE"]         # gfx950: FETCH_SIZE = 1/2 of a wide coalesced read
        e["hbm_write_bytes"] = e["WRITE_SIZE"]
        e["hbm_bytes_per_launch"] = (e["hbm_read_bytes"] + e["hbm_write_bytes"]) / max(e["launches"], 1)
    fam = {"launches": 0, "hbm_read_bytes": 0.0, "hbm_write_bytes": 0.0}
    for k in ("conv_fwd_dgrad", "conv_wgrad"):
        for f in fam:
            fam[f] += res[k][f]
    fam["hbm_bytes_per_launch"] = (fam["hbm_read_bytes"] + fam["hbm_write_bytes"]) / fam["launches"]
    res["conv_family"] = fam
    # calibration: correlation over 5 FPN levels of 8 x 800x1024 in bf16: 8*17064*256 elements, 2 B read + 2 B written
    cal = 8 * 17064 * 256 * 2.0
    res["calibration"] = {"kernel": "correlate_kernel (forward + backward d_feat)", "expected_read_bytes_per_pass": cal,
                          "measured_read_bytes": res["correlate"]["hbm_read_bytes"],
                          "measured_write_bytes": res["correlate"]["hbm_write_bytes"],
                          "launches": res["correlate"]["launches"]}
    # the GPU box has no .git: tools/stamp_commit.sh (run in the build container before gpurun) leaves the commit in
    # .commit_stamp at the repo root, which travels with the snapshot
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    commit = ""
    try:
        commit = open(os.path.join(root, ".commit_stamp")).read().strip()
    except OSError:
        try:
            commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=root).stdout.strip()
        except Exception:
            commit = ""
    res["commit"] = commit or "unknown"
    json.dump(res, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as f:
        f.write("# HBM traffic from PMC counters (rocprofv3 --pmc, separate passes), one training step, bf16, bs=8\n\n")
        f.write("FETCH_SIZE doubled (gfx950 counts wide coalesced reads at 1/2), WRITE_SIZE as is; KiB -> bytes.\n\n")
        f.write("| kernel group | launches/step | HBM read MB | HBM write MB | MB per launch |\n|---|---|---|---|---|\n")
        for k in ("conv_fwd_dgrad", "conv_wgrad", "conv_family", "correlate"):
            e = res[k]
            f.write("| %s | %d | %.1f | %.1f | %.2f |\n" % (k, e["launches"], e["hbm_read_bytes"] / 1e6,
                                                           e["hbm_write_bytes"] / 1e6, e["hbm_bytes_per_launch"] / 1e6))
        c = res["calibration"]
        f.write("\nCalibration: %d correlate launches (forward + backward d_feat, all 5 levels each) should read and write %.1f MB each "
                "per pass (x2 passes); measured read %.1f MB, write %.1f MB.\n" % (
                    c["launches"], cal / 1e6, c["measured_read_bytes"] / 1e6, c["measured_write_bytes"] / 1e6))
    print(open(out + ".md").read())


if __name__ == "__main__":
    main()
