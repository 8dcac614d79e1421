"""GPU box: is the default bench's 0.2 - 0.3 s timed window representative of a sustained run?  N training steps (bs 8, bf16,
800 x 1024) back to back; per block of 100 steps: median / mean step time from HIP events at the step boundaries, the host's
enqueue time per step, and the device's clock / power / temperature as rocm-smi reports them (when it is allowed to).
python tools/sustained.py [steps=3000]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oneshotdet_amd import ops, spec, synth, train

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
B = 8
LR = float(os.environ.get("OSD_SUSTAINED_LR", "0.0005"))      # 0: the weights never change (is a slowdown over time data-dependent?)
eng = train.TrainEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=torch.bfloat16, lr=LR)
print("learning rate", LR)
images = torch.from_numpy(synth.make_images("bench.target", B, 800, 1024, seed=1000)).cuda()
queries = torch.from_numpy(synth.make_images("bench.query", B, 127, 127, seed=1000)).cuda()
gts = synth.make_gt_boxes(B, 800, 1024, seed=1000, max_boxes=6)
gtb = np.zeros((B, 6, 4), np.float32)
for i, g in enumerate(gts):
    gtb[i, :len(g)] = g
gtb, gtc = torch.from_numpy(gtb).cuda(), torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()
with ops.tuning():
    eng.forward_backward(images, queries, gtb, gtc)
torch.cuda.synchronize()
eng.defer_join = True


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True, text=True, timeout=20).stdout
        rows = [r for r in out.splitlines() if r and not r.startswith("WARNING")]
        if len(rows) >= 2:
            head, val = rows[0].split(","), rows[1].split(",")
            keep = [i for i, h in enumerate(head) if any(k in h.lower() for k in ("sclk", "power", "junction", "edge"))]
            return "; ".join("%s=%s" % (head[i].strip(), val[i].strip()) for i in keep)[:300]
    except Exception as e:      # noqa: BLE001
        return "rocm-smi: %r" % (e,)
    return "rocm-smi: no data"


def hwmon():
    """power cap / average power / memory temperature straight from sysfs (what rocm-smi may not print)"""
    import glob
    out = []
    for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_average",
                "/sys/class/drm/card*/device/hwmon/hwmon*/temp*_input", "/sys/class/drm/card*/device/pp_dpm_mclk",
                "/sys/class/drm/card*/device/pp_dpm_sclk", "/sys/class/drm/card*/device/pp_dpm_fclk"):
        for f in sorted(glob.glob(pat))[:6]:
            try:
                out.append("%s=%s" % (f.split("/")[-1], open(f).read().strip().replace("\n", ",")))
            except Exception:      # noqa: BLE001
                pass
    return "; ".join(out)[:600]


# two yardsticks run ALONE on the chip between blocks: an MFMA-bound launch (3x3 256 -> 256 conv over a P3-sized batch) and an
# HBM-bound one (512 MB copy) — do the kernels themselves slow down, or does the step lose time between them?
_x = torch.randn(8, 100, 128, 256, device="cuda").bfloat16()
_pc = ops.pack_conv(torch.randn(256, 256, 3, 3, device="cuda") / 48, bias=torch.zeros(256, device="cuda"), dtype=torch.bfloat16)
_big = torch.zeros(128 * 1024 * 1024, device="cuda")
_big2 = torch.empty_like(_big)
with ops.tuning():
    ops.conv2d(_x, _pc, pad=1)


def yardsticks():
    torch.cuda.synchronize()
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    a.record()
    for _ in range(10):
        ops.conv2d(_x, _pc, pad=1)
    b.record()
    for _ in range(4):
        _big2.copy_(_big)
    c.record()
    torch.cuda.synchronize()
    return "conv3x3 P3 %.1f us, 512 MB copy %.0f us (%.2f TB/s)" % (a.elapsed_time(b) * 100, b.elapsed_time(c) * 250, 1.074 / (b.elapsed_time(c) / 4 * 1e-3) / 1e3)


print("before:", smi(), flush=True)
print("hwmon:", hwmon(), flush=True)
print("yardsticks (idle chip):", yardsticks(), flush=True)
for _ in range(5):
    eng.train_step(images, queries, gtb, gtc)
torch.cuda.synchronize()
t_start = time.time()
for blk in range(N // 100):
    evs = []
    t0 = time.perf_counter()
    for _ in range(100):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)
        eng.train_step(images, queries, gtb, gtc)
    t1 = time.perf_counter()
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(e)
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
    mem = torch.cuda.memory_allocated() / 2**20
    st = torch.cuda.memory_stats()
    alloc_info = "reserved %.0f MiB, device mallocs %d, frees %d, retries %d, ooms %d" % (
        st.get("reserved_bytes.all.current", 0) / 2**20, st.get("num_device_alloc", 0), st.get("num_device_free", 0), st.get("num_alloc_retries", 0), st.get("num_ooms", 0))
    # the host alone: enqueue one step from an idle device (clock stops before any wait)
    h0 = time.perf_counter()
    eng.train_step(images, queries, gtb, gtc)
    h1 = time.perf_counter()
    torch.cuda.synchronize()
    h2 = time.perf_counter()
    alloc_info += "; one step from idle: host %.2f ms, wall %.2f ms" % ((h1 - h0) * 1e3, (h2 - h0) * 1e3)
    # the same forward + backward without the training proposals (score / top-k / NMS on their own stream): is the data-dependent part
    # of a slowdown the proposal pipeline?  (no update: gradients accumulate into the next step's, harmless for a timing probe)
    eng.join()
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    eng.forward_backward(images, queries, gtb, gtc, with_proposals=False)
    torch.cuda.synchronize()
    h1 = time.perf_counter()
    eng.forward_backward(images, queries, gtb, gtc, with_proposals=True)
    torch.cuda.synchronize()
    h2 = time.perf_counter()
    eng.flat_g.zero_()
    eng._grads_clean = True
    alloc_info += "; forward + backward alone from idle: without proposals %.2f ms, with %.2f ms" % ((h1 - h0) * 1e3, (h2 - h1) * 1e3)
    print("steps %5d-%5d (t = %5.1f s): median %.3f ms  mean %.3f  p90 %.3f  host enqueue %.3f ms/step  allocated %.0f MiB  %s"
          % (blk * 100, blk * 100 + 99, time.time() - t_start, ms[50], sum(ms) / 100, ms[90], (t1 - t0) * 10, mem,
             (yardsticks() + "; " + smi()) if blk % 5 == 4 else ""), flush=True)
    if blk % 5 == 4:
        print("   ", alloc_info, flush=True)
ops.gn_onepass_check("sustained run")
if os.environ.get("OSD_SAVE_HEAD"):      # the head outputs of the last step: what the proposal pipeline sees after N steps of training
    torch.save([(a.float().cpu(), b.float().cpu()) for a, b in eng.last_head_out], os.environ["OSD_SAVE_HEAD"])
