"""bench.py — throughput of the siamese-FCOS hot path on MI355X (contract: see the task's bench section / DESIGN.md §5).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|f32] [--batch 8] [--no-cpu-baseline]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
              bench.py --gpus N --steps K --warmup W
          or simply `python bench.py --gpus N`: with no WORLD_SIZE in the environment the process starts N ranks of itself
          (one per GPU, rendezvous on 127.0.0.1) BEFORE touching the GPU and exits with their worst return code.
  A WORLD_SIZE that disagrees with --gpus is an error (exit 2): the line's n_gpus is always the world size that ran.

One "step" = one pass of the hot path over one batch of synthetic (target, query) pairs already resident in HBM:
two ResNet-50-FPN backbones, query pooling, correlation, FCOS towers + predictions, score/decode/top-k/NMS proposals.
Weak scaling: every rank processes its own batch (forward needs no collective); value = images of all ranks / max time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "images/sec/GPU fwd+bwd, 800x1024 target + 127x127 query, bs=8; 1->8 GPU scaling"
# --workload: (target geometries cycled step by step, query shots per image, default batch per GPU)
WORKLOADS = {"config3": (((800, 1024),), 1), "config5": (((640, 832), (800, 1024), (1024, 1312)), 5)}
WORKLOAD_BATCH = {"config3": 8, "config5": 4}
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}   # MI355X_MICROARCH.md: dense MFMA peaks (f32-in MFMA = vector rate)
PEAK_HBM_TBS = 8.0                             # MI355X_MICROARCH.md: HBM3E spec peak (6.3 measured by a streaming copy)


def conv_bytes(x, y, pc, res=None, mask=None, x2=None):
    """Algorithmic HBM bytes of one conv launch: every tensor once — the input pixels the taps touch (a strided 1x1 reads a
    quarter of its map), the packed weights, the output, the residual / mask operands of the epilogue."""
    m = y.shape[0] * y.shape[1] * y.shape[2]
    b = min(x.numel(), m * pc.r * pc.s * x.shape[-1]) * x.element_size() + pc.w.numel() * pc.w.element_size() + y.numel() * y.element_size()
    for t in (res, mask):
        if t is not None:
            b += min(t.numel(), y.numel()) * t.element_size()
    if x2 is not None:
        b += min(x2.numel(), m * x2.shape[-1]) * x2.element_size()
    return float(b)


class ConvTimer(object):
    """HIP-event timing of every conv_igemm launch on the stream it is launched on (torch's current stream is the
    stream handed to the C-ABI).  Algorithmic FLOPs = 2 * M * Cout * Cin * R * S with the REAL (unpadded) channels."""

    def __init__(self):
        self.records = []
        self.flops = 0.0
        self.launches = 0
        self.labels = []         # (kind, M, Cout, K) per record, for --layer-table

    def install(self, ops):
        self._orig = ops.conv2d
        timer = self

        def timed_conv2d(x, pc, *a, **kw):
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            y = timer._orig(x, pc, *a, **kw)
            e.record()
            m = y.shape[0] * y.shape[1] * y.shape[2]
            k_real = 147 if pc.stem else pc.r * pc.s * getattr(pc, "cin_real", pc.cin_k)
            timer.records.append((s, e))
            timer.labels.append(("conv%dx%d" % (pc.r, pc.s), m, pc.cout, k_real, 2.0 * m * pc.cout * k_real,
                                 conv_bytes(x, y, pc, kw.get("res", a[3] if len(a) > 3 else None), kw.get("mask", a[10] if len(a) > 10 else None),
                                            kw.get("x2", a[12] if len(a) > 12 else None))))
            timer.flops += 2.0 * m * pc.cout * k_real
            timer.launches += 1
            return y
        ops.conv2d = timed_conv2d
        self._orig_cg = ops.conv2d_multi

        def timed_conv2d_multi(xs, pcs, *a, **kw):
            one = len(xs) == 1 and not kw.get("_whole") and kw.get("algo") is None     # goes out as a plain conv2d launch
            if one or not (kw.get("_whole") or len(xs) < 3 or kw.get("algo") is not None):
                return timer._orig_cg(xs, pcs, *a, **kw)     # dispatcher call: the launches inside are bracketed
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            ys = timer._orig_cg(xs, pcs, *a, **kw)
            e.record()
            pc = pcs[0]
            m = sum(y.shape[0] * y.shape[1] * y.shape[2] for y in ys)
            k_real = pc.r * pc.s * getattr(pc, "cin_real", pc.cin_k)
            timer.records.append((s, e))
            rs_, ms_ = kw.get("residuals", a[3] if len(a) > 3 else None), kw.get("masks", a[5] if len(a) > 5 else None)
            by = sum(conv_bytes(xi, yi, pc, None if not rs_ else rs_[i], None if not ms_ else ms_[i]) for i, (xi, yi) in enumerate(zip(xs, ys)))
            by -= (len(xs) - 1) * pc.w.numel() * pc.w.element_size() if all(q is pc for q in pcs) else 0.0      # levels share one weight
            # "_grouped": the FPN levels of ONE conv (an FCOS tower layer); "_levels": every segment its own conv (the FPN's P3 + P4
            # output convs as one launch, round 5) — a backbone / FPN row for roofline.backbone_convs, which skips the tower rows
            kind = "conv%dx%d_grouped" if len(set(id(q) for q in pcs)) < len(pcs) or len(pcs) == 1 else "conv%dx%d_levels"
            timer.labels.append((kind % (pc.r, pc.s), m, pc.cout, k_real, 2.0 * m * pc.cout * k_real, by))
            timer.flops += 2.0 * m * pc.cout * k_real
            timer.launches += 1
            return ys
        ops.conv2d_multi = timed_conv2d_multi

    def uninstall(self, ops):
        ops.conv2d = self._orig
        ops.conv2d_multi = self._orig_cg

    def reset(self):
        self.records, self.flops, self.launches, self.labels = [], 0.0, 0, []

    def layer_table(self, steps):
        """Per-shape totals of the bracketed launches: rows (kind, M, Cout, K, launches/step, us/launch, TFLOP/s, ms/step, MB per
        launch (algorithmic: conv_bytes; 0 for the weight-gradient rows), bound TFLOP/s = min(MFMA peak, 8 TB/s x FLOP per byte),
        fraction of that bound)."""
        ov = self.bracket_overhead_ms()
        agg = {}
        for (s, e), lab in zip(self.records, self.labels):
            a = agg.setdefault(lab[:4], [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += s.elapsed_time(e) - ov
            a[2] += lab[4]
            a[3] += lab[5] if len(lab) > 5 else 0.0
        rows = []
        for k, v in agg.items():
            tf = v[2] / (v[1] * 1e-3) / 1e12
            bound = min(self.peak_tflops, PEAK_HBM_TBS * v[2] / v[3]) if v[3] > 0 else self.peak_tflops
            rows.append((k[0], k[1], k[2], k[3], v[0] / steps, v[1] * 1e3 / v[0], tf, v[1] / steps, v[3] / v[0] / 1e6, bound, tf / bound))
        return sorted(rows, key=lambda r: -r[7])

    peak_tflops = PEAK_TFLOPS["bf16"]

    def total_ms(self):
        return sum(s.elapsed_time(e) for s, e in self.records) - self.bracket_overhead_ms() * len(self.records)

    @staticmethod
    def launch_bracket_overhead_ms(n=60, burst=16):
        """What a start/stop event bracket adds to ONE kernel launch inside it (dispatch of the kernel behind the start marker,
        completion signal ahead of the stop marker): with a tiny kernel K, bracket(K x burst) - bracket(K) = (burst - 1) launches
        back to back, so one launch costs s = that / (burst - 1) and the bracket's fixed part is bracket(K) - s.  rocprofv3's kernel
        trace reports durations without it; `roofline.kernel_time` subtracts it so that the figure can be checked against profiles/."""
        if not hasattr(ConvTimer, "_launch_overhead"):
            from oneshotdet_amd import ops
            probe = torch.zeros(64, device="cuda", dtype=torch.float32)

            def bracket(k):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(k):
                    ops.add_mask(probe, None, None, out=probe)
                b.record()
                return a, b
            for _ in range(5):
                bracket(burst)
            torch.cuda.synchronize()
            one, many = [], []
            for _ in range(n):
                torch.cuda._sleep(int(2e6))            # park the stream: the host enqueues the bracket ahead of the GPU
                one.append(bracket(1))
                many.append(bracket(burst))
            torch.cuda.synchronize()
            t1 = sorted(a.elapsed_time(b) for a, b in one)[n // 2]
            tn = sorted(a.elapsed_time(b) for a, b in many)[n // 2]
            ConvTimer._launch_overhead = max(0.0, t1 - (tn - t1) / (burst - 1))
        return ConvTimer._launch_overhead

    @staticmethod
    def bracket_overhead_ms(n=200):
        """Elapsed time of an EMPTY start/stop event bracket on this stream (two marker packets with nothing between):
        subtracted from every bracketed launch so avg_launch_us is the kernel's own duration, the figure rocprofv3's
        kernel trace reports."""
        if not hasattr(ConvTimer, "_overhead"):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            torch.cuda.synchronize()
            for s, e in ev:
                s.record()
                e.record()
            torch.cuda.synchronize()
            ts = sorted(s.elapsed_time(e) for s, e in ev)
            ConvTimer._overhead = ts[len(ts) // 2]
        return ConvTimer._overhead


def cpu_baseline(dtype_name, seconds_budget=25.0, train=False, shapes=((800, 1024),), shots=1):
    """The oracle (CPU restatement of the reference, kind 'port') timed on this box's host cores on a bounded sample of the same
    workload: single (target, `shots` x 127x127 query) pairs, one per shape in `shapes` — forward incl. proposals (forward
    mode), or forward + training proposals + FCOS loss + backward through autograd (train mode: the headline metric's step
    without the optimiser).  SURVEY.md 8d's protocol: bs = 1, torch.set_num_threads(n) for n = 8 and n = all cores, median of
    5 timed passes after one warm-up each; `value` is the faster of the two settings, both are reported."""
    import statistics
    import numpy as np
    from oneshotdet_amd import spec, synth
    from oracle import hotpath_ref as orc
    sd = orc.to_torch_state_dict(synth.make_state_dict(spec.hot_path_shapes()))
    if train:
        for k, v in sd.items():
            v.requires_grad_(not spec.is_frozen(k))
    samples = []
    for i, (h, w) in enumerate(shapes):
        img = torch.from_numpy(synth.make_images("bench.target", 1, h, w, seed=1000 + i))
        q = torch.from_numpy(synth.make_images("bench.query", shots, 127, 127, seed=1000 + i))
        samples.append((img, q, synth.make_gt_boxes(1, h, w, seed=1000 + i, max_boxes=6), (h, w)))
    all_cores = torch.get_num_threads()
    post_s = [0.0]             # seconds inside the proposal post-processing (score / top-k / decode / NMS: numpy O(n^2) NMS)

    def one(sample):
        img, q, gts, hw = sample
        if not train:
            with torch.no_grad():
                o = orc.hot_path_forward(img, q, sd, shots=shots)
                t1 = time.time()
                orc.fcos_postprocess(o["logits"], o["bbox_reg"], o["centerness"], [hw])
                post_s[0] += time.time() - t1
            return
        o = orc.hot_path_forward(img, q, sd, shots=shots)
        with torch.no_grad():
            t1 = time.time()
            orc.fcos_postprocess(o["logits"], o["bbox_reg"], o["centerness"], [hw], pre_nms_top_n=spec.PRE_NMS_TOP_N_TRAIN,
                                 post_nms_top_n=spec.POST_NMS_TOP_N_TRAIN)
            post_s[0] += time.time() - t1
        c, r, t, _ = orc.fcos_loss(o["logits"], o["bbox_reg"], o["centerness"], gts, focal="cuda")
        (c + r + t).backward()
        for v in sd.values():
            v.grad = None

    def pass_seconds():
        t0 = time.time()
        for sm in samples:
            one(sm)
        return time.time() - t0
    results, total_s = {}, 0.0
    settings = sorted({min(8, all_cores), all_cores})
    per_setting = seconds_budget / len(settings)
    for nthreads in settings:
        torch.set_num_threads(nthreads)
        pass_seconds()                      # warm-up (oneDNN primitive creation for these shapes at this thread count)
        post_s[0] = 0.0
        ts, t_begin = [], time.time()
        while len(ts) < 5 and (len(ts) < 1 or time.time() - t_begin < per_setting):
            ts.append(pass_seconds())
        results[nthreads] = dict(images_per_sec=round(len(samples) / statistics.median(ts), 4), passes=len(ts),
                                 post_share=round(post_s[0] / max(sum(ts), 1e-9), 3))
        total_s += sum(ts)
    torch.set_num_threads(all_cores)
    best = max(results, key=lambda n: results[n]["images_per_sec"])
    what = "forward + training proposals + FCOS loss + backward (autograd)" if train else "forward incl. proposals"
    shp = " / ".join("1x%dx%d" % hw for hw in shapes)
    return {"value": results[best]["images_per_sec"], "unit": "images/sec", "cores": best, "kind": "port",
            "by_threads": {str(n): results[n] for n in settings},
            "sample": "bs = 1: (%s target + %dx127x127 query) %s, oracle/hotpath_ref.py (torch CPU fp32); median of <= 5 passes per "
                      "thread setting (%s threads), %.0f s of CPU work; post_share = fraction spent in the proposal post-processing "
                      "(score / top-k / decode / the oracle's numpy NMS)"
                      % (shp, shots, what, " and ".join(str(n) for n in settings), total_s)}


def self_launch(argv, n):
    """`python bench.py --gpus N` without a launcher (tools/train_net.py:224-226 reads WORLD_SIZE the same way): start N
    ranks of this script as child processes, one per GPU, and return the worst exit code.  The parent never initialises
    the GPU (no HIP call, no exec of a GPU process: children are ordinary subprocesses)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OSD_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    # poll: the first rank that exits non-zero takes its siblings down (they would otherwise sit in the rendezvous or the
    # first collective until the process-group timeout) and its code is the result
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in live:
                        q.terminate()
            if live:
                time.sleep(0.05)
    except KeyboardInterrupt:
        for q in procs:
            q.terminate()
        raise
    finally:
        for q in procs:
            if q.poll() is None:
                try:
                    q.wait(timeout=10)
                except Exception:      # noqa: BLE001
                    q.kill()
    return rc


def visible_gpus():
    """Number of GPUs a child rank could use, WITHOUT any HIP call in this process: the KFD topology in sysfs (a node with
    simd_count > 0 is a GPU), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set.  None = cannot tell (no
    sysfs): the children then fail by themselves with exit code 2."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                for ln in f:
                    if ln.startswith("simd_count") and int(ln.split()[1]) > 0:
                        n += 1
        return n
    except Exception:      # noqa: BLE001
        return None


def check_world(args):
    """--gpus N is the contract: N ranks run, or nothing does.  Returns (rank, local_rank, world) from the environment;
    launches the ranks itself when there is no launcher; exits 2 when the launcher's WORLD_SIZE disagrees with --gpus."""
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            if not (args.dry_run_cpu or os.environ.get("OSD_BENCH_SHARE_GPU") == "1"):
                have = visible_gpus()                      # sysfs / environment only: the parent never makes a HIP call
                if have is not None and have < args.gpus:
                    print("bench.py: --gpus %d but %d GPU(s) visible" % (args.gpus, have), file=sys.stderr)
                    sys.exit(2)
            sys.exit(self_launch(sys.argv[1:], args.gpus))
        return 0, 0, 1
    world = int(os.environ["WORLD_SIZE"])
    if world != args.gpus:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; pass --gpus %d or launch %d ranks"
              % (args.gpus, world, world, args.gpus), file=sys.stderr)
        sys.exit(2)
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world


def dist_setup(backend):
    """One process per GPU, launched by torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the env)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        # generous timeout: rank 0 autotunes (every candidate kernel of every geometry) BEFORE it joins, the other ranks wait in
        # the rendezvous meanwhile (config5's three geometries or --second-stage take minutes on a cold box)
        import datetime
        tmo = datetime.timedelta(minutes=int(os.environ.get("OSD_PG_TIMEOUT_MIN", "60")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    return rank, local_rank, world


STEP_STATS = {}      # filled by timed_steps: per-step durations from HIP events on the main stream, per-rank bracket times


def timed_steps(step, steps, warmup, world, device_sync, reduce_device):
    """W untimed steps, then exactly K steps bracketed by barrier + device sync on both sides; MAX over ranks.  Beside the
    contract's bracket, one HIP event per step boundary on the main stream (recorded, never waited for inside the region)
    gives the per-step durations -> median / p10 / p90 (SURVEY.md 8d reports the median), and every rank's own bracket time
    is gathered so a slow rank is visible in the line."""
    def sync():
        device_sync()
        if world > 1:
            dist.barrier()
            device_sync()
    for _ in range(warmup):
        step()
    sync()
    # the host's cyclic garbage collector stays out of the timed region (a generation-2 pass over the engine's object graph is
    # tens of milliseconds; the step allocates no cycles of its own)
    import gc
    gc.collect()
    gc_was = gc.isenabled()
    gc.disable()
    use_ev = reduce_device != "cpu" and torch.cuda.is_available()
    evs = []
    t0 = time.perf_counter()
    for _ in range(steps):
        if use_ev:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
        step()
    if use_ev:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)
    sync()
    elapsed = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    STEP_STATS.clear()
    if use_ev and len(evs) > 1:
        ms = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
        STEP_STATS["step_ms"] = {"median": round(ms[len(ms) // 2], 3), "p10": round(ms[len(ms) // 10], 3),
                                 "p90": round(ms[min(len(ms) - 1, (9 * len(ms)) // 10)], 3), "n": len(ms),
                                 "source": "HIP events at the step boundaries of rank 0's main stream (not synchronised inside "
                                           "the timed region); `ms_per_step` is the contract's wall bracket / steps"}
    if world > 1:
        every = [None] * world
        dist.all_gather_object(every, float(elapsed))
        STEP_STATS["rank_ms_per_step"] = [round(t / steps * 1e3, 3) for t in every]
        t = torch.tensor([elapsed], device=reduce_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def result_line(args, world, batch, elapsed, workload, launch, roofline=None, cpu=None):
    pg_world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    assert pg_world == world == args.gpus, (pg_world, world, args.gpus)
    if world == 1:
        par = "dp1: one rank, no collective anywhere in the step"
    else:
        par = "dp%d: batch sharded over ranks (weak scaling), process-group world size %d, backend %s" % (
            world, pg_world, dist.get_backend())
        if dist.get_backend() == "nccl":       # "nccl" IS RCCL on ROCm: say which build moved the gradients
            try:
                par += " (RCCL %s over xGMI, %d ranks after init)" % (".".join(str(v) for v in torch.cuda.nccl.version()), pg_world)
            except Exception:      # noqa: BLE001
                par += " (RCCL)"
    line = {
        "metric": METRIC, "value": round(batch * world * args.steps / elapsed, 2), "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": workload, "global_batch": batch * world,
                   "parallelism": par, "world_size": pg_world, "launch": launch},
        "roofline": roofline,
    }
    if cpu is not None:
        line["cpu_baseline"] = cpu
    line.update(STEP_STATS)
    return line


def measured_peaks(dtype):
    """Yardsticks measured on the box (profiles/r1_peak_probe.json: bare MFMA loop, streaming copy, vendor GEMM) reported
    beside the datasheet peak the `frac` is taken against."""
    path = os.path.join(ROOT, "profiles", "r1_peak_probe.json")
    try:
        j = json.load(open(path))
    except Exception:
        return None
    if dtype == "bf16":
        return {"mfma_loop_tflops": j["mfma_bf16_16x16x32_tflops"]["2_waves_per_simd"],
                "vendor_gemm_same_shape_as_tower_conv_tflops": j["vendor_gemm_bf16_tflops"]["M128000_N256_K2304"],
                "vendor_gemm_best_tflops": j["vendor_gemm_bf16_tflops"]["M16000_N1024_K6272"],
                "copy_tb_per_s": j["copy_tb_per_s_read_plus_write"]["1024_workgroups"], "source": "profiles/r1_peak_probe.json"}
    return {"mfma_loop_tflops": j["mfma_f32_16x16x4_tflops"]["2_waves_per_simd"],
            "copy_tb_per_s": j["copy_tb_per_s_read_plus_write"]["1024_workgroups"], "source": "profiles/r1_peak_probe.json"}


def measured_traffic(mode, dtype, launches_per_step=None):
    """HBM bytes per conv-family launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, gfx950 corrections).  Counters
    cannot be read from inside the process: they are collected OFFLINE by tools/prof_pmc.sh + tools/summarize_pmc.py in
    separate rocprofv3 --pmc passes of this same command and committed under profiles/ with the commit they were taken
    at.  Returns (bytes per launch or None, provenance string): the newest round's file for this mode is used."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s_%s.json" % (mode, dtype))),
                   key=lambda f: int(os.path.basename(f)[1:].split("_", 1)[0]))
    if not files:
        return None, "no PMC profile committed for this mode"
    try:
        with open(files[-1]) as f:
            j = json.load(f)
        src = "OFFLINE, not measured by this run: profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, at commit %s)" % (
            os.path.basename(files[-1]), j.get("commit") or "unknown")
        # the counter passes launch the conv family N times per step; a file whose N is not this run's was taken on another tree
        # state (round 4's line quoted 230 launches against the run's 265): per-launch bytes of another launch list are not this
        # run's traffic — report null and say why (the per-step total of the file stays in the provenance string)
        n_file = j["conv_family"].get("launches")
        # (the counter passes count kernel DISPATCHES between two loss launches, this run counts bracketed host launches: the two differ
        # by one at the same commit — 263 against 262 in round 5 — so a 1 % band separates "the same launch list" from "another tree")
        if launches_per_step is not None and n_file is not None and abs(int(n_file) - int(launches_per_step)) > max(2, int(launches_per_step) // 100):
            return None, ("STALE, refused: profiles/%s counted %d conv-family launches per step (%.2f GB per step) but this run makes %d; "
                          "regenerate with tools/prof_all.sh at this commit" % (os.path.basename(files[-1]), n_file,
                                                                               (j["conv_family"]["hbm_read_bytes"] + j["conv_family"]["hbm_write_bytes"]) / 1e9,
                                                                               launches_per_step))
        return round(j["conv_family"]["hbm_bytes_per_launch"]), src
    except Exception as e:
        return None, "unreadable PMC profile: %r" % (e,)


class TrainTimer(ConvTimer):
    """ConvTimer that also brackets the weight-gradient kernel (2*M*Cout*Cin*R*S FLOP per launch) and, separately, the
    correlation kernel (HBM-bound: elements * (read + write) bytes per launch)."""

    def install(self, ops):
        ConvTimer.install(self, ops)
        self.corr, self.corr_bytes = [], 0.0
        self._orig_c = ops.correlate_levels
        tm = self

        def timed_correlate(xs, qs):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ys = tm._orig_c(xs, qs)
            b.record()
            tm.corr.append((a, b))
            tm.corr_bytes += sum(2.0 * x.numel() * x.element_size() for x in xs) + sum(q.numel() * 4 for q in qs)
            return ys
        ops.correlate_levels = timed_correlate
        self._orig_w = ops.conv2d_wgrad
        timer = self

        def timed_wgrad(x, dy, dw, r, s, stride, pad, cout, scale=None, db=None):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            timer._orig_w(x, dy, dw, r, s, stride, pad, cout, scale=scale, db=db)
            b.record()
            timer.records.append((a, b))
            fl = 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * cout * x.shape[-1] * r * s
            timer.labels.append(("wgrad%dx%d" % (r, s), dy.shape[0] * dy.shape[1] * dy.shape[2], cout, x.shape[-1] * r * s, fl))
            timer.flops += fl
            timer.launches += 1
        ops.conv2d_wgrad = timed_wgrad
        self._orig_g = ops.conv2d_wgrad_grouped

        def timed_grouped(pairs, dw, r, s, stride, pad, cout, scale=None, db=None, **kw):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            timer._orig_g(pairs, dw, r, s, stride, pad, cout, scale=scale, db=db, **kw)
            b.record()
            timer.records.append((a, b))
            mm = sum(dy.shape[0] * dy.shape[1] * dy.shape[2] for x, dy in pairs)
            fl = 2.0 * mm * cout * pairs[0][0].shape[-1] * r * s
            timer.labels.append(("wgrad%dx%d_grouped" % (r, s), mm, cout, pairs[0][0].shape[-1] * r * s, fl))
            timer.flops += fl
            timer.launches += 1
        ops.conv2d_wgrad_grouped = timed_grouped
        self._orig_b = ops.conv2d_wgrad_batched

        def timed_batched(items, r, s, stride, pad, cout, algo=None):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            timer._orig_b(items, r, s, stride, pad, cout, algo=algo)
            b.record()
            timer.records.append((a, b))
            x, dy = items[0][0], items[0][1]
            m = dy.shape[0] * dy.shape[1] * dy.shape[2]
            fl = 2.0 * m * cout * x.shape[-1] * r * s * len(items)
            timer.labels.append(("wgrad%dx%d_x%d" % (r, s, len(items)), m, cout, x.shape[-1] * r * s, fl))
            timer.flops += fl
            timer.launches += 1
        ops.conv2d_wgrad_batched = timed_batched
        self._orig_m = ops.conv2d_wgrad_multi

        def timed_multi(items, r, s, stride, pad, cout, algo=None):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            timer._orig_m(items, r, s, stride, pad, cout, algo=algo)
            b.record()
            timer.records.append((a, b))
            m = sum(it[1].shape[0] * it[1].shape[1] * it[1].shape[2] for it in items)
            kk = items[0][0].shape[-1] * r * s
            fl = 2.0 * m * cout * kk
            timer.labels.append(("wgrad%dx%d_multi%d" % (r, s, len(items)), m, cout, kk, fl))
            timer.flops += fl
            timer.launches += 1
        ops.conv2d_wgrad_multi = timed_multi
        self._orig_x = ops.conv2d_wgrad_mixed

        def timed_mixed(items, algo=None):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            timer._orig_x(items, algo=algo)
            b.record()
            timer.records.append((a, b))
            fl, mm = 0.0, 0
            for x, dy, dw, sc, db, r, s, stride, pad, cout in items:
                m = dy.shape[0] * dy.shape[1] * dy.shape[2]
                fl += 2.0 * m * cout * x.shape[-1] * r * s
                mm += m
            timer.labels.append(("wgrad_mixed%d" % len(items), mm, 0, 0, fl))
            timer.flops += fl
            timer.launches += 1
        ops.conv2d_wgrad_mixed = timed_mixed

    def uninstall(self, ops):
        ConvTimer.uninstall(self, ops)
        ops.conv2d_wgrad = self._orig_w
        ops.conv2d_wgrad_grouped = self._orig_g
        ops.conv2d_wgrad_batched = self._orig_b
        ops.conv2d_wgrad_multi = self._orig_m
        ops.conv2d_wgrad_mixed = self._orig_x
        ops.correlate_levels = self._orig_c

    def correlation_roofline(self):
        ms = sum(a.elapsed_time(b) for a, b in self.corr) - self.bracket_overhead_ms() * len(self.corr)
        gbs = self.corr_bytes / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                "kernel": "correlate_levels_kernel (forward y = x*q and backward d_feat = g*q; one launch covers all 5 FPN levels)",
                "avg_launch_us": round(ms * 1e3 / max(len(self.corr), 1), 2), "launches": len(self.corr),
                "bytes_per_launch": round(self.corr_bytes / max(len(self.corr), 1))}


def main_train(args, rank, world, backend="nccl"):
    """BASELINE.json configs[2]/[3]: bs=8/GPU, forward + FCOS loss + backward + gradient all-reduce (RCCL) + SGD."""
    import numpy as np
    from oneshotdet_amd import ops, spec, synth, train
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    two = bool(args.second_stage)      # opt-in: the reference's COMPLETE training step (roi_heads.box losses + backward)
    wire = torch.bfloat16 if args.grad_wire == "bf16" else None
    eng = train.TrainEngine(synth.make_state_dict(spec.full_model_shapes() if two else spec.hot_path_shapes()), dtype=dtype,
                            second_stage=two)
    B = args.batch
    # config3 (default, the headline): one geometry.  config5 = BASELINE.json configs[4]'s per-GPU workload: multi-scale
    # targets (short edge cycling 640 / 800 / 1024, long edge = short x 1.28 rounded up to /32: SURVEY.md 8d) and S = 5 queries
    # per image, mean-pooled (generalized_rcnn.py:100-104); step i runs geometry i mod 3
    shapes, S = WORKLOADS[args.workload]
    batches = []
    for gi, (H, W) in enumerate(shapes):
        images = torch.from_numpy(synth.make_images("bench.target", B, H, W, seed=1000 + rank + 97 * gi)).cuda()
        queries = torch.from_numpy(synth.make_images("bench.query", B * S, 127, 127, seed=1000 + rank + 97 * gi)).cuda()
        gts = synth.make_gt_boxes(B, H, W, seed=1000 + rank + 97 * gi, max_boxes=6)
        gtb = np.zeros((B, 6, 4), np.float32)
        for i, g in enumerate(gts):
            gtb[i, :len(g)] = g
        batches.append((images, queries, torch.from_numpy(gtb).cuda(), torch.tensor([len(g) for g in gts], dtype=torch.int32).cuda()))
    images, queries, gt_boxes, gt_count = batches[0]

    def init_pg():
        """The process group of this run: N ranks under a launcher, or a ONE-rank RCCL group for --live-exchange."""
        if args.live_exchange and world == 1:
            if "WORLD_SIZE" not in os.environ or "MASTER_ADDR" not in os.environ:
                os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                                  MASTER_PORT=os.environ.get("MASTER_PORT", "29531"))
            dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        else:
            dist_setup(backend)
    pg_early = os.environ.get("OSD_BENCH_PG_AFTER") == "warm"      # A/B: warm_streams() instead of a whole step before RCCL
    if pg_early:
        eng.warm_streams()
        init_pg()
    # Autotuning: every conv shape met for the first time is timed over its candidate kernels.  With N > 1 ranks the choices
    # must not depend on each rank's own timing noise (ranks running different kernels = the slowest one sets every step): rank
    # 0's tuner cache is broadcast and replayed by the other ranks (tune_step below), so all ranks launch identical kernels.
    def tune_all():
        for bt in batches:
            with ops.tuning():
                eng.forward_backward(*bt)
            torch.cuda.synchronize()
    cache_file = os.environ.get("OSD_TUNER_CACHE")      # optional: a tuner-cache file (dist_utils.save_tuner_choices)
    if cache_file and os.path.exists(cache_file):
        from oneshotdet_amd import dist_utils
        dist_utils.load_tuner_choices(ops, cache_file)
    refined = None
    if rank == 0:
        tune_all()
        if os.environ.get("OSD_REFINE"):      # opt-in: 77 swaps tried on the default workload, none kept — 90 s for nothing
            # second tuning phase (tuner.refine_in_step): for the shapes that weigh most in the step, the runners-up of the isolated
            # timing are tried INSIDE the step and kept where the step itself gets faster twice in a row.  Before the timed region,
            # like all tuning; the other ranks replay the result
            rs = [0]

            def rstep():
                bt = batches[rs[0] % len(batches)]
                rs[0] += 1
                eng.train_step(*bt)
            # The refinement runs a few thousand steps: with the learning rate at 0 and the momentum buffer restored afterwards they leave
            # the model exactly as it was.  (They must: the step time is DATA-dependent — trained for ~1,500 steps on one synthetic batch
            # the model's scores sharpen and the training proposals' NMS has to look several times deeper, 10.3 -> 14.2 ms per step,
            # tools/sustained.py / profiles/r6_sustained.txt — and a yardstick that drifts while it measures ranks nothing.)
            dj, eng.defer_join = eng.defer_join, True
            lr0, eng.lr = eng.lr, 0.0
            buf0, steps0 = eng._sgd["buf"].clone(), eng._sgd["steps"]
            try:
                refined = ops.refine_in_step(rstep, torch.cuda.synchronize, n_steps=12, cycle=len(batches),
                                             budget_s=float(os.environ.get("OSD_REFINE_BUDGET_S", "90")),
                                             verbose=bool(os.environ.get("OSD_TUNE_VERBOSE")))
            finally:
                eng.join()
                torch.cuda.synchronize()
                eng.defer_join, eng.lr = dj, lr0
                eng._sgd["buf"].copy_(buf0)
                eng._sgd["steps"] = steps0
            torch.cuda.synchronize()
        if cache_file and not os.path.exists(cache_file):
            from oneshotdet_amd import dist_utils
            dist_utils.save_tuner_choices(ops, cache_file)
    else:
        eng.warm_streams()             # every stream gets its hardware queue before RCCL creates its own
    # Process group AFTER the first use of every stream: with RCCL initialised first the same step measured 12-14 % slower on
    # one GPU (the engine's streams land on other hardware queues).
    if not pg_early:
        init_pg()
    if world > 1:
        from oneshotdet_amd import dist_utils
        dist_utils.broadcast_tuner_choices(ops, src=0)
        if rank != 0:                  # cache hits only: nothing is timed here, a miss raises (ops.replaying)
            for bt in batches:
                with ops.replaying():
                    eng.forward_backward(*bt)
                torch.cuda.synchronize()
        assert dist_utils.tuner_choices_agree(ops), "ranks ended with different kernel choices"
    if dist.is_initialized():
        eng.attach_exchange(None, single_rank=bool(args.live_exchange), wire_dtype=wire)
    torch.cuda.synchronize()
    if os.environ.get("OSD_DUMP_ALGOS") and rank == 0:
        with open(os.environ["OSD_DUMP_ALGOS"], "w") as f:
            for k, a in ops.WGRAD_ALGO_CACHE.items():
                f.write("wgrad %s -> variant %d target_code %d\n" % (k, (a - 1) & 15, (a - 1) >> 4))
            for k, a in ops.ALGO_CACHE.items():
                f.write("conv %s -> %d\n" % (k, a))
            for k, a in ops.SPLIT_CACHE.items():
                f.write("split %s -> %s\n" % (k, a))

    launch = ("eager, 7 streams (target backbone + head chain; query backbone; 2 x weight gradients; proposals%s; "
              "pooled-query gradient chain; exchange + update)%s%s"
              % (" + second-stage box head" if args.second_stage else "",
                 "; backbones in lockstep (one launch per layer pair)" if eng.lockstep else "",
                 "; both towers per launch" if eng.towers_merged else ""))
    # the step's tail (last weight gradients, exchange, update, repack) overlaps the next step's frozen layers; an
    # explicit device synchronisation brackets the timed region as always (OSD_NO_DEFER_JOIN=1: A/B switch)
    eng.defer_join = not os.environ.get("OSD_NO_DEFER_JOIN") and not args.graph
    step_no = [0]

    def step():
        bt = batches[step_no[0] % len(batches)]
        step_no[0] += 1
        return eng.train_step(*bt)
    if args.graph and len(batches) > 1:
        raise SystemExit("--graph captures one geometry: not with --workload config5")
    if args.graph:
        try:
            eng.capture(images, queries, gt_boxes, gt_count)
            step = lambda: eng.replay_step()                                # noqa: E731
            launch = "hipGraph replay (forward+backward graph, RCCL all-reduce, optimiser graph), 2 streams"
        except Exception as e:     # capture is an optimisation: report and fall back to eager launches
            print("graph capture failed, running eagerly: %r" % (e,), file=sys.stderr)
            torch.cuda.synchronize()

    elapsed = timed_steps(step, args.steps, args.warmup, world, torch.cuda.synchronize, "cuda")
    # outside the timed region: a one-pass GroupNorm launch whose hand-off timed out has written NaN and set its error word — such a
    # run is not a measurement (raises OsdError -> non-zero exit, no JSON line)
    ops.gn_onepass_check("bench.py, after the timed training steps")
    roofline = None
    if not args.no_conv_timing:
        timer = TrainTimer()
        timer.install(ops)
        torch.cuda.synchronize()
        nst = max(2, min(args.steps, 5))
        if len(batches) > 1:
            nst = len(batches) * max(1, nst // len(batches))      # every geometry equally often
        eng.wstream = eng.wstream2 = eng.s1 = None    # one stream: concurrent kernels would stretch each other's durations
        eng._overlap = False                          # and no gradient exchange: this pass only times kernels
        per_shape = []
        for it in range(nst):
            n0 = len(timer.records)
            torch.cuda._sleep(int(150e6))
            eng.forward_backward(*batches[it % len(batches)])
            torch.cuda.synchronize()
            per_shape.append((it % len(batches), n0, len(timer.records)))
        conv_ms = timer.total_ms()
        tflops = timer.flops / (conv_ms * 1e-3) / 1e12
        corr_roofline = timer.correlation_roofline()
        traffic, traffic_src = measured_traffic("train", args.dtype, timer.launches // nst)
        roofline = {"bound": "mfma", "achieved": round(tflops, 2), "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                    "frac": round(tflops / PEAK_TFLOPS[args.dtype], 4), "traffic": traffic,
                    "peak_measured": measured_peaks(args.dtype), "traffic_source": traffic_src,
                    "kernel": "conv_sp_kernel / conv_dma_kernel / conv_pred_kernel / conv_igemm_kernel (forward + data gradient) and conv_wgrad_sk_kernel / conv_wgrad_kernel",
                    "avg_launch_us": round(conv_ms * 1e3 / max(timer.launches, 1), 2),
                    "launches_per_step": timer.launches // nst,
                    "gflop_per_step": round(timer.flops / nst / 1e9, 1),
                    "conv_ms_per_step": round(conv_ms / nst, 3),
                    "measured": "HIP events per launch (minus the %.1f us empty-bracket overhead), %d eager steps after "
                                "the timed region" % (ConvTimer.bracket_overhead_ms() * 1e3, nst)}
        # the same brackets minus what a bracket adds to a kernel launch (calibrated with a tiny kernel, see
        # ConvTimer.launch_bracket_overhead_ms): the sum of kernel durations a rocprofv3 kernel trace of these launches reports
        lov = ConvTimer.launch_bracket_overhead_ms()
        kt_ms = sum(s_.elapsed_time(e_) for s_, e_ in timer.records) - lov * len(timer.records)
        roofline["kernel_time"] = {"ms_per_step": round(kt_ms / nst, 3), "achieved": round(timer.flops / (kt_ms * 1e-3) / 1e12, 1),
                                   "frac": round(timer.flops / (kt_ms * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype], 4),
                                   "bracket_overhead_us_per_launch": round(lov * 1e3, 2),
                                   "what": "sum of kernel durations = event brackets minus the bracket's calibrated per-launch cost; "
                                           "compare with the kernel-trace row of profiles/r*_bench_train_bf16.md"}
        # the single shape that takes the most time per step: the family figures above average 264 launches, most of them
        # small or HBM-bound; this is the dominant kernel launch by itself
        timer.peak_tflops = PEAK_TFLOPS[args.dtype]
        table = timer.layer_table(nst)
        top = next((r for r in table if r[0].startswith("conv")), None)
        if top is not None:
            roofline["dominant_launch"] = {
                "kind": "%s M=%d Cout=%d K=%d (FCOS tower layer over P3+P4: forward and data gradient)" % top[:4]
                        if top[0] == "conv3x3_grouped" else "%s M=%d Cout=%d K=%d" % top[:4],
                "launches_per_step": int(round(top[4])), "avg_launch_us": round(top[5], 1),
                "gflop_per_launch": round(2.0 * top[1] * top[2] * top[3] / 1e9, 1),
                "achieved": round(top[6], 1), "frac": round(top[6] / PEAK_TFLOPS[args.dtype], 4),
                "ms_per_step": round(top[7], 3)}
        # the rows north_star's 0.6 target names — the target backbone's and FPN's forward / data-gradient convs (single launches
        # with more than 2,048 pixels; the towers are the grouped rows, the query branch the small ones) — as one figure
        bb = [r for r in table if r[0].startswith("conv") and "grouped" not in r[0] and r[1] > 2048 and r[3] > 0]
        if bb:
            gf = sum(2.0 * r[1] * r[2] * r[3] * r[4] / 1e9 for r in bb)
            ms = sum(r[7] for r in bb)
            # every row against ITS roof, min(MFMA peak, 8 TB/s x FLOP per algorithmic byte): the time the rows would take at their
            # bounds over the time they took; and how the time splits between MFMA-bound and HBM-bound rows
            ideal_ms = sum(2.0 * r[1] * r[2] * r[3] * r[4] / 1e9 / r[9] for r in bb)
            hbm_rows = [r for r in bb if r[9] < PEAK_TFLOPS[args.dtype]]
            hbm_ms = sum(r[7] for r in hbm_rows)
            hbm_gb = sum(r[8] * r[4] / 1e3 for r in hbm_rows)
            roofline["backbone_convs"] = {"gflop_per_step": round(gf, 1), "ms_per_step": round(ms, 3), "achieved": round(gf / ms, 1),
                                          "frac": round(gf / ms / PEAK_TFLOPS[args.dtype], 4),
                                          "frac_of_bound": round(ideal_ms / ms, 4), "ms_at_bound": round(ideal_ms, 3),
                                          "hbm_bound_rows": {"ms_per_step": round(hbm_ms, 3), "gb_per_step": round(hbm_gb, 2),
                                                             "achieved_tb_s": round(hbm_gb / max(hbm_ms, 1e-9), 2),
                                                             "frac_of_8_tb_s": round(hbm_gb / max(hbm_ms, 1e-9) / PEAK_HBM_TBS, 4)},
                                          "mfma_bound_rows": {"ms_per_step": round(ms - hbm_ms, 3),
                                                              "achieved": round((gf - sum(2.0 * r[1] * r[2] * r[3] * r[4] / 1e9 for r in hbm_rows)) / max(ms - hbm_ms, 1e-9), 1)},
                                          "what": "forward + data-gradient launches of the backbones / FPN with M > 2048 (--layer-table rows "
                                                  "conv1x1 / conv3x3 / conv7x1, not grouped); bound per row = min(MFMA peak, 8 TB/s x FLOP per "
                                                  "algorithmic byte), frac_of_bound = time at the bounds / time taken"}
        if len(batches) > 1:
            # the dominant launch of every geometry by itself (the tower layer over that geometry's P3 + P4, or P3..P7)
            roofline["per_geometry"] = []
            ov = ConvTimer.bracket_overhead_ms()
            for gi, (H, W) in enumerate(shapes):
                agg = {}
                for g2, a, b in per_shape:
                    if g2 != gi:
                        continue
                    for (s_, e_), lab in zip(timer.records[a:b], timer.labels[a:b]):
                        if lab[0].startswith("conv"):
                            v = agg.setdefault(lab[:4], [0, 0.0, 0.0])
                            v[0] += 1
                            v[1] += s_.elapsed_time(e_) - ov
                            v[2] += lab[4]
                k, v = max(agg.items(), key=lambda kv: kv[1][1])
                fam_ms = sum(v2[1] for v2 in agg.values())
                fam_fl = sum(v2[2] for v2 in agg.values())
                roofline["per_geometry"].append({
                    "target": "%dx%d" % (H, W), "dominant_launch": "%s M=%d Cout=%d K=%d" % k,
                    "avg_launch_us": round(v[1] * 1e3 / v[0], 1), "achieved": round(v[2] / (v[1] * 1e-3) / 1e12, 1),
                    "frac": round(v[2] / (v[1] * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype], 4),
                    "fwd_dgrad_conv_family_tflops": round(fam_fl / (fam_ms * 1e-3) / 1e12, 1)})
        if args.layer_table and rank == 0:
            with open(args.layer_table, "w") as f:
                f.write("| kind | M (pixels) | Cout | K | launches/step | us/launch | TFLOP/s | ms/step | MB/launch | TB/s | bound TFLOP/s | frac of bound |\n"
                        "|---|---|---|---|---|---|---|---|---|---|---|---|\n")
                for r in table:
                    f.write("| %s | %d | %d | %d | %.0f | %.1f | %.0f | %.3f | %.1f | %.2f | %.0f%s | %.2f |\n"
                            % (r[:9] + (r[8] / max(r[5], 1e-9), r[9], " (hbm)" if r[9] < PEAK_TFLOPS[args.dtype] else "", r[10])))
        timer.uninstall(ops)
    if rank == 0:
        what = ("%s MFMA convs, forward (two R-50-FPN backbones, query pooling, correlation, FCOS head, training proposals) + FCOS "
                "loss + backward (stem/layer1 frozen) + gradient all-reduce + SGD(momentum) + weight repack" % args.dtype)
        if args.workload == "config5":
            workload = ("BASELINE.json configs[4] (per-GPU workload): bs=%d/GPU, multi-scale targets %s cycling step by step + %d "
                        "x 127x127 queries per image (mean-pooled correlation), %s"
                        % (B, " / ".join("%dx%d" % hw for hw in shapes), S, what))
        else:
            workload = "BASELINE.json configs[2]: bs=%d/GPU, 800x1024 target + 127x127 query, %s" % (B, what)
        if two:
            workload += (" + second stage (subsample 128 ROIs per image, few-shot ROI box head forward, cross-entropy + "
                         "smooth-L1, backward into both backbones)")
        cpu = cpu_baseline(args.dtype, train=True, shapes=shapes, shots=S) if (world == 1 and not args.no_cpu_baseline) else None
        line = result_line(args, world, B, elapsed, workload, launch, roofline, cpu)
        if args.workload == "config5":
            line["metric"] = "images/sec/GPU fwd+bwd, multi-scale {640,800,1024} short-edge target + 5-shot 127x127 query stack, bs=4"
        # which part of the reference's training step (engine/trainer.py:79-93) the line covers: the north_star hot path is
        # the siamese-FCOS first stage; the reference's trainer also runs roi_heads on its proposals
        line["config"]["tuner"] = ("per-shape kernel choice by isolated timing before the timed region (latency-sized convs timed with cold "
                                   "weights)" + ("" if refined is None else "; second phase inside the step (tuner.refine_in_step): %d of the "
                                   "heaviest shapes' choices replaced by a runner-up that made the step itself faster twice" % len(refined)))
        line["config"]["stages"] = ("first + second stage: the reference's complete training step" if two else
                                    "first stage (SURVEY.md 8a R0-R12 = BASELINE.json north_star); the reference's trainer also "
                                    "runs the second-stage roi_heads losses/backward: that step is `--second-stage`, do not "
                                    "compare this line with a full-step number")
        if roofline is not None:
            line["roofline_correlation"] = corr_roofline
            # the same algorithmic FLOP against the WALL time of the timed multi-stream step (the figure above divides by
            # the sum of isolated kernel times, which exceeds the wall step because streams overlap)
            in_step = roofline["gflop_per_step"] / (elapsed / args.steps * 1e3)          # GFLOP / ms = TFLOP/s
            roofline["in_step"] = {"achieved": round(in_step, 1), "frac": round(in_step / PEAK_TFLOPS[args.dtype], 4),
                                   "what": "conv-family GFLOP per step / ms_per_step of the timed region (everything else "
                                           "in the step - GroupNorm, loss, proposals, update - counts as time, not FLOP)"}
        if world > 1:
            assert eng.exchange.active and eng.exchange.world == world
            line["config"]["parallelism"] += ("; fp32 gradient averaging (%s all-reduce): %d buckets of the flat buffer, each "
                                              "exchanged behind backward as soon as it is final"
                                              % (dist.get_backend(), len(eng.exchange.ranges)))
        elif args.live_exchange:
            assert eng.exchange.active and eng.exchange.world == 1
            line["config"]["parallelism"] = ("dp1 with the gradient exchange LIVE: one rank, every bucket all-reduced by %s on the "
                                             "communication stream behind backward (%s on the wire) - the N > 1 step's stream "
                                             "topology on one GPU" % (dist.get_backend(), args.grad_wire))
        else:
            assert not eng.exchange.active
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default=os.environ.get("OSD_BENCH_DTYPE", ""), choices=["f32", "bf16"])
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default: 8; 4 for --workload config5)")
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS),
                    help="train mode: config3 = BASELINE.json configs[2]/[3], the headline (800x1024, 1 shot, bs 8); config5 = "
                         "configs[4]'s per-GPU workload (multi-scale 640/800/1024 short edge, 5-shot query stack, bs 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-conv-timing", action="store_true")
    ap.add_argument("--layer-table", default="", help="train mode: write the per-shape conv table (markdown) here")
    ap.add_argument("--no-graph", action="store_true", help="forward mode: eager launches instead of hipGraph replay")
    ap.add_argument("--graph", action="store_true",
                    help="train mode: replay the step from hipGraphs (measured slower than eager 2-stream launches: the "
                         "step is GPU-bound, so eager is the default)")
    ap.add_argument("--mode", default=os.environ.get("OSD_BENCH_MODE", "train"), choices=["train", "forward"],
                    help="train = forward + loss + backward + gradient all-reduce + SGD (the headline metric); "
                         "forward = BASELINE.json configs[1] (inference forward incl. proposals)")
    ap.add_argument("--second-stage", action="store_true",
                    help="forward mode: also run the few-shot ROI box head on the 2000 proposals per image (SURVEY.md 8f "
                         "#1) = the reference's complete eval forward; train mode: also the second stage's losses and "
                         "backward = the reference's complete training step (engine/trainer.py:79-93)")
    ap.add_argument("--live-exchange", action="store_true",
                    help="N = 1, train mode: initialise a ONE-rank RCCL process group and run every bucket's all-reduce on the "
                         "communication stream anyway (what the step costs with the exchange's streams and kernels live)")
    ap.add_argument("--grad-wire", default="f32", choices=["f32", "bf16"], help="dtype of the gradient buckets on the wire")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="exercise only the multi-process plumbing (gloo, no GPU work): used by tests/test_dist_cpu.py")
    args = ap.parse_args()

    if not args.batch:
        args.batch = WORKLOAD_BATCH[args.workload]
    rank, local_rank, world = check_world(args)      # may start the ranks and exit; never touches the GPU
    if args.dry_run_cpu:
        dist_setup("gloo")
        work = torch.zeros(1)

        def fake_step():
            work.add_(1.0 + rank)
            time.sleep(0.002 * (1 + rank))       # rank-dependent so MAX-over-ranks is observable
        elapsed = timed_steps(fake_step, args.steps, args.warmup, world, lambda: None, "cpu")
        if rank == 0:
            print(json.dumps(result_line(args, world, args.batch, elapsed, "dry run (no GPU work)", "cpu dry run")),
                  flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    # OSD_BENCH_SHARE_GPU=1 (tests only): every rank on cuda:0 with the gloo backend, so the exact N > 1 code path
    # (rank-seeded data, overlapped exchange, barrier-bracketed timing, MAX over ranks) runs on a one-GPU box
    share = os.environ.get("OSD_BENCH_SHARE_GPU") == "1"
    torch.cuda.set_device(0 if share else local_rank)
    if not args.dtype:      # configs[2] (train) is bf16, configs[1] (forward parity config) is fp32
        args.dtype = "bf16" if args.mode == "train" else "f32"
    if args.mode == "train":
        # the process group is initialised INSIDE main_train, after the engine's streams have been used once
        return main_train(args, rank, world, "gloo" if share else "nccl")
    dist_setup("gloo" if share else "nccl")

    from oneshotdet_amd import model, ops, spec, synth
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    shapes = spec.full_model_shapes() if args.second_stage else spec.hot_path_shapes()
    eng = model.HotPathEngine(synth.make_state_dict(shapes), dtype=dtype)
    B = args.batch
    images = torch.from_numpy(synth.make_images("bench.target", B, 800, 1024, seed=1000 + rank)).cuda()
    queries = torch.from_numpy(synth.make_images("bench.query", B, 127, 127, seed=1000 + rank)).cuda()

    two = bool(args.second_stage)
    eng.tune(images, queries, second_stage=two)   # per-shape conv algorithm selection by measurement (untimed, once)
    use_graph = not args.no_graph
    if use_graph:
        # production path: the whole forward captured once into a hipGraph (multi-stream branches), then replayed
        runner = model.GraphedDetect(eng, images, queries, second_stage=two)

        def step():
            return runner()
    else:
        def step():
            return eng.detect(images, queries, second_stage=two)

    elapsed = timed_steps(step, args.steps, args.warmup, world, torch.cuda.synchronize, "cuda")
    ops.gn_onepass_check("bench.py, after the timed forward passes")      # see main_train

    # Roofline of the dominant kernel family (conv_igemm): HIP events around every conv launch, on the stream the
    # kernel is launched on, over `steps` further steps of the SAME workload in this process, run eagerly on one
    # stream (kernels inside a replayed hipGraph cannot be bracketed individually, and concurrent branches would
    # inflate each other's durations).
    roofline = None
    if not args.no_conv_timing:
        timer = ConvTimer()
        timer.install(ops)
        torch.cuda.synchronize()
        for _ in range(args.steps):
            # park the stream behind a spin kernel while the host enqueues the whole step, so every bracket measures
            # GPU time only (the host needs ~40 us per bracketed launch, more than the short kernels run)
            torch.cuda._sleep(int(60e6))
            eng.detect(images, queries, concurrent=False, second_stage=two)
            torch.cuda.synchronize()
        conv_ms = timer.total_ms()
        tflops = timer.flops / (conv_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": round(tflops, 2), "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                    "frac": round(tflops / PEAK_TFLOPS[args.dtype], 4), "traffic": None,
                    "peak_measured": measured_peaks(args.dtype),
                    "kernel": "conv_igemm_kernel (all instantiations)",
                    "avg_launch_us": round(conv_ms * 1e3 / max(timer.launches, 1), 2),
                    "launches_per_step": timer.launches // max(args.steps, 1),
                    "gflop_per_step": round(timer.flops / max(args.steps, 1) / 1e9, 1),
                    "conv_ms_per_step": round(conv_ms / max(args.steps, 1), 3),
                    "measured": "HIP events per launch (minus the %.1f us empty-bracket overhead), %d eager "
                                "single-stream steps after the timed region"
                                % (ConvTimer.bracket_overhead_ms() * 1e3, args.steps)}
        timer.uninstall(ops)

    if rank == 0:
        workload = ("BASELINE.json configs[1] (forward-only parity/inference config; --mode train is the headline): bs=%d/GPU, 800x1024 target + "
                    "127x127 query, two R-50-FPN backbones + query pooling + correlation + FCOS head + proposals "
                    "(top-k, NMS 0.8, top-2000)%s, %s MFMA convs"
                    % (B, " + second-stage ROI box head on the proposals (7x7 level-routed ROIAlign, concat with the "
                          "query ROI map, 3 conv+GN+LeakyReLU, fc6, fc7, predictor, decode, NMS 0.5)" if two else "",
                       args.dtype))
        cpu = cpu_baseline(args.dtype) if (world == 1 and not args.no_cpu_baseline) else None
        line = result_line(args, world, B, elapsed, workload,
                           "hipGraph replay, 4 streams" if use_graph else "eager, 4 streams", roofline, cpu)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
