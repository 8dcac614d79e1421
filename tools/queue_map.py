"""GPU box: which hardware queue each engine stream landed on, from a rocprofv3 kernel trace of bench.py (train mode): a marker
kernel per stream — main: fcos_loss_finalize, s1 (query branch + bbox tower): roialign_bwd, p (proposals): nms_scan, u (update):
sgd_pack_multi, w / w2 (weight gradients): the two queues conv_wgrad_sk runs on.  usage: python tools/queue_map.py run_kernel_trace.csv"""
import collections
import csv
import sys

marks = {"fcos_loss_finalize": "main", "roialign_bwd": "s1", "nms_scan": "p", "sgd_pack_multi": "u", "conv_wgrad_sk": "w/w2", "pred_dy_gather": "w/w2"}
seen = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    for k, v in marks.items():
        if k in r["Kernel_Name"]:
            seen[v][r["Queue_Id"]] += 1
print({k: dict(v) for k, v in seen.items()})
