"""GPU box: tune the bs=8 workload and print the chosen conv algorithm per shape."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oneshotdet_amd import model, ops, spec, synth
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
eng = model.HotPathEngine(synth.make_state_dict(spec.hot_path_shapes()), dtype=dt)
images = torch.from_numpy(synth.make_images("bench.target", 8, 800, 1024, seed=1000)).cuda()
queries = torch.from_numpy(synth.make_images("bench.query", 8, 127, 127, seed=1000)).cuda()
eng.tune(images, queries)
names = {0: "dma", 1: "reg"}
for k, a in sorted(ops.ALGO_CACHE.items(), key=lambda kv: -kv[0][1]):
    a0 = a - 1
    print("M=%8d N=%5d cin=%5d %dx%d s%d res%d act%d -> %s v%d tile%d" % (k[1], k[2], k[3], k[4], k[5], k[6], k[8], k[9], names[a0 >> 5], (a0 >> 3) & 3, a0 & 7))
