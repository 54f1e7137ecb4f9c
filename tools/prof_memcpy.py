import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench
from heatnet_amd import optim
wl = sys.argv[1] if len(sys.argv) > 1 else "seg"
dev = torch.device("cuda:0")
model = bench.build(wl, dev)
arenas = optim.arenas_for(model)
opt = optim.FlatRMSprop(arenas, lr=1e-4)
batch = bench.synthetic_batch(0, dev, 2, 320, 704)
step = bench.make_step(wl, model, opt, batch, None)
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
evs = [e for e in prof.events() if "copy" in e.name.lower() or "Memcpy" in e.name]
from collections import Counter
c = Counter()
for e in evs:
    st = [f for f in (e.stack or []) if "heatnet" in f or "bench" in f]
    c[(e.name, tuple(st[:3]))] += 1
for (name, st), n in c.most_common(25):
    print(n, name, " <- ".join(s.split("/")[-1] for s in st))
