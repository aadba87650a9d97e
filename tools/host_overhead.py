"""Host enqueue time vs GPU time of one training step (is the step launch-bound?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd import configs
from upsparts_amd.model import TrainModel, Trainer
dev = torch.device("cuda:0")
cfg = configs.cub_config(n_parts=10, batch_size=64); cfg["precision"] = "bf16"
model = TrainModel(cfg, device=dev, seed=0); tr = Trainer(cfg, None, model)
g = torch.Generator().manual_seed(1)
batch = {k: (torch.rand(64, 128, 128, 3, generator=g) * 2 - 1).to(dev) for k in ("view0", "view1", "view0_target")}
for _ in range(2): tr.train_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4): tr.train_step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue ms/step", (t1 - t0) / 4 * 1e3, "total ms/step", (t2 - t0) / 4 * 1e3)
if len(sys.argv) > 1 and sys.argv[1] == "profile":      # where the host time goes: cProfile over 6 steps, top functions by own time
    import cProfile, pstats
    pr = cProfile.Profile()
    with torch.autograd.set_multithreading_enabled(False):      # backward on this thread, so that the profile sees it
        tr.train_step(batch)
        pr.enable()
        for _ in range(6): tr.train_step(batch)
        pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
