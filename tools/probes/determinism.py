"""Race detector for the multi-stream schedule: two trainers from the same seed run N full-size steps (B = 64, 128x128, the bench
config) on the same inputs / noise seeds; every parameter must end bit-identical.  Usage: python3 tools/probes/determinism.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import upsparts_amd  # noqa: E402,F401
from upsparts_amd import configs  # noqa: E402
from upsparts_amd.model import TrainModel, Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
sums = []
for run in range(2):
    cfg = configs.cub_config(n_parts=10, batch_size=64)
    cfg["precision"] = os.environ.get("PRECISION", "bf16")
    model = TrainModel(cfg, device=dev, seed=0)
    tr = Trainer(cfg, None, model)
    tr._gen = torch.Generator(device=dev).manual_seed(7) if hasattr(tr, "_gen") and tr._gen is not None and tr._gen.device.type == "cuda" else tr._gen
    g = torch.Generator().manual_seed(1234)
    batch = {k: (torch.rand(64, 128, 128, 3, generator=g) * 2 - 1).to(dev) for k in model.inputs}
    torch.manual_seed(99); torch.cuda.manual_seed_all(99)
    for s in range(steps):
        tr.train_step(batch)
    torch.cuda.synchronize()
    sums.append({k: grp["flat"]["p"].detach().double().sum().item() for k, grp in model.bank.groups.items()})
    sums[-1]["_bits"] = {k: grp["flat"]["p"].detach().clone() for k, grp in model.bank.groups.items()}
ok = all(torch.equal(sums[0]["_bits"][k], sums[1]["_bits"][k]) for k in sums[0]["_bits"])
finite = all(torch.isfinite(v).all().item() for v in sums[0]["_bits"].values())
print("steps", steps, "precision", os.environ.get("PRECISION", "bf16"), "bit-identical", ok, "finite", finite,
      {k: round(v, 4) for k, v in sums[0].items() if k != "_bits"})
sys.exit(0 if ok and finite else 1)
