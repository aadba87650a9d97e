"""RCCL call pattern of the trainer on ONE GPU: a world-size-1 NCCL group with UPS_FORCE_COLLECTIVES=1 runs every bucket
all-reduce (identity) exactly where the multi-GPU run issues it -- asynchronously, from inside the backward pass, beside the
side streams, with the early encoder_0 head slice -- and the result must equal the run without collectives bit for bit.  A third
run adds `hip_graph: True`: the step captured as a sequence of HIP graphs cut at the collectives, RCCL all-reduces issued eagerly
between the replayed segments (model.Trainer._capture_step).
Usage: UPS_FORCE_COLLECTIVES=1 python tools/nccl_trainer_check.py [towers]"""
import copy, os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("UPS_FORCE_COLLECTIVES", "1")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29544"), RANK="0", WORLD_SIZE="1")
import upsparts_amd  # noqa
from upsparts_amd import configs, dist as D
from upsparts_amd.model import TrainModel, Trainer
assert D.FORCE_COLLECTIVES
dev = torch.device("cuda:0")
cfg = copy.deepcopy(configs.cub_config(n_parts=4, batch_size=2, spatial_size=32))      # the CUB yaml at reduced widths
cfg.update(precision="bf16", vgg_widths=(8, 8, 16, 16, 16), patch_size=8, z0_size=16, local_app_size=16)
if len(sys.argv) > 1 and sys.argv[1] == "towers":      # the yaml's latent widths (256 / 64): the critics run as grouped launches (ops.TowersFn)
    cfg.update(z0_size=256, local_app_size=64)
cfg["encoder0"].update(config=[16, 32, 32, 64], extra_resnets=1)
cfg["encoder1"].update(config=[16, 32, 32, 64], extra_resnets=1)
cfg["dv"].update(config=[8, 16, 32, 40], upsample_config=["linear"] * 3)
cfg["final_hour"].update(config=[16, 32])
g = torch.Generator().manual_seed(3)
B, S = cfg["batch_size"], cfg["spatial_size"]
batches = [{k: torch.rand(B, S, S, 3, generator=g) * 2 - 1 for k in ("view0", "view1", "view0_target")} for _ in range(5)]
Z, P = cfg["z0_size"], cfg["n_parts"]
noises = [{"eps_pi0": torch.randn(7, B, Z, generator=g), "eps_pi1": torch.randn(B, Z, generator=g),
           "eps_l0": torch.randn(B, S, S, P, generator=g), "eps_l1": torch.randn(B, S, S, P, generator=g)} for _ in range(5)]
res = []
for use_nccl, use_graph in ((False, False), (True, False), (True, True)):
    D.FORCE_COLLECTIVES = use_nccl
    if use_nccl and not dist.is_initialized():
        dist.init_process_group("nccl", world_size=1, rank=0)
    c = dict(cfg, hip_graph=use_graph)
    model = TrainModel(c, device=dev, seed=0)
    tr = Trainer(c, None, model)
    for b, nz in zip(batches, noises):
        tr.train_step(b, nz)
    torch.cuda.synchronize()
    res.append({k: grp["flat"]["p"].detach().cpu().clone() for k, grp in model.bank.groups.items()})
    if use_nccl:
        assert tr._early_hooked
    if use_graph:
        nseg = len(tr._g["graph"]["graphs"])
        assert nseg >= 5, nseg            # four bucket boundaries (world size 1 has no scalar boundary) -> five graphs
        print("graph mode: {} segments".format(nseg))
dist.destroy_process_group()
for k in res[0]:
    assert torch.equal(res[0][k], res[1][k]), k
    assert torch.equal(res[0][k], res[2][k]), "graph + RCCL: " + k
print("nccl trainer check ok")
