"""Data-parallel training step on the real HIP path: two ranks (two processes) share the one GPU of the test box and
all-reduce through gloo (RCCL refuses two ranks on one device; the Trainer code path -- per-key asynchronous bucket
all-reduce launched as each backward segment completes, 1/world folded into Adam, averaged Lagrangian scalars -- is the
same one bench.py drives over RCCL).  Checks:
  * both ranks hold bit-identical parameters and Lagrangian state after two steps;
  * the all-reduced gradient of every optimizer key equals the sum of the two ranks' local gradients, each recomputed by
    a single-process trainer fed that rank's shard and noise."""
import copy
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VGG_W = (8, 8, 16, 16, 16)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg():
    from oracle import configs
    cfg = copy.deepcopy(configs.tiny_config())
    cfg.update(precision="fp32", vgg_widths=VGG_W)
    return cfg


def _shard(rank):
    from oracle import configs as oc, ref_model as R
    cfg = oc.tiny_config()
    views = R.synthetic_views(cfg, seed=1234 + rank)
    noise = R.synthetic_noise(cfg, seed=4321 + rank)
    return views, noise


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import upsparts_amd  # noqa: F401
    from upsparts_amd import dist as D
    from upsparts_amd.model import TrainModel, Trainer
    D.init_from_env("gloo")
    dev = torch.device("cuda:0")
    cfg = _cfg()
    model = TrainModel(cfg, device=dev, seed=0)
    tr = Trainer(cfg, None, model, world_size=world, rank=rank)
    views, noise = _shard(rank)
    grads0 = None
    for step in range(2):
        tr.train_step(views, noise)
        if step == 0:
            grads0 = {k: g["flat"]["g"].detach().cpu().clone() for k, g in model.bank.groups.items()}
    torch.cuda.synchronize()
    # the encoder_0 head slice is all-reduced early (Trainer._hook_early_reduce): the hook must be installed and consumed
    assert tr._early_hooked and not tr._early
    assert any(lay.after_wgrad is not None for lay in model.nets.layers.values())
    out[rank] = {"params": {k: g["flat"]["p"].detach().cpu() for k, g in model.bank.groups.items()},
                 "grads0": grads0, "state": {k: float(v) for k, v in tr.state.items()}}
    torch.distributed.destroy_process_group()


def test_two_rank_step_matches_sum_of_local_gradients(dev):
    import upsparts_amd  # noqa: F401
    from upsparts_amd.model import TrainModel, Trainer
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), "replicas diverged: " + k
        assert torch.equal(r0["grads0"][k], r1["grads0"][k]), "all-reduced gradients differ between ranks: " + k
    for k in r0["state"]:
        assert r0["state"][k] == r1["state"][k], "Lagrangian / EMA state diverged: " + k
    # local gradients of step 0, recomputed without any collective
    local = []
    for rank in range(2):
        cfg = _cfg()
        model = TrainModel(cfg, device=dev, seed=0)
        tr = Trainer(cfg, None, model)
        views, noise = _shard(rank)
        tr.train_step(views, noise)
        local.append({k: g["flat"]["g"].detach().cpu().clone() for k, g in model.bank.groups.items()})
    for k in r0["grads0"]:
        want = local[0][k] + local[1][k]
        err = float((r0["grads0"][k] - want).abs().max() / max(float(want.abs().max()), 1e-12))
        assert err <= 1e-5, "bucket {}: all-reduced gradient vs sum of local gradients, rel err {:.2e}".format(k, err)


def _graph_worker(rank, world, port, out, use_graph, steps):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import upsparts_amd  # noqa: F401
    from upsparts_amd import dist as D
    from upsparts_amd.model import TrainModel, Trainer
    D.init_from_env("gloo")
    dev = torch.device("cuda:0")
    cfg = _cfg()
    cfg["hip_graph"] = use_graph
    model = TrainModel(cfg, device=dev, seed=0)
    tr = Trainer(cfg, None, model, world_size=world, rank=rank)
    views, noise = _shard(rank)
    for _ in range(steps):
        losses = tr.train_step(views, noise)
    torch.cuda.synchronize()
    nseg = len(tr._g["graph"]["graphs"]) if use_graph and tr._g and tr._g["graph"] else 0
    out[(use_graph, rank)] = {"params": {k: g["flat"]["p"].detach().cpu() for k, g in model.bank.groups.items()},
                              "state": {k: float(v) for k, v in tr.state.items()}, "segments": nseg,
                              "losses": {k: float(v) for k, v in losses.items()}, "step": tr.global_step}
    torch.distributed.destroy_process_group()


def test_hip_graph_replay_under_data_parallelism(dev):
    """`hip_graph: True` with world_size 2: the step is captured as a sequence of HIP graphs cut at the collectives (bucket
    all-reduces after each backward segment, the averaged Lagrangian scalars) and replayed with the collectives run eagerly in
    between.  After five steps (two eager warm-up steps, the capture, replays) parameters, Lagrangian / EMA state and losses are
    bit-identical to the eager two-rank run, on both ranks."""
    mgr = mp.Manager()
    out = mgr.dict()
    for use_graph in (False, True):
        mp.spawn(_graph_worker, args=(2, _free_port(), out, use_graph, 5), nprocs=2, join=True)
    for rank in range(2):
        e, g = out[(False, rank)], out[(True, rank)]
        assert g["segments"] >= 5 and g["step"] == e["step"] == 5, (g["segments"], g["step"])
        for k in e["params"]:
            assert torch.equal(e["params"][k], g["params"][k]), "graph replay diverged from the eager DP step: {} (rank {})".format(k, rank)
        assert e["state"] == g["state"] and e["losses"] == g["losses"]
    for k in out[(True, 0)]["params"]:
        assert torch.equal(out[(True, 0)]["params"][k], out[(True, 1)]["params"][k]), "replicas diverged: " + k


def test_rccl_call_pattern_at_world_size_one(dev):
    """tools/nccl_trainer_check.py in a child process: a world-size-1 NCCL (= RCCL) group with UPS_FORCE_COLLECTIVES=1 issues
    every bucket all-reduce where the multi-GPU run does (asynchronously, inside backward, early encoder_0 head slice) and
    must leave the parameters bit-identical to a run without collectives."""
    import subprocess
    for attempt in range(2):
        env = dict(os.environ, UPS_FORCE_COLLECTIVES="1", MASTER_PORT=str(_free_port()))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_trainer_check.py")], env=env, capture_output=True,
                           text=True, timeout=600)
        if r.returncode == 0 and "nccl trainer check ok" in r.stdout:
            return
        # a numerical mismatch (the script's own asserts) is a failure at once; a rendezvous / RCCL start-up hiccup of the child
        # process (seen once in ~10 full-suite runs on the pool's boxes) gets one more try on a fresh port
        if "AssertionError" in r.stderr or attempt == 1:
            break
    assert False, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_two_ranks_over_rccl():
    """`python bench.py --gpus 2` on a box with >= 2 GPUs: the launcher starts two ranks over RCCL and rank 0's line says so."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["config"]["rccl_world_size"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["global_batch"] == 8 and out["value"] > 0


def _runner_rank(rank, world, port, ypath, root):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      UPS_DIST_BACKEND="gloo")
    import upsparts_amd  # noqa: F401
    from upsparts_amd import runner
    it = runner.main(["-t", ypath, "-p", root, "--num_steps", "3"])
    assert it.world_size == world and it.rank == rank
    torch.save({k: g["flat"]["p"].detach().cpu() for k, g in it.model.bank.groups.items()}, os.path.join(root, "params_rank{}.pt".format(rank)))
    torch.distributed.destroy_process_group()


def test_runner_wires_data_parallelism(dev, tmp_path):
    """`torchrun ... -m upsparts_amd.runner -t yaml`: every rank joins the process group, trains on its own shard with
    all-reduced gradients (identical replicas), and only rank 0 writes the log and the checkpoints."""
    import yaml
    from oracle import configs
    cfg = copy.deepcopy(configs.tiny_config())
    cfg.update(precision="fp32", vgg_widths=list(VGG_W), ckpt_freq=2, dataset="no.such.Dataset")     # -> synthetic pairs, per-rank seed
    ypath = tmp_path / "t.yaml"
    ypath.write_text(yaml.safe_dump(cfg))
    root = str(tmp_path / "run")
    os.makedirs(root)
    mp.spawn(_runner_rank, args=(2, _free_port(), str(ypath), root), nprocs=2, join=True)
    p0, p1 = torch.load(os.path.join(root, "params_rank0.pt")), torch.load(os.path.join(root, "params_rank1.pt"))
    for k in p0:
        assert torch.equal(p0[k], p1[k]), "replicas diverged: " + k
    log = open(os.path.join(root, "train", "log.txt")).read()
    assert log.count("global_step: 0\n") == 1                      # one writer
    assert sorted(os.listdir(os.path.join(root, "train", "checkpoints"))) == ["model.ckpt-2", "model.ckpt-3"]


def _resume_worker(rank, world, port, out, ckpt):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import upsparts_amd  # noqa: F401
    from upsparts_amd import dist as D
    from upsparts_amd.model import TrainModel, Trainer
    D.init_from_env("gloo")
    dev = torch.device("cuda:0")
    cfg = _cfg()
    model = TrainModel(cfg, device=dev, seed=0)
    tr = Trainer(cfg, None, model, world_size=world, rank=rank)
    views, noise = _shard(rank)
    tr.train_step(views, noise)
    before = tr._gen.get_state().cpu()
    if rank == 0:                       # only rank 0 writes checkpoints (Trainer._checkpoint)
        tr.save_checkpoint(ckpt)
    torch.distributed.barrier()
    tr.initialize(ckpt)
    after = tr._gen.get_state().cpu()
    draw = torch.randint(0, 1 << 30, (4,), generator=tr._gen, device=dev).cpu()
    out[rank] = {"before": before, "after": after, "draw": draw, "noise_seed": tr._noise.seed if hasattr(tr._noise, "seed") else None}
    torch.distributed.destroy_process_group()


def test_resume_keeps_the_ranks_generators_apart(dev, tmp_path):
    """Round-5 advisor: only rank 0 writes the checkpoint, and every rank used to restore ITS generator state from it -- after a
    resume all ranks drew the same TPS uniforms and crop windows.  Now the state is restored on the rank / world size that wrote it
    only; the others re-seed from their own shard seed and the restored step."""
    mgr = mp.Manager()
    out = mgr.dict()
    ckpt = str(tmp_path / "model.ckpt-1")
    mp.spawn(_resume_worker, args=(2, _free_port(), out, ckpt), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert torch.equal(r0["after"], r0["before"]), "rank 0 must continue its own stream"
    assert not torch.equal(r1["after"], r0["after"]), "rank 1 resumed with rank 0's generator state"
    assert not torch.equal(r0["draw"], r1["draw"])
    # a single-rank restore of the same file (another world size) does not take rank 0's state of the two-rank run either
    from upsparts_amd.model import TrainModel, Trainer
    cfg = _cfg()
    tr = Trainer(cfg, None, TrainModel(cfg, device=dev, seed=0))
    tr.initialize(ckpt)
    assert not torch.equal(tr._gen.get_state().cpu(), r0["after"])
