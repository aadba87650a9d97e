"""Headline benchmark: images/sec of full training steps (forward, seven per-key losses, per-key
gradients, TF-Adam, gradient all-reduce) of the part-discovery model on synthetic 128x128 batches.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): CUB yaml, 128x128, n_parts 10, batch 64 per GPU, bf16 activations /
weights with fp32 accumulation and fp32 master weights, use_tps False, synthetic U(-1,1) views resident in
HBM before the timed region, noise drawn on device inside the step.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAIN_GFLOP_PER_IMAGE = 268.9      # BASELINE.md section 3, 128^2, P=10 (F_alg)
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (guide)
PEAK_F32_TFLOPS = 157.3


def cpu_baseline(cfg_fn, batch, threads, budget_s=20.0):
    """The CPU oracle (restatement of the reference path, kind 'port') timed on a bounded sample:
    one warm-up step, then whole training steps until ~budget_s seconds of CPU work (at most 8 steps)."""
    from oracle import ref_model as R
    cfg = cfg_fn(batch)
    torch.set_num_threads(threads)
    params = R.init_params(cfg, 0)
    vp = R.vgg_params(7)
    views = R.synthetic_views(cfg, smooth=False)
    noise = R.synthetic_noise(cfg)
    adam = R.init_adam(params)
    state = R.initial_state(cfg)
    params, adam, state = R.train_step(params, adam, cfg, views, noise, state, 0, vp, dtype=torch.float32, scheme="merged")[:3]
    steps, t0 = 0, time.time()
    while steps < 8 and (steps == 0 or time.time() - t0 < budget_s):
        params, adam, state = R.train_step(params, adam, cfg, views, noise, state, steps + 1, vp, dtype=torch.float32,
                                           scheme="merged")[:3]
        steps += 1
    dt = time.time() - t0
    return {"value": round(batch * steps / dt, 4), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "{} training steps (after 1 warm-up), batch {}, 128x128, P=10, fp32 torch-CPU restatement (oracle), "
                      "{} threads, {:.1f} s".format(steps, batch, threads, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--parts", type=int, default=10)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        torch.distributed.init_process_group("nccl", init_method="env://", world_size=world, rank=rank)
    dev = torch.device("cuda:{}".format(local))
    torch.cuda.set_device(dev)

    import upsparts_amd  # noqa: F401
    from upsparts_amd import configs, ops
    from upsparts_amd.model import TrainModel, Trainer

    cfg = configs.cub_config(n_parts=args.parts, batch_size=args.batch)
    cfg["precision"] = args.precision
    model = TrainModel(cfg, device=dev, seed=0)
    trainer = Trainer(cfg, None, model, world_size=world, rank=rank)
    g = torch.Generator().manual_seed(1234 + rank)
    batch = {k: (torch.rand(args.batch, 128, 128, 3, generator=g) * 2 - 1).to(dev)
             for k in ("view0", "view1", "view0_target")}

    ops.KernelTimer.layer = "decoder_visualize/conv2d_8"       # the 258->256 3x3 conv at 128x128 (37% of forward FLOPs)
    for _ in range(args.warmup):
        trainer.train_step(batch)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    ops.KernelTimer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.train_step(batch)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.KernelTimer.enabled = False
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax)

    if rank == 0:
        images = args.batch * world * args.steps
        value = images / dt
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
        traffic = None
        try:   # HBM bytes per launch of the roofline kernel from the committed PMC passes (profiles/), not measured live
            with open(os.path.join(ROOT, "profiles", "round1_v15_pmc_dv_rb128.json")) as f:
                traffic = json.load(f)["traffic_bytes_per_launch"] * (args.batch / 64.0) if args.precision == "bf16" else None
        except Exception:
            traffic = None
        kms = ops.KernelTimer.mean_ms()
        ach = ops.KernelTimer.flops / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        out = {"metric": "images/sec training (CUB 128x128, 10 parts)", "value": round(value, 2), "unit": "images/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
               "config": {"workload": "CUB yaml 128x128 n_parts={} batch {}/GPU, full train step (7 per-key losses, "
                                      "per-key grads, TF-Adam), use_tps False, VGG19-topology perceptual trunk with "
                                      "stand-in weights at native 128x128".format(args.parts, args.batch),
                          "global_batch": args.batch * world, "parallelism": "dp{}".format(world)},
               "model_tflops_per_gpu": round(value * TRAIN_GFLOP_PER_IMAGE / 1e3 / world, 2),
               "roofline": {"bound": "mfma", "kernel": "conv3x3_patch_kernel<{},128,2,16> @ {}".format(
                                "bf16" if args.precision == "bf16" else "f32", ops.KernelTimer.layer),
                            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                            "kernel_ms": round(kms, 4), "launches_timed": len(ops.KernelTimer.events), "traffic": traffic,
                            "flop_per_launch": ops.KernelTimer.flops}}
        if world == 1 and not args.no_cpu_baseline:
            def cfg_fn(b):
                from oracle import configs as oc
                return oc.cub_config(n_parts=args.parts, batch_size=b)
            # torch-CPU scales badly past a few dozen threads on this graph (256 threads: 500 s per step): cap at 16
            out["cpu_baseline"] = cpu_baseline(cfg_fn, args.cpu_batch, min(os.cpu_count() or 1, args.cpu_threads))
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
