"""Per-layer time of one training step: HIP events around every conv forward / dgrad / wgrad call, grouped by
(layer, pass, shape).  Usage (GPU box): python tools/layer_profile.py [--batch 64] [--steps 2]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa: E402,F401
from upsparts_amd import configs, ops  # noqa: E402
from upsparts_amd.model import TrainModel, Trainer  # noqa: E402

REC = []
ON = [False]


def wrap(name, fn, shape_of):
    def inner(*a, **k):
        if not ON[0]:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        REC.append((name, shape_of(*a, **k), e0, e1))
        return r
    return inner


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--top", type=int, default=70)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = configs.cub_config(n_parts=10, batch_size=args.batch)
    cfg["precision"] = "bf16"
    model = TrainModel(cfg, device=dev, seed=0)
    tr = Trainer(cfg, None, model)
    g = torch.Generator().manual_seed(1)
    batch = {k: (torch.rand(args.batch, 128, 128, 3, generator=g) * 2 - 1).to(dev) for k in ("view0", "view1", "view0_target")}

    def sh_f(x, layer, **k):
        return (layer.name, tuple(x.shape), layer.k, layer.stride, layer.cin_v, layer.co)

    def sh_b(g_, x, layer, **k):
        return (layer.name, tuple(x.shape), layer.k, layer.stride, layer.cin_v, layer.co)

    ops.conv_forward = wrap("fwd", ops.conv_forward, sh_f)
    ops.conv_dgrad = wrap("dgrad", ops.conv_dgrad, sh_b)
    ops.conv_wgrad = wrap("wgrad", ops.conv_wgrad, sh_b)
    for _ in range(2):
        tr.train_step(batch)
    torch.cuda.synchronize()
    ON[0] = True
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(args.steps):
        tr.train_step(batch)
    t1.record()
    torch.cuda.synchronize()
    step_ms = t0.elapsed_time(t1) / args.steps
    agg = collections.OrderedDict()
    for name, sh, e0, e1 in REC:
        key = (sh, name)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1; a[1] += e0.elapsed_time(e1)
    rows = []
    for (sh, name), (cnt, ms) in agg.items():
        lname, xs, k, st, cin_v, co = sh
        n, h, w, _ = xs
        ho, wo = -(-h // st), -(-w // st)
        fl = 2.0 * n * ho * wo * k * k * cin_v * co
        rows.append((ms / args.steps, cnt / args.steps, lname, name, xs, k, st, cin_v, co, fl))
    rows.sort(key=lambda r: -r[0])
    tot = sum(r[0] for r in rows)
    print("step {:.2f} ms (with events); conv calls total {:.2f} ms".format(step_ms, tot))
    for kind in ("fwd", "dgrad", "wgrad"):
        print("  {:6s} {:.2f} ms".format(kind, sum(r[0] for r in rows if r[3] == kind)))
    print("{:34s} {:6s} {:>5s} {:>20s} k s {:>4s} {:>5s} {:>8s} {:>8s}".format("layer", "pass", "calls", "x", "cin", "co", "ms/step", "TF/s"))
    for ms, cnt, lname, name, xs, k, st, cin_v, co, fl in rows[:args.top]:
        print("{:34s} {:6s} {:5.1f} {:>20s} {} {} {:4d} {:5d} {:8.3f} {:8.1f}".format(
            lname[-34:], name, cnt, "x".join(map(str, xs)), k, st, cin_v, co, ms, fl * cnt / ms / 1e9))
    # by network prefix
    nets = collections.defaultdict(float)
    for r in rows:
        nets[(r[2].split("/")[0], r[3])] += r[0]
    for k2, v in sorted(nets.items(), key=lambda kv: -kv[1]):
        print("  net {:24s} {:6s} {:8.2f} ms".format(k2[0], k2[1], v))


if __name__ == "__main__":
    main()
