"""Achieved HBM bandwidth of the HBM-bound kernel families, timed in isolation with HIP events: algorithmic bytes (SURVEY 8a rows
5-9, 12, 16: every tensor read once + written once) / mean launch time, as a fraction of the 8 TB/s HBM3E peak AND of the ~6.3 TB/s
the chip delivers from HBM (MI355X_MICROARCH.md).

Round 6 (round-5 verdict, weak 6): every row ROTATES at least three independent operand sets whose sum exceeds the 256 MiB Infinity
Cache by a wide margin (--sets-mib, default 800) -- until round 5 each launch re-ran on the same buffers, 42-252 MB a row, and the table
measured the cache as much as HBM (mask_parts_fwd "7.05 TB/s").  --same-buffers restores the old protocol for comparison.
    python tools/hbm_roofline.py [--shape B,S,P] [--json out.json] [--same-buffers]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa: E402,F401
from upsparts_amd import lib as L, ops  # noqa: E402

PEAK = 8000.0   # GB/s
ACHIEVABLE = 6300.0     # GB/s a streaming kernel gets from HBM on this chip (guide)
dev = torch.device("cuda:0")
B, S, P, A = 64, 128, 10, 64
T = torch.bfloat16


def timeit(fns, iters=20):
    """fns: independent closures over DIFFERENT operand sets, run round-robin."""
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = max(iters, 2 * len(fns))
    e0.record()
    for i in range(n):
        fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def prior_desc(view, n, l, lm, m, hard, px, per_np, sums, g_hard=None, dl=None, dl_rec=None):
    d = L.PriorDesc()
    d.n, d.h, d.w, d.P, d.view, d.entropy_ce, d.gamma = n, S, S, P, view, 0, 10.0
    d.half_h = d.half_w = S // 8
    d.ms_alpha, d.ms_lambda = 1.0, 1e-2
    d.w_kl, d.w_entropy, d.w_ms, d.w_area, d.w_patch, d.w_gmrf, d.w_var = 1.0, 1.0, 1e-5, 1e-12, 1e-4, 1e-3, 1.0
    g = lambda t: t.data_ptr() if t is not None else None
    d.l, d.l_mean, d.m, d.hard, d.px, d.per_np, d.sums = g(l), g(lm), g(m), g(hard), g(px), g(per_np), g(sums)
    d.g_hard, d.dl, d.dl_rec = g(g_hard), g(dl), g(dl_rec)
    return d


def build(seed):

    """One complete, independent operand set: [(row name, algorithmic bytes, closure, note)]."""
    g = torch.Generator(device=dev).manual_seed(seed)
    rows = []

    def add(name, nbytes, fn, note=""):
        rows.append((name, nbytes, fn, note))

    n2 = 2 * B
    CI = (A + P + 7) // 8 * 8            # channels of the image decoder's input (appearance + one-hot masks, padded to 8)
    mean = torch.randn(n2, S, S, P, device=dev, generator=g)
    eps = torch.randn(n2, S, S, P, device=dev, generator=g)
    map_b = mean.numel() * 4
    add("part_softmax (l, m, hard) + hard-mask moments", 5 * map_b, lambda: ops.part_softmax(mean, eps, moments_gamma=10.0),
        "2 reads + 3 writes of a [2B,S,S,P] fp32 map; the spatial soft-max moments of the hard map come out of the same pass "
        "(round 2: a separate 0.068 ms pass over the hard map at 15 % of the roof)")
    l, m, hard, _ = ops.part_softmax(mean, eps)
    px = ops.moments_to_px(ops.spatial_moments(hard, 10.0), S)
    l0, m0, h0, lm0, px0 = l[:B].contiguous(), m[:B].contiguous(), hard[:B].contiguous(), mean[:B].contiguous(), px[:B].contiguous()
    nfl = L.load().ups_prior_sums_floats(B, P)
    sums = torch.empty(nfl, dtype=torch.float32, device=dev)
    per_np = torch.empty((B, P, 8), dtype=torch.float32, device=dev)
    half_b = map_b // 2
    d_f = prior_desc(0, B, l0, lm0, m0, h0, px0, per_np, sums)
    add("prior_fwd (view 0)", 4 * half_b, lambda: L.call("ups_prior_fwd", C.byref(d_f), L.stream()), "l, l_mean, m, hard read once")
    gh = torch.randn(B, S, S, P, device=dev, generator=g)
    dl, dlr = torch.empty_like(l0), torch.empty_like(l0)
    d_b = prior_desc(0, B, l0, lm0, m0, h0, px0, per_np, sums, gh, dl, dlr)
    add("prior_bwd (view 0, dl_tot + dl_rec)", 7 * half_b, lambda: L.call("ups_prior_bwd", C.byref(d_b), L.stream()),
        "5 reads + 2 writes (neighbour taps served by cache)")
    x = torch.randn(n2, 64, 64, 256, device=dev, generator=g).to(T)
    up_b = x.numel() * 2 * 5
    add("bilinear2x_fwd 64->128, 256 ch", up_b, lambda: ops.BilinearFn.apply(x), "read h*w, write 4*h*w")
    gy = torch.randn(n2, 128, 128, 256, device=dev, generator=g).to(T)
    gx = torch.empty_like(x)
    add("bilinear2x_bwd 128->64, 256 ch", up_b,
        lambda: L.call("ups_bilinear2x_bwd", L.ptr(gy), L.ptr(gx), L.dt(gy), n2, 64, 64, 256, L.stream()))
    npar = 33_100_000
    p_, g_, m_, v_ = (torch.randn(npar, device=dev, generator=g) for _ in range(4))
    v_.abs_()
    add("adam (33.1 M parameters)", npar * 28, lambda: ops.adam_step(p_, g_, m_, v_, 1e-4, 0.5, 0.9, 1e-8), "p, g, m, v read; p, m, v written")
    view = torch.rand(B, S, S, 3, device=dev, generator=g)
    add("mask_parts_fwd", view.numel() * 4 + half_b + P * B * S * S * 8 * 2, lambda: ops.MaskPartsFn.apply(view, h0, T),
        "view + hard read, [P*B,S,S,8] bf16 written")
    feat = torch.randn(B, P, A, device=dev, generator=g)
    add("unpool_fwd", half_b + B * S * S * CI * 2, lambda: ops.UnpoolFn.apply(h0, feat, T), "hard read, [B,S,S,A+P] bf16 written")
    gparts = torch.randn(P * B, S, S, 8, device=dev, generator=g).to(T)
    ghp = torch.empty(B, S, S, P, device=dev)
    add("mask_parts_bwd", gparts.numel() * 2 + view.numel() * 4 + half_b,
        lambda: L.call("ups_mask_parts_bwd", L.ptr(view), L.ptr(gparts), L.ptr(ghp), L.dt(gparts), B, S * S, P, L.stream()),
        "[P*B,S,S,8] bf16 gradient + view read, g_hard written")
    ginj = torch.randn(B, S, S, CI, device=dev, generator=g).to(T)
    gfe = torch.empty(L.load().ups_unpool_bwd_floats(B, P, A), dtype=torch.float32, device=dev)
    add("unpool_bwd (g_hard + g_feat)", ginj.numel() * 2 + 2 * half_b,
        lambda: L.call("ups_unpool_bwd", L.ptr(h0), L.ptr(feat), L.ptr(ginj), L.ptr(ghp), L.ptr(gfe), L.dt(ginj), B, S * S, P, A, CI,
                       L.stream()),
        "one pass since round 4: [B,S,S,80] bf16 gradient read ONCE, hard read, g_hard written (until round 5 this row still counted "
        "the gradient twice, the two-kernel form of round 3: its fractions were overstated by 1.67x)")
    m1 = m[B:].contiguous()
    px1 = px[B:].contiguous()
    sums1 = torch.empty(nfl, dtype=torch.float32, device=dev)
    add("spatial_moments + KL (view 1, masked soft map)", half_b, lambda: ops.spatial_moments(m1, 10.0, rect_px=px1, half=S // 8, kl_sums=sums1),
        "1 read of the map: the variance moments and view 1's categorical KL from one pass (round 4: the separate prior_fwd launch "
        "of view 1, another read of the same map, is gone)")
    st1 = ops.spatial_moments(m1, 10.0, rect_px=px1, half=S // 8)
    d_b1 = prior_desc(1, B, l[B:].contiguous(), None, m1, None, px1, st1, sums1, gh, dl, dlr)
    add("prior_bwd (view 1, dl_tot + dl_rec)", 4 * half_b, lambda: L.call("ups_prior_bwd", C.byref(d_b1), L.stream()),
        "m, g_hard read; 2 writes")
    fa = torch.randn(B, S, S, 64, device=dev, generator=g).to(T)
    fb = torch.randn(B, S, S, 64, device=dev, generator=g).to(T)
    add("l1_fwd (VGG block1 features)", 2 * fa.numel() * 2, lambda: ops.L1MeanFn.apply(fa, fb, 64, L.ACT_RELU), "two reads")
    add("maxpool2_fwd 128->64, 64 ch", int(fa.numel() * 2 * 1.25), lambda: ops.MaxPoolFn.apply(fa), "read + quarter-size write")

    return rows


def main():
    global B, S, P
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--shape", default="64,128,10", help="B,S,P of the part-path maps (configs: 64,128,10 | 32,256,16 | 16,256,20 | 64,128,25)")
    ap.add_argument("--sets-mib", type=int, default=800, help="rotate operand sets until their rows' bytes sum to at least this (>> 256 MiB)")
    ap.add_argument("--same-buffers", action="store_true", help="the protocol of rounds 1-5: one operand set, re-used by every launch")
    ap.add_argument("--only", default="", help="comma-separated substrings of row names")
    args = ap.parse_args()
    B, S, P = (int(v) for v in args.shape.split(","))
    sets = [build(0)]
    smallest = min(nb for _, nb, _, _ in sets[0])
    nsets = 1 if args.same_buffers else max(3, -(-args.sets_mib * (1 << 20) // smallest))
    nsets = min(nsets, 12)
    for k in range(1, nsets):
        sets.append(build(k))
    rows = []
    for i, (name, nbytes, _, note) in enumerate(sets[0]):
        if args.only and not any(t in name for t in args.only.split(",")):
            continue
        ms = timeit([st[i][2] for st in sets])
        gbs = nbytes / ms / 1e6
        rows.append({"kernel": name, "algorithmic_MB": round(nbytes / 1e6, 1), "ms": round(ms, 4), "GBps": round(gbs, 1),
                     "frac_of_8TBps": round(gbs / PEAK, 3), "frac_of_6.3TBps": round(gbs / ACHIEVABLE, 3), "note": note})
    print("shape B={} S={} P={}; {} operand set(s) rotated per row ({})".format(
        B, S, P, nsets, "same buffers every launch: the Infinity Cache serves part of these bytes" if nsets == 1 else
        "every launch reads operands last touched >= {} MB of traffic ago".format(int((nsets - 1) * smallest / 1e6))))
    print("{:52s} {:>10s} {:>9s} {:>9s} {:>7s} {:>9s}".format("kernel", "alg. MB", "ms", "GB/s", "of 8T", "of 6.3T"))
    for r in rows:
        print("{:52s} {:10.1f} {:9.4f} {:9.1f} {:7.3f} {:9.3f}".format(r["kernel"][:52], r["algorithmic_MB"], r["ms"], r["GBps"], r["frac_of_8TBps"], r["frac_of_6.3TBps"]))
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"peak_GBps": PEAK, "achievable_GBps": ACHIEVABLE, "operand_sets": nsets,
                       "shapes": "B={}, {}x{}, P={}, bf16 activations".format(B, S, S, P), "kernels": rows}, f, indent=1)


if __name__ == "__main__":
    main()
