"""Where the one-tile row-stream forward differs from run to run (hazard hunt, tools/probes/rows_hazard.sh): the 64-channel / 64-column
case of test_conv_rows_kernel repeated N times; every run is compared with the patch kernel's result (the truth up to rounding) and
the positions of the elements that are far off are summarised.   usage: UPS_LIB=ab/<name>/libupsparts_hip.so python3 rows_repro.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import upsparts_amd  # noqa
from upsparts_amd import lib, ops
dev = torch.device("cuda:0")
n, h, w, c = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (3, 64, 64, 64)))
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
g = torch.Generator().manual_seed(701)
V = torch.randn(3, 3, c, c, generator=g) / math.sqrt(c * 9)
b = torch.randn(c, generator=g) * 0.1
lay = ops.ConvLayer("t/conv2d_0", V.to(dev), b.to(dev), 3, 1, False, "leaky_relu")
lay.in_post, lay.out_act = True, lib.ACT_LRELU
xd = torch.nn.functional.leaky_relu(torch.randn(n, h, w, c, generator=g), 0.2).to(torch.bfloat16).to(dev)
os.environ["UPS_ROWS_KERNEL"] = "0"
y0 = ops.conv_forward(xd, lay, res=xd, res_post=True).float()
os.environ["UPS_ROWS_KERNEL"] = "force"
bad_runs = 0
for r in range(reps):
    y = ops.conv_forward(xd, lay, res=xd, res_post=True).float()
    torch.cuda.synchronize()
    d = (y - y0).abs()
    bad = d > 0.05 * (y0.abs() + 0.1)
    if os.environ.get("UPS_REPRO_CH"):      # observation builds overwrite a neighbouring channel on purpose: look at this one only
        keep = torch.zeros_like(bad)
        keep[..., int(os.environ["UPS_REPRO_CH"])] = True
        bad &= keep
    nb = int(bad.sum())
    if nb:
        bad_runs += 1
        idx = bad.nonzero()
        print("run %2d: %6d elements off  images %s  rows %s  cols %s  channels %s" % (
            r, nb, sorted(set(idx[:, 0].tolist())), sorted(set(idx[:, 1].tolist()))[:24], sorted(set(idx[:, 2].tolist()))[:24],
            sorted(set(idx[:, 3].tolist()))[:16]))
        # what do the wrong values look like: equal to a neighbouring row's / an earlier image's result?
        i0 = idx[0].tolist()
        print("        first: (img %d, y %d, x %d, c %d) got %.4f want %.4f" % (*i0, float(y[tuple(i0)]), float(y0[tuple(i0)])))
        # accumulator view (round 6): undo the stored activation and take the residual off -- what the MFMA chain held
        inv = lambda t: torch.where(t > 0, t, t / 0.2)
        res = inv(xd.float())
        for j in range(min(3, idx.shape[0])):
            k = tuple(idx[j].tolist())
            print("        acc view %s: got %.4f want %.4f (residual %.4f, bias %.4f)" % (
                k, float(inv(y)[k] - res[k]), float(inv(y0)[k] - res[k]), float(res[k]), float(b[k[3]])))
            k1 = k[:3] + (k[3] + 1,)        # the neighbouring channel: where the observation builds (ab/d_*) put a copy of a register
            print("          channel +1: pre-activation got %.4f want %.4f (its residual %.4f; want - residual = %.4f)" % (
                float(inv(y)[k1]), float(inv(y0)[k1]), float(res[k1]), float(inv(y0)[k1] - res[k1])))
print("%d of %d runs differ from the patch kernel" % (bad_runs, reps))
