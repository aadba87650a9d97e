"""Where a block of the patch kernel spends its life (debug build with -DUPS_PHASE_TIMING, see tools/probes/phase_timing.sh):
thread 0 of every block records the 100 MHz wall clock at phase boundaries; this script runs one launch of a bench_conv case and
prints the mean duration of each phase and the number of blocks resident at once.
Usage (GPU box): UPS_LIB=tools/probes/lib_phase.so python tools/probes/phase_timing.py <case> [fwd|dgrad]"""
import ctypes, math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd import ops, lib
from bench_conv import CASES
dev = torch.device("cuda:0")
case = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"
name, n, h, cin, cout, k, stride, coords, act = [c for c in CASES if c[0] == case][0]
if case == "dv_out" and mode == "dgrad" and "noact" not in sys.argv[3:]:
    act = "leaky_relu"      # as the model runs it: the logit convolution reads the stored post-activation tensor, its input gradient takes act' from sign bytes
g = torch.Generator().manual_seed(1)
cin_v = cin + (2 if coords else 0)
V = (torch.randn(k, k, cin_v, cout, generator=g) / math.sqrt(k * k * cin_v)).to(dev)
b = torch.randn(cout, generator=g).to(dev)
lay = ops.ConvLayer("x/conv2d_0", V, b, k, stride, coords, act)
f16 = case.startswith("dv_")
fmt = lib.F16 if f16 else None
lay.f16 = f16
if act == "leaky_relu":
    lay.in_post, lay.out_act = True, lib.ACT_LRELU
x = torch.randn(n, h, h, ops.round8(cin), device=dev).to(torch.bfloat16)
if f16:
    x = x.to(torch.float16).view(torch.bfloat16)
res_self = act is not None and stride == 1 and cin == cout
out_f32 = case == "dv_out"
fwd = lambda: ops.conv_forward(x, lay, res=x if res_self else None, fmt=fmt, res_post=lay.in_post, out_f32=out_f32)
y = fwd()
gy = torch.zeros(y.shape[:-1] + (ops.round8(cout),), device=dev)
gy[..., :cout] = torch.randn(y.shape[:-1] + (cout,), device=dev)
gy = gy.to(torch.bfloat16)
xb = None
if "bits" in sys.argv[3:]:       # act' from the producer's sign bytes, as the step runs it (res_patch == 2 on the wide instances)
    pos = (x.view(torch.int16) > 0).view(*x.shape[:-1], -1, 8).to(torch.uint8)
    xb = (pos * (2 ** torch.arange(8, device=dev, dtype=torch.uint8))).sum(-1).to(torch.uint8).contiguous()
run = fwd if mode == "fwd" else (lambda: ops.conv_dgrad(gy, x, lay, res=gy if res_self else None, x_bits=xb))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
L = ctypes.CDLL(os.environ["UPS_LIB"])
nb = 65536
buf = np.zeros(nb * 8, dtype=np.uint64)
rc = L.ups_phase_dump(buf.ctypes.data_as(ctypes.c_void_p), nb)
t = buf.reshape(nb, 8).astype(np.int64)
live = t[:, 0] > 0
t = t[live]
t0 = t[:, 0].min()
end = t[:, 6]
print("%s %s: launch %.3f ms, %d blocks recorded, span %.3f ms" % (case, mode, e0.elapsed_time(e1), len(t), (end.max() - t0) / 1e5))
names = ["setup", "loads landed", "main loop", "epi: res/dact tile", "epi: acc -> LDS", "epi: store"]
for i in range(6):
    d = (t[:, i + 1] - t[:, i]) / 100.0
    print("  %-20s mean %6.2f us   p10 %6.2f  p90 %6.2f" % (names[i], d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
life = (t[:, 6] - t[:, 0]) / 100.0
print("  block life mean %.2f us; resident blocks (sum of lives / span) %.1f = %.2f per CU" % (life.mean(), life.sum() / ((end.max() - t0) / 100.0), life.sum() / ((end.max() - t0) / 100.0) / 256))

# per-slot timelines: slot = (XCC, SE, CU, LDS base); gap = next block's first instruction - this block's last timestamp
hw = buf.reshape(nb, 8)[live][:, 7]
cu = (hw >> np.uint64(8)) & np.uint64(0xf); se = (hw >> np.uint64(13)) & np.uint64(0x7); xcc = (hw >> np.uint64(32)) & np.uint64(0xf)
lb = (hw >> np.uint64(40)) & np.uint64(0xff)
slot = (xcc.astype(np.int64) << 24) | (se.astype(np.int64) << 16) | (cu.astype(np.int64) << 8) | lb.astype(np.int64)
cuid = slot >> 8
gaps, per_slot = [], []
for sid in np.unique(slot):
    m = slot == sid
    st, en = np.sort(t[m, 0]), np.sort(t[m, 6])
    per_slot.append(m.sum())
    if m.sum() > 1:
        gaps.extend(((st[1:] - en[:-1]) / 100.0).tolist())
gaps = np.array(gaps)
print("  slots seen %d (CUs %d), blocks per slot %.1f..%.1f; gap between consecutive blocks of a slot: mean %.2f us, p10 %.2f, p50 %.2f, p90 %.2f"
      % (len(per_slot), len(np.unique(cuid)), min(per_slot), max(per_slot), gaps.mean(), np.percentile(gaps, 10), np.percentile(gaps, 50), np.percentile(gaps, 90)))
ends = np.array([t[slot == sid, 6].max() for sid in np.unique(slot)])
print("  last block of a slot ends %.1f .. %.1f us before the launch's end (mean %.1f)" % ((end.max() - ends.max()) / 100.0, (end.max() - ends.min()) / 100.0, (end.max() - ends).mean() / 100.0))
xs = xcc.astype(np.int64)
print("  per XCD: blocks, first start, last end (us from launch start), mean block life, mean main loop")
for k in np.unique(xs):
    m = xs == k
    print("    xcd %d: %5d blocks  start %7.1f  end %8.1f  life %6.2f  main %6.2f" % (k, m.sum(), (t[m, 0].min() - t0) / 100.0, (t[m, 6].max() - t0) / 100.0,
                                                                                 life[m].mean(), ((t[m, 3] - t[m, 2]) / 100.0).mean()))
