"""part_softmax at the benchmark shape: with / without the hard-mask moments, random and spatially coherent logits, both kernel forms."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import upsparts_amd  # noqa
from upsparts_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
n2, S, P = 128, 128, 10
eps = torch.randn(n2, S, S, P, device=dev, generator=g)
low = torch.randn(n2, P, S // 16, S // 16, device=dev, generator=g)
means = {"random": torch.randn(n2, S, S, P, device=dev, generator=g),
         "coherent": (6.0 * torch.nn.functional.interpolate(low, size=(S, S), mode="bilinear", align_corners=True)).permute(0, 2, 3, 1).contiguous()}


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for px in ("1", "0"):
    os.environ["UPS_SOFTMAX_PX"] = px
    for name, mean in means.items():
        t0 = timeit(lambda: ops.part_softmax(mean, eps))
        t1 = timeit(lambda: ops.part_softmax(mean, eps, want_bits=True, moments_gamma=10.0))
        print("UPS_SOFTMAX_PX={} {:9s}: plain {:.1f} us, + bits + hard-mask moments {:.1f} us".format(px, name, t0, t1))
