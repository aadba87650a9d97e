"""Data-parallel stream budget, measured on ONE GPU (round-4 verdict, item 3).

Under data parallelism the step's seven HIP streams are joined by the collectives' stream; the hardware runs four queues
(GPU_MAX_HW_QUEUES) and time-slices more than four ACTIVE ones (+30 % on the step).  This probe runs the
headline step (CUB 128x128, 10 parts, B = 64, bf16) in four arrangements, each in a fresh process so that the stream creation
order is each arrangement's own:

    full              seven streams, no collectives                  (the single-rank default: the reference point)
    full + stand-in   seven streams + the collective stand-in        (what a rank of a DP run did until round 5)
    compact + stand-in  three streams + the collective stand-in      (the DP default since round 5: <= 3 step queues + 1)
    compact           three streams, no collectives                  (what the compact plan costs by itself)

The stand-in (UPS_DP_STANDIN=1, dist.py): every bucket all-reduce is an out-of-place device copy of the bucket (132 MB per step
over the seven buckets) on a stream of its own, launched where the real all-reduce starts -- behind the segment's weight
gradients, on the weight-gradient stream's position -- and waited for where the real one is.  `dp_wait` = Trainer.dp_wait_ms():
what the launching stream waited at the end of the backward pass, for the side streams and per bucket.

    python tools/probes/stream_dp.py [steps]            (prints one line per arrangement; ~1 min)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(steps):
    import time
    import torch
    sys.path.insert(0, ROOT)
    import upsparts_amd  # noqa: F401
    from upsparts_amd import configs
    from upsparts_amd.model import TrainModel, Trainer
    dev = torch.device("cuda:0")
    cfg = configs.cub_config(n_parts=10, batch_size=64)
    cfg["precision"] = "bf16"
    model = TrainModel(cfg, device=dev, seed=0)
    tr = Trainer(cfg, None, model)
    g = torch.Generator().manual_seed(1234)
    batch = {k: (torch.rand(64, 128, 128, 3, generator=g) * 2 - 1).to(dev) for k in model.inputs}
    for _ in range(5):
        tr.train_step(batch)
    torch.cuda.synchronize()
    tr._dp_wait_events = []
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({"plan": tr.stream_plan, "ms_per_step": round(ms, 3), "img_s": round(64e3 / ms, 1), "dp_wait": tr.dp_wait_ms()}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]))
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    arrangements = (("full", "full", "0"), ("full + stand-in", "full", "1"), ("compact + stand-in", "compact", "1"),
                    ("compact", "compact", "0"))
    for rep in range(2):            # two passes: box drift shows as the spread between them
        for name, plan, standin in arrangements:
            env = dict(os.environ, UPS_STREAM_PLAN=plan, UPS_DP_STANDIN=standin)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(steps)], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print("{:20s} {}".format(name, line[-1] if line else "FAILED: " + out.stderr[-400:]))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
