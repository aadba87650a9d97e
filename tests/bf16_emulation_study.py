"""CPU study (test infrastructure, runs in the build container): which bf16 STORES of encoder_0 -> decoder_visualize carry
the logit error that moves the part-mask boundaries of the confident fixture?  The oracle graph is run in fp32 with the HIP
path's roundings emulated (operands of every convolution rounded to bf16, fp32 accumulation, fp32 epilogue, outputs rounded
on store) under a policy that names the tensors kept at higher precision.

    python tests/bf16_emulation_study.py [policy ...]      policies: see POLICIES
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import configs, ref_model as R  # noqa: E402
import make_golden_full as G  # noqa: E402


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


class EmuScope(R.Scope):
    """R.Scope with the storage roundings of the HIP bf16 path.  policy: dict
       store_res  : precision of a residual block's output / of the bilinear output ("bf16" | "hilo" | "fp32")
       store_plain: precision of a plain conv output that feeds the trunk
       operand    : "bf16" (MFMA operand rounding) | "fp32"
       min_hw     : the policy's high precision applies to tensors with h >= min_hw (below: always high precision if lowfp32)
    """

    def __init__(self, *a, policy=None, **kw):
        super().__init__(*a, **kw)
        self.policy = policy or {}

    def _store(self, y, kind):
        mode = self.policy.get(kind, "bf16")
        lim = self.policy.get("hi_below", 0)
        if y.shape[1] < lim:
            mode = "fp32"
        if mode == "bf16":
            return bf(y)
        if mode == "hilo":      # hi = bf16(y), lo = bf16(y - hi): 16 mantissa bits
            hi = bf(y)
            return hi + bf(y - hi)
        return y

    def conv2d(self, x, num_filters, k=3, stride=1, _kind="store_plain", _res=None):
        name = "{}/conv2d_{}".format(self.prefix, self.counter)
        self.counter += 1
        V = self.params[name + "/V"]
        b = self.params[name + "/b"]
        cin = x.shape[-1]
        xo = bf(x) if self.policy.get("operand", "bf16") == "bf16" and not x.shape[1] < self.policy.get("hi_below", 0) else x
        Vm = bf(V[:, :, :cin]) if self.policy.get("operand", "bf16") == "bf16" and not x.shape[1] < self.policy.get("hi_below", 0) else V[:, :, :cin]
        y = R.conv2d_same(xo, Vm, b, stride)
        if self.coords:
            zc = self.add_coordinates(torch.zeros_like(x[..., :0]))
            y = y + R.conv2d_same(zc, V[:, :, cin:], torch.zeros_like(b), stride)
        if _res is not None:
            y = y + _res
        return self._store(y, _kind)

    def nin(self, x, n):
        return self.conv2d(x, n, k=1)

    def residual_block(self, x, skipin=None, conv=None):
        assert skipin is None
        c = x.shape[-1]
        return self.conv2d(self.activate(x), c, _kind="store_res", _res=x)

    def upsample_linear(self, x):
        return self._store(R.bilinear_up2(x), "store_res")


class EmuNets(R.Nets):
    def __init__(self, config, params, pol_e, pol_d):
        super().__init__(config, params, None)
        self.pol_e, self.pol_d = pol_e, pol_d

    def _scope(self, name, kw):
        pol = self.pol_e if name == "encoder_0" else self.pol_d
        return EmuScope(self.params, name, kw.get("activation", "relu"), kw.get("coords", False), None, policy=pol)


BF = {"store_res": "bf16", "store_plain": "bf16", "operand": "bf16"}
POLICIES = {
    "fp32": ({"store_res": "fp32", "store_plain": "fp32", "operand": "fp32"},) * 2,
    "bf16": (BF, BF),
    "operand_only": ({"store_res": "fp32", "store_plain": "fp32", "operand": "bf16"},) * 2,
    "dv_res_hilo": (BF, dict(BF, store_res="hilo")),
    "dv_all_hilo": (BF, dict(BF, store_res="hilo", store_plain="hilo")),
    "enc_fp32_dv_bf16": ({"store_res": "fp32", "store_plain": "fp32", "operand": "fp32"}, BF),
    "enc_bf16_dv_fp32": (BF, {"store_res": "fp32", "store_plain": "fp32", "operand": "fp32"}),
    "both_res_hilo": (dict(BF, store_res="hilo"), dict(BF, store_res="hilo")),
    "both_all_hilo": (dict(BF, store_res="hilo", store_plain="hilo"), dict(BF, store_res="hilo", store_plain="hilo")),
    "dv_lowfp32_16": (BF, dict(BF, hi_below=16)),
    "dv_lowfp32_32": (BF, dict(BF, hi_below=32)),
    "dv_lowfp32_64": (BF, dict(BF, hi_below=64)),
    "enc_hilo_dv_lowfp32_32_hilo": (dict(BF, store_res="hilo", store_plain="hilo"), dict(BF, hi_below=32, store_res="hilo")),
}


def iou(a, gold, P):
    out = []
    for b in range(a.shape[0]):
        for p in range(P):
            inter = np.logical_and(a[b] == p, gold[b] == p).sum()
            union = np.logical_or(a[b] == p, gold[b] == p).sum()
            if union:
                out.append(inter / union)
    return float(np.mean(out))


def run(policy_name, cfg, params, views, noise, z, ref=None):
    pol_e, pol_d = POLICIES[policy_name]
    nets = EmuNets(cfg, params, pol_e, pol_d)
    Z, P = cfg.get("z0_size", 256), cfg["n_parts"]
    with torch.no_grad():
        v0 = views["view0"].float()
        x = v0 if pol_e.get("operand") == "fp32" else bf(v0)
        pe = nets.e_pi(x)
        d0 = R.FullLatent(pe, Z)
        pi0 = d0.sample(noise["eps_pi0"][0].float())
        zin = pi0 if pol_d.get("operand") == "fp32" else bf(pi0)
        lm = nets.dv(zin)
        l0 = lm + noise["eps_l0"].float()
    a = lm.argmax(-1).numpy()
    s = l0.argmax(-1).numpy()
    res = {"iou_mean": iou(a, z["out_parts_hard"], P), "agree": float((a == z["out_parts_hard"]).mean()),
           "iou_sampled": iou(s, z["hard0_argmax"], P)}
    if ref is not None:
        res["logit_relerr_max"] = float((lm - ref).abs().max() / ref.abs().max())
        res["logit_relerr_rms"] = float((lm - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        res["pe_rel_rms"] = float((pe - ref_pe[0]).pow(2).mean().sqrt() / ref_pe[0].pow(2).mean().sqrt())
    return res, lm, pe


ref_pe = [None]

if __name__ == "__main__":
    torch.set_num_threads(8)
    cfg = configs.cub_config(n_parts=10, batch_size=2)
    params = G.confident_params(cfg)
    views, noise = R.synthetic_views(cfg), R.synthetic_noise(cfg)
    z = np.load(os.path.join(ROOT, "tests", "golden", "full_cub128_confident.npz"))
    names = sys.argv[1:] or list(POLICIES)
    r, ref, pe = run("fp32", cfg, params, views, noise, z)
    ref_pe[0] = pe
    print("fp32", r, flush=True)
    for n in names:
        if n == "fp32":
            continue
        r, _, _ = run(n, cfg, params, views, noise, z, ref)
        print(n, {k: round(v, 5) for k, v in r.items()}, flush=True)
