"""Where does a resumed step differ from the uninterrupted one?  (diagnostic behind tests/test_gpu_model.py::
test_resumed_run_continues_the_noise_stream): per optimizer key the largest difference of parameters and Adam moments right after
the restore and after one more step."""
import copy, os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd.model import TrainModel, Trainer
from oracle import ref_model as R, configs
dev = torch.device("cuda:0")
cfg = copy.deepcopy(configs.tiny_config())
cfg.update(precision="bf16", vgg_widths=[8, 8, 16, 16, 16], use_tps=True)
cfg.setdefault("tps_parameters", {"scal": 0.8, "tps_scal": 0.15, "rot_scal": 0.2, "off_scal": 0.2, "scal_var": 0.1, "augm_scal": 1.0})
views = R.synthetic_views(cfg)
model = TrainModel(cfg, device=dev, seed=0)
tr = Trainer(cfg, None, model)
for _ in range(2):
    tr.train_step(views)
path = os.path.join(tempfile.mkdtemp(), "model.ckpt-2")
tr.save_checkpoint(path)
from upsparts_amd import ops as _ops
REC = {}
_orig_wgrad = _ops.conv_wgrad


def _rec_wgrad(g, x, layer, *a, **k):
    if True:
        torch.cuda.synchronize()
        REC.setdefault(TAG[0], []).append((layer.name, g.detach().clone(), x.detach().clone(), tuple(g.shape), g.dtype, g.data_ptr() % 4096))
    return _orig_wgrad(g, x, layer, *a, **k)


_ops.conv_wgrad = _rec_wgrad
FREC = {}
_orig_fwd = _ops.conv_forward


def _rec_fwd(x, layer, *a, **k):
    out = _orig_fwd(x, layer, *a, **k)
    torch.cuda.synchronize()
    res = k.get("res")
    FREC.setdefault(TAG[0], []).append((layer.name, x.detach().clone(), out.detach().clone(), None if res is None else res.detach().clone()))
    return out


_ops.conv_forward = _rec_fwd
TAG = ["warm"]
ORDER = sys.argv[1] if len(sys.argv) > 1 else "interleaved"       # "test": the uninterrupted run's third step first, as the test does
if ORDER.startswith("test"):
    TAG[0] = "a"
    a = tr.train_step(views)
    a = {k: float(v) for k, v in a.items()}
model2 = TrainModel(cfg, device=dev, seed=3)
tr2 = Trainer(cfg, None, model2)
tr2.initialize(path)
if ORDER == "test_sync":
    torch.cuda.synchronize()
if ORDER.startswith("test"):
    TAG[0] = "b"
    b = tr2.train_step(views)
    b = {k: float(v) for k, v in b.items()}
    print("forward calls:", len(FREC.get("a", [])), len(FREC.get("b", [])))
    shown = 0
    for i, (fa, fb) in enumerate(zip(FREC.get("a", []), FREC.get("b", []))):
        ex, eo = torch.equal(fa[1], fb[1]), torch.equal(fa[2], fb[2])
        er = fa[3] is None or torch.equal(fa[3], fb[3])
        if not (ex and eo and er) and shown < 12:
            shown += 1
            print("  fwd call {:3d} {:40s} x equal {} res equal {} out equal {} (x {} out {})".format(i, fa[0], ex, er, eo, tuple(fa[1].shape), tuple(fa[2].shape)))
    for ra, rb in zip(REC.get("a", []), REC.get("b", [])):
        if torch.equal(ra[1], rb[1]) and torch.equal(ra[2], rb[2]):
            continue
        print("wgrad call", ra[0], ra[3], ra[4], "ptr%4096", ra[5], rb[5], "g equal", torch.equal(ra[1], rb[1]), "x equal", torch.equal(ra[2], rb[2]),
              "max |dg|", float((ra[1].float() - rb[1].float()).abs().max()))
        if not torch.equal(ra[2], rb[2]):
            d = (ra[2].float() - rb[2].float()).abs()
            print("    x", tuple(ra[2].shape), "differing lanes (last dim):", sorted(set(torch.nonzero(d)[:, -1].tolist())), "count", int((d > 0).sum()),
                  "max", float(d.max()), "nan", int(torch.isnan(ra[2].float()).sum()), int(torch.isnan(rb[2].float()).sum()))
            nz = torch.nonzero(d)
            print("    first:", nz[:5].tolist(), [float(ra[2][tuple(i)]) for i in nz[:5]], [float(rb[2][tuple(i)]) for i in nz[:5]])
        if not torch.equal(ra[1], rb[1]):
            d = (ra[1].float() - rb[1].float()).abs()
            print("    differing lanes (last dim):", sorted(set(torch.nonzero(d)[:, -1].tolist())), "count", int((d > 0).sum()))
    print({k: (a[k], b[k]) for k in a if a[k] != b[k]})
    bad = [(n, float((p.detach() - model2.variables[n].detach()).abs().max())) for n, p in model.variables.items()
           if not torch.equal(p.detach(), model2.variables[n].detach())]
    print("no sync: differing variables:", len(bad), "of", len(model.variables), bad[:8])
    torch.cuda.synchronize()
    bad = [(n, float((p.detach() - model2.variables[n].detach()).abs().max())) for n, p in model.variables.items()
           if not torch.equal(p.detach(), model2.variables[n].detach())]
    print("synced:  differing variables:", len(bad), bad[:8])
    for key, grp in model.bank.groups.items():
        g2 = model2.bank.groups[key]
        for k in ("g", "m", "v"):
            d = (grp["flat"][k] - g2["flat"][k]).abs()
            if float(d.max()) > 0:
                idx = torch.nonzero(d).flatten()
                print("  {} {}: {} elements differ, first at {}, max {:.3e} (|.| max {:.3e})".format(
                    key, k, idx.numel(), idx[:6].tolist(), float(d.max()), float(grp["flat"][k].abs().max())))
    for n, g in model.bank.grads.items():
        d = (g - model2.bank.grads[n]).abs()
        if float(d.max()) > 0:
            print("   at (ky, kx, ci, co):", torch.nonzero(d)[:30].tolist())
            print("  grad", n, tuple(g.shape), "differs in", int((d > 0).sum()), "elements, max", float(d.max()), "|g| max", float(g.abs().max()))
    sys.exit(0)


def report(tag):
    torch.cuda.synchronize()
    print("----", tag)
    for key, grp in model.bank.groups.items():
        g2 = model2.bank.groups[key]
        d = {k: float((grp["flat"][k].float() - g2["flat"][k].float()).abs().max()) for k in grp["flat"] if torch.is_tensor(grp["flat"][k])}
        print("  {:22s} t {} / {}  {}".format(key, grp["t"], g2["t"], {k: "{:.3e}".format(v) for k, v in d.items()}))
    bad = [n for n, p in model.variables.items() if not torch.equal(p.detach(), model2.variables[n].detach())]
    print("  differing variables:", len(bad), bad[:12])
    print("  state:", {k: (float(v), float(tr2.state[k])) for k, v in tr.state.items() if float(v) != float(tr2.state[k])})


report("after restore")
a = tr.train_step(views); b = tr2.train_step(views)
print({k: (float(a[k]), float(b[k])) for k in a if float(a[k]) != float(b[k])})
report("after one more step")
a = tr.train_step(views); b = tr2.train_step(views)
report("after two more steps")
