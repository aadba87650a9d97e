#!/bin/bash
# HBM traffic of ONE launch shape from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE
# reports half the bytes of wide coalesced reads on gfx950 -> x2; WRITE_SIZE exact).
# usage (GPU box): bash tools/pmc_traffic.sh <out.json> <kernel-name-substring> <algorithmic-bytes> -- <args of tools/one_conv.py>
#   e.g. bash tools/pmc_traffic.sh gpurun_out/w8.json conv_wgrad3x3_f8 1612709888 -- wgrad dv_rb128 f8
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; KEY=$2; ALG=$3; shift 4
cd $R
T=$(mktemp -d /tmp/pmct.XXXXXX)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $T/$c -- python3 tools/one_conv.py "$@" > /dev/null 2>&1
done
python3 - $T "$KEY" "$ALG" "$OUT" "$*" <<'PY'
import csv, glob, json, sys
T, key, alg, out, args = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(T + "/%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"] and r["Counter_Name"] == c:
                v.append(float(r["Counter_Value"]))
    vals[c] = v
rd = 2 * 1024 * sum(vals["FETCH_SIZE"]) / max(1, len(vals["FETCH_SIZE"]))
wr = 1024 * sum(vals["WRITE_SIZE"]) / max(1, len(vals["WRITE_SIZE"]))
json.dump({"kernel": key, "command": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python3 tools/one_conv.py " + args,
           "launches_seen": [len(vals["FETCH_SIZE"]), len(vals["WRITE_SIZE"])],
           "tree": __import__("os").environ.get("UPS_TREE", "unrecorded"),
           "correction": "gfx950: FETCH_SIZE x2 (it reports half the bytes of wide coalesced reads), WRITE_SIZE exact",
           "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
           "algorithmic_bytes_per_launch_tensor_once": alg, "traffic_over_algorithmic": (rd + wr) / alg if alg else None},
          open(out, "w"), indent=1)
print(open(out).read())
PY
rm -rf $T
