#!/bin/bash
# usage (GPU box): bash tools/profile_round.sh <round-tag>          e.g.  bash tools/profile_round.sh round2
#   1. rocprofv3 --kernel-trace --stats of the default bench.py run (three HIP streams) -> <tag>_bench_b64_kernel_stats.csv,
#      the bench line, and the per-(kernel, grid) durations of the patch kernel from the same trace
#   2. the same with UPS_NO_OVERLAP=1 (everything on one stream: per-kernel durations without CU sharing)
#   3. two --pmc passes (FETCH_SIZE, WRITE_SIZE) on the hot conv launch -> <tag>_pmc_dv_rb128.json (HBM bytes per launch)
#   4. SQ counter passes on the same layer, forward / input gradient / weight gradient -> <tag>_sq_dv_rb128.txt (MFMA-pipe busy
#      cycles vs CU-busy cycles, instruction counts, LDS conflicts); tools/bench_conv.py tables (bf16 and the fp8 forward)
#   5. isolated timings of the HBM-bound kernel families -> <tag>_hbm_kernels.json
# Everything lands in gpurun_out/profiles_<tag>/ on the box (merged back); copy what is to be judged into profiles/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-round3}
cd $R
O=$R/gpurun_out/profiles_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline > $O/bench.log 2>&1
grep '"metric"' $O/bench.log > $O/${TAG}_bench_b64.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_b64_kernel_stats.csv
# the roofline launch is the 8388608-thread grid of conv3x3_patch_kernel<bf16,128,2,16> (one forward + two input-gradient
# launches of decoder_visualize/conv2d_8 per step)
python3 tools/trace_summary.py $(find $O/stats -name "*kernel_trace.csv" | head -1) conv3x3_patch > $O/${TAG}_patch_kernel_by_grid.txt
python3 tools/trace_summary.py $(find $O/stats -name "*kernel_trace.csv" | head -1) bilinear > $O/${TAG}_bilinear_by_grid.txt
rm -rf $O/stats
UPS_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 bench.py --no-cpu-baseline > $O/bench_nooverlap.log 2>&1
grep '"metric"' $O/bench_nooverlap.log > $O/${TAG}_bench_b64_nooverlap.json
cp $(find $O/stats1 -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_b64_nooverlap_kernel_stats.csv
python3 tools/trace_summary.py $(find $O/stats1 -name "*kernel_trace.csv" | head -1) bilinear > $O/${TAG}_bilinear_by_grid_nooverlap.txt
python3 tools/trace_summary.py $(find $O/stats1 -name "*kernel_trace.csv" | head -1) conv3x3_patch > $O/${TAG}_patch_kernel_by_grid_nooverlap.txt
python3 tools/by_grid.py $(find $O/stats1 -name "*kernel_trace.csv" | head -1) 10 0.1 > $O/${TAG}_by_kernel_and_grid_nooverlap.txt
rm -rf $O/stats1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 tools/one_conv.py fwd > /dev/null 2>&1
done
python3 - $O $TAG <<'PY'
import csv, glob, json, sys
O, tag = sys.argv[1], sys.argv[2]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(O + "/pmc_%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if "conv3x3_patch" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v.append(float(r["Counter_Value"]))
    vals[c] = v
rd = 2 * 1024 * sum(vals["FETCH_SIZE"]) / max(1, len(vals["FETCH_SIZE"]))
wr = 1024 * sum(vals["WRITE_SIZE"]) / max(1, len(vals["WRITE_SIZE"]))
alg = {"input (also the residual: read once)": 1073741824, "weights": 1179648, "output": 1073741824}
json.dump({"kernel": "conv3x3_patch_kernel<f16,128,2,16,0,false,1,true>",
           "launch": "decoder_visualize res-block conv as the model issues it: n=128 images, 128x128, 256(+2 CoordConv)->256, fp16 forward tensors in post-activation storage, LDS-DMA patch, residual taken from the resident patch (tools/one_conv.py fwd)",
           "command": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --output-format csv -- python3 tools/one_conv.py fwd",
           "FETCH_SIZE_kb_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_kb_per_launch": vals["WRITE_SIZE"],
           "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE exact",
           "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
           "algorithmic_bytes_per_launch_tensor_once": alg, "algorithmic_total": sum(alg.values()),
           "traffic_over_algorithmic": (rd + wr) / sum(alg.values())},
          open(O + "/%s_pmc_dv_rb128.json" % tag, "w"), indent=1)
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
# the same two passes on the fp8 mode's launch of that layer: since round 4 its INPUT GRADIENT (the forward stays fp16): operand = a
# producer's e5m2 copy of the gradient, block-scaled K = 128 MFMA
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc8_$c -- python3 tools/one_conv.py dgrad dv_rb128 f8 > /dev/null 2>&1
done
python3 - $O $TAG <<'PY'
import csv, glob, json, sys
O, tag = sys.argv[1], sys.argv[2]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(O + "/pmc8_%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if "conv3x3_patch" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v.append(float(r["Counter_Value"]))
    vals[c] = v
rd = 2 * 1024 * sum(vals["FETCH_SIZE"]) / max(1, len(vals["FETCH_SIZE"]))
wr = 1024 * sum(vals["WRITE_SIZE"]) / max(1, len(vals["WRITE_SIZE"]))
# every DISTINCT tensor the launch touches, once (round-3 verdict: the old table counted the input twice)
alg = {"e5m2 copy of the gradient (the MFMA operand)": 536870912, "the bf16 gradient itself (residual of the block's input gradient)": 1073741824,
       "the forward input, read for the sign of act' (fp16)": 1073741824, "weights (e4m3, transposed)": 589824, "output (bf16)": 1073741824}
json.dump({"kernel": "conv3x3_patch_kernel<bf16,128,2,16,F8=3,PRE> (input gradient)",
           "launch": "decoder_visualize res-block conv, INPUT GRADIENT in fp8 mode: n=128 images, 128x128, 256->256, the operand arrives as a producer's e5m2 copy, v_mfma_scale_f32_16x16x128_f8f6f4 (tools/one_conv.py dgrad dv_rb128 f8); the forward launch of this layer stays fp16 in fp8 mode",
           "command": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --output-format csv -- python3 tools/one_conv.py dgrad dv_rb128 f8",
           "FETCH_SIZE_kb_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_kb_per_launch": vals["WRITE_SIZE"],
           "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE exact",
           "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
           "algorithmic_bytes_per_launch_tensor_once": alg, "algorithmic_total": sum(alg.values()),
           "traffic_over_algorithmic": (rd + wr) / sum(alg.values())},
          open(O + "/%s_pmc_dv_rb128_fp8.json" % tag, "w"), indent=1)
PY
rm -rf $O/pmc8_FETCH_SIZE $O/pmc8_WRITE_SIZE
for m in fwd dgrad wgrad; do
  bash tools/pmc_conv.sh ${TAG}_sq_$m $m > /dev/null 2>&1
  (echo "== $m"; cat $R/gpurun_out/pmc_${TAG}_sq_$m.txt) >> $O/${TAG}_sq_dv_rb128.txt 2>/dev/null
  rm -rf $R/gpurun_out/pmc_${TAG}_sq_$m $R/gpurun_out/pmc_${TAG}_sq_$m.txt
done
for m in dgrad; do          # fp8 mode's launch of the same layer (its forward stays fp16)
  bash tools/pmc_conv.sh ${TAG}_sq8_$m $m dv_rb128 f8 > /dev/null 2>&1
  (echo "== $m, fp8 copy in, block-scaled MFMA"; cat $R/gpurun_out/pmc_${TAG}_sq8_$m.txt) >> $O/${TAG}_sq_dv_rb128_fp8.txt 2>/dev/null
  rm -rf $R/gpurun_out/pmc_${TAG}_sq8_$m $R/gpurun_out/pmc_${TAG}_sq8_$m.txt
done
(echo "== bf16, activation-on-load (round-2 form)"; python3 tools/bench_conv.py --iters 10; echo "== the model form: post-activation storage; fp16 forward tensors for the 3x3 layers"; python3 tools/bench_conv.py --iters 10 --post --f16) > $O/${TAG}_bench_conv.txt 2>&1
python3 tools/bench_conv.py --fp8 --iters 10 --only dv_rb128,dv_rb64,dv_rb32,dv_rb16,ea_rb2,vgg3_2,vgg4_2 > $O/${TAG}_bench_conv_fp8.txt 2>&1
(echo "== forward operand arrives as an e4m3 copy: block-scaled K = 128 MFMA"; python3 tools/bench_conv.py --fp8 --fp8-copy --iters 10 --only dv_rb128,dv_rb64,dv_rb32,dv_rb16,ea_rb2,vgg3_2,vgg4_2; echo "== the same with UPS_F8_SCALED=0 (K = 32 fp8 MFMA)"; UPS_F8_SCALED=0 python3 tools/bench_conv.py --fp8 --fp8-copy --iters 10 --only dv_rb128,dv_rb64,dv_rb32,dv_rb16,ea_rb2,vgg3_2,vgg4_2) > $O/${TAG}_bench_conv_fp8_copy.txt 2>&1
python3 tools/bench_first.py > $O/${TAG}_bench_first.txt 2>&1
# fp8 mode: one-stream kernel trace grouped by (kernel, grid)
UPS_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats8 -- python3 bench.py --precision fp8 --no-cpu-baseline --steps 10 --warmup 6 > $O/bench_fp8_nooverlap.log 2>&1
grep '"metric"' $O/bench_fp8_nooverlap.log > $O/${TAG}_bench_b64_fp8_nooverlap.json
python3 tools/by_grid.py $(find $O/stats8 -name "*kernel_trace.csv" | head -1) 16 0.1 > $O/${TAG}_by_kernel_and_grid_fp8_nooverlap.txt
rm -rf $O/stats8
python3 tools/hbm_roofline.py --json $O/${TAG}_hbm_kernels.json > $O/${TAG}_hbm_kernels.txt 2>&1
bash tools/pmc_part.sh $TAG > /dev/null 2>&1; cp $R/gpurun_out/pmc_part_$TAG.txt $O/${TAG}_pmc_part_kernels.txt
# the other perceptual-input readings and the other BASELINE configs (one bench line each)
for pi in resize256 resize256_crop224; do python3 bench.py --no-cpu-baseline --perceptual-input $pi 2>/dev/null | grep '"metric"' > $O/${TAG}_bench_b64_$pi.json; done
python3 bench.py --no-cpu-baseline --precision fp8 2>/dev/null | grep '"metric"' > $O/${TAG}_final_bench_b64_fp8.json
for cf in deepfashion256p16 pennaction128 cub256p20; do python3 bench.py --no-cpu-baseline --config $cf 2>/dev/null | grep '"metric"' > $O/${TAG}_bench_$cf.json; done
python3 bench.py --no-cpu-baseline --config cub256p20 --precision bf16 2>/dev/null | grep '"metric"' > $O/${TAG}_bench_cub256p20_bf16.json
# round 5: the un-profiled headline line (with the CPU baseline leg), the fp32 line, and the fp8 weight gradient of the roofline layer
# (SQ counters, HBM traffic, isolated timings against the bf16 kernel)
python3 bench.py --steps 20 --warmup 5 2>/dev/null | grep '"metric"' > $O/${TAG}_final_bench_b64.json
python3 bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '"metric"' > $O/${TAG}_bench_b64_fp32.json
bash tools/pmc_conv.sh ${TAG}_sq8_wgrad wgrad dv_rb128 f8 > /dev/null 2>&1
(echo "== wgrad, fp8 kernel (conv_wgrad3x3_f8.hip): e5m2 copy of the gradient in, fp16 forward tensor quantised while staged"; cat $R/gpurun_out/pmc_${TAG}_sq8_wgrad.txt) > $O/${TAG}_sq_wgrad_dv_rb128_fp8.txt 2>/dev/null
rm -rf $R/gpurun_out/pmc_${TAG}_sq8_wgrad $R/gpurun_out/pmc_${TAG}_sq8_wgrad.txt
# tensor-once bytes of that launch: fp16 forward tensor 1 073 741 824 + e5m2 copy 536 870 912 + 128 fp32 slabs of (9 x 258 x 256 + 256) floats 304 218 112
bash tools/pmc_traffic.sh $O/${TAG}_pmc_wgrad_dv_rb128_fp8.json conv_wgrad3x3_f8 1914830848 -- wgrad dv_rb128 f8 > /dev/null 2>&1
(echo "== bf16 weight gradient (conv_wgrad3x3<64,128,8,SLIDE>), the mask decoder's residual blocks as the model issues them"; python3 tools/bench_conv.py --f16 --post --only dv_rb128,dv_rb64,dv_rb32,dv_rb16 --iters 10; echo "== fp8 weight gradient (wgrad column; the other columns are bench_conv's in-kernel-conversion forms)"; python3 tools/bench_conv.py --f16 --post --fp8 --fp8-wgrad --only dv_rb128,dv_rb64,dv_rb32,dv_rb16 --iters 10) > $O/${TAG}_bench_conv_wgrad_fp8.txt 2>&1
cat $O/${TAG}_bench_b64.json | cut -c1-300; cat $O/${TAG}_bench_b64_nooverlap.json | cut -c1-200
head -14 $O/${TAG}_bench_b64_kernel_stats.csv | cut -c1-160
cat $O/${TAG}_patch_kernel_by_grid.txt | head -8; cat $O/${TAG}_bilinear_by_grid.txt | head -8; cat $O/${TAG}_bilinear_by_grid_nooverlap.txt | head -8
cat $O/${TAG}_sq_dv_rb128.txt; cat $O/${TAG}_pmc_dv_rb128.json | tail -12
