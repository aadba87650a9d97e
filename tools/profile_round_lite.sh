#!/bin/bash
# usage (GPU box): bash tools/profile_round_lite.sh <tag>
# The trace / bench-line part of tools/profile_round.sh (no counter passes, no per-layer tables): re-run after a change that moves
# launches between streams or removes launches but leaves the kernels themselves alone.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-round4}
cd $R
O=$R/gpurun_out/profiles_$TAG
rm -rf $O; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 2>/dev/null | grep '"metric"' > $O/${TAG}_final_bench_b64.json
python3 bench.py --no-cpu-baseline --precision fp8 2>/dev/null | grep '"metric"' > $O/${TAG}_final_bench_b64_fp8.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline > $O/bench.log 2>&1
grep '"metric"' $O/bench.log > $O/${TAG}_bench_b64.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_b64_kernel_stats.csv
python3 tools/trace_summary.py $(find $O/stats -name "*kernel_trace.csv" | head -1) conv3x3_patch > $O/${TAG}_patch_kernel_by_grid.txt
python3 tools/timeline.py $(find $O/stats -name "*kernel_trace.csv" | head -1) > $O/${TAG}_timeline.txt 2>&1
rm -rf $O/stats
UPS_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 bench.py --no-cpu-baseline > $O/bench_nooverlap.log 2>&1
grep '"metric"' $O/bench_nooverlap.log > $O/${TAG}_bench_b64_nooverlap.json
cp $(find $O/stats1 -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_b64_nooverlap_kernel_stats.csv
python3 tools/trace_summary.py $(find $O/stats1 -name "*kernel_trace.csv" | head -1) conv3x3_patch > $O/${TAG}_patch_kernel_by_grid_nooverlap.txt
python3 tools/by_grid.py $(find $O/stats1 -name "*kernel_trace.csv" | head -1) 10 0.1 > $O/${TAG}_by_kernel_and_grid_nooverlap.txt
rm -rf $O/stats1
UPS_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats8 -- python3 bench.py --precision fp8 --no-cpu-baseline --steps 10 --warmup 6 > $O/bench_fp8_nooverlap.log 2>&1
grep '"metric"' $O/bench_fp8_nooverlap.log > $O/${TAG}_bench_b64_fp8_nooverlap.json
python3 tools/by_grid.py $(find $O/stats8 -name "*kernel_trace.csv" | head -1) 16 0.1 > $O/${TAG}_by_kernel_and_grid_fp8_nooverlap.txt
rm -rf $O/stats8
for pi in resize256 resize256_crop224; do python3 bench.py --no-cpu-baseline --perceptual-input $pi 2>/dev/null | grep '"metric"' > $O/${TAG}_bench_b64_$pi.json; done
for cf in deepfashion256p16 pennaction128 cub256p20; do python3 bench.py --no-cpu-baseline --config $cf 2>/dev/null | grep '"metric"' > $O/${TAG}_bench_$cf.json; done
python3 bench.py --no-cpu-baseline --config cub256p20 --precision bf16 2>/dev/null | grep '"metric"' > $O/${TAG}_bench_cub256p20_bf16.json
for f in $O/*.json; do echo "$(basename $f): $(cut -c1-120 $f)"; done
head -3 $O/${TAG}_by_kernel_and_grid_nooverlap.txt | cut -c1-100; head -1 $O/${TAG}_by_kernel_and_grid_fp8_nooverlap.txt; tail -12 $O/${TAG}_timeline.txt
