#!/bin/bash
# Regenerates the evidence under profiles/ for one round tag: kernel-trace stats, the two HBM-traffic PMC passes
# (FETCH_SIZE and WRITE_SIZE in separate runs, per MI355X_MICROARCH.md) and one SQ counter pass, all over the same
# eager single-stream bench command.  usage: tools/profile_all.sh r02   (run from the repo root on the GPU box)
set -eo pipefail
tag=${1:-rXX}
out=gpurun_out/prof_$tag
mkdir -p "$out"
cmd="bench.py --graph 0 --streams 1 --steps 30 --warmup 5 --settle-steps 30 --no-cpu-baseline --no-roofline --no-extras"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/stats" -o run --output-format csv -- python3 $cmd > "$out/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmcF" -o run --output-format csv -- python3 $cmd > "$out/pmcF.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmcW" -o run --output-format csv -- python3 $cmd > "$out/pmcW.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY \
  -d "$out/pmcS" -o run --output-format csv -- python3 $cmd > "$out/pmcS.log" 2>&1
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" profiles/${tag}_final_bench_cnn_bf16_b256_kernel_stats.csv
python3 tools/pmc_traffic.py "$(find "$out/pmcF" -name '*counter_collection.csv' | head -1)" \
  "$(find "$out/pmcW" -name '*counter_collection.csv' | head -1)" profiles/${tag}_hbm_traffic_pmc.json "python3 $cmd" > "$out/traffic.txt"
python3 tools/sq_summary.py "$(find "$out/pmcS" -name '*counter_collection.csv' | head -1)" profiles/${tag}_final_bench_cnn_bf16_b256_pmc_sq.csv > "$out/sq.txt"
# kernel stats of the other benchmarked shapes: config 4's per-GPU shape (arcface, 1024 faces, 10 000 IDs) and config 5 (hybrid, 256 faces)
rocprofv3 --kernel-trace --stats -d "$out/stats4" -o run --output-format csv -- python3 $cmd --model arcface --batch 1024 --gallery 10000 > "$out/stats4.log" 2>&1
cp "$(find "$out/stats4" -name '*kernel_stats.csv' | head -1)" profiles/${tag}_config4_arcface_b1024_g10000_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d "$out/stats5" -o run --output-format csv -- python3 $cmd --model hybrid > "$out/stats5.log" 2>&1
cp "$(find "$out/stats5" -name '*kernel_stats.csv' | head -1)" profiles/${tag}_config5_hybrid_b256_kernel_stats.csv
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_* gpurun_out/profiles_$tag/
echo "profiles for $tag regenerated"
