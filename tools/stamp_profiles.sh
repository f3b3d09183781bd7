#!/bin/bash
# Copy the outputs of tools/final_round.sh <src-tag> from gpurun_out/ (scratch) into profiles/ (tracked) under the round's names.
# usage: bash tools/stamp_profiles.sh r06a r06
src=${1:?source tag}; dst=${2:?round tag}
cp gpurun_out/bench_$src.json profiles/${dst}_bench.json
cp gpurun_out/bench_details_$src.txt profiles/${dst}_bench_details.txt
cp gpurun_out/rocprof_kernel_stats_$src.csv profiles/${dst}_rocprofv3_kernel_stats_bench.csv
for p in f16x2 bf16; do cp gpurun_out/rocprof_kernel_stats_${src}_$p.csv profiles/${dst}_rocprofv3_kernel_stats_$p.csv; done
for p in fp32 f16x2 bf16; do cp gpurun_out/pmc_bench_${src}_$p.json profiles/${dst}_pmc_bench_$p.json; done
cp gpurun_out/b1_timeline_$src.txt profiles/${dst}_b1_timeline.txt
cp gpurun_out/latency_$src.txt profiles/${dst}_latency_sweep.txt
cp gpurun_out/torchrun_n1_$src.txt profiles/${dst}_torchrun_n1.txt
grep -v amdgpu.ids gpurun_out/final_round_$src.log | cut -c1-400 > profiles/${dst}_final_round_log.txt
ls -la profiles/${dst}_*
