#!/bin/bash
# Round-end sequence on the GPU box: parity tests, smoke, the default bench line (+ per-class table), rocprofv3 kernel-trace summary, PMC
# passes of the same step (stamped with the library's build id), one-face latency sweep.  Usage (repo root): bash tools/final_round.sh <tag>
tag=${1:-r06}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/pytest_gpu_$tag.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench"; timeout 900 python bench.py --details > gpurun_out/bench_$tag.json 2> gpurun_out/bench_details_$tag.txt; echo "rc=$?"; cut -c1-600 gpurun_out/bench_$tag.json
echo "== rocprofv3 kernel trace"; rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-exact-leg --no-config3-leg --no-parity-gate > gpurun_out/rocprof_run_$tag.log 2>&1; echo "rc=$?"
f=$(find /tmp/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/rocprof_kernel_stats_$tag.csv && head -8 "$f" | cut -c1-160
echo "== pmc (fp32: the headline)"; bash tools/pmc_bench.sh $tag fp32 > gpurun_out/pmc_bench_${tag}_fp32.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/pmc_bench_${tag}_fp32.log | cut -c1-300
# the secondary legs get the same evidence: kernel stats and PMC passes of bench.py --precision f16x2 / bf16
for prec in f16x2 bf16; do
  wflag=""; [ "$prec" = "bf16" ] && wflag="--w 0.7"
  echo "== rocprofv3 kernel trace, precision $prec"; rm -rf /tmp/prof_$prec && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$prec -o bench -- python bench.py --precision $prec $wflag --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-gate > gpurun_out/rocprof_run_${tag}_$prec.log 2>&1; echo "rc=$?"
  f=$(find /tmp/prof_$prec -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/rocprof_kernel_stats_${tag}_$prec.csv && head -6 "$f" | cut -c1-160
  echo "== pmc, precision $prec"; bash tools/pmc_bench.sh $tag $prec > gpurun_out/pmc_bench_${tag}_$prec.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/pmc_bench_${tag}_$prec.log | cut -c1-300
done
echo "== one-face timeline"; bash tools/b1_timeline.sh 1 $tag | head -12
echo "== latency"; timeout 300 python tools/latency.py f16x2,fp32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/latency_$tag.txt
# the N > 1 launch path with one rank (torchrun, RCCL initialised, gather to rank 0): what the driver's scaling bench runs per rank
echo "== torchrun, one rank"; HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-f16x2-leg --no-config3-leg 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300 | tee gpurun_out/torchrun_n1_$tag.txt
