#!/bin/bash
# Re-measure everything under profiles/ on the GPU box (run through gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh'
# Raw traces go to /tmp on the box; only the condensed summaries land in gpurun_out/ (copy them to profiles/).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" && python bench.py > gpurun_out/r01_bench_n1.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 "$R/bench.py" --steps 20 --warmup 3 --profile > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 "$R/bench.py" --steps 5 --warmup 1 --profile > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 "$R/bench.py" --steps 5 --warmup 1 --profile > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ops -- python3 "$R/tests/perf/ops_table.py" > /dev/null 2>&1
cd "$R"
python tools/rocprof_summary.py /tmp/prof_bench > gpurun_out/r01_bench_kernel_stats.txt
python tools/rocprof_summary.py /tmp/prof_ops > gpurun_out/r01_ops_kernel_stats.txt
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write loss_grad_kernel 4194304 1024000000 > gpurun_out/pmc_traffic.json
head -4 gpurun_out/r01_bench_kernel_stats.txt
python -c 'import json; r=json.load(open("gpurun_out/r01_bench_n1.json")); print(r["value"], r["roofline"]["frac"], r["roofline"]["traffic"], r["cpu_baseline"]["value"])'
