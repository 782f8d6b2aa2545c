#!/bin/bash
# Re-measure everything under profiles/ on the GPU box (run through gpurun from the repo root):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh r03'
# Raw traces go to /tmp on the box; only the condensed summaries land in gpurun_out/ (copy them to profiles/).
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O="$R/gpurun_out"
mkdir -p "$O"
cd "$R" && python bench.py > "$O/${TAG}_bench_n1.json" 2> "$O/${TAG}_bench_n1.err" && echo "bench done"
python bench.py --two_launch_sym --no_cpu_baseline --steps 50 > "$O/${TAG}_bench_n1_two_launch.json" 2>/dev/null && echo "two-launch bench done"
cd /tmp && export TMPDIR=/tmp
# per-kernel durations of the bench step (same command, profiling mode: only the K batched steps)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 "$R/bench.py" --steps 20 --warmup 3 --profile > /dev/null 2>&1 && echo "trace done"
# HBM traffic: separate PMC passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 "$R/bench.py" --steps 4 --warmup 1 --profile > /dev/null 2>&1 && echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 "$R/bench.py" --steps 4 --warmup 1 --profile > /dev/null 2>&1 && echo "write done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch2 -- python3 "$R/bench.py" --steps 4 --warmup 1 --profile --two_launch_sym > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write2 -- python3 "$R/bench.py" --steps 4 --warmup 1 --profile --two_launch_sym > /dev/null 2>&1 && echo "two-launch traffic done"
# every streaming entry point at 2^26 points under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ops -- python3 "$R/tools/kbench.py" --table --S 1 --N 67108864 --libs 3:0,2:2 --reps 3 > "$O/${TAG}_ops_roofline.md" 2>/dev/null && echo "ops done"
# HBM traffic of vjp with grad_x (16-byte nt loads of x and g, 16-byte nt stores of grad_x): 24 algorithmic bytes per point
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch_vjp -- python3 "$R/tools/kbench.py" --op vjp --S 1 --N 67108864 --order 3 --reps 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write_vjp -- python3 "$R/tools/kbench.py" --op vjp --S 1 --N 67108864 --order 3 --reps 3 > /dev/null 2>&1 && echo "vjp traffic done"
# weak-SINDy contraction (N1): one trajectory of 10 000 time points, 50 test functions, and a 2^20-point one for the rate
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_weak -- python3 "$R/tools/kbench.py" --op weak_gram --S 1 --N 10000 --order 3 --K 50 --reps 20 > "$O/${TAG}_weak_gram.txt" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_weak_big -- python3 "$R/tools/kbench.py" --op weak_gram --S 1 --N 1048576 --order 3 --K 50 --reps 20 >> "$O/${TAG}_weak_gram.txt" 2>/dev/null && echo "weak gram done"
# VALU counters of the arithmetic-bound kernels, MFMA counters of the Gram
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_valu -- python3 "$R/tools/kbench.py" --table --S 1 --N 67108864 --libs 3:0 --reps 1 --only euler_jvp euler_jvp_vjp odeint odeint_rk4 symreg_linear loss_grad symreg_reversed vjp jvp_vjp > /dev/null 2>&1 && echo "valu done"
export SYMODE_GRAM_VALU=0      # order 3 (F = 12) takes the vector-pipe Gram by default: this pass records the MFMA form it replaced
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma3 -- python3 "$R/tools/kbench.py" --op gram --S 1024 --N 125000 --order 3 --reps 3 > /dev/null 2>&1
unset SYMODE_GRAM_VALU
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_gvalu -- python3 "$R/tools/kbench.py" --op gram --S 1024 --N 125000 --order 3 --reps 3 > /dev/null 2>&1
export SYMODE_GRAM_M4=0        # order 5 (F = 23) takes the 4x4-tile matrix-core Gram by default: this pass records the split vector-pipe form it replaced
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_gvalu5 -- python3 "$R/tools/kbench.py" --op gram --S 1024 --N 125000 --order 5 --reps 3 > /dev/null 2>&1
unset SYMODE_GRAM_M4
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma5 -- python3 "$R/tools/kbench.py" --op gram --S 1024 --N 125000 --order 5 --reps 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gram5 -- python3 "$R/tools/kbench.py" --op gram --S 1024 --N 125000 --order 5 --reps 5 > /dev/null 2>&1 && echo "gram pmc done"
cd "$R"
python tools/rocprof_summary.py /tmp/prof_bench > "$O/${TAG}_bench_kernel_stats.txt"
python tools/rocprof_summary.py /tmp/prof_ops > "$O/${TAG}_ops_kernel_stats.txt"
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write 1024000000 "closure_reversed=symreg_reversed_kernel<symode::Library<2, 5, 0>, true" > "$O/pmc_traffic.json"
python tools/pmc_traffic.py /tmp/pmc_fetch2 /tmp/pmc_write2 1024000000 loss_grad=loss_grad_kernel "symreg_reversed=symreg_reversed_kernel<symode::Library<2, 5, 0>, false" > "$O/pmc_traffic_two_launch.json"
python tools/pmc_valu.py /tmp/pmc_valu euler_jvp_kernel euler_jvp_vjp_kernel odeint_kernel symreg_linear_kernel loss_grad_kernel symreg_reversed_kernel "symode::vjp_kernel" "symode::jvp_vjp_kernel" > "$O/${TAG}_valu_pmc.json"
python tools/pmc_traffic.py /tmp/pmc_fetch_vjp /tmp/pmc_write_vjp 67108864 "vjp_grad_x=symode::vjp_kernel" > "$O/pmc_traffic_vjp.json"
python tools/rocprof_summary.py /tmp/prof_weak > "$O/${TAG}_weak_gram_kernel_stats.txt"
python tools/rocprof_summary.py /tmp/prof_weak_big >> "$O/${TAG}_weak_gram_kernel_stats.txt"
python tools/pmc_mfma.py /tmp/pmc_mfma3 aug_gram_kernel > "$O/${TAG}_gram_o3_mfma_pmc.json"
python tools/pmc_valu.py /tmp/pmc_gvalu aug_gram_valu_kernel > "$O/${TAG}_gram_o3_valu_pmc.json"
python tools/pmc_valu.py /tmp/pmc_gvalu5 aug_gram_split_kernel > "$O/${TAG}_gram_o5_split_pmc.json"
python tools/pmc_mfma.py /tmp/pmc_mfma5 aug_gram_m4_kernel > "$O/${TAG}_gram_o5_m4_mfma_pmc.json"
python tools/rocprof_summary.py /tmp/prof_gram5 > "$O/${TAG}_gram_o5_kernel_stats.txt"
python tools/latency_bench.py --orders 3 5 > "$O/${TAG}_latency.txt" 2>&1
head -6 "$O/${TAG}_bench_kernel_stats.txt"
cat "$O/pmc_traffic.json" | head -30
cat "$O/${TAG}_valu_pmc.json" | head -60
