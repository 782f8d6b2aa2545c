# kernel trace of three 64 x 50 000-point L-BFGS sweeps (tools/micro/prof_sweep.py): time per kernel of an inner iteration
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O="$R/gpurun_out"; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sweep -- python3 "$R/tools/micro/prof_sweep.py" > "$O/r03_sweep_prof.txt" 2>&1
cd "$R"
python tools/rocprof_summary.py /tmp/prof_sweep > "$O/r03_sweep_kernel_stats.txt"
grep "^graph" "$O/r03_sweep_prof.txt"; head -12 "$O/r03_sweep_kernel_stats.txt"
