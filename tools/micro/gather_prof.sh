# kernel trace of the index-table Gram at the config[3] shape (tools/micro/gather_gram.py): what the kernel itself takes,
# beside the event-bracketed call times the script prints
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O="$R/gpurun_out"; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gather -- python3 "$R/tools/micro/gather_gram.py" > "$O/r03_gather_gram.txt" 2>&1
cd "$R"
python tools/rocprof_summary.py /tmp/prof_gather > "$O/r03_gather_gram_kernel_stats.txt"
tail -5 "$O/r03_gather_gram.txt"; head -14 "$O/r03_gather_gram_kernel_stats.txt"
