set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O="$R/gpurun_out"; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma5 -- python3 "$R/tools/kbench.py" --op gram --S 1024 --N 125000 --order 5 --reps 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gram5 -- python3 "$R/tools/kbench.py" --op gram --S 1024 --N 125000 --order 5 --reps 5 > /dev/null 2>&1
cd "$R"
python tools/pmc_mfma.py /tmp/pmc_mfma5 aug_gram_m4_kernel > "$O/r03_gram_o5_m4_mfma_pmc.json"
python tools/rocprof_summary.py /tmp/prof_gram5 > "$O/r03_gram_o5_kernel_stats.txt"
cat "$O/r03_gram_o5_m4_mfma_pmc.json"; head -8 "$O/r03_gram_o5_kernel_stats.txt"
# HBM traffic of the same launch (separate passes, as for the bench kernel): 16 algorithmic bytes per point
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch_gram5 -- python3 "$R/tools/kbench.py" --op gram --S 1024 --N 125000 --order 5 --reps 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write_gram5 -- python3 "$R/tools/kbench.py" --op gram --S 1024 --N 125000 --order 5 --reps 3 > /dev/null 2>&1
cd "$R"
python tools/pmc_traffic.py /tmp/pmc_fetch_gram5 /tmp/pmc_write_gram5 128000000 "gram_m4=aug_gram_m4_kernel" > "$O/r03_pmc_traffic_gram_o5.json"
cat "$O/r03_pmc_traffic_gram_o5.json" | head -30
