#!/bin/bash
# Seeds 0-49 of selkov/noise20_eq_wsindy.cfg, one process per seed like the reference's script of the same name.
# Run from a directory that holds run_configs/ (this package directory does) with the repository root on
# PYTHONPATH.
for i in {0..49}; do
    echo "Running seed $i"
    python -m symode_amd.main_wsindy --seed "$i" --config selkov/noise20_eq_wsindy.cfg
done
