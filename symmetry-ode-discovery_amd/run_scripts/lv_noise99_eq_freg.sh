#!/bin/bash
# Seeds 0-49 of lv/noise99_eq_fsymreg.cfg, one process per seed like the reference's script of the same name.
# Run from a directory that holds run_configs/ (this package directory does) with the repository root on
# PYTHONPATH.
for i in {0..49}; do
    echo "Running seed $i"
    python -m symode_amd.main --seed "$i" --config lv/noise99_eq_fsymreg.cfg
done
