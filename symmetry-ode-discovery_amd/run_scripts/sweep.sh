#!/bin/bash
# Seeds 0-49 of one equation-discovery config (the reference's run_scripts/*.sh loop over `python main.py --seed $i`).
#   L-BFGS SINDy / EquivSINDy-c configs run all seeds in ONE process on the batched kernels;
#   configs with an autoencoder / symmetry regulariser fall back to the per-seed loop.
# usage (from symmetry-ode-discovery_amd/):  bash run_scripts/sweep.sh dosc/noise20_sindy.cfg
set -e
cfg=$1
export PYTHONPATH=${PYTHONPATH:-..}
if grep -q -- "--load_laligan\|--w_sym_reg 0\.[1-9]\|--use_latent" "run_configs/$cfg"; then
    for i in $(seq 0 49); do
        echo "Running seed $i"
        python -m symode_amd.main --seed "$i" --config "$cfg"
    done
else
    python -m symode_amd.main_sweep --seed 0 --n_seeds 50 --config "$cfg"
fi
