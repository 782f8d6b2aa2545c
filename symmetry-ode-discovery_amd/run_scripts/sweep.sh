#!/bin/bash
# Seeds 0-49 of one equation-discovery config (the reference's run_scripts/*.sh loop over `python main.py --seed $i`).
#   L-BFGS SINDy / EquivSINDy-c configs (no autoencoder, no latent model, no symmetry regulariser) run all seeds in ONE
#   process on the batched kernels (main_sweep); weak-SINDy configs loop over main_wsindy, everything else over main.
# usage (from symmetry-ode-discovery_amd/):  bash run_scripts/sweep.sh dosc/noise20_sindy.cfg
set -e
cfg=$1
export PYTHONPATH=${PYTHONPATH:-..}
file="run_configs/$cfg"
per_seed() {
    for i in $(seq 0 49); do
        echo "Running seed $i"
        python -m "symode_amd.$1" --seed "$i" --config "$cfg"
    done
}
case "$cfg" in
    *wsindy*) per_seed main_wsindy; exit 0 ;;
esac
if grep -q -- "--sindy_optimizer lbfgs" "$file" && ! grep -q -- "--load_laligan\|--use_latent\|--w_sym_reg 0\.0*[1-9]\|--w_sym_reg [1-9]" "$file"; then
    python -m symode_amd.main_sweep --seed 0 --n_seeds 50 --config "$cfg"
else
    per_seed main
fi
