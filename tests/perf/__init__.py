"""GPU-vs-CPU timing scripts that use the oracle as the CPU side (the oracle is test infrastructure, so they live here)."""
