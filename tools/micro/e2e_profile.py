"""cProfile of the single-seed driver (symode_amd.main.main) on the reference's config names, second run (data files
exist, kernels warm):  python tools/micro/e2e_profile.py dosc/noise20_sindy.cfg [extra main args ...]"""
import cProfile, io, os, pstats, shutil, sys, tempfile, time, contextlib
sys.path.insert(0, os.getcwd())
import torch
import importlib
M = importlib.import_module("symode_amd." + os.environ.get("E2E_MODULE", "main"))     # main | main_sindy | main_wsindy | main_sweep

cfg = sys.argv[1]
extra = sys.argv[2:]
work = os.environ.get("E2E_WORKDIR") or tempfile.mkdtemp(prefix="e2e_")     # E2E_WORKDIR: reuse data files across processes
os.makedirs(work, exist_ok=True)
if not os.path.exists(os.path.join(work, "run_configs")):
    shutil.copytree(os.path.join(os.path.dirname(os.path.abspath(M.__file__)), "run_configs"), os.path.join(work, "run_configs"))
os.chdir(work)
FIRST = os.environ.get("E2E_FIRST_ONLY") == "1"          # profile the FIRST run of this process (per-seed process cost)
argv = ["--seed", "0", "--config", cfg, "--gpu", "0"] + extra        # (main_sweep: pass --n_seeds / --method among the extras)
if os.environ.get("E2E_PREP"):                      # e.g. E2E_PREP="lv/noise99_sym.cfg --num_epochs 1": a run whose outputs cfg loads
    prep = os.environ["E2E_PREP"].split()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        M.main(["--seed", "0", "--config", prep[0], "--gpu", "0"] + prep[1:])
    print(f"prep {' '.join(prep)}: {time.perf_counter() - t0:.2f} s", flush=True)
for rep in range(1 if FIRST else 2):
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        pr.enable()
        M.main(list(argv))
        torch.cuda.synchronize()
        pr.disable()
    dt = time.perf_counter() - t0
    print(f"{cfg} {' '.join(extra)} run {rep}: {dt:.3f} s", flush=True)
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(45)
print("\n".join(l[:170] for l in st.getvalue().splitlines()[:75]))
