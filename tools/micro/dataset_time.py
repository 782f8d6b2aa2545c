"""First-run cost of the reference's data recipes (RK4 + noise + GP smoothing), by phase (cProfile)."""
import cProfile, io, os, pstats, sys, tempfile, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from symode_amd.parser_utils import get_args
from symode_amd.dataset import get_dataset
task, noise = sys.argv[1], sys.argv[2]
os.chdir(tempfile.mkdtemp(prefix="ds_"))
args = vars(get_args(argv=["--task", task, "--noise", noise, "--smoothing", "gp", "--gpu", "0"]))
pr = cProfile.Profile()
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable(); ds, _, args = get_dataset(args); torch.cuda.synchronize(); pr.disable()
print(f"{task} noise {noise} gp: get_dataset (first run) {time.perf_counter() - t0:.2f} s; x {tuple(ds.x.shape)}", flush=True)
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(18)
print("\n".join(l[:150] for l in st.getvalue().splitlines()[:40]))
