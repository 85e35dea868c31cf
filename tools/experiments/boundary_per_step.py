import sys, os, json; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["MW_BENCH_PER_STEP"]="1"
import bench
args = bench.parse_args(["--no-cpu-baseline"])
r = bench.boundary_rates(args, None, 0)
for k in ("host_numpy","torch_device"): print(k, round(r[k]["value"]), r[k]["median_ms_per_step"], r[k]["per_step_ms"])
