"""The `node_uwb_pose_T500` leg of bench.py alone (cfg/uwb_pose.yaml at trajectory_length 500: fill the window, four solves), without the CPU
baseline — the program tools/profile_r04.sh puts under rocprofv3 for that leg's kernel-trace statistics and PMC passes."""
import importlib.util, json, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
D = types.SimpleNamespace(rank=0, world=1, local_rank=0)
args = types.SimpleNamespace(no_cpu_baseline=True, seed=0)
print(json.dumps(bench.leg_node_se3(D, args, "node_uwb_pose_T500")))
