"""configs[4] as one chained step (bench.py config5_leg) on its own: python3 tools/bench_c5.py [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.config5_leg(torch.device("cuda:0"), int(sys.argv[1]) if len(sys.argv) > 1 else 10), indent=1))
