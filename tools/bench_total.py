"""total_loss from raw pyramids (bench.py's end_to_end leg alone), with a kernel-time breakdown: GPU time of one call
from events around a HIP-graph replay vs the eager wall time.  usage (GPU box): python3 tools/bench_total.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
print(json.dumps(bench.total_loss_leg(dev, reps=100), indent=1))
