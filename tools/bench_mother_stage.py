import sys, json, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
print(json.dumps(bench.mother_stage_leg(dev, 0, steps=5, warmup=2), indent=1))
