import json, sys, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
out = bench.decode_b1(torch, dev)
print(json.dumps(out, indent=1))
