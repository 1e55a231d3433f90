#!/bin/bash
python - <<'P'
import sys, json, torch
sys.path.insert(0, '.')
dummies = [torch.cuda.Stream() for _ in range(7)]      # emulate a process that has created streams before
import bench
print(json.dumps(bench.vae_leg(torch.device("cuda:0"))))
P
python tools/gpu_decode_pair.py 2>/dev/null
