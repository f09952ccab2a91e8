"""rocprofv3 workload: the edge-sharded / global-BA leg of bench.py alone (update_lowmem over 64 keyframes, 372 edges)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(bench.edge_sharded_leg(torch.device("cuda:0"), 0, 1, steps=2))
