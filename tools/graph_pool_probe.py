"""How large is a captured hipGraph's private allocator pool, and which counter sees it?  (GPU diagnostic.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
x = torch.randn(64, 1024, 1024, device=dev)
torch.cuda.synchronize()
r0, f0 = torch.cuda.memory_reserved(dev), torch.cuda.mem_get_info(dev)[0]
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = (x * 2).sum(0) + x[0]
    z = torch.cat([x, x])
torch.cuda.synchronize()
print("reserved delta", torch.cuda.memory_reserved(dev) - r0, "free delta", f0 - torch.cuda.mem_get_info(dev)[0], "pool id", g.pool())
snap = torch.cuda.memory_snapshot()
print("snapshot keys", sorted(snap[0].keys()) if snap else None)
by = {}
for seg in snap:
    by.setdefault(tuple(seg.get("segment_pool_id", ("?",))), 0)
    by[tuple(seg.get("segment_pool_id", ("?",)))] += seg["total_size"]
print("bytes by pool id", by)
