"""Which HIP calls does torch make for a pinned host tensor (fresh and recycled)?  AMD_LOG_LEVEL=3 python tools/probe/pinned_calls.py"""
import sys, torch
torch.cuda.init(); torch.zeros(4, device="cuda:0"); torch.cuda.synchronize()
a = torch.zeros(1 << 20, dtype=torch.uint8, pin_memory=True)   # warm: the allocator's one-off set-up
sys.stderr.write("=== MARK fresh\n"); sys.stderr.flush()
b = torch.zeros(1 << 21, dtype=torch.uint8, pin_memory=True)
sys.stderr.write("=== MARK free\n"); sys.stderr.flush()
del b
sys.stderr.write("=== MARK recycled\n"); sys.stderr.flush()
c = torch.zeros(1 << 21, dtype=torch.uint8, pin_memory=True)
sys.stderr.write("=== MARK end\n"); sys.stderr.flush()
