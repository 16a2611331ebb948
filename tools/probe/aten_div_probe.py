"""(x - 0.5) / C0 with a python float: which float sequence does ATen run on this GPU build?"""
import numpy as np, torch
C0 = 0.28209479177387814
x = torch.rand(1 << 20, device="cuda")
ref = (x - 0.5) / C0
c0f = np.float32(C0)
cands = {
    "(x-0.5) * f32(1/f32(C0))": (x - 0.5) * torch.tensor(np.float32(1.0) / c0f, device="cuda"),
    "(x-0.5) * f32(1/C0 in double)": (x - 0.5) * torch.tensor(np.float32(1.0 / C0), device="cuda"),
    "(x-0.5) / tensor(f32(C0))": (x - 0.5) / torch.tensor(c0f, device="cuda"),
    "double then round": ((x.double() - 0.5) / C0).float(),
}
for k, v in cands.items():
    print("%-32s mismatches: %d" % (k, int((v != ref).sum())))
print("f32(1/f32(C0)) =", repr(np.float32(1.0) / c0f), " f32(1/C0) =", repr(np.float32(1.0 / C0)))
