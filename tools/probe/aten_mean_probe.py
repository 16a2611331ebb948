"""Which float sequence does torch.mean(x, -1) over 3 channels / x.sum(2) use on this GPU build?"""
import torch
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(480, 640, 3, device="cuda", generator=g)
x[::7, ::5] = x[::7, ::5] * 0 + torch.tensor([0.1, 0.1, 0.1], device="cuda") + 1e-8 * torch.rand(1, device="cuda")
m = torch.mean(x, -1, True)[..., 0]
e0, e1, e2 = x[..., 0], x[..., 1], x[..., 2]
third = torch.tensor(1.0 / 3.0, dtype=torch.float32, device="cuda")
cands = {
    "((a+b)+c)*f(1/3)": ((e0 + e1) + e2) * third,
    "((a+b)+c)/3": ((e0 + e1) + e2) / 3.0,
    "(a+(b+c))*f(1/3)": (e0 + (e1 + e2)) * third,
    "(a+(b+c))/3": (e0 + (e1 + e2)) / 3.0,
    "((a+c)+b)*f(1/3)": ((e0 + e2) + e1) * third,
    "((a+c)+b)/3": ((e0 + e2) + e1) / 3.0,
    "(a*f+b*f)+c*f": (e0 * third + e1 * third) + e2 * third,
    "f64 mean -> f32": x.double().mean(-1).float(),
    "f64 sum * f32(1/3)": (x.double().sum(-1) * float(third.double())).float(),
}
for k, v in cands.items():
    print("%-22s mismatches vs torch.mean: %d" % (k, int((v != m).sum())))
s = x.sum(2)
for k, v in {"(a+b)+c": (e0 + e1) + e2, "a+(b+c)": e0 + (e1 + e2), "(a+c)+b": (e0 + e2) + e1, "f64": x.double().sum(2).float()}.items():
    print("sum: %-10s mismatches vs x.sum(2): %d" % (k, int((v != s).sum())))
