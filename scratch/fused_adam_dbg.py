import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch, numpy as np
from tests.test_splat_gpu import _setup, T, _dev
from gps_slam_amd import gsplat_ops as ops
from gps_slam_amd._lib import check, lib
N, W, H = 10007, 160, 120
g, vm, K, c2w = _setup(N, W, H, seed=11)
sh = T(g["sh"])
P = dict(means=T(g["means"]), ls=T(g["log_scales"]), q=T(g["quats"]), ol=T(g["opac_logit"]).view(-1).contiguous(), dc=sh[:, 0].contiguous(), rest=sh[:, 1:].contiguous())
vmT, KT, cp = T(vm), T(K), T(c2w[:3, 3].copy())
radii, m2, depths, conics, colors, opac = ops.gauss_preprocess_fwd(P["means"], P["ls"], P["q"], P["ol"], P["dc"], P["rest"], 3, vmT, KT, cp, W, H)
gen = torch.Generator().manual_seed(5)
rnd = lambda *s: torch.randn(*s, generator=gen).to(_dev())
v_m2, v_con, v_col, v_op = rnd(N, 2), rnd(N, 3), rnd(N, 4), rnd(N)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream); p = lambda t: C.c_void_p(t.data_ptr())
restA, mA, vA = P["rest"].clone(), torch.zeros_like(P["rest"]), torch.zeros_like(P["rest"])
restB, mB, vB = P["rest"].clone(), torch.zeros_like(P["rest"]), torch.zeros_like(P["rest"])
gA = ops.gauss_preprocess_bwd(P["means"], P["ls"], P["q"], P["ol"], P["dc"], restA, 3, vmT, KT, cp, W, H, 0.3, radii, conics, v_m2, v_con, v_col, v_op)
ops.adam_step([restA], [gA[5]], [mA], [vA], [5e-4], 1)
outB = [torch.empty_like(x) for x in gA[:5]]; g_restB = torch.empty_like(restB)
check(lib.gps_gauss_preprocess_bwd_adam(N, 16, 3, p(P["means"]), p(P["ls"]), p(P["q"]), p(P["ol"]), p(P["dc"]), p(restB), p(vmT), p(KT), p(cp), W, H, 0.3, p(radii), p(conics), p(v_m2), p(v_con), p(v_col), p(v_op), p(outB[0]), p(outB[1]), p(outB[2]), p(outB[3]), p(outB[4]), p(g_restB), p(mB), p(vB), 5e-4, 0.9, 0.999, 1e-15, 1, st), "x")
torch.cuda.synchronize()
for name, a, b in (("g", gA[5], g_restB), ("m", mA, mB), ("v", vA, vB), ("p", restA, restB)):
    ne = (a != b)
    print(name, "mismatch", int(ne.sum()), "of", a.numel(), "max abs diff", float((a - b).abs().max()))
    if ne.any():
        idx = ne.nonzero()[:5]
        print(idx.tolist(), a[ne][:5].tolist(), b[ne][:5].tolist())
