"""Adam against THE optimiser: libtorch's torch::optim::Adam, constructed the way the reference constructs it
(src/raw_gs_model.cpp:654-674 -- one optimizer per tensor, lr / eps / betas through float variables) and stepped the way
optimizersStep() steps it (:696-705), on the GPU (oracle/libtorch_adam.cpp: the installed libtorch's own Adam::step, nothing
restated), plus Python's torch.optim.Adam(foreach=False, fused=False) for the record.

What is pinned: exp_avg and exp_avg_sq BIT-identical to the library's after every one of 5 steps (zero gradients and a re-created
optimiser included), for gps_adam_step and for the fused preprocessing-backward kernel's Adam; the parameters identical except
where ATen-on-ROCm's float division is not the IEEE quotient -- settled element by element with exact rational arithmetic
(test_parameter_update_is_the_ieee_one): wherever the two differ, the HIP kernel holds the correctly rounded
fma(-step_size, RN(m / denom), p) and the library does not.
"""
import math
from fractions import Fraction

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref_adam():
    from oracle import libtorch_adam_build
    return libtorch_adam_build.load()


def _f32(x):
    return float(np.float32(x))


# the reference's learning rates as it holds them: float members, meansLr * scene_scale a float product (raw_gs_model.cpp:26-32, :666)
LRS = [_f32(np.float32(1.6e-4) * np.float32(3.3)), _f32(5e-3), _f32(1e-3), _f32(2.5e-3), _f32(5e-4), _f32(5e-2)]
SHAPES = [(50001, 3), (50001, 3), (50001, 4), (50001, 3), (50001, 15, 3), (50001, 1)]


def _rn_f32(q):
    """round-to-nearest-even float32 of an exact Fraction (normal range)"""
    if q == 0:
        return np.float32(0.0)
    sign = -1 if q < 0 else 1
    a = abs(q)
    e = math.floor(math.log2(a))
    while Fraction(2) ** e > a:
        e -= 1
    while Fraction(2) ** (e + 1) <= a:
        e += 1
    e = max(e, -126)                       # subnormals share the exponent -126
    ulp = Fraction(2) ** (e - 23)
    n = a / ulp
    k = n.numerator // n.denominator
    r = n - k
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and k % 2 == 1):
        k += 1
    return np.float32(sign * float(k * ulp))   # k * ulp is exactly representable: the float() is exact


def test_host_constants_are_the_references():
    from gps_slam_amd import gs_model
    eps, b1, b2 = _ref_adam().RefAdam.scalars()
    assert (gs_model.ADAM_BETA1, gs_model.ADAM_BETA2, gs_model.ADAM_EPS) == (b1, b2, eps)
    assert b1 == 0.8999999761581421 and b2 == 0.9990000128746033 and eps == 1.0000000036274937e-15


def _within_one_quotient_ulp(a, b, lr, step):
    """|a - b| <= one ulp of the quotient m / denom (|q| <= ~4 after bias correction) times the step size, plus the one ulp of
    the parameter itself that the final rounding of p - step_size * q can then differ by -- element by element"""
    bound = 2.0 ** -21 * lr / (1 - 0.9 ** step) + 2.0 ** -22 * a.abs()
    return bool(((a - b).abs() <= bound).all())


def _grads(gen, step, scale=1e-3):
    out = []
    for k, s in enumerate(SHAPES):
        g = torch.randn(s, generator=gen) * scale
        if step == 3 and k % 2:
            g.zero_()                      # a tensor nothing was seen of in this iteration
        if step == 4:
            g[::7] = 0.0                   # Gaussians outside the view: exact zeros among live gradients
        out.append(g.to(DEV))
    return out


def test_gps_adam_step_equals_torch_optim_adam_over_five_steps_and_a_recreated_optimiser():
    from gps_slam_amd import gs_model, gsplat_ops as ops
    mod = _ref_adam()
    gen = torch.Generator().manual_seed(11)
    P = [torch.randn(s, generator=gen).to(DEV) for s in SHAPES]
    ref = mod.RefAdam(P, LRS)
    pyp = [p.clone().requires_grad_(True) for p in P]
    mk_py = lambda: [torch.optim.Adam([p], lr=lr, betas=(gs_model.ADAM_BETA1, gs_model.ADAM_BETA2), eps=gs_model.ADAM_EPS,
                                      foreach=False, fused=False) for p, lr in zip(pyp, LRS)]
    py = mk_py()
    M = [torch.full_like(p, float("nan")) for p in P]    # step 1 must not read them
    V = [torch.full_like(p, float("nan")) for p in P]
    n_diff, n_el = 0, 0
    for generation in range(2):                           # initOptimizers() runs again before every localOptimize
        for step in range(1, 6):
            G = _grads(gen, step)
            ref.step(G)
            for p, g, o in zip(pyp, G, py):
                p.grad = g.clone()
                o.step()
            ops.adam_step(P, G, M, V, LRS, step, (gs_model.ADAM_BETA1, gs_model.ADAM_BETA2), gs_model.ADAM_EPS)
            for a, b in zip(M, ref.exp_avg()):
                assert torch.equal(a, b), "exp_avg differs from torch::optim::Adam's (generation %d, step %d)" % (generation, step)
            for a, b in zip(V, ref.exp_avg_sq()):
                assert torch.equal(a, b), "exp_avg_sq differs from torch::optim::Adam's (generation %d, step %d)" % (generation, step)
            for a, b, lr in zip(P, ref.parameters(), LRS):
                ne = a != b
                n_diff += int(ne.sum()); n_el += a.numel()
                # one ulp of the quotient m / denom (<= 1 in magnitude after bias correction ~ O(1)) scaled by the step size
                assert _within_one_quotient_ulp(a, b, lr, step)
                b.copy_(a)                                 # the library continues from the HIP parameters: compare steps, not drift
            # Python's optimiser: _single_tensor_adam updates exp_avg with lerp_ (m + (g - m) * (1 - beta1)), not the C++ frontend's
            # mul_ / add_ -- a different rounding of the same value: close, not bit-equal; kept off the trajectory
            for a, p in zip(P, pyp):
                torch.testing.assert_close(a, p.detach(), rtol=0, atol=1e-5 * 5.0)
                p.data.copy_(a)
        ref.init()
        py = mk_py()
        M = [m.fill_(float("nan")) for m in M]
        V = [v.fill_(float("nan")) for v in V]
    print("parameters: %d of %d elements differ from the library's by the last bit of the quotient" % (n_diff, n_el))
    assert n_diff <= 2e-3 * n_el


def test_parameter_update_is_the_ieee_one():
    """Which side of the one-ulp parameter differences is IEEE-exact.  For a sample of elements -- every element where HIP and
    libtorch-on-ROCm disagree after one step from identical state, plus as many where they agree -- the update is recomputed in
    exact rational arithmetic from the float32 inputs both sides share bit for bit (p, exp_avg, exp_avg_sq after the step):
        denom = RN(RN(RN(sqrt(v)) * RN32(1 / sqrt(bc2))) + eps),  q = RN(m / denom),  p' = RN(p - step_size * q)   (an fma)
    (sqrt taken with numpy float32, which is correctly rounded; libtorch divides by sqrt(bc2) where the kernel multiplies by its
    float reciprocal -- test above: the moments and the agreeing parameters show the two denominators are the same floats, and
    this test only uses the kernel's formula to decide between two candidate results that differ in the LAST step).  HIP must
    equal the exact result on every sampled element."""
    from gps_slam_amd import gs_model, gsplat_ops as ops
    mod = _ref_adam()
    gen = torch.Generator().manual_seed(5)
    shape, lr, step = (200000, 3), LRS[1], 2
    p0 = torch.randn(shape, generator=gen).to(DEV)
    ref = mod.RefAdam([p0], [lr])
    M, V = [torch.zeros_like(p0)], [torch.zeros_like(p0)]
    P = [p0.clone()]
    for s in range(1, step + 1):
        G = [(torch.randn(shape, generator=gen) * 1e-3).to(DEV)]
        p_before = P[0].clone()
        ref.step(G)
        ops.adam_step(P, G, M, V, [lr], s, (gs_model.ADAM_BETA1, gs_model.ADAM_BETA2), gs_model.ADAM_EPS)
        if s < step:
            ref.parameters()[0].copy_(P[0])
    assert torch.equal(M[0], ref.exp_avg()[0]) and torch.equal(V[0], ref.exp_avg_sq()[0])
    hip, lib = P[0].cpu().numpy().ravel(), ref.parameters()[0].cpu().numpy().ravel()
    pb, m, v = p_before.cpu().numpy().ravel(), M[0].cpu().numpy().ravel(), V[0].cpu().numpy().ravel()
    differ = np.flatnonzero(hip != lib)
    agree = np.flatnonzero(hip == lib)[: max(200, differ.size)]
    b1, b2, eps = gs_model.ADAM_BETA1, gs_model.ADAM_BETA2, gs_model.ADAM_EPS
    bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
    inv = np.float32(1.0 / math.sqrt(bc2))
    step_size = np.float32(lr / bc1)
    eps32 = np.float32(eps)
    hip_exact = lib_exact = 0
    for i in np.concatenate([differ, agree]):
        sq = np.sqrt(v[i])                                             # float32, correctly rounded
        denom = _rn_f32(Fraction(float(_rn_f32(Fraction(float(sq)) * Fraction(float(inv))))) + Fraction(float(eps32)))
        q = _rn_f32(Fraction(float(m[i])) / Fraction(float(denom)))
        want = _rn_f32(Fraction(float(pb[i])) - Fraction(float(step_size)) * Fraction(float(q)))
        hip_exact += int(want == hip[i])
        lib_exact += int(want == lib[i])
        assert want == hip[i], (i, float(want), float(hip[i]), float(lib[i]))
    print("sampled %d elements (%d where the two differ): HIP IEEE-exact on %d, libtorch-on-ROCm on %d"
          % (differ.size + agree.size, differ.size, hip_exact, lib_exact))
    assert lib_exact == agree.size   # every disagreement is the library's division


@pytest.mark.parametrize("fuse", [0, 2])
def test_train_step_adam_equals_torch_optim_adam(fuse):
    """The train step's Adam (fuse 0: gps_adam_step behind the gradient-writing backward; fuse 2: inside preprocess_bwd_kernel,
    the timed path) against torch::optim::Adam fed the SAME gradients (those of the fuse-0 twin, which are bit-identical to the
    ones the fused kernel consumes in registers: tests/test_train_step_full_gpu.py), 4 iterations, then a re-created optimiser."""
    from gps_slam_amd.gs_model import Camera, SLAMGaussianModel
    from tests import scenes
    mod = _ref_adam()
    N, W, H = 30000, 320, 240
    g = scenes.random_gaussians(N, seed=3, scale_range=(0.004, 0.02))
    c2w, K = scenes.default_camera(W, H, seed=3)
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    models = []
    for mode in (0, fuse):
        m = SLAMGaussianModel(dict(capacity=1 << 16, fuse_sh_rest_adam=mode, strip_backward=True), device=DEV)
        m.add_params(dict(means=T(g["means"]), scales=T(g["log_scales"]), quats=T(g["quats"]), featuresDc=T(g["sh"][:, 0].copy()),
                          featuresRest=T(g["sh"][:, 1:].copy()), opacities=T(g["opac_logit"])))
        models.append(m)
    mG, mF = models
    gen = torch.Generator().manual_seed(3)
    gt = torch.rand((H, W, 3), generator=gen).to(DEV)
    base = torch.rand((H, W, 3), generator=gen).to(DEV)
    refd = (torch.rand((H, W, 1), generator=gen) * 3.5 + 0.5).to(DEV)
    cam = Camera(0, W, H, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), c2w, image=gt, device=DEV)
    for generation in range(2):
        for m in models:
            m.initOptimizers(-1, 3.3)
        lrs = mG._opt["lrs"]
        assert lrs == LRS
        ref = mod.RefAdam([t.clone() for t in mF.opt_gs_params.tensors()], lrs)
        for it in range(1, 5):
            for m in models:
                m.train_step(cam, refd, base, gt)
            torch.cuda.synchronize()
            ref.step([t.clone() for t in mG.grads()])
            for a, b, c in zip(mF._opt["m"], mG._opt["m"], ref.exp_avg()):
                assert torch.equal(a[:N], c) and torch.equal(b[:N], c), (generation, it)
            for a, b, c in zip(mF._opt["v"], mG._opt["v"], ref.exp_avg_sq()):
                assert torch.equal(a[:N], c) and torch.equal(b[:N], c), (generation, it)
            for a, b, c, lr in zip(mF.opt_gs_params.tensors(), mG.opt_gs_params.tensors(), ref.parameters(), lrs):
                assert torch.equal(a, b)
                assert float((a != c).float().mean()) < 2e-2   # (the library's division: test_parameter_update_is_the_ieee_one)
                assert _within_one_quotient_ulp(a, c, lr, it)
                c.copy_(a)
