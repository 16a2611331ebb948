import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gps_slam_amd import gsplat_ops as ops
from bench_kernels import _time_launches
for (H, W) in ((480, 640), (720, 1280)):
    a = torch.rand((1, H, W, 3), device='cuda'); b = torch.rand((1, H, W, 3), device='cuda')
    m, d1, d2, d3 = ops.fusedssim(1e-4, 9e-4, a, b, train=True, channels_last=True)
    dL = torch.randn_like(a)
    st = torch.cuda.current_stream()
    import ctypes as C
    from gps_slam_amd._lib import lib
    sp = C.c_void_p(st.cuda_stream); p = lambda t: C.c_void_p(t.data_ptr())
    g = torch.empty_like(a)
    for cl in (1, 0):
        f = lambda: lib.gps_ssim_fwd(1, 3, H, W, cl, 1e-4, 9e-4, p(a), p(b), p(m), p(d1), p(d2), p(d3), sp)
        r = lambda: lib.gps_ssim_bwd(1, 3, H, W, cl, p(a), p(b), p(dL), p(d1), p(d2), p(d3), p(g), sp)
        tf, tb = _time_launches(f, 50, st), _time_launches(r, 50, st)
        P = H * W * 3 * 4
        print("%dx%d %s: fwd %.1f us (%.0f GB/s of 6 images), bwd %.1f us (%.0f GB/s of 7 images)" % (W, H, "HWC" if cl else "CHW", tf * 1e6, 6 * P / tf / 1e9, tb * 1e6, 7 * P / tb / 1e9))
