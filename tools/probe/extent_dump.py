"""Radius and {alpha >= 1/255} pixel bounds (pack_record) of the Gaussians an optimise view sees, late in a whole-sequence run:
what a strip-backward task costs under the shipped classes (by radius) and would cost under classes by the bounds' half width.
python tools/probe/extent_dump.py [frames] [W H] [out.npz]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from bench_kernels import _python_twin

dev = "cuda:0"
torch.cuda.set_device(0)
bench.prime(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480)
out = sys.argv[4] if len(sys.argv) > 4 else "gpurun_out/extents_%dx%d.npz" % (W, H)
seq = bench.synthetic_sequence_device(W, H, n, 1234, dev)
sc = bench.Scene(seq, None, 1234, False, overlap=False, n_frames=n, keyframe_theta=1.0, keyframe_trans=0.02, capacity=1 << 20)
sc.run(0, n)
torch.cuda.synchronize()
model, cam, rc = _python_twin(sc, dev)
model.initOptimizers(-1, 1.0)
model.train_step(cam, rc["depth_map"], rc["color_map"], cam.image, ref_depth_clamped=rc["depth_map_clamped"])
torch.cuda.synchronize()
N = model.getGaussianNum()
B = model._B
rec = B["records"][:N * 12].view(N, 12).cpu().numpy() if B["records"].dim() == 1 else B["records"][:N].reshape(N, 12).cpu().numpy()
radii = B["radii"][:N].cpu().numpy()
xb = rec[:, 10].view(np.int32); yb = rec[:, 11].view(np.int32)
x_lo = (xb & 0xffff).astype(np.int16).astype(np.int32); x_hi = xb >> 16
y_lo = (yb & 0xffff).astype(np.int16).astype(np.int32); y_hi = yb >> 16
np.savez_compressed(out, radii=radii.astype(np.int16), mx=rec[:, 0].astype(np.float32), my=rec[:, 1].astype(np.float32),
                    x_lo=x_lo.astype(np.int16), x_hi=x_hi.astype(np.int16), y_lo=y_lo.astype(np.int16), y_hi=y_hi.astype(np.int16), W=W, H=H)
v = radii > 0
print("N %d visible %d; mean radius %.1f, mean half width of the bounds %.1f, mean rows %.1f (2 r = %.1f)" %
      (N, v.sum(), radii[v].mean(), ((x_hi - x_lo + 1)[v] / 2).mean(), (y_hi - y_lo + 1)[v].mean(), 2 * radii[v].mean()))
sc.close()
