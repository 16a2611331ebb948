"""What, done right before a run, is followed by the ~100 ms hold of the process's queues?  A scene is built, 0.5 s pass (no hold
pending), ONE action runs, the run starts.  Printed: the stall and how long after the action's end it ended."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench

dev = "cuda:0"
torch.cuda.set_device(0)
bench.prime(dev)
n = 300
reps = int(os.environ.get("REPS", "6"))
seq = bench.synthetic_sequence_device(640, 480, n, 1234, dev)
hip = ctypes.CDLL("libamdhip64.so")
keepalive = []
def act(kind):
    if kind.startswith("torch_pinned_"):
        k = int(kind.rsplit("_", 1)[1])
        keepalive.append([torch.zeros(1 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(k)])
    elif kind.startswith("raw_pinned_"):
        k = int(kind.rsplit("_", 1)[1])
        for _ in range(k):
            p = ctypes.c_void_p()
            assert hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20), 0) == 0
            ctypes.memset(p, 1, 1 << 20)
    elif kind == "host_churn":       # plain host memory: 600 MB allocated, touched, freed (munmap)
        a = np.ones(600 << 20, np.uint8); del a
    elif kind == "host_keep":
        keepalive.append(np.ones(600 << 20, np.uint8))
    elif kind == "pinned_recycled":  # 600 pinned tensors freed and taken again from torch's pool (no driver call)
        t = [torch.zeros(1 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(600)]; del t
        t = [torch.zeros(1 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(600)]; keepalive.append(t)
        time.sleep(0.5)
        del keepalive[-1]
        keepalive.append([torch.zeros(1 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(600)])
for kind in os.environ.get("ACTIONS", "none,torch_pinned_600,torch_pinned_20,torch_pinned_1,raw_pinned_600,host_churn,host_keep,pinned_recycled").split(","):
    for rep in range(reps):
        sc = bench.Scene(seq, None, 1234, False, overlap=True, n_frames=n, keyframe_theta=1.0, keyframe_trans=0.02)
        sc.pipe.keep_frame_ms = True
        torch.cuda.synchronize()
        time.sleep(0.5)
        t_a0 = time.perf_counter()
        act(kind)
        torch.cuda.synchronize()
        t_a = time.perf_counter()
        tm = sc.pipe.SLAMTrainCamsTimed(sc.model, sc.cams)
        ms = np.asarray(sc.pipe.frame_ms)
        w = int(ms[:60].argmax())
        end = t_a + 1e-3 * ms[:w + 1].sum()
        print("%-18s rep %d: action %.0f ms; %.0f frames/s, longest call %.1f ms at frame %d, ending %.0f ms after the action's end (%.0f after its start)" %
              (kind, rep, 1e3 * (t_a - t_a0), tm.fps(), ms[w], w, 1e3 * (end - t_a), 1e3 * (end - t_a0)), flush=True)
        sc.close(); del sc
        torch.cuda.empty_cache()
        keepalive.clear()
