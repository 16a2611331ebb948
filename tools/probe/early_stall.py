"""The early-run stall (one ~90 ms or ~2.08 s gap within the first ~20 frames of some whole runs): does it follow the START OF THE
RUN or the CREATION OF THE SCENE?  Scenes are created, then the run starts after SLEEP seconds (alternating 0 / SLEEP).
MODE = seq | ovl; REPS scenes each; TRIGGER = an allocation burst right before the run; the container's throttled CPU periods
(cgroup cpu.stat) are printed per run.  GPS_BENCH_NO_THREAD_CAP=1 keeps libtorch's default intra-op pool (the stalls' cause:
LABBOOK section 14); the stalls only reproduce with a build whose createTsdfEngine zero-fills its images with torch::zeros."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench

dev = "cuda:0"
torch.cuda.set_device(0)
bench.prime(dev)
n = int(os.environ.get("FRAMES", "300"))
reps = int(os.environ.get("REPS", "12"))
sleep = float(os.environ.get("SLEEP", "0.5"))
keep = os.environ.get("KEEP", "0") == "1"       # keep every scene alive (no close / empty_cache between the runs)
seq = bench.synthetic_sequence_device(640, 480, n, 1234, dev)
kept = []
import glob
def cpu_stat():   # the container's CPU accounting: throttled periods so far
    try:
        return {l.split()[0]: l.split()[1] for l in open("/sys/fs/cgroup/cpu.stat") if l.startswith(("nr_throttled", "throttled_usec"))}
    except OSError:
        return {}
print("intra-op threads %d; cpu.stat at start: %s" % (torch.get_num_threads(), cpu_stat()), flush=True)
for rep in range(reps):
    ov = os.environ.get("MODE", "ovl") == "ovl"
    t_c = time.time()
    sc = bench.Scene(seq, None, 1234, False, overlap=ov, n_frames=n, keyframe_theta=1.0, keyframe_trans=0.02)
    sc.pipe.keep_frame_ms = True
    torch.cuda.synchronize()
    t_c = time.time() - t_c
    s = sleep if rep % 2 else 0.0
    if s: time.sleep(s)
    trig = os.environ.get("TRIGGER", "none")
    junk = None
    if trig == "pinned":      # 600 fresh pinned host allocations (what createTsdfEngine does for the frames), kept alive
        junk = [torch.zeros(1 << 20, dtype=torch.uint8).pin_memory() for _ in range(600)]
    elif trig == "pinned1":   # ONE fresh pinned allocation of the same total
        junk = torch.zeros(600 << 20, dtype=torch.uint8).pin_memory()
    elif trig == "vram":      # one fresh 1 GB device allocation straight from the driver
        junk = torch.cuda.caching_allocator_alloc(1 << 30); 
    elif trig == "vramfree":
        j = torch.cuda.caching_allocator_alloc(1 << 30); torch.cuda.caching_allocator_delete(j); torch.cuda.empty_cache()
    elif trig == "memset":    # first touch of a fresh 1 GB device allocation
        junk = torch.zeros(1 << 28, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    k0 = cpu_stat()
    tm = sc.pipe.SLAMTrainCamsTimed(sc.model, sc.cams)
    k1 = cpu_stat()
    print("   throttled during the run:", {k: int(v) - int(k0.get(k, 0)) for k, v in k1.items()}, flush=True)
    ms = np.asarray(sc.pipe.frame_ms)
    w = int(ms.argmax())
    print("rep %2d sleep %.1f create %.2f s: %.0f frames/s, first 30 frames %.1f ms, slowest %.1f ms (frame %d), start offset of it %.1f ms" %
          (rep, s, t_c, tm.fps(), ms[:30].sum(), ms[w], w, ms[:w].sum()), flush=True)
    if keep: kept.append(sc)
    else:
        sc.close(); del sc
        torch.cuda.empty_cache()
