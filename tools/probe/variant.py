"""Build a variant of the C-ABI library for an A/B on one GPU box: ONE source file replaced (or recompiled with extra -D flags),
every other object taken from gps_slam_amd/build/, the flags exactly those of gps_slam_amd/_build.py (the splat files are
compiled WITH fp contraction, the tsdf files without -- a variant built with other flags measures the flags).

usage: python tools/probe/variant.py <name> <file.hip | path to a replacement source> [extra hipcc flags...]
   ->  tools/probe/libgps_<name>.so     (run `python -c "import __graft_entry__ as g; g.build()"` first)
e.g.   python tools/probe/variant.py inflight4 splat_raster.hip -DGPS_BWD_INFLIGHT=4
       git show HEAD~1:gps_slam_amd/csrc/tsdf_fusion.hip > /tmp/tsdf_fusion.hip && python tools/probe/variant.py prev /tmp/tsdf_fusion.hip
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gps_slam_amd import _build as B

name, src = sys.argv[1], sys.argv[2]
base = os.path.basename(src)
path = src if os.path.sep in src else os.path.join(B.CSRC, src)
assert base in B.sources(), "a replacement source keeps the name of the file it replaces (e.g. /tmp/x/splat_raster.hip)"
flags = list(B.COMMON) + [f for pre, fl in B.PER_FILE.items() if base.startswith(pre) for f in fl] + sys.argv[3:]
obj = "/tmp/variant_%s.o" % name
subprocess.check_call([B.HIPCC] + flags + ["-I" + os.path.join(ROOT, "include"), "-I" + B.CSRC, "-c", path, "-o", obj])
objs = [os.path.join(B.HERE, "build", f[:-4] + ".o") for f in B.sources() if f != base] + [obj]
out = os.path.join(ROOT, "tools", "probe", "libgps_%s.so" % name)
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
