"""combat_amd/libcombat_hip_stamps.so = the library with conv3x3_dma.hip (and whatever else is named on the command
line) compiled with -DCOMBAT_STAMPS; use it through COMBAT_HIP_LIB.  The product library is not touched."""
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
from combat_amd import build as b
b.build(verbose=False)
names = sys.argv[1:] or ["conv3x3_dma.hip"]
objs = []
for src in b.SOURCES:
    o = os.path.join(b.OBJ, os.path.splitext(src)[0] + ".o")
    if src in names:
        o = os.path.join(b.OBJ, os.path.splitext(src)[0] + ".stamps.o")
        subprocess.run([b.HIPCC] + b.FLAGS + ["-DCOMBAT_STAMPS", "-c", os.path.join(b.CSRC, src), "-o", o], check=True)
    objs.append(o)
out = os.path.join(b.HERE, "libcombat_hip_stamps.so")
subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print(out)
