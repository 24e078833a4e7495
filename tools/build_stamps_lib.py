"""A second library beside the product one, with some sources compiled with extra flags (profiling / A-B builds):
    python tools/build_stamps_lib.py [--flags=-DCOMBAT_STAMPS] [--out=libcombat_hip_stamps.so] [conv3x3_dma.hip ...]
-> combat_amd/<out>; use it through COMBAT_HIP_LIB.  The product library is not touched."""
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
from combat_amd import build as b
b.build(verbose=False)
flags, out_name, names = ["-DCOMBAT_STAMPS"], "libcombat_hip_stamps.so", []
for a in sys.argv[1:]:
    if a.startswith("--flags="):
        flags = a[8:].split()
    elif a.startswith("--out="):
        out_name = a[6:]
    else:
        names.append(a)
names = names or ["conv3x3_dma.hip"]
tag = os.path.splitext(out_name)[0].replace("libcombat_hip_", "")
objs = []
for src in b.SOURCES:
    o = os.path.join(b.OBJ, os.path.splitext(src)[0] + ".o")
    if src in names:
        o = os.path.join(b.OBJ, os.path.splitext(src)[0] + "." + tag + ".o")
        subprocess.run([b.HIPCC] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, src), "-o", o], check=True)
    objs.append(o)
out = os.path.join(b.HERE, out_name)
subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print(out)
