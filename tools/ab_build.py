r"""Builds variants of libazula_amd.so for A/B experiments:  python tools/ab_build.py NAME -DMACRO[=v] ...
-> azula_amd/csrc/_ab/libazula_amd_NAME.so (travels with gpurun); select with AZULA_AMD_LIB=<path>."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from azula_amd.csrc import build as B

name, defs = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(B.HERE, "_ab")
obj_dir = os.path.join(out_dir, "obj_" + name)
os.makedirs(obj_dir, exist_ok=True)
objs = []
procs = []
for s in B.SOURCES:
    obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
    objs.append(obj)
    procs.append(subprocess.Popen([B.hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(s, []), *defs, "-x", "hip", "-c", os.path.join(B.HERE, s), "-o", obj]))
assert all(p.wait() == 0 for p in procs)
lib = os.path.join(out_dir, f"libazula_amd_{name}.so")
subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], check=True)
print(lib)
