"""A/B build that differs from the product library in ONE translation unit (seconds instead of the full rebuild of build_variant.py):
    python tools/build_one.py <name> <file.hip> [-DFLAG ...]   ->  tools/_ab/liboffk_<name>.so
The other objects are the product build's (optical-flow-guided-feature-pytorch_amd/csrc/_obj): build the product library first.
Only for flags that do not change a struct shared with other translation units."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("offk_build", os.path.join(ROOT, "optical-flow-guided-feature-pytorch_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)

name, src, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
out_dir = os.path.join(ROOT, "tools", "_ab")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, "%s_%s.o" % (name, src.replace(".hip", "")))
cmd = [b._hipcc(), "-c", os.path.join(b.CSRC, src), "-o", obj] + b.FLAGS + b.EXTRA_FLAGS.get(src, []) + defs
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    raise SystemExit(r.stderr)
objs = [obj if s == src else os.path.join(b.OBJ, s.replace(".hip", ".o")) for s in b.SOURCES]
lib = os.path.join(out_dir, "liboffk_%s.so" % name)
subprocess.run([b._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs, check=True)
print(lib)
