"""Second build of liboffk with extra -D flags (timing / tuning builds), loaded through OFFK_LIB:
    python tools/build_variant.py timing -DOFFK_PT_TIMING        -> tools/_bin/liboffk_timing.so
Objects go to tools/_bin/<name>_obj; the product library is untouched."""
import importlib.util
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("offk_build", os.path.join(ROOT, "optical-flow-guided-feature-pytorch_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)

name, defs = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(ROOT, "tools", os.environ.get("OFFK_VARIANT_DIR", "_bin"))
obj_dir = os.path.join(out_dir, name + "_obj")
os.makedirs(obj_dir, exist_ok=True)


def one(src):
    o = os.path.join(obj_dir, src.replace(".hip", ".o"))
    cmd = [b._hipcc(), "-c", os.path.join(b.CSRC, src), "-o", o] + b.FLAGS + b.EXTRA_FLAGS.get(src, []) + defs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr)
    return o


with ThreadPoolExecutor(max_workers=4) as ex:
    objs = list(ex.map(one, b.SOURCES))
lib = os.path.join(out_dir, "liboffk_%s.so" % name)
subprocess.run([b._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs, check=True)
print(lib)
