"""A/B a differently-built libmmg: python scripts/ab_lib.py libmmg_x.so  -> bench numbers with that library."""
import os, sys, json, subprocess
lib = sys.argv[1]
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = "import sys; sys.path.insert(0, %r); from multimodalgame_amd import _lib; _lib.LIB_PATH = %r; import bench; sys.argv=['bench.py','--steps','300','--warmup','20','--no-cpu-baseline']; bench.main()" % (here, os.path.join(here, "multimodalgame_amd", lib))
out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout.strip().splitlines()[-1]
d = json.loads(out)
print(lib, round(d["value"]), round(d["ms_per_step"] * 1e3, 1), d["roofline"]["kernels_us"])
