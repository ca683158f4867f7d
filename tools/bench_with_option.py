"""bench.py with process-wide rasterizer options set first (A/B of an option on one box):
    python tools/bench_with_option.py stream_policy=0 -- --steps 300 --warmup 30 --no-cpu-baseline"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussiancity_amd import _native as N
i = sys.argv.index("--")
for kv in sys.argv[1:i]:
    k, v = kv.split("=")
    N.set_option(k, int(v))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[i + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")
