#!/usr/bin/env python3
"""tools/hostapi_bench.py [GiB] -- the `host_api` leg of bench.py alone: sz_compress / sz_uncompress on pageable numpy buffers
through ctypes, per-call wall times, bytes against the oracle (bench.host_api_rates)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from pkg_loader import load_package
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 2
print(json.dumps(bench.host_api_rates(load_package(), gib << 30), indent=1))
