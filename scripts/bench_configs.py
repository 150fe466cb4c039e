#!/usr/bin/env python3
"""Prints the secondary BASELINE configs (configs[0] / [2] / [3]) as bench-format JSON lines of their own:
`python3 scripts/bench_configs.py [0] [2] [3]`.  The measurement code lives in bench_configs.py at the repo root (bench.py
carries the same objects in `secondary.configs`); this entry point is what the rocprofv3 summaries under profiles/ were
taken on (`rocprofv3 --kernel-trace --stats -- python3 scripts/bench_configs.py 3`)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench_configs  # noqa: E402

if __name__ == "__main__":
    for line in bench_configs.run(tuple(sys.argv[1:]) or ("0", "2", "3")).values():
        print(json.dumps(line), flush=True)
