#!/usr/bin/env python3
"""Run ONE conv shape a few times (target for rocprofv3 --pmc)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_conv  # noqa: E402

if __name__ == "__main__":
    idx = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    bench_conv.run(bench_conv.SHAPES[idx], iters=3)
