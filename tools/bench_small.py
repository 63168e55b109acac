#!/usr/bin/env python3
"""Small-image convs through flowse_op_conv2d (the model handle's kernel choice for the shape); run under
rocprofv3 --kernel-trace --stats for kernel-only times."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_conv  # noqa: E402

SHAPES = [(1, 4, 4, 256, 0, 256, 9), (1, 8, 8, 256, 0, 256, 9), (1, 16, 16, 256, 0, 256, 9), (1, 32, 32, 256, 0, 256, 9),
          (8, 4, 4, 256, 0, 256, 9), (8, 8, 8, 256, 0, 256, 9), (8, 4, 4, 256, 256, 256, 9), (1, 16, 16, 256, 256, 256, 1)]
if __name__ == "__main__":
    for s in SHAPES:
        bench_conv.run(s, iters=10, check="--check" in sys.argv)
