"""Wide parity sweep from the command line (the implementation and its gates live in tests/: tests/parity_sweep.py,
tests/test_gpu_sweep.py).  Writes one JSON line (kept under profiles/).

    python tools/parity_sweep.py [B=4096] [iters=12] [dist=random|perturbed] [problem=step|acro]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from aircraftoptimalcontrol_amd import batch as aoc, problems
import parity_sweep


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dist = sys.argv[3] if len(sys.argv) > 3 else "random"
    prob = sys.argv[4] if len(sys.argv) > 4 else "step"
    out = parity_sweep.sweep(aoc, problems, B, n_it, dist, prob, log=lambda r: print(json.dumps(r), file=sys.stderr))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
