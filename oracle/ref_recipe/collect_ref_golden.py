"""oracle/_ref/ref_outputs.bin (written by dump_ref_golden on a ROS box) -> tests/golden/ref_golden.npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import refio  # noqa: E402


def main():
    out = refio.read(os.path.join(ROOT, "oracle", "_ref", "ref_outputs.bin"))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_golden.npz"), **out)
    print("wrote tests/golden/ref_golden.npz with", len(out), "arrays:", ", ".join(sorted(out)[:8]), "...")


if __name__ == "__main__":
    main()
