#!/usr/bin/env python3
"""Headless equivalent of `python lqr_tracking.py` (lqr_tracking.py:321-342) on the HIP library: loads
Data/xx_star.npy, Data/uu_star.npy, tracks them from a perturbed start, saves the tracked trajectory.

    python examples/run_lqr_tracking.py [--data Data]
"""
import argparse
import os

import numpy as np

import _common  # noqa: F401
from lqr_tracking import lqr_tracking


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="Data")
    a = ap.parse_args()
    xx_opt = np.load(os.path.join(a.data, "xx_star.npy"))
    uu_opt = np.load(os.path.join(a.data, "uu_star.npy"))
    tt = np.linspace(0, 1, xx_opt.shape[1])
    xx_lqr, uu_lqr = lqr_tracking(xx_opt, uu_opt, tt)
    np.save(os.path.join(a.data, "xx_lqr.npy"), xx_lqr)
    np.save(os.path.join(a.data, "uu_lqr.npy"), uu_lqr)
    print("initial offset %s\nfinal tracking error %s" % (xx_lqr[:, 0] - xx_opt[:, 0], xx_lqr[:, -1] - xx_opt[:, -1]))


if __name__ == "__main__":
    main()
