#!/usr/bin/env python
"""Per-wave timeline of k_icp_step from a -DSRRG2_TIMELINE build (SRRG2_AMD_TIMELINE=file).

usage: timeline.py FILE [iteration ...]
Stamps are wall_clock64() ticks (100 MHz); printed as microseconds since the earliest stamp of the launch:
median / p90 / max over the waves, and the median time spent between consecutive stamps."""
import sys

import numpy as np

NAMES = ["state+T", "p+prior", "row ranges", "first W0", "phase1 done", "open lanes", "finish_point", "reduce"]


def main():
    raw = np.fromfile(sys.argv[1], dtype=np.uint64)
    nw = int(raw[0])
    st = raw[1:].reshape(32, nw, 16).astype(np.int64)
    its = [int(a) for a in sys.argv[2:]] or [0, 1, 2, 5]
    for it in its:
        t = st[it][:, :8]
        ok = t[:, 0] > 0
        if not ok.any():
            continue
        t = t[ok].astype(np.float64)
        t0 = t[:, 0].min()
        # stamps that were skipped by a wave (e.g. inactive) stay 0: carry the previous one forward
        for k in range(1, 8):
            t[:, k] = np.where(t[:, k] > 0, t[:, k], t[:, k - 1])
        rel = (t - t0) / 100.0
        print("iteration %d: %d waves, launch span %.2f us" % (it, len(t), rel[:, 7].max()))
        cen = st[it][ok][:, 8:14].sum(axis=0)
        print("  census: active %d | skipped with neighbour %d, certified unmatched %d | stragglers %d | open near %d far %d" % tuple(cen))
        prev = np.zeros(len(t))
        for k in range(8):
            d = rel[:, k] - (rel[:, k - 1] if k else 0)
            print("  %-13s at med %6.2f p90 %6.2f max %6.2f | step med %5.2f p90 %5.2f max %5.2f" %
                  (NAMES[k], np.median(rel[:, k]), np.percentile(rel[:, k], 90), rel[:, k].max(),
                   np.median(d), np.percentile(d, 90), d.max()))


if __name__ == "__main__":
    main()
