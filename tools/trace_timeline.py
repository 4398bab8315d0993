"""Reads a rocprofv3 --kernel-trace CSV of tools/step_time.py and prints, for the last training step in it, how the
GPU's time divides: which kernel classes were resident when (union of intervals), per-queue busy time, and the
longest stretches with no blend kernel resident.
    python tools/trace_timeline.py gpurun_out/r2/trace/step_kernel_trace.csv"""
import csv
import sys
from collections import defaultdict


def short(name):
    for key, s in (("blend_fwd", "K6"), ("blend_bwd", "K7"), ("preprocess", "K1"), ("param_bwd", "K8"), ("radix_scatter", "scatter"),
                   ("radix_hist", "hist"), ("radix_scan", "rscan"), ("culled_pairs", "pairs"), ("count_", "count"),
                   ("sorted_", "scan"), ("scan_blocks", "scan"), ("ranges", "ranges"), ("image_loss", "loss"),
                   ("tile_visited", "order"), ("tile_order", "order"), ("unpack", "unpack"), ("pack", "pack")):
        if key in name:
            return s
    return "torch"


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]))
    rows.sort()
    # steps: split at K8 launches that are followed by a K1
    k1 = [i for i, r in enumerate(rows) if r[2] == "K1"]
    # a step begins with the first K1 after a K8
    starts = [i for j, i in enumerate(k1) if j == 0 or any(r[2] == "K8" for r in rows[k1[j - 1]:i])]
    a, b = starts[-2], starts[-1]
    step = rows[a:b]
    t0 = step[0][0]
    t1 = max(r[1] for r in step)
    print(f"step span {1e-6 * (t1 - t0):.3f} ms, {len(step)} kernels")
    cls = defaultdict(list)
    for s, e, n, q in step:
        cls[n].append((s, e))
    blend = cls["K6"] + cls["K7"]
    every = [(s, e) for s, e, _, _ in step]
    print(f"GPU busy (any kernel) {1e-6 * union(every):.3f} ms; a blend kernel resident {1e-6 * union(blend):.3f} ms; "
          f"K7 resident {1e-6 * union(cls['K7']):.3f}; K6 resident {1e-6 * union(cls['K6']):.3f}")
    for n in sorted(cls, key=lambda n: -sum(e - s for s, e in cls[n])):
        iv = cls[n]
        print(f"  {n:8s} launches {len(iv):4d}  sum {1e-6 * sum(e - s for s, e in iv):7.3f} ms  union {1e-6 * union(iv):7.3f} ms  "
              f"mean {1e-3 * sum(e - s for s, e in iv) / len(iv):8.1f} us  max {1e-3 * max(e - s for s, e in iv):8.1f} us")
    qs = defaultdict(list)
    for s, e, n, q in step:
        qs[q].append((s, e))
    for q in sorted(qs):
        print(f"  queue {q}: busy {1e-6 * union(qs[q]):.3f} ms of {1e-6 * (t1 - t0):.3f}, {len(qs[q])} kernels")
    # number of blend kernels resident over time, sampled
    import bisect
    ev = sorted([(s, 1) for s, e in blend] + [(e, -1) for s, e in blend])
    hist = defaultdict(int)
    cur, last = 0, t0
    for t, d in ev:
        hist[cur] += t - last
        last = t
        cur += d
    hist[cur] += t1 - last
    print("  time with k blend kernels resident:", {k: round(1e-6 * v, 3) for k, v in sorted(hist.items())})
    if len(sys.argv) > 2:
        for s, e, n, q in step:
            print(f"{1e-3 * (s - t0):9.1f} {1e-3 * (e - t0):9.1f} {1e-3 * (e - s):8.1f} q{q} {n}")


if __name__ == "__main__":
    main(sys.argv[1])
