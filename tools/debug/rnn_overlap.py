#!/usr/bin/env python3
"""Do the GRU step launches of different layers overlap on the GPU?  Reads a rocprofv3 kernel trace of tools/rnn_bench.py:
per kernel name: launches, sum of durations, union of busy intervals, queues used."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    key = "gru_fwd_step" if "gru_fwd_step" in n else "gru_bwd_step" if "gru_bwd_step" in n else None
    if key:
        by[key].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
for key, iv in by.items():
    iv.sort()
    tot = sum(e - s for s, e, _ in iv)
    union, cur_s, cur_e = 0, None, None
    for s, e, _ in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    gaps = sorted(iv[i + 1][0] - iv[i][1] for i in range(len(iv) - 1))
    print("%s: %d launches, sum of durations %.2f ms, union of busy time %.2f ms, queues %s, median start-to-previous-end gap %.2f us, median duration %.2f us"
          % (key, len(iv), tot / 1e6, union / 1e6, sorted(set(q for _, _, q in iv)), gaps[len(gaps) // 2] / 1e3, sorted(e - s for s, e, _ in iv)[len(iv) // 2] / 1e3))
