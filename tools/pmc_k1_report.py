import sqlite3, collections, sys
for dd in (1, 2, 3, 0):
    c = sqlite3.connect('gpurun_out/pmc_k1_dbg%d/k1_results.db' % dd)
    rows = list(c.execute("select counter_name, dispatch_id, value from counters_collection where kernel_name like '%kstrongest%' order by dispatch_id"))
    d = collections.defaultdict(list)
    for n, i, v in rows:
        d[n].append((i, v))
    print("DBG", dd)
    for n, l in sorted(d.items()):
        l.sort(); half = len(l) // 2
        u = sum(v for _, v in l[:half]) / half; w = sum(v for _, v in l[half:]) / (len(l) - half)
        print("  %-20s per row: uniform %.1f  world %.1f" % (n, u / 102400, w / 102400))
