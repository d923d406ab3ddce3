"""Dumps the per-kernel statistics of a rocprofv3 (rocpd sqlite) result as a markdown table."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["| kernel | calls | total (us) | avg (us) | % |", "|---|---|---|---|---|"]
    for name, calls, tot, avg, pct in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        lines.append("| %s | %d | %.1f | %.2f | %.1f |" % (short, calls, tot, avg, pct))
    try:
        pm = list(c.execute("select * from counters_collection limit 0"))
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        if "counter_name" in cols:
            q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                 "group by kernel_name, counter_name order by kernel_name")
            pmc = list(c.execute(q))
            import os
            last = int(os.environ.get("ROCPD_LAST", "0"))  # average over the last N dispatches of every kernel only (a warm-up in front)
            if last:
                from collections import defaultdict
                acc = defaultdict(list)
                key = next((x for x in ("dispatch_id", "start", "id", "event_id") if x in cols), None)
                for k, n, v in c.execute("select kernel_name, counter_name, value from counters_collection" + (" order by " + key if key else "")):
                    acc[(k, n)].append(v)
                pmc = [(k, n, sum(v[-last:]) / len(v[-last:]), len(v[-last:])) for (k, n), v in sorted(acc.items())]
            if pmc:
                lines += ["", "| kernel | counter | avg per dispatch | dispatches |", "|---|---|---|---|"]
                for k, n, v, cnt in pmc:
                    short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                    lines.append("| %s | %s | %.6g | %d |" % (short, n, v, cnt))
    except sqlite3.Error as e:
        lines.append("(counters: %s)" % e)
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
