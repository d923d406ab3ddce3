"""Prints the last N kernel dispatches of a rocprofv3 (rocpd sqlite) kernel trace: start/end (us), queue, stream."""
import sqlite3, sys
db = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
c = sqlite3.connect(db)
views = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
v = "kernels" if "kernels" in views else [x for x in views if "kernel" in x.lower()][0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % v)]
print("view", v, cols)
want = [x for x in ("name", "start", "end", "queue_id", "stream_id", "queue", "stream") if x in cols]
rows = list(c.execute("select %s from %s order by start" % (",".join(want), v)))
t0 = rows[-N][want.index("start")]
for r in rows[-N:]:
    d = dict(zip(want, r))
    nm = d["name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]
    print("%-28s %9.1f %9.1f  dur %7.1f  q=%s s=%s" % (nm, (d["start"] - t0) / 1e3, (d["end"] - t0) / 1e3, (d["end"] - d["start"]) / 1e3,
                                                 d.get("queue_id", d.get("queue")), d.get("stream_id", d.get("stream"))))
print("stream->queue:", sorted(set((r[want.index("stream_id")], r[want.index("queue_id")]) for r in rows)))
