"""HBM traffic of the k-strongest kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> JSON.
usage: pmc_k1_traffic_report.py <fetch.db> <write.db> <scans_per_launch> <out.json>"""
import json, sqlite3, sys
ALGO = 400 * 3360 + 400 * 12 * 4


def avg(db, counter):
    c = sqlite3.connect(db)
    rows = [v for (v,) in c.execute("select value from counters_collection where kernel_name like '%kstrongest%' and counter_name = ?", (counter,))]
    return sum(rows) / len(rows), len(rows)


fetch, nf = avg(sys.argv[1], "FETCH_SIZE")
write, nw = avg(sys.argv[2], "WRITE_SIZE")
n = int(sys.argv[3])
fb = fetch * 1024 * 2  # KB units; gfx950 reports 1/2 of a wide coalesced stream (MI355X_MICROARCH.md, HBM section)
wb = write * 1024
out = {
    "kernel": "kstrongest_kernel<4,7>", "scans_per_launch": n,
    "counters": {"FETCH_SIZE": {"dispatches": nf, "avg_KB": fetch}, "WRITE_SIZE": {"dispatches": nw, "avg_KB": write}},
    "fetch_bytes_corrected_per_launch": fb, "write_bytes_per_launch": wb,
    "hbm_bytes_per_scan": (fb + wb) / n, "algorithmic_bytes_per_scan": ALGO,
    "traffic_over_algorithmic": (fb + wb) / n / ALGO,
    "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/refresh_profiles.sh); "
              "KB units x1024; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a 16 B/lane coalesced stream); "
              "WRITE_SIZE uncalibrated; average over uniform-random and synthetic-world sweeps",
}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
