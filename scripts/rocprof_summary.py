"""Turn rocprofv3 rocpd databases (gpurun_out/...) into the small text/JSON summaries committed under profiles/.

  python scripts/rocprof_summary.py stats  <results.db>  <out.txt>
  python scripts/rocprof_summary.py pmc    <fetch.db> <write.db> <out.json>
  python scripts/rocprof_summary.py pmc_step <fetch.db> <write.db> <steps> <out.json>          (bytes per kernel and STEP: the training leg's `traffic`)
  python scripts/rocprof_summary.py sq     <results.db> <kernel regex> <out.txt> [append]
  python scripts/rocprof_summary.py timeline <results.db> <first-kernel substring> <out.txt>     (dispatches of the LAST call that starts with that kernel)

PMC post-processing follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB, collected in
separate passes; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced stream, so reads are
doubled.  WRITE_SIZE is uncalibrated (taken as reported)."""
import json
import sqlite3
import sys


def _median(v):
    v = sorted(v)
    n = len(v)
    return 0.0 if n == 0 else (v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2]))


def stats(db, out):
    """Per kernel symbol AND launch grid (a 1-tile calibration launch must not dilute a batch-32 average): calls, total, average,
    median, min, max.  Persistent kernels launch the same grid whatever the batch, so the profiling legs also run with CERB_AUTO_PRECISION=0
    (scripts/profile_r06.sh): then every launch of a symbol in a `--mode batch` run is a batch launch and avg == median up to noise."""
    c = sqlite3.connect(db)
    groups = {}
    total = 0
    for n, gx, gy, gz, a, b in c.execute("select name, grid_x, grid_y, grid_z, start, end from kernels"):
        groups.setdefault((n, gx, gy, gz), []).append((b - a) / 1e3)
        total += b - a
    rows = sorted(((k, v) for k, v in groups.items()), key=lambda kv: -sum(kv[1]))
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace summary, one row per kernel symbol and launch grid (durations in microseconds)\n")
        f.write("%-84s %-16s %7s %13s %11s %11s %11s %11s %7s\n" % ("kernel", "grid", "calls", "total_us", "avg_us", "median_us", "min_us", "max_us", "pct"))
        for (n, gx, gy, gz), d in rows:
            f.write("%-84s %-16s %7d %13.1f %11.3f %11.3f %11.3f %11.3f %7.2f\n" % (n[:84], "%dx%dx%d" % (gx, gy, gz), len(d), sum(d), sum(d) / len(d), _median(d), min(d), max(d),
                                                                                  100.0 * sum(d) * 1e3 / max(total, 1)))
        # register / LDS footprint per kernel
        f.write("\n# per-kernel launch footprint (first dispatch of each symbol)\n")
        f.write("%-90s %6s %6s %8s %10s\n" % ("kernel", "vgpr", "sgpr", "lds", "wg"))
        seen = set()
        for n, v, s, lds, wg in c.execute("select name, vgpr_count, sgpr_count, lds_size, workgroup_x from kernels"):
            if n in seen:
                continue
            seen.add(n)
            f.write("%-90s %6s %6s %8s %10s\n" % (n[:90], v, s, lds, wg))
    print("wrote", out)


def pmc(fetch_db, write_db, out):
    res = {}
    for key, db in (("FETCH_SIZE", fetch_db), ("WRITE_SIZE", write_db)):
        c = sqlite3.connect(db)
        for name, n, avg in c.execute(
            "select kernel_name, count(*), avg(value) from counters_collection where counter_name=? group by kernel_name", (key,)
        ):
            if "at::native" in name:
                continue
            r = res.setdefault(name, {})
            r[key + "_KiB_avg_per_launch"] = avg
            r[key + "_launches"] = n
    for name, r in res.items():
        f = r.get("FETCH_SIZE_KiB_avg_per_launch", 0.0) or 0.0
        w = r.get("WRITE_SIZE_KiB_avg_per_launch", 0.0) or 0.0
        r["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0  # gfx950: FETCH_SIZE counts 64 B per 128-B request
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)


def pmc_step(fetch_db, write_db, steps, out):
    """HBM bytes per kernel symbol and per STEP of a bench run profiled over `steps` steps in total (warm-up + timed + the profiled one):
    {symbol: {launches_per_step, hbm_bytes_per_launch, hbm_bytes_per_step}} -- what bench.py --mode train reads as `traffic`."""
    res = {}
    for key, db in (("FETCH_SIZE", fetch_db), ("WRITE_SIZE", write_db)):
        c = sqlite3.connect(db)
        for name, n, tot in c.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (key,)):
            if "at::native" in name:
                continue
            r = res.setdefault(name, {})
            r[key + "_KiB_total"] = tot
            r[key + "_launches"] = n
    steps = float(steps)
    for name, r in res.items():
        f, w = r.get("FETCH_SIZE_KiB_total", 0.0) or 0.0, r.get("WRITE_SIZE_KiB_total", 0.0) or 0.0
        n = max(r.get("FETCH_SIZE_launches", 0), r.get("WRITE_SIZE_launches", 0), 1)
        r["launches_per_step"] = n / steps
        r["hbm_bytes_per_step"] = (2.0 * f + w) * 1024.0 / steps  # gfx950: FETCH_SIZE counts 64 B per 128-B request
        r["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0 / n
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)


def pmc_table(pmc_json, calls, px, algorithmic_b_px, out):
    """Bytes per pixel and pass of one call: the pmc() summary of a run that made `calls` calls over `px` pixels each, as a table
    sorted by traffic, beside the algorithmic bytes per pixel of the whole call."""
    d = json.load(open(pmc_json))
    rows = []
    for k, v in d.items():
        n = max(v.get("FETCH_SIZE_launches", 0), v.get("WRITE_SIZE_launches", 0)) / float(calls)
        b = v.get("hbm_bytes_per_launch", 0.0) * n
        rd = 2.0 * (v.get("FETCH_SIZE_KiB_avg_per_launch", 0.0) or 0.0) * 1024.0 * n
        rows.append((b, rd, n, k))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    with open(out, "w") as f:
        f.write("# HBM bytes per pixel and pass of ONE call (PMC FETCH_SIZE x 2 + WRITE_SIZE, separate passes; %d pixels per call, %d calls profiled)\n" % (px, calls))
        f.write("# whole call: %.2f B/px measured against %.1f B/px algorithmic = %.2fx\n" % (tot / px, algorithmic_b_px, tot / px / algorithmic_b_px))
        f.write("%8s %8s %8s %7s  %s\n" % ("B/px", "read", "written", "n/call", "kernel"))
        for b, rd, n, k in rows:
            if b <= 0:
                continue
            f.write("%8.2f %8.2f %8.2f %7.1f  %s\n" % (b / px, rd / px, (b - rd) / px, n, k[:110]))
    print("wrote", out)


def sq(db, pattern, out, append=False):
    """SQ / TCP / TCC counters per kernel symbol matching `pattern` (regex) and launch grid: launches, average, min, max per launch."""
    import re

    c = sqlite3.connect(db)
    rows = list(c.execute("select kernel_name, grid_size, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                          "group by kernel_name, grid_size, counter_name"))
    d = {}
    for k, g, n, cnt, v, lo, hi in rows:
        if re.search(pattern, k):
            d.setdefault("%s  [grid %d]" % (k[:100], g), {})[n] = (cnt, v, lo, hi)
    with open(out, "a" if append else "w") as f:
        for k, m in sorted(d.items()):
            f.write(k + "\n")
            for n, (cnt, v, lo, hi) in sorted(m.items()):
                f.write("   %-36s n=%5d avg=%.5g min=%.5g max=%.5g\n" % (n, cnt, v, lo, hi))
    print("wrote", out)


def timeline(db, first, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end, queue_id from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if first in r[0]]
    rows = rows[idx[-1]:]
    t0 = rows[0][1]
    with open(out, "w") as f:
        f.write("# dispatches of the last call (rocprofv3 --kernel-trace): start offset, duration (ms), queue, kernel; wall %.3f ms, sum of durations %.3f ms\n"
                % ((max(r[2] for r in rows) - t0) / 1e6, sum(r[2] - r[1] for r in rows) / 1e6))
        for n, a, b, q in rows:
            f.write("%8.3f %8.3f q%-3s %s\n" % ((a - t0) / 1e6, (b - a) / 1e6, q, n[:70]))
    print("wrote", out)


def fills(db, first, out):
    """The fillBuffer / copyBuffer dispatches of the last step (from the last kernel whose name contains `first`): how many, how long, and which
    kernel follows each -- finds the zero fills a step still issues and whose they are."""
    import collections

    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end, grid_x from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if first in r[0]]
    rows = rows[idx[-1]:]
    cnt, tot, grid = collections.Counter(), collections.Counter(), collections.Counter()
    for i, r in enumerate(rows):
        if "fillBuffer" in r[0] or "copyBuffer" in r[0]:
            k = (r[0][13:24], rows[i + 1][0][:70] if i + 1 < len(rows) else "")
            cnt[k] += 1
            tot[k] += (r[2] - r[1]) / 1e3
            grid[k] += r[3]
    with open(out, "w") as f:
        f.write("# fills / copies of the last step (%d dispatches, %.1f ms wall): count, total us, mean grid_x, kind, the kernel that follows\n" % (len(rows), (rows[-1][2] - rows[0][1]) / 1e6))
        for k, v in sorted(cnt.items(), key=lambda kv: -tot[kv[0]]):
            f.write("%4d %9.1f %10d  %-12s %s\n" % (v, tot[k], grid[k] // v, k[0], k[1]))
        f.write("# total: %d dispatches, %.1f us\n" % (sum(cnt.values()), sum(tot.values())))
    print("wrote", out)


def busy(db, out):
    """Device occupancy of the last second of a trace: union of the kernel intervals / wall, the overlap between queues, the largest gaps."""
    c = sqlite3.connect(db)
    rows = sorted(c.execute("select start, end, queue_id, name from kernels"))
    t1 = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= t1 - 1.0e9]
    t0 = rows[0][0]
    wall = (t1 - t0) / 1e6
    ev = sorted([(r[0], 1) for r in rows] + [(r[1], -1) for r in rows])
    depth, last, by_depth = 0, t0, {}
    for t, d in ev:
        by_depth[depth] = by_depth.get(depth, 0) + (t - last)
        depth += d
        last = t
    gaps, cur_end = [], rows[0][1]
    for a, b, q, n in rows[1:]:
        if a > cur_end:
            gaps.append((a - cur_end, n))
        cur_end = max(cur_end, b)
    gaps.sort(reverse=True)
    with open(out, "w") as f:
        f.write("# last %.1f ms of the trace: time with k kernels in flight (ms): %s\n" % (wall, {k: round(v / 1e6, 2) for k, v in sorted(by_depth.items())}))
        f.write("# idle (0 in flight) %.2f %% ; sum of kernel durations / wall = %.3f\n" % (100.0 * by_depth.get(0, 0) / (t1 - t0), sum(r[1] - r[0] for r in rows) / (t1 - t0)))
        f.write("# largest gaps (us) and the kernel that ended them:\n")
        for g, n in gaps[:12]:
            f.write("%8.1f  %s\n" % (g / 1e3, n[:80]))
        per = {}
        for a, b, q, n in rows:
            per[q] = per.get(q, 0) + (b - a)
        f.write("# busy per queue (ms): %s\n" % {q: round(v / 1e6, 1) for q, v in per.items()})
    print("wrote", out)


if __name__ == "__main__":
    if sys.argv[1] == "busy":
        busy(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "fills":
        fills(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmc_step":
        pmc_step(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
    elif sys.argv[1] == "pmc_table":
        pmc_table(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), sys.argv[6])
    elif sys.argv[1] == "sq":
        sq(sys.argv[2], sys.argv[3], sys.argv[4], len(sys.argv) > 5)
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
