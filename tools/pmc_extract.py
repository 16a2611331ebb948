"""Per-kernel PMC summaries of bench.py's roofline micro-benchmark loops from the rocpd databases of separate `rocprofv3
--kernel-trace --pmc ...` passes over the SAME bench command (tools/profile.sh).

bench.py drops a marker kernel (at::cuda::spin_kernel) around every timed window, one before its measurement scaffolding and one
in front of every micro-benchmark loop of bench_kernels.roofline_section (`roofline.micro_order` of the bench line).  The
dispatches of loop j are those between the j-th micro marker and the next marker; a loop's first launch is its warm-up.

usage: pmc_extract.py <bench log of the FETCH pass> <db FETCH_SIZE> <db WRITE_SIZE> <db SQ1> <db SQ2> <out dir>
writes <out dir>/pmc_<kernel>.json for the kernels below."""
import glob
import json
import os
import sqlite3
import sys

# micro loop -> kernels summarised from it (substring of the kernel name); a list = several kernels summed into one row
TARGETS = {
    "bwd": [("raster_ges_bwd_strip_kernel", ["raster_ges_bwd_strip_kernel"]), ("raster_ges_bwd_gs_kernel", ["raster_ges_bwd_gs_kernel"])],
    "fwd": [("raster_ges_fwd_pk_kernel", ["raster_ges_fwd_pk_kernel"])],
    "pre": [("preprocess_fwd_kernel", ["preprocess_fwd_kernel"])],
    "render": [("binning (sb_scan_kernel + sb_scatter_kernel)", ["sb_scan_kernel", "sb_scatter_kernel"])],
    "pbwd": [("preprocess_bwd_kernel", ["preprocess_bwd_kernel"])],
    "integrate": [("integrate_kernel", ["integrate_kernel"])],
    "raycast": [("raycast_kernel", ["raycast_kernel"])],
    "freeview": [("colour_kernel", ["colour_kernel"]), ("expected_depths_partial_kernel", ["expected_depths_partial_kernel"]),
                 ("raycast_kernel<false> (free views)", ["raycast_kernel"])],
    "tracked": [("track_eval_poll_kernel", ["track_eval_poll_kernel"])],
}


def pmc_file_name(kernel):
    """profiles/pmc_<...>.json of a roofline row (bench_kernels._pmc uses the same rule)"""
    import re
    if kernel.startswith("binning"):
        return "pmc_binning.json"
    return "pmc_%s.json" % re.sub(r"[^A-Za-z0-9_]+", "_", kernel.split(" (")[0]).strip("_")


def build_commit():
    """the commit the profiled library was built from: `git rev-parse HEAD > .build_commit` before the gpurun call (the GPU box's
    snapshot has no .git); "unknown" when the file is missing"""
    try:
        return open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".build_commit")).read().strip()
    except OSError:
        return "unknown"


def bench_line(log):
    for l in reversed(open(log).read().splitlines()):
        if l.startswith('{"metric"'):
            return json.loads(l)
    raise SystemExit("no bench line in %s" % log)


def dispatches(db_dir, counter):
    """[(kernel name, value)] of every dispatch in dispatch order"""
    path = glob.glob(os.path.join(db_dir, "**", "*.db"), recursive=True)[0]
    db = sqlite3.connect(path)
    return db.execute("select c.kernel_name, c.value from counters_collection c where c.counter_name=? order by c.dispatch_id",
                      (counter,)).fetchall()


def loops(rows, order, n_window_markers):
    """{loop name: [(kernel, value)] without the markers}"""
    marks = [i for i, (n, _) in enumerate(rows) if "spin_kernel" in n]
    base = n_window_markers + 1   # window markers, then the scaffolding marker
    out = {}
    for j, name in enumerate(order):
        if base + j >= len(marks):
            break
        lo = marks[base + j]
        hi = marks[base + j + 1] if base + j + 1 < len(marks) else len(rows)
        out[name] = rows[lo + 1:hi]
    return out


def per_launch(seg, names):
    """mean over the loop's launches (first = warm-up dropped) of the summed counter of the named kernels"""
    per = {}
    for nm in names:
        vals = [v for k, v in seg if nm in k]
        if not vals:
            return None, 0
        per[nm] = vals[1:] if len(vals) > 1 else vals
    n = min(len(v) for v in per.values())
    return sum(sum(v[:n]) / n for v in per.values()), n


def main():
    log, d_fetch, d_write, d_sq1, d_sq2, out_dir = sys.argv[1:7]
    line = bench_line(log)
    roof = line["roofline"]
    order = roof["micro_order"]
    cfg = line["config"]
    n_window_markers = 2 * cfg.get("marker_windows_per_schedule", cfg["windows"]) * len(cfg["schedules"])   # (+ the in-loop window's pair)
    live = {r["kernel"]: r for r in roof["kernels"]}
    sets = {"FETCH_SIZE": d_fetch, "WRITE_SIZE": d_write}
    sq1 = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES")
    sq2 = ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT")
    seg = {c: loops(dispatches(d, c), order, n_window_markers) for c, d in sets.items()}
    for c in sq1:
        seg[c] = loops(dispatches(d_sq1, c), order, n_window_markers)
    for c in sq2:
        seg[c] = loops(dispatches(d_sq2, c), order, n_window_markers)
    os.makedirs(out_dir, exist_ok=True)
    for loop, targets in TARGETS.items():
        for kname, parts in targets:
            f, n = per_launch(seg["FETCH_SIZE"].get(loop, []), parts)
            if f is None:
                continue
            w, _ = per_launch(seg["WRITE_SIZE"].get(loop, []), parts)
            fb, wb = f * 1024.0, (w or 0.0) * 1024.0   # rocprofv3 reports both in KiB
            rec = {"kernel": kname, "loop": loop, "launches": n, "library_commit": build_commit(),
                   "command": "bench.py as tools/profile.sh runs it; counters of the roofline micro-benchmark loop `%s`" % loop,
                   # FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies 128-B requests at 64 B on gfx950 (MI355X guide, HBM section): doubled
                   "fetch_bytes_raw": fb, "fetch_bytes_x2": 2 * fb, "write_bytes": wb, "hbm_bytes_per_launch": 2 * fb + wb,
                   "units": roof["units"], "algorithmic_bytes": (live.get(kname) or {}).get("algorithmic_bytes"),
                   "live_avg_us_of_the_profiled_run": (live.get(kname) or {}).get("avg_us")}
            sq = {}
            for c in sq1 + sq2:
                v, _ = per_launch(seg[c].get(loop, []), parts)
                if v is not None:
                    sq[c] = v
            if sq.get("SQ_WAVE_CYCLES"):
                sq["wait_any_frac"] = sq.get("SQ_WAIT_ANY", 0.0) / sq["SQ_WAVE_CYCLES"]
            if sq.get("SQ_ACTIVE_INST_LDS"):
                sq["lds_conflict_per_active_lds"] = sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / sq["SQ_ACTIVE_INST_LDS"]
            rec["sq"] = sq
            fn = pmc_file_name(kname)
            json.dump(rec, open(os.path.join(out_dir, fn), "w"), indent=1)
            print(fn, json.dumps({k: rec[k] for k in ("launches", "hbm_bytes_per_launch", "algorithmic_bytes")}), json.dumps(sq)[:300])


if __name__ == "__main__":
    main()
