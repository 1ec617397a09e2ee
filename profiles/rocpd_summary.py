"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace into the `--stats`-style table committed under profiles/.
usage: python profiles/rocpd_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    span = con.execute("select min(start), max(end) from kernels").fetchone()
    lines = ["# rocprofv3 --kernel-trace --stats summary (%s)" % db.split("/")[-1], "",
             "GPU kernel time %.1f ms over a %.1f ms span (%d dispatches)" % (total / 1e6, (span[1] - span[0]) / 1e6,
                                                                              sum(r[1] for r in rows)), "",
             "| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds | scratch |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows[:40]:
        name = r[0]
        if len(name) > 90:
            name = name[:87] + "..."
        lines.append("| %s | %d | %.2f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s | %s |" % (
            name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10]))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("--pmc", "--layers", "--traffic")):
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)


def pmc(db, out=None, like="%"):
    """per-kernel average of every collected counter (rocprofv3 --pmc run): python rocpd_summary.py --pmc db out.md"""
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                       "where kernel_name like ? group by kernel_name, counter_name order by 5 desc", (like,)).fetchall()
    lines = ["# rocprofv3 --pmc summary (%s)" % db.split("/")[-1], "", "| kernel | counter | dispatches | avg per dispatch | sum |",
             "|---|---|---|---|---|"]
    for r in rows[:60]:
        lines.append("| %s | %s | %d | %.6g | %.6g |" % (r[0][:70], r[1], r[2], r[3], r[4]))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--pmc":
    pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)


def layers(db, out=None, n=106):
    """the last n convolution dispatches in launch order (one forward of backbone + scorer): --layers db out.md [n]"""
    con = sqlite3.connect(db)
    rows = con.execute("select name, grid_x, grid_y, duration from kernels where name like '%spconv%' order by start").fetchall()[-n:]
    lines = ["# convolution launches of one step, in order (%s)" % db.split("/")[-1], "", "| # | kernel | blocks x col groups | us |", "|---|---|---|---|"]
    tot = 0
    for i, (name, gx, gy, dur) in enumerate(rows):
        tot += dur
        lines.append("| %d | %s | %d x %d | %.1f |" % (i, name[5:32], gx // 256, gy, dur / 1e3))
    lines.append("")
    lines.append("total %.2f ms" % (tot / 1e6))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--layers":
    layers(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None, int(sys.argv[4]) if len(sys.argv) > 4 else 106)


def traffic(db_fetch, db_write, out):
    """HBM traffic per launch of the dominant kernel family (k_spconv_fwd*), from the two PMC passes.
    FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B: the 128-byte requests of wide,
    coalesced streaming loads are tallied at 64 bytes (MI355X_MICROARCH.md, HBM section), while 64-byte row gathers -- the
    A operand of the convolution -- are counted in full.  Calibrated on this access pattern in
    profiles/r02_fetch_calibration.md (profiles/calibrate_fetch.sh): known / raw = 2.000 for streaming reads, 1.059 for
    random 64-byte row gathers, 1.000 for WRITE_SIZE.  So the raw counter is stored here and bench.py adds half of the
    bytes the kernel streams with wide loads (its kernel map, known exactly per launch):
        traffic = WRITE_SIZE + FETCH_SIZE_raw + 0.5 * map_bytes      (lower bound: raw, upper bound: 2 x raw)."""
    import json
    res = {"classes": {}}
    # round 5: two kernel families -- k_spconv_fwd3 (fp32 MFMAs: the <= 32-channel layers) and k_spconv_x3 (split operands on the
    # bf16 pipe: the wide layers); totals over both, and per family so that bench.py can set each beside its algorithmic bytes
    # round 6: k_spconv_x3f (the split-operand arithmetic with full-line gathers) as a class of its own: its gathers are 128-byte
    # requests, which the counter tallies at 64 bytes -- bench.py doubles that class' raw counter instead of adding back map bytes
    fams = {"fwd3": "%k_spconv_fwd%", "x3": "%k_spconv_x3<%", "x3f": "%k_spconv_x3f<%"}
    tot = {"fetch": [0, 0.0], "write": [0, 0.0]}
    for fam, like in fams.items():
        ent = {}
        for name, db, counter in (("fetch", db_fetch, "FETCH_SIZE"), ("write", db_write, "WRITE_SIZE")):
            con = sqlite3.connect(db)
            n, kib = con.execute("select count(*), sum(value) from counters_collection where kernel_name like ? "
                                 "and counter_name = ?", (like, counter)).fetchone()
            n, kib = int(n or 0), float(kib or 0.0)
            ent[name + "_launches"] = n
            ent[name + ("_raw_bytes_per_launch" if name == "fetch" else "_bytes_per_launch")] = 1024.0 * kib / max(n, 1)
            tot[name][0] += n
            tot[name][1] += kib
        res["classes"][fam] = ent
    res["fetch_launches"], res["fetch_KiB_total"] = tot["fetch"]
    res["write_launches"], res["write_KiB_total"] = tot["write"]
    fetch_raw = 1024.0 * res["fetch_KiB_total"] / max(res["fetch_launches"], 1)
    write_b = 1024.0 * res["write_KiB_total"] / max(res["write_launches"], 1)
    res.update({"kernel": "k_spconv_fwd3 + k_spconv_x3 + k_spconv_x3f", "fetch_raw_bytes_per_launch": fetch_raw, "write_bytes_per_launch": write_b,
                "calibration": {"streaming_known_over_raw": 2.0, "gather64_known_over_raw": 1.059, "write_known_over_raw": 1.0,
                                "source": "profiles/r05_fetch_calibration.md (64 .. 384-byte rows) / r02_fetch_calibration.md",
                                "rule": "FETCH_SIZE tallies 64 bytes per request; a request is <= 128 bytes: full-line (128-byte) "
                                        "requests count half, 64-byte requests -- every row gather of the convolutions, whatever "
                                        "the row width: a lane quad reads one 64-byte piece -- count in full"},
                "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 1 --warmup 1` (all launches of "
                        "all passes averaged); raw counters, see bench.py for the correction"})
    open(out, "w").write(json.dumps(res, indent=1) + "\n")
    print(res)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--traffic":
    traffic(sys.argv[2], sys.argv[3], sys.argv[4])
