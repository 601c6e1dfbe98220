#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into small text files under profiles/.

usage: summarize_prof.py <round-tag> <kernel-trace.db> [<pmc FETCH_SIZE .db> <pmc WRITE_SIZE .db>]
"""
import collections
import sqlite3
import sys

CONV_SLOTS = {  # launch order of the tensor-product convs inside one dp_forward, per kernel instantiation
    "tp_conv_kernel<0>": ["lig-lig", "phore->lig", "phore->lig(norm)", "phore-phore", "lig->phore", "lig->phore(norm)"],
    "tp_conv_kernel<1>": ["lig-lig", "phore->lig", "phore->lig(norm)", "phore-phore", "lig->phore", "lig->phore(norm)"],
    "tp_conv_kernel<2>": ["lig-lig", "phore->lig", "phore->lig(norm)", "phore-phore", "lig->phore", "lig->phore(norm)"],
    "tp_conv_kernel<3>": ["lig-lig", "phore->lig", "phore->lig(norm)"],
    "tp_conv_kernel<4>": ["centre (final_conv)"],
    "tp_conv_kernel<5>": ["torsion bonds (tor_bond_conv)"],
}


def short(name):
    name = name.replace("void ", "").replace("dp::", "")
    return name.split("(")[0][:60]


def _has(cur, name):
    return bool(list(cur.execute("select name from sqlite_master where name=?", (name,))))


def main():
    tag, trace = sys.argv[1], sys.argv[2]
    con = sqlite3.connect(trace)
    cur = con.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    out = [f"# rocprofv3 --kernel-trace --stats summary ({tag}); durations in microseconds",
           "# command: rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-alt --steps 20 --warmup 3",
           "kernel,calls,total_us,avg_us,percent"]
    for n, c, t, a, p in rows:
        if p < 0.001:
            continue
        out.append(f"\"{short(n)}\",{c},{t:.1f},{a:.2f},{p:.3f}")
    open(f"profiles/{tag}_kernel_stats.csv", "w").write("\n".join(out) + "\n")

    cols = [d[0] for d in cur.execute("select * from kernels limit 1").description] if _has(cur, "kernels") else []
    namecol = "name" if "name" in cols else "kernel_name"
    disp = list(cur.execute(f"select {namecol}, start, end from kernels order by start")) if cols else []
    lines = [f"# per-launch-slot durations of the fused TP conv ({tag}); slot = position inside one dp_forward",
             "kernel,slot,launches,avg_us,min_us,max_us"]
    per = collections.defaultdict(list)
    for n, s, e in disp:
        for k in CONV_SLOTS:
            if k in n:
                per[k].append((e - s) / 1e3)
    for k, durs in per.items():
        slots = CONV_SLOTS[k]
        for i, sname in enumerate(slots):
            d = durs[i::len(slots)]
            if d:
                lines.append(f"{k},{sname},{len(d)},{sum(d) / len(d):.2f},{min(d):.2f},{max(d):.2f}")
    open(f"profiles/{tag}_conv_slots.csv", "w").write("\n".join(lines) + "\n")

    if len(sys.argv) >= 5:
        res = ["# HBM traffic counters per launch slot (separate --pmc passes), KiB as reported by rocprofv3",
               "# FETCH_SIZE on gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md HBM section): 'hbm_read_MiB' doubles it",
               "kernel,slot,launches,FETCH_SIZE_KiB,WRITE_SIZE_KiB,hbm_read_MiB_corrected,hbm_write_MiB"]
        vals = {}
        for cname, db in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
            c2 = sqlite3.connect(db).cursor()
            rr = list(c2.execute("select kernel_name, start, value from counters_collection where counter_name=? order by start", (cname,)))
            per = collections.defaultdict(list)
            for n, s, v in rr:
                for k in CONV_SLOTS:
                    if k in n:
                        per[k].append(v)
            vals[cname] = per
        for k in CONV_SLOTS:
            slots = CONV_SLOTS[k]
            for i, sname in enumerate(slots):
                f = vals["FETCH_SIZE"].get(k, [])[i::len(slots)]
                w = vals["WRITE_SIZE"].get(k, [])[i::len(slots)]
                if f and w:
                    fa, wa = sum(f) / len(f), sum(w) / len(w)
                    res.append(f"{k},{sname},{len(f)},{fa:.1f},{wa:.1f},{2 * fa / 1024:.2f},{wa / 1024:.2f}")
        open(f"profiles/{tag}_conv_hbm_traffic.csv", "w").write("\n".join(res) + "\n")


if __name__ == "__main__":
    main()
