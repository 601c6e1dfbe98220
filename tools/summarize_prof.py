#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into small text files.

usage: summarize_prof.py <out-prefix> <kernel-trace.db> [--fetch <pmc FETCH_SIZE .db>] [--write <pmc WRITE_SIZE .db>] [--bench <bench json>]

Writes <out-prefix>_kernel_stats.csv (top kernels), <out-prefix>_conv_slots.csv (per launch slot of the fused TP conv inside one
dp_forward: duration, and - with --bench, which carries the edge counts of the benched batch - algorithmic TFLOP/s by SURVEY.md 8d's
edge-side count) and, with the PMC passes, <out-prefix>_conv_hbm_traffic.csv plus one JSON record for profiles/conv_hbm_traffic.json."""
import argparse
import collections
import json
import sqlite3

# launch order of the convs of one kernel instantiation inside one dp_forward (dp_graph.hip)
EDGE_ALL = ["lig-lig", "phore->lig", "phore->lig(norm)", "phore-phore", "lig->phore", "lig->phore(norm)"]
SLOTS = {}
# node-side kernels: tp_conv_node_kernel<KIND, HAS_LL, XMODE, HALFOK> (and the _split_ twin for small batches; HALFOK = false is the lean
# instantiation without the half-pass variant); default dp_forward fuses the convs of a side into one launch, so every instantiation
# is launched once per forward
LIG3, LIG2, PH2 = "lig-lig + phore->lig + phore->lig(norm)", "phore->lig + phore->lig(norm)", "lig->phore + lig->phore(norm)"
for k in range(4):
    for kn in ("tp_conv_node_kernel", "tp_conv_node_split_kernel"):
        for h in ("true", "false"):
            SLOTS[f"{kn}<{k}, true, 1, {h}>"] = [LIG3]
            SLOTS[f"{kn}<{k}, false, 1, {h}>"] = [LIG2]
            SLOTS[f"{kn}<{k}, false, 2, {h}>"] = [PH2]
    for h in ("true", "false"):                               # precision="bf16": <KIND, HAS_LL, XMODE, HALFOK> (round 4: without HALFOK)
        SLOTS[f"tp_conv_node_bf16_kernel<{k}, true, 1, {h}>"] = [LIG3]
        SLOTS[f"tp_conv_node_bf16_kernel<{k}, false, 1, {h}>"] = [LIG2]
        SLOTS[f"tp_conv_node_bf16_kernel<{k}, false, 2, {h}>"] = [PH2]
    SLOTS[f"tp_conv_node_bf16_kernel<{k}, true, 1>"] = [LIG3]
    SLOTS[f"tp_conv_node_bf16_kernel<{k}, false, 1>"] = [LIG2]
    SLOTS[f"tp_conv_node_bf16_kernel<{k}, false, 2>"] = [PH2]
    for h in ("true", "false"):                               # the f16x3 node-side convs of an fp32 model: <KIND, HAS_LL, XMODE, HALFOK>
        SLOTS[f"tp_conv_node_f16_kernel<{k}, true, 1, {h}>"] = [LIG3]
        SLOTS[f"tp_conv_node_f16_kernel<{k}, false, 1, {h}>"] = [LIG2]
        SLOTS[f"tp_conv_node_f16_kernel<{k}, false, 2, {h}>"] = [PH2]
# round 6: the 16-bit node-side kernels carry a fifth template argument (S0K: the s0-only instantiation of a normals conv, its own launch) and the
# normals convs of a big batch run as that launch + a DP_EDGES_LIST launch over their listed pairs (tp_conv_bf16_kernel<KIND, 2, false, true, EMIT =
# true, SKIPL0 = true>); the phore-phore convs of small pharmacophores as a list launch over the whole batch's edges (SKIPL0 = false)
LIG2L, LIGN0, PHN0, PH1 = "lig-lig + phore->lig", "phore->lig(norm), l = 0 part", "lig->phore(norm), l = 0 part", "lig->phore"
for k in range(4):
    for fam in ("tp_conv_node_f16_kernel", "tp_conv_node_bf16_kernel"):
        for h in ("true", "false"):
            SLOTS[f"{fam}<{k}, true, 1, {h}, false>"] = [LIG3]            # re-labelled below when the batch took the split
            SLOTS[f"{fam}<{k}, false, 1, {h}, false>"] = [LIG2]
            SLOTS[f"{fam}<{k}, false, 2, {h}, false>"] = [PH2]
            SLOTS[f"{fam}<{k}, false, 1, {h}, true>"] = [LIGN0]
            SLOTS[f"{fam}<{k}, false, 2, {h}, true>"] = [PHN0]
    SLOTS[f"tp_conv_bf16_kernel<{k}, 2, false, true, true, true>"] = ["normals convs: listed pairs (both sides)"]
    SLOTS[f"tp_conv_bf16_kernel<{k}, 2, false, true, true, false>"] = ["phore-phore (list launch)"]
SLOTS["tp_conv_kernel<4>"] = ["centre (final_conv)"]
SLOTS["tp_conv_kernel<5>"] = ["torsion bonds (tor_bond_conv)"]
# SURVEY.md 8d per edge-conv (edge-side count): 2*(n_attr*hid + hid*numel) + 2*sum_paths mul_in*mul_out*(2 l_out + 1); kinds 4 / 5 = the
# centre head (fc 40->40->200, out 2x1o+2x1e) and the torsion head (fc 60->60->1600, scalar outputs)
FLOPS = {0: 79200 + 2000, 1: 139200 + 4200, 2: 199200 + 6400, 3: 271200 + 8400, 4: 19200 + 1200, 5: 199200 + 3200}
EDGE_SET = {LIG3: ("ll", "x", "x"), LIG2: ("x", "x"), PH2: ("x", "x"), "lig-lig + phore->lig": ("ll", "x"), "lig->phore": ("x",), "phore-phore (list launch)": "pp", "lig-lig": "ll", "phore-phore": "pp", "phore->lig": "x", "phore->lig(norm)": "x", "lig->phore": "x", "lig->phore(norm)": "x",
            "centre (final_conv)": "center", "torsion bonds (tor_bond_conv)": "tor"}


def short(name):
    name = name.replace("void ", "").replace("dp::", "").replace("dpn::", "").replace("dpb::", "").replace("dpf::", "")
    return name.split("(")[0][:60]


def slot_names(kernels_seen):
    """edge-side layer kernels: all six slots when the node-side kernel is absent, only phore-phore (and lig-lig when that ran
    edge-side) otherwise; in an fp32 model those edge-side convs run as tp_conv_bf16_kernel<KIND, 3, false> (bf16x6)"""
    s = dict(SLOTS)
    split = any(k.startswith(("tp_conv_node_f16_kernel<", "tp_conv_node_bf16_kernel<")) and k.endswith(", true>") for k in kernels_seen)
    if split:                                        # the fused launches carry one conv less each
        for k in list(s):
            if k.startswith(("tp_conv_node_f16_kernel<", "tp_conv_node_bf16_kernel<")) and k.endswith(", false>") and k.count(",") == 4:
                s[k] = [{LIG3: LIG2L, LIG2: "phore->lig", PH2: PH1}[s[k][0]]]
    node = any("tp_conv_node" in k for k in kernels_seen)
    node_ll = any("tp_conv_node" in k and ", true, " in k.split("<")[1][:12] for k in kernels_seen)
    x6 = any(k.startswith("tp_conv_bf16_kernel<") and k.endswith(", 3, false>") for k in kernels_seen)
    x1 = any(k.startswith("tp_conv_bf16_kernel<") and k.endswith(", 1, false>") for k in kernels_seen)      # precision="bf16": one-product edge-side kernel
    for k in range(4):
        names = [f"tp_conv_kernel<{k}>"] + ([f"tp_conv_bf16_kernel<{k}, 3, false>"] if node and x6 else []) + ([f"tp_conv_bf16_kernel<{k}, 1, false>"] if node and x1 else [])
        names += [n_ for n_ in (f"tp_conv_bf16_kernel<{k}, 2, false, true, false, false>", f"tp_conv_bf16_kernel<{k}, 1, false, false, false, false>") if n_ in kernels_seen]
        for n in names:
            if not node:
                s[n] = EDGE_ALL if k < 3 else EDGE_ALL[:3]
            elif node_ll:
                s[n] = ["phore-phore"]
            else:
                s[n] = ["lig-lig", "phore-phore"] if k < 3 else ["lig-lig"]
    if node and x6:
        s["tp_conv_bf16_kernel<5, 3, false>"] = ["torsion bonds (tor_bond_conv)"]
    if node and x1:
        s["tp_conv_bf16_kernel<5, 1, false>"] = ["torsion bonds (tor_bond_conv)"]
        s["tp_conv_bf16_kernel<4, 1, false>"] = ["centre (final_conv)"]
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("prefix")
    ap.add_argument("trace")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--bench")
    ap.add_argument("--command", default="")
    ap.add_argument("--config-id", type=int, default=2, help="bench.py --config value of the profiled run (keys the traffic record)")
    a = ap.parse_args()
    cur = sqlite3.connect(a.trace).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    out = [f"# rocprofv3 --kernel-trace --stats summary; durations in microseconds", f"# command: {a.command}", "kernel,calls,total_us,avg_us,percent"]
    for n, c, t, av, p in rows:
        if p >= 0.001:
            out.append(f"\"{short(n)}\",{c},{t:.1f},{av:.2f},{p:.3f}")
    open(f"{a.prefix}_kernel_stats.csv", "w").write("\n".join(out) + "\n")

    cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
    namecol = "name" if "name" in cols else "kernel_name"
    disp = [(short(n), s, e) for n, s, e in cur.execute(f"select {namecol}, start, end from kernels order by start")]
    slots = slot_names({n for n, _, _ in disp})
    bench = json.load(open(a.bench)) if a.bench else None
    edges = (bench or {}).get("config", {}).get("edge_counts", {})
    per = collections.defaultdict(list)
    for n, s, e in disp:
        if n in slots:
            per[n].append((e - s) / 1e3)
    lines = ["# per-launch-slot durations of the fused TP conv; slot = position inside one dp_forward; TFLOP/s = SURVEY.md 8d's edge-side",
             "# algorithmic flops (2*(60*60+60*numel)+TP per edge-conv) x edges of the slot's edge set / duration - the node-side kernel executes",
             f"# ~3.5x fewer flops for the same result (DESIGN.md 5).  edge counts of the benched batch: {edges}", f"# command: {a.command}",
             "kernel,slot,launches,avg_us,min_us,max_us,alg_TFLOPs"]
    for k in sorted(per):
        names = slots[k]
        kind = int(k.split("<")[1][0])
        for i, sname in enumerate(names):
            d = per[k][i::len(names)]
            if not d:
                continue
            avg = sum(d) / len(d)
            tf = ""
            es = EDGE_SET.get(sname)
            es = es if isinstance(es, tuple) else (es,)
            if kind in FLOPS and all(x in edges for x in es):
                tf = f"{FLOPS[kind] * sum(edges[x] for x in es) / (avg * 1e-6) / 1e12:.1f}"
            lines.append(f"{k},{sname},{len(d)},{avg:.2f},{min(d):.2f},{max(d):.2f},{tf}")
    open(f"{a.prefix}_conv_slots.csv", "w").write("\n".join(lines) + "\n")

    if a.fetch and a.write:
        res = ["# HBM traffic counters per launch slot (separate --pmc passes), KiB as reported by rocprofv3",
               "# FETCH_SIZE on gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md HBM section): 'hbm_read_MiB' doubles it",
               "kernel,slot,launches,FETCH_SIZE_KiB,WRITE_SIZE_KiB,hbm_read_MiB_corrected,hbm_write_MiB"]
        vals = {}
        for cname, db in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
            c2 = sqlite3.connect(db).cursor()
            pr = collections.defaultdict(list)
            for n, s, v in c2.execute("select kernel_name, start, value from counters_collection where counter_name=? order by start", (cname,)):
                if short(n) in slots:
                    pr[short(n)].append(v)
            vals[cname] = pr
        rec = None
        for k in sorted(slots):
            names = slots[k]
            for i, sname in enumerate(names):
                f = vals["FETCH_SIZE"].get(k, [])[i::len(names)]
                w = vals["WRITE_SIZE"].get(k, [])[i::len(names)]
                if f and w:
                    fa, wa = sum(f) / len(f), sum(w) / len(w)
                    res.append(f"{k},{sname},{len(f)},{fa:.1f},{wa:.1f},{2 * fa / 1024:.2f},{wa / 1024:.2f}")
                    if (k.startswith("tp_conv_node_kernel<3, true, 1, ")) or (k == "tp_conv_kernel<3>" and sname == "phore->lig") or \
                            (k.startswith(("tp_conv_node_bf16_kernel<3, true, 1", "tp_conv_node_f16_kernel<3, true, 1, ")) and not k.endswith(", true>")):      # bench.py PROF_LAYER = 3
                        rec = int((2 * fa + wa) * 1024)
        open(f"{a.prefix}_conv_hbm_traffic.csv", "w").write("\n".join(res) + "\n")
        if rec is not None and bench is not None:
            import os
            import sys
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
            import bench as benchmod
            json.dump({"source_hash": benchmod.kernel_source_hash(), "config": a.config_id, "poses": bench["config"]["poses_per_gpu"],
                       "precision": bench["config"]["precision"], "conv_algo": bench["config"]["conv_algo"], "traffic_bytes": rec,
                       "profile": os.path.basename(a.prefix) + "_conv_hbm_traffic.csv"}, open(f"{a.prefix}_traffic_record.json", "w"))


if __name__ == "__main__":
    main()
