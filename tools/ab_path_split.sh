#!/bin/bash
# A/B of the node-side fp32 kernels at configs[1] (512 poses): unsplit lean kernels (default) vs the LEAN path-split build (build first:
# tools/build_variant.sh splitlean "-DDP_SPLIT_LEAN" dp_conv_node_k0.hip dp_conv_node_k1.hip dp_conv_node_k2.hip dp_conv_node_k3.hip) - two workgroups per node group
# each owns half of the out slots -> half the accumulators, no spills) - step rate, per-kernel time, HBM bytes per launch (PMC).
# -> gpurun_out/r4_ab_split_{0,1}_{bench.json,trace.txt,fetch.txt,write.txt}
cd "$(dirname "$0")/.." || exit 1
export PYTHONPATH=. TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out
ROOT=$PWD
for v in 0 1; do
  if [ $v == 1 ]; then export DIFFPHORE_PATH_SPLIT=1 DIFFPHORE_LIB=$ROOT/build_abl/libsplitlean.so; else export DIFFPHORE_PATH_SPLIT=auto; fi
  python bench.py --steps 60 --warmup 10 --no-alt --no-e2e --no-cpu-baseline --no-other > gpurun_out/r4_ab_split_${v}_bench.json 2>/dev/null
  for c in trace FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/ab_$v_$c
    if [ $c == trace ]; then P="--kernel-trace --stats"; else P="--kernel-trace --pmc $c"; fi
    (cd /tmp && timeout 600 rocprofv3 $P -d /tmp/ab_${v}_$c -o r -- python $ROOT/bench.py --steps 8 --warmup 2 --no-alt --no-e2e --no-cpu-baseline --no-other > /dev/null 2>&1)
    db=$(find /tmp/ab_${v}_$c -name '*.db' | head -1)
    if [ $c == trace ]; then
      python - "$db" > gpurun_out/r4_ab_split_${v}_trace.txt <<'PY'
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
acc = collections.defaultdict(list)
for n, s, e in cur.execute(f"select k.kernel_name, d.start, d.end from {kd} d join {ks} k on d.kernel_id = k.id"):
    acc[n.replace("void ", "").split("(")[0][:60]].append((e - s) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if "tp_conv" in k: print(f"{k},{len(v)},{sum(v)/len(v):.1f}")
PY
    else
      python tools/pmc_dump.py "$db" tp_conv_node > gpurun_out/r4_ab_split_${v}_$c.txt
    fi
  done
done
python - <<'PY'
import json
for v in (0, 1):
    d = json.load(open(f"gpurun_out/r4_ab_split_{v}_bench.json"))
    print("path_split", v, "pose-steps/s", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "layer-3 launch ms", round(d["roofline"]["kernel_ms"], 4))
    for f in ("trace", "FETCH_SIZE", "WRITE_SIZE"):
        print(open(f"gpurun_out/r4_ab_split_{v}_{f}.txt").read().strip())
PY
