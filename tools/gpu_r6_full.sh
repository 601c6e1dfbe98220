#!/bin/bash
# the GPU suite + the default bench line (what the driver runs at round end)
export PYTHONPATH=$PWD
mkdir -p gpurun_out
tag=${1:-full}
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_suite.txt 2>&1
tail -4 gpurun_out/${tag}_suite.txt
timeout 1500 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "exact", d.get("value_exact_fp32"))
print("roofline", r["frac"], r["kernel_ms"], "whole", r["whole_step"], "\nsingle", {k: v for k, v in (r["step_weighted_single_stream"] or {}).items() if k != "launches"})
print("single launches", [(x["layer"], x["convs"], x["ms"], round(x["TFLOPs"], 1)) for x in (r["step_weighted_single_stream"] or {}).get("launches", [])])
print("e2e", {k: v for k, v in (d["config"]["e2e"] or {}).items() if k in ("value", "seconds", "breakdown_s")})
print("one", d["config"]["e2e_one_complex"])
print("other", [(o.get("config"), round(o.get("value", 0)), o.get("error")) for o in d["config"]["other_configs"] or []])
print("alt", [(x.get("precision", "")[:20], round(x.get("value", 0))) for x in d["config"]["alt_runs"] or []])
print("cpu", d.get("cpu_baseline"))
PY
