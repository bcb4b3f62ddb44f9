#!/bin/bash
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-secondary --steps 50 2>/dev/null > /tmp/n.json
  (cd ab_prev && python bench.py --no-cpu-baseline --no-secondary --steps 50 2>/dev/null > /tmp/o.json)
  python - <<PY
import json
for n in ["n","o"]:
    j=json.loads(open("/tmp/%s.json"%n).read().strip().splitlines()[-1]); print(n, j["ms_per_step"], {k:v["ms_per_step"] for k,v in j["kernels"].items() if "igemm<bf16,1" in k or "igemm<bf16,2" in k or "rows" in k})
PY
done
