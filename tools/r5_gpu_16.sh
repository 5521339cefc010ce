#!/bin/bash
for env in "FC_XQ=1" "FC_XQ=0" "FC_QUAD=0"; do
  echo "== $env"; env $env timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > /tmp/b.json 2>/tmp/b.err
  python - <<'P'
import json
d=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], {k:(v.get("value"), v.get("error")) for k,v in d["secondary"].items()})
P
done
