#!/usr/bin/env python3
"""profiles/traffic.json (what bench.py's roofline.traffic quotes) from a rocprofv3 summary of the headline run.
usage: python scripts/update_traffic.py <tag> "<what changed>"      (reads profiles/<tag>_summary.json)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, what = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
s = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_summary.json")))
d = s["derived"]
p = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(p))
total = int(round(d["fetch_bytes_x2_gfx950"] + d["write_bytes"]))
t["source"] = (f"profiles/{tag}_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, the headline kernel "
               f"{s.get('kernel_stats', [{}])[0].get('kernel', 'k_query_direct<2,false,true,1,false,true>') if isinstance(s.get('kernel_stats'), list) else 'k_query_direct<2,false,true,1,false,true>'}, "
               "1048576 rays in 8x8 tiles + device-chosen split slots, 1310720 tris; bench.py --steps 1000 --warmup 50 --no-companions); "
               "c5ii: round 5's figure (k_query_wide<CLOSEST> on one 12.5 M-ray shard), not re-measured under contract 3")
t["round"] = f"round 6 ({tag})"
t["fetch_bytes_raw"], t["fetch_bytes_x2_gfx950"], t["write_bytes"] = d["fetch_bytes_raw"], d["fetch_bytes_x2_gfx950"], d["write_bytes"]
t["closest_hbm_bytes_per_launch"] = total
t["history"][f"{tag} (round 6: {what})"] = total
json.dump(t, open(p, "w"), indent=1, ensure_ascii=False)
print(total)
