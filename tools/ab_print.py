"""One line per run of a tools/ab_variants.sh record of bench.py lines: variant, frames/s, frame latency, per-stage ms (in flight, alone).
    python tools/ab_print.py gpurun_out/<tag>_ab.jsonl"""
import json,sys
v=None
for l in open(sys.argv[1]):
    try: d=json.loads(l)
    except Exception: continue
    if "variant" in d: v=d["variant"]; continue
    st=d.get("stages_ms",{})
    print(v, d["value"], d.get("frame_latency_ms"), {k:(st[k]["ms"],st[k]["ms_single_stream"]) for k in st})
