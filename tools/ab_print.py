import json,sys
v=None
for l in open(sys.argv[1]):
    try: d=json.loads(l)
    except Exception: continue
    if "variant" in d: v=d["variant"]; continue
    st=d.get("stages_ms",{})
    print(v, d["value"], d.get("frame_latency_ms"), {k:(st[k]["ms"],st[k]["ms_single_stream"]) for k in st})
