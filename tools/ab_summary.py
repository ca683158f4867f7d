"""Prints value / other entry point / per-stage alone times of every bench line under a directory (tools/ab_libs.sh)."""
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.jsonl")):
    for ln in open(f):
        ln = ln.strip()
        if not ln.startswith("{"):
            continue
        d = json.loads(ln)
        st = d.get("stages_ms", {})
        print("%-60s %9.1f %9.1f  alone us: %s" % (f.split("/")[-1], d["value"], d.get("other_entry_point", {}).get("value", 0),
              " ".join("%s %.1f" % (k[:5], 1e3 * v["ms_single_stream"]) for k, v in st.items() if v.get("ms_single_stream"))))
