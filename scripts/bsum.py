import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([x for x in open(f) if x.startswith("{")][-1])
    except Exception as e:
        print(f, "no json", e); continue
    print(f, d["value"], d["unit"], d["ms_per_step"], "ms |", " ".join("%s%d:%.3f" % (x["kernel"][2:6], x["gen"], x["ms"]) for x in d["roofline"]["launches"]))
