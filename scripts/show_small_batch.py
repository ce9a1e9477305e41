import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    if d["kernel"] == "default" and d["batch"] in (1, 2, 4, 8, 16, 32): print(d["op"], d["batch"], round(d["us_per_call"], 2))
