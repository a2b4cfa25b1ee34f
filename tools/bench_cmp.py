import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    kb = d["kernel_breakdown"]
    print(f, d["value"], "pw_fwd", kb["pw_fwd"], "sum_ms", round(sum(v["ms_per_step"] for v in kb.values()), 3))
