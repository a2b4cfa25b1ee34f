import json,sys
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{"metric"'):
        d=json.loads(line)
        print("value", d["value"], "ms/step", d["ms_per_step"], "mem", d["peak_mem_GB"], "loss", d["loss"])
        if d.get("roofline"): print("roofline", {k:v for k,v in d["roofline"].items() if k!="whole_path"})
        tot=0
        for k,v in (d.get("kernel_breakdown") or {}).items():
            tot+=v["ms_per_step"]; print(f'  {k:16s} {v["ms_per_step"]:8.3f} ms  n={v["launches_per_step"]:3d}  {v["GBps"]:8.1f} GB/s {v["TFLOPs"]:7.2f} TF')
        print("  total kernel ms", round(tot,2))
        for r in (d.get("top_launches") or []): print("   top", r)
        if d.get("cpu_baseline"): print("cpu", d["cpu_baseline"])
    elif line: print(line[:300])
