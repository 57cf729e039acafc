import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d.get("valid"), "bwd", d["roofline"]["avg_launch_ms"], "chains", d["config"].get("chains"), d["config"].get("block"))
km=d.get("detail",{}).get("kernel_ms") or d["roofline"].get("kernel_ms")
print(km)
