import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["detail"]["kernel_ms"]; print(sys.argv[1], "%.1f Msamples/s %.2f ms  gsum %.3f vfb %.3f"%(d["value"], d["ms_per_step"], k["k_gsum"], k["k_vfb_chain"]))
