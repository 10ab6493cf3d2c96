import json,sys
d=json.load(open(sys.argv[1]))
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],3))
for k in d["kernels"]: print(f'{k["kernel"]:16s} {k["ms_per_step"]:.4f} ms  {k["achieved_GBs"] or 0:.1f} GB/s')
