import json, sys
d = json.load(open(sys.argv[1]))
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 3))
for r in d.get("stages", []):
    print(f'  stage {r["stage"][:34]:34s} {r["ms_per_step"]:.4f} ms  {r["achieved"]:.1f} {r["unit"]}  frac {r["frac"]:.4f}')
for k in d["kernels"]:
    print(f'{k["kernel"]:16s} {k["ms_per_step"]:.4f} ms')
