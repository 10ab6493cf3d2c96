"""Per-kernel means of an SQ counter pass (tools/pmc_sq.sh).  usage: pmc_sq_sum.py <counter_collection.csv> [kernel substring]"""
import csv, sys, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_[a-z0-9_]+(<\d>)?)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
want = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in acc.items():
    if want not in k or "SQ_WAVE_CYCLES" not in v: continue
    big = [i for i, x in enumerate(v["GRBM_GUI_ACTIVE"]) if x > 0.5 * max(v["GRBM_GUI_ACTIVE"])]          # the large launches only
    f = lambda name: sum(v[name][i] for i in big) / len(big)
    wc = f("SQ_WAVE_CYCLES")
    print(f"{k:22s} n={len(big):3d} gui/8={f('GRBM_GUI_ACTIVE') / 8:.3e} wave_cycles={wc:.3e} wait_any={f('SQ_WAIT_ANY') / wc:.2f} wait_inst={f('SQ_WAIT_INST_ANY') / wc:.2f} "
          f"active_any={f('SQ_ACTIVE_INST_ANY') / wc:.2f} active_valu={f('SQ_ACTIVE_INST_VALU') / wc:.2f} insts_valu={f('SQ_INSTS_VALU'):.3e} busy={f('SQ_BUSY_CYCLES'):.3e}")
