cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_lds; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/l -- python3 tools/decode_trace.py > $O/l.log 2>&1
python3 - $(find $O/l -name "*counter_collection.csv" | head -1) <<'PY'
import csv, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:30]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
print("# rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/decode_trace.py")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:10]:
    a = v.get("SQ_LDS_IDX_ACTIVE", 0)
    print(f"{k:22s} launches {n[k]:5d}  LDS-array cycles {a:14.0f}  bank-conflict cycles {v.get('SQ_LDS_BANK_CONFLICT', 0):14.0f} ({100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / a if a else 0:5.1f} %)  unaligned stall {v.get('SQ_LDS_UNALIGNED_STALL', 0):10.0f}  GUI-active {v.get('GRBM_GUI_ACTIVE', 0):14.0f}")
PY
rm -rf $O
