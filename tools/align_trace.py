"""One teacher-forced alignment (256 clips, Whisper-small size, 24-48 tokens per clip as bench.py draws them) for a rocprofv3 trace:
  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/align_trace.py
then  python3 tools/align_trace.py --summarise OUT  lists the kernels of the last whisper_align_run in launch order (duration, grid) and per-kernel sums."""
import collections, csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarise(d, verbose=True):
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    K = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
    dtw = [i for i, r in enumerate(K) if "k_dtw" in r["Kernel_Name"]]
    # a run = everything after the previous k_dtw up to this one
    lo, hi = dtw[-2] + 1, dtw[-1]
    run = K[lo:hi + 1]
    span = (int(run[-1]["End_Timestamp"]) - int(run[0]["Start_Timestamp"])) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in run) / 1e3
    print(f"last alignment: {len(run)} kernels, {span:.1f} us from first start to last end, {busy:.1f} us of kernel time")
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in run:
        m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", r["Kernel_Name"])
        name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:40]
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        acc[name][0] += 1; acc[name][1] += us
        if verbose:
            print(f"  {name:44s} {us:9.2f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9s} x {r.get('Grid_Size_Y', '1'):>5s} x {r.get('Grid_Size_Z', '1'):>4s}  wg {r.get('Workgroup_Size_X', '?')}")
    print("per kernel:")
    for name, (cnt, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"  {name:44s} {cnt:4d} launches  {us / cnt:9.2f} us each  {us:9.1f} us")


if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    summarise(sys.argv[2], verbose="--brief" not in sys.argv); sys.exit(0)

import time
import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth, whisper_weights as WW
n = 256
edims, tdims = WW.DIMS["small"], WW.TEXT_DIMS["small"]
eng = pkg.ProsodyEngine(0)
eng.upload(synth.synth_batch(n, 10.0, 16000, first=0), 16000)
eng.logmel_run(80)
eng.whisper_load(edims, WW.pack(WW.synthetic_weights(edims), edims))
eng.whisper_decoder_load(tdims, WW.pack_decoder(WW.synthetic_decoder_weights(tdims), tdims))
eng.whisper_encode_run()
trng = np.random.default_rng(7)
toks = [trng.integers(0, tdims["n_vocab"], size=int(trng.integers(24, 48))).tolist() for _ in range(n)]
for _ in range(4):
    eng.whisper_align_run(toks, [1000] * n, 3)
    eng.sync(); time.sleep(0.02)
print("rows:", sum(len(t) for t in toks))
eng.close()
