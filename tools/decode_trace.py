"""One device-resident decoding loop (256 clips, Whisper-small size, 32 steps, end-of-text suppressed) for a rocprofv3 trace:
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python3 tools/decode_trace.py
then  python3 tools/decode_trace.py --summarise OUT  counts what happened between the first and the last step of the loop."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarise(d):
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    mc = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
    K = list(csv.DictReader(open(kt)))
    adv = [r for r in K if "k_step_advance" in r["Kernel_Name"]]
    t0, t1 = int(adv[0]["Start_Timestamp"]), int(adv[-1]["End_Timestamp"])
    # the loops of the run: a gap of more than 5 ms between two advances separates them
    loops, cur = [], [adv[0]]
    for a, b in zip(adv, adv[1:]):
        if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 5_000_000:
            loops.append(cur); cur = []
        cur.append(b)
    loops.append(cur)
    copies = list(csv.DictReader(open(mc[0]))) if mc else []
    print(f"{len(adv)} k_step_advance launches in {len(loops)} loops; trace spans {(t1 - t0) / 1e6:.1f} ms")
    for li, lp in enumerate(loops):
        a, b = int(lp[0]["Start_Timestamp"]), int(lp[-1]["End_Timestamp"])
        inside = [c for c in copies if a <= int(c["Start_Timestamp"]) <= b]
        nbytes = [int(c.get("Bytes", c.get("Size", 0)) or 0) for c in inside]
        kin = [r for r in K if a <= int(r["Start_Timestamp"]) <= b]
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kin)
        print(f"loop {li}: {len(lp)} steps in {(b - a) / 1e6:.2f} ms = {(b - a) / 1e6 / max(len(lp) - 1, 1):.3f} ms per step; {len(kin)} kernels, device busy {busy / (b - a) * 100:.1f} %; "
              f"memory copies between the first and the last step: {len(inside)} ({sum(nbytes)} bytes: {sorted(set(nbytes))})")


    # where a step's time goes (last loop): per kernel launches / mean duration / time per step, and the gaps between consecutive kernels
    import re, collections
    lp = loops[-1]
    a, b = int(lp[0]["Start_Timestamp"]), int(lp[-1]["End_Timestamp"])
    kin = sorted((r for r in K if a <= int(r["Start_Timestamp"]) <= b), key=lambda r: int(r["Start_Timestamp"]))
    steps = max(len(lp) - 1, 1)
    acc = collections.defaultdict(lambda: [0, 0])
    for r in kin:
        m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", r["Kernel_Name"])
        name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:40]
        acc[name][0] += 1; acc[name][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    gaps = [int(y["Start_Timestamp"]) - int(x["End_Timestamp"]) for x, y in zip(kin, kin[1:])]
    print(f"last loop, per step: {len(kin) / steps:.1f} kernels; gaps between consecutive kernels: mean {sum(gaps) / len(gaps) / 1e3:.2f} us, "
          f"sum per step {sum(g for g in gaps if g > 0) / steps / 1e3:.1f} us, overlapped (negative) {sum(1 for g in gaps if g < 0)}")
    for name, (cnt, ns) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"  {name:60s} {cnt / steps:6.1f} per step  {ns / cnt / 1e3:8.2f} us each  {ns / steps / 1e3:8.1f} us per step")


if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    summarise(sys.argv[2]); sys.exit(0)

import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth, whisper_weights as WW
from prosody_control_french_tts_amd.Aligners import decoding as DEC
n, steps = 256, 32
edims, tdims = WW.DIMS["small"], WW.TEXT_DIMS["small"]
eng = pkg.ProsodyEngine(0)
eng.upload(synth.synth_batch(n, 10.0, 16000, first=0), 16000)
eng.logmel_run(80)
eng.whisper_load(edims, WW.pack(WW.synthetic_weights(edims), edims))
eng.whisper_decoder_load(tdims, WW.pack_decoder(WW.greedy_test_decoder_weights(tdims), tdims))
eng.whisper_encode_run()
mask = DEC.vocab_mask(tdims["n_vocab"], list(range(50258, 50363)) + [50257], [220, 50257], 50363)
init = [[50258, 50265, 50359] for _ in range(n)]
for _ in range(3):
    toks, _, _ = eng.whisper_decode_loop(init, 3, 50257, 50364, mask, steps, 50)
    eng.sync()
    import time; time.sleep(0.02)
print("steps run:", toks.shape[1])
eng.close()
