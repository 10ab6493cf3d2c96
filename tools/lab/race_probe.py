"""Stage-by-stage repeat test of the Whisper leg on the miniature checkpoint, meant to be run by TWO processes at once on one GPU:
every iteration re-runs log-mel -> encoder -> a prefix decoding step -> 8 incremental steps -> alignment on the same resident batch and
compares every stage's output with iteration 0.  usage: race_probe.py iters [tag]   (PROBE_PAR=2 race_probe.py: the parent starts two)"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if os.environ.get("PROBE_PAR") and len(sys.argv) < 3:
    n = int(os.environ["PROBE_PAR"]); env = {k: v for k, v in os.environ.items() if k != "PROBE_PAR"}
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), sys.argv[1] if len(sys.argv) > 1 else "40", f"p{k}"], env=env) for k in range(n)]
    sys.exit(max(p.wait() for p in ps))
import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import whisper_weights as WW, tagger as T
from prosody_control_french_tts_amd.Aligners import decoding as DEC
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
tag = sys.argv[2] if len(sys.argv) > 2 else "solo"
from scipy.signal import resample_poly
z = np.load(os.path.join(ROOT, "tests", "golden", "demo_full.npz"))
names = sorted((k for k in z.files if k != "rate"), key=T.segment_sort_key)[:3]
clips = [np.clip(np.round(resample_poly(z[n].astype(np.float64), 160, 441)), -32768, 32767).astype(np.int16) for n in names]
edims = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=2)
tdims = dict(n_vocab=384, n_text_ctx=128, n_state=128, n_head=2, n_layer=2)
We, Wd = WW.synthetic_weights(edims, seed=77), WW.greedy_test_decoder_weights(tdims, seed=79)
eng = pkg.ProsodyEngine(0)
eng.upload(clips, 16000)
eng.whisper_load(edims, WW.pack(We, edims)); eng.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
V = tdims["n_vocab"]; eot, tsb = 300, 310
mask = DEC.vocab_mask(V, [eot + 1, eot + 2], [5, eot], tsb - 1)
prompts = [[301, 302, 303]] * len(clips)
h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
ref, bad = None, {}
for it in range(iters):
    out = {}
    eng.logmel_run(80)
    mels = [eng.logmel_fetch(i) for i in range(len(clips))]
    for i in range(len(clips)): out[f"mel{i}"] = h(mels[i])
    if it == 0: ref_mels = mels
    elif os.environ.get("PROBE_DETAIL"):
        for i in range(len(clips)):
            d = np.argwhere(mels[i] != ref_mels[i])
            if len(d):
                fr = np.unique(d[:, 1]); bd = np.unique(d[:, 0])
                print(tag, "it", it, "clip", i, "differing cells", len(d), "frames", fr[:6], "..", fr[-3:], "n_frames", len(fr), "bands", len(bd),
                      "max abs diff", float(np.max(np.abs(mels[i] - ref_mels[i]))), "content frames", len(clips[i]) // 160, flush=True)
    eng.whisper_encode_run()
    for i in range(len(clips)): out[f"enc{i}"] = h(eng.whisper_encode_fetch(i))
    toks, lps, steps = eng.whisper_decode_loop(prompts, 3, eot, tsb, mask, 12, 50)
    out["tok"] = h(toks); out["lp"] = h(lps)
    al = eng.whisper_align([p + [7, 9, 11, 13, 15, 17, eot] for p in prompts], [len(c) // 160 for c in clips], 3, want_cost=True)
    for i, a in enumerate(al): out[f"cost{i}"] = h(a["cost"]); out[f"path{i}"] = h(a["time_indices"])
    if ref is None: ref = out
    for k in out:
        if out[k] != ref[k]: bad[k] = bad.get(k, 0) + 1
print(tag, "iterations", iters, "stages that ever differed from iteration 0:", bad or "none")
eng.close()
