"""Repeat test of the prosody kernels (energy, LUFS, F0, STFT-dB, pYIN-free) meant to be run by several processes at once on one GPU.
usage: PROBE_PAR=3 race_probe_c2.py iters"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if os.environ.get("PROBE_PAR") and len(sys.argv) < 3:
    n = int(os.environ["PROBE_PAR"]); env = {k: v for k, v in os.environ.items() if k != "PROBE_PAR"}
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), sys.argv[1] if len(sys.argv) > 1 else "40", f"p{k}"], env=env) for k in range(n)]
    sys.exit(max(p.wait() for p in ps))
import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth
iters = int(sys.argv[1]); tag = sys.argv[2] if len(sys.argv) > 2 else "solo"
clips = synth.synth_batch(24, 10.0, 16000, first=0)
eng = pkg.ProsodyEngine(0); eng.upload(clips, 16000); sl = eng.whole_clip_slices()
params = pkg.PitchParams.praat(150.0, 600.0)
h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
ref, bad = None, {}
for it in range(iters):
    out = {}
    out["energy"] = h(eng.energy(sl, 500))
    lu, st = eng.lufs(sl); out["lufs"] = h(lu)
    pi = eng.pitch(sl, params); out["f0"] = h(pi["f0"]); out["summary"] = h(pi["summary"])
    eng.stft_db_run(1024, 256); out["stft"] = h(np.stack([eng.stft_db_fetch(i) for i in range(4)]))
    eng.frame_energy_run(800, 800, requantize=True); out["vad"] = h(np.concatenate([eng.frame_energy_fetch(i)[0] for i in range(4)]))
    if ref is None: ref = out
    for k in out:
        if out[k] != ref[k]: bad[k] = bad.get(k, 0) + 1
print(tag, "iterations", iters, "stages that ever differed from iteration 0:", bad or "none")
eng.close()
