"""The world-2 phase of tests/test_gpu_world2.py (its worker, cut before the solo run), repeated; prints the hashes of every raw JSON per run.
usage: world2_probe.py runs [steps-comma-list] [voices-comma-list] [inject 0/1]"""
import hashlib, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_gpu_world2 as W
src = W._WORKER
cut = src.index("failed = AP.run_all(cfg_for(list(VOICES), model_root), base=base)")
head = src[:cut]
STEPS = os.environ.get("PROBE_STEPS", "Align+Transcribe,Raw Synthesis,Measure & Build SSML,Final Transcribe").split(",")
VOICESEL = os.environ.get("PROBE_VOICES", "v1,bad,v3").split(",")
INJECT = os.environ.get("PROBE_INJECT", "1") == "1"
tail = f'''
import hashlib, json
def _cfg(v, m):
    c = cfg_for(v, m); c["steps_to_run"] = {STEPS!r}; return c
if not {INJECT!r}:
    AP.AudioPipeline.run = _run
failed = AP.run_all(_cfg({VOICESEL!r}, model_root), base=base)
dist.barrier()
if rank == 0:
    out = {{}}
    for p in sorted((base / "Data" / "voice").rglob("*.raw.json")):
        out[str(p.relative_to(base / "Data" / "voice"))] = hashlib.sha1(p.read_bytes()).hexdigest()[:10]
    print("RESULT " + json.dumps(out))
dist.barrier(); dist.destroy_process_group()
'''
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seen = {}
for r in range(runs):
    td = tempfile.mkdtemp(); script = os.path.join(td, "w.py"); open(script, "w").write(head + tail)
    port = str(29700 + (os.getpid() + r) % 200)
    procs = [subprocess.Popen([sys.executable, script, str(k), "2", ROOT, port, td], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=W._env()) for k in range(2)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT ")]
    if not line:
        print("failed:", outs[0][-1200:]); continue
    for name, h in json.loads(line[0][7:]).items():
        seen.setdefault(name, {}).setdefault(h, 0); seen[name][h] += 1
bad = {k: v for k, v in seen.items() if len(v) > 1}
print("runs", runs, "files", len(seen), "varying", len(bad))
for k, v in sorted(bad.items()):
    print("  ", k, v)
