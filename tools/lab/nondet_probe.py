"""Hunt for run-to-run differences of the aligner on the miniature checkpoint: fresh child processes transcribe the same recordings
(optionally two at a time on the one GPU) and the raw JSON results are hashed.  usage: nondet_probe.py [runs] [parallel] [first] [count]"""
import hashlib, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    base, first, count = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    import logging
    import numpy as np
    from pathlib import Path
    from prosody_control_french_tts_amd import audio_pipeline as AP, tagger as T
    from tests.test_gpu_c5 import WORDS, write_wav
    from tests.test_gpu_aligner import write_model_dir
    logging.basicConfig(level=logging.ERROR)
    base = Path(base)
    z = np.load(Path(ROOT) / "tests" / "golden" / "demo_full.npz")
    rate = int(z["rate"])
    names = sorted((k for k in z.files if k != "rate"), key=T.segment_sort_key)[first:first + count]
    for k, n in enumerate(names):
        write_wav(base / "Data" / "voice" / "v" / "audio" / f"segment_ph{k + 1}.wav", z[n], rate)
    write_model_dir(base / "whisper_dir", merges=WORDS, word_gain=3.0, eot_gain=0.3)
    cfg = {"data_dir": "Data/voice", "out_dir": "Out", "whisper_device": "cuda", "whisper_model": "medium", "whisper_dir": str(base / "whisper_dir"),
           "voice_names": ["v"], "steps_to_run": ["Align+Transcribe"]}
    assert AP.run_all(cfg, base=base) == []
    out = {}
    for p in sorted((base / "Data" / "voice" / "v").rglob("*.raw.json")):
        out[p.name] = hashlib.sha1(p.read_bytes()).hexdigest()[:12]
    print("RESULT " + json.dumps(out))
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "seq":
    # ONE process, several jobs one after the other on the same engine: "first:count first:count ..." -> hashes per job (state carried over?)
    import logging
    import numpy as np
    from pathlib import Path
    from prosody_control_french_tts_amd import audio_pipeline as AP, tagger as T
    from tests.test_gpu_c5 import WORDS, write_wav
    from tests.test_gpu_aligner import write_model_dir
    logging.basicConfig(level=logging.ERROR)
    z = np.load(Path(ROOT) / "tests" / "golden" / "demo_full.npz")
    allnames = sorted((k for k in z.files if k != "rate"), key=T.segment_sort_key)
    for job in sys.argv[2:]:
        first, count = (int(x) for x in job.split(":"))
        base = Path(tempfile.mkdtemp())
        for k, n in enumerate(allnames[first:first + count]):
            write_wav(base / "Data" / "voice" / "v" / "audio" / f"segment_ph{k + 1}.wav", z[n], int(z["rate"]))
        write_model_dir(base / "whisper_dir", merges=WORDS, word_gain=3.0, eot_gain=0.3)
        cfg = {"data_dir": "Data/voice", "out_dir": "Out", "whisper_device": "cuda", "whisper_model": "medium", "whisper_dir": str(base / "whisper_dir"),
               "voice_names": ["v"], "steps_to_run": ["Align+Transcribe"]}
        assert AP.run_all(cfg, base=base) == []
        out = {int(p.name.split("ph")[1].split(".")[0]) - 1 + first: hashlib.sha1(p.read_bytes()).hexdigest()[:12] for p in sorted((base / "Data" / "voice" / "v").rglob("*.raw.json"))}
        print("job", job, out)
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "--rank":
    # one of two gloo ranks on cuda:0, the five recordings of "v1" sharded 3 / 2, the Align+Transcribe step only
    rank, port, base = int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, PCE_DIST_BACKEND="gloo", PCE_RANK_DEVICE="0")
    import logging
    import numpy as np
    from pathlib import Path
    import torch.distributed as dist
    from prosody_control_french_tts_amd import audio_pipeline as AP, shard, tagger as T
    from tests.test_gpu_c5 import WORDS, write_wav
    from tests.test_gpu_aligner import write_model_dir
    logging.basicConfig(level=logging.ERROR)
    base = Path(base)
    if rank == 0:
        z = np.load(Path(ROOT) / "tests" / "golden" / "demo_full.npz")
        names = sorted((k for k in z.files if k != "rate"), key=T.segment_sort_key)[:5]
        for k, n in enumerate(names):
            write_wav(base / "Data" / "voice" / "v" / "audio" / f"segment_ph{k + 1}.wav", z[n], int(z["rate"]))
        write_model_dir(base / "whisper_dir", merges=WORDS, word_gain=3.0, eot_gain=0.3)
    shard.init_from_env(); dist.barrier()
    cfg = {"data_dir": "Data/voice", "out_dir": "Out", "whisper_device": "cuda", "whisper_model": "medium", "whisper_dir": str(base / "whisper_dir"),
           "voice_names": ["v"], "steps_to_run": ["Align+Transcribe"]}
    assert AP.run_all(cfg, base=base) == []
    dist.barrier()
    if rank == 0:
        out = {p.name: hashlib.sha1(p.read_bytes()).hexdigest()[:12] for p in sorted((base / "Data" / "voice" / "v").rglob("*.raw.json"))}
        print("RESULT " + json.dumps(out))
    dist.barrier(); dist.destroy_process_group()
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "lockstep":
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    seen = {}
    for r in range(runs):
        td = tempfile.mkdtemp(); port = str(29600 + (os.getpid() + r) % 300)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(k), port, td], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for k in range(2)]
        outs = [p.communicate(timeout=600)[0].decode() for p in procs]
        line = [l for l in outs[0].splitlines() if l.startswith("RESULT ")]
        if not line:
            print("failed:", outs[0][-800:], outs[1][-800:]); continue
        for name, h in json.loads(line[0][7:]).items():
            seen.setdefault(name, {}).setdefault(h, 0); seen[name][h] += 1
    for name, hs in sorted(seen.items()):
        print(name, hs, "<-- DIFFERS" if len(hs) > 1 else "")
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "ranks":
    # the shape of tests/test_gpu_world2.py: clips [0:3] and [3:5] by two concurrent processes, and [0:5] by one; per clip the hashes must agree
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    seen = {}
    def launch(first, count):
        td = tempfile.mkdtemp()
        return first, subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", td, str(first), str(count)], stdout=subprocess.PIPE,
                                       stderr=subprocess.STDOUT, env=dict(os.environ, PCE_RANK_DEVICE="0"))
    def collect(first, p, tag):
        o = p.communicate(timeout=600)[0].decode()
        line = [l for l in o.splitlines() if l.startswith("RESULT ")]
        if not line:
            print("child failed:", o[-1500:]); return
        for name, h in json.loads(line[0][7:]).items():
            k = int(name.split("ph")[1].split(".")[0]) - 1 + first           # global clip index
            seen.setdefault(k, {}).setdefault((tag, h), 0); seen[k][(tag, h)] += 1
    for r in range(runs):
        a, b = launch(0, 3), launch(3, 2)
        collect(*a, "w2"); collect(*b, "w2")
        c = launch(0, 5); collect(*c, "w1")
    for k in sorted(seen):
        hs = {h for (_, h) in seen[k]}
        print("clip", k, dict(seen[k]), "<-- DIFFERS" if len(hs) > 1 else "")
    sys.exit(0)

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
par = int(sys.argv[2]) if len(sys.argv) > 2 else 2
first = int(sys.argv[3]) if len(sys.argv) > 3 else 3
count = int(sys.argv[4]) if len(sys.argv) > 4 else 2
seen = {}
for r in range(0, runs, par):
    procs = []
    for k in range(par):
        td = tempfile.mkdtemp()
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", td, str(first), str(count)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      env=dict(os.environ, PCE_RANK_DEVICE="0")))
    for p in procs:
        o = p.communicate(timeout=600)[0].decode()
        line = [l for l in o.splitlines() if l.startswith("RESULT ")]
        if not line:
            print("child failed:", o[-1500:]); continue
        for name, h in json.loads(line[0][7:]).items():
            seen.setdefault(name, {}).setdefault(h, 0); seen[name][h] += 1
for name, hs in seen.items():
    print(name, hs, "<-- DIFFERS" if len(hs) > 1 else "")
