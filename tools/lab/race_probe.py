"""Stage-by-stage repeat test of the Whisper leg on the miniature checkpoint, meant to be run by SEVERAL processes at once on one GPU:
every iteration re-runs log-mel -> encoder -> a prefix decoding step -> incremental steps -> alignment on the same resident batch and
compares every stage's output with iteration 0.
usage: race_probe.py iters [tag]      PROBE_PAR=n: the parent starts n of itself (fresh processes; the parent never touches the GPU)
       PROBE_STAGES=mel,enc,dec,align  (default: all; "mel" alone = only the log-mel kernels, "dec" alone = encoder once, then only the loop)
       PROBE_CU_MASKS="0:0-79;0:80-159;0:160-255"  child k gets HSA_CU_MASK = the k-th entry (disjoint compute units per process)
       PROBE_IDLE=m  m more processes that create an engine context and then sleep (what the pytest parent is during tests/test_gpu_world2.py)
       PROBE_ROLES="mel;enc;dec"  child k runs only the stages of the k-th entry (victim / culprit experiments); an entry "torch:gemm", "torch:stft",
                       "torch:ln", "torch:copy" is a plain PyTorch loop of that operation for as long as child 0 runs
       PROBE_DETAIL=1  print what a differing log-mel looks like (which frames, and whether the wrong frame equals another frame's values)"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if os.environ.get("PROBE_PAR") and len(sys.argv) < 3:
    n = int(os.environ["PROBE_PAR"]); env = {k: v for k, v in os.environ.items() if k not in ("PROBE_PAR", "PROBE_IDLE")}
    masks = [m for m in os.environ.get("PROBE_CU_MASKS", "").split(";") if m]
    iters = sys.argv[1] if len(sys.argv) > 1 else "40"
    roles = [r for r in os.environ.get("PROBE_ROLES", "").split(";") if r]
    ps, helpers = [], []
    import tempfile, time as _t
    flagdir = tempfile.mkdtemp()
    is_h = lambda r: r.startswith("torch:") or r.startswith("st:")
    order = sorted(range(n), key=lambda k: not (roles and is_h(roles[k % len(roles)])))     # PyTorch neighbours first: they take longest to come up
    for k in order:
        e = dict(env, PROBE_FLAG=os.path.join(flagdir, f"up{k}"))
        if masks:
            e["HSA_CU_MASK"] = masks[k % len(masks)]
        if roles:
            e["PROBE_STAGES"] = roles[k % len(roles)]
        is_helper = is_h(e.get("PROBE_STAGES", ""))
        if not is_helper and helpers:                       # the libpce children start once every PyTorch neighbour is launching kernels
            t0 = _t.time()
            while _t.time() - t0 < 240 and not all(os.path.exists(os.path.join(flagdir, f"up{j}")) for j in order[:len(helpers)]):
                _t.sleep(0.2)
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), iters, f"p{k}"], env=e)
        (helpers if is_helper else ps).append(p)
    idle = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "0", f"idle{k}"], env=env) for k in range(int(os.environ.get("PROBE_IDLE", "0")))]
    rc = max(p.wait() for p in ps)
    for p in idle + helpers:
        p.kill(); p.wait()
    sys.exit(rc)
if os.environ.get("PROBE_STAGES", "").startswith("torch:"):          # a plain PyTorch neighbour (no libpce in this process), until the parent ends it
    import torch
    op = os.environ["PROBE_STAGES"].split(":")[1]
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(4096, 1024, device="cuda", generator=g, dtype=torch.float16); b = torch.randn(1024, 2048, device="cuda", generator=g, dtype=torch.float16)
    x = torch.randn(48000, device="cuda", generator=g); win = torch.hann_window(400, device="cuda")
    (a @ b); torch.cuda.synchronize()
    open(os.environ["PROBE_FLAG"], "w").close()
    print(sys.argv[2], "neighbour", op, flush=True)
    while True:
        for _ in range(50):
            if op == "gemm": c = a @ b
            elif op == "stft": c = torch.stft(x, 400, 160, window=win, return_complex=True)
            elif op == "ln": c = torch.nn.functional.layer_norm(a.float(), (1024,))
            elif op == "copy": c = a.cpu()
        torch.cuda.synchronize()
import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import whisper_weights as WW, tagger as T
from prosody_control_french_tts_amd.Aligners import decoding as DEC
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
tag = sys.argv[2] if len(sys.argv) > 2 else "solo"
stages = set(os.environ.get("PROBE_STAGES", "mel,enc,dec,align").split(","))
from scipy.signal import resample_poly
z = np.load(os.path.join(ROOT, "tests", "golden", "demo_full.npz"))
names = sorted((k for k in z.files if k != "rate"), key=T.segment_sort_key)[:3]
clips = [np.clip(np.round(resample_poly(z[n].astype(np.float64), 160, 441)), -32768, 32767).astype(np.int16) for n in names]
edims = dict(n_mels=80, n_ctx=1500, n_state=128, n_head=2, n_layer=2)
tdims = dict(n_vocab=384, n_text_ctx=128, n_state=128, n_head=2, n_layer=2)
We, Wd = WW.synthetic_weights(edims, seed=77), WW.greedy_test_decoder_weights(tdims, seed=79)
if os.environ.get("PROBE_STAGES", "").startswith("st:"):
    # a neighbour that launches ONE kind of libpce kernel in a tight loop through the self-test entry points (operands converted once), as long as child 0 runs:
    # st:gemm128 (k_gemm_bf16, 66 KB of LDS-DMA ring), st:gemm128deep (k_gemm_bf16<.,4>: few rows, 128 KB ring), st:wide (k_gemm_wide), st:flat (k_gemm_flat 256 x 256,
    # 128 KB ring), st:skinny (k_gemm_skinny), st:attn (k_attention_lean16), st:attn32 (PCE_ATTN_M16=0 in the environment: k_attention_lean)
    import ctypes as C, torch
    kind = os.environ["PROBE_STAGES"].split(":")[1]
    eng = pkg.ProsodyEngine(0)
    rng = np.random.default_rng(3)
    def t16(*shape):
        return torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(torch.float16).contiguous()
    lib, ctx = eng._lib, eng._ctx
    if kind == "attn" or kind == "attn32":
        q, k, v = t16(3, 1500, 128), t16(3, 1500, 128), t16(3, 1500, 128); o = torch.zeros_like(q); fb = C.c_int32(0)
        call = lambda: lib.pce_selftest_attention(ctx, q.data_ptr(), k.data_ptr(), v.data_ptr(), 3, 2, 1500, 1500, 0, 0, o.data_ptr(), C.addressof(fb))
    else:
        M, N, K, epi = {"gemm128": (4500, 128, 128, 16), "gemm128deep": (96, 128, 512, 16), "wide": (4500, 1536, 128, 16), "flat": (4608, 256, 256, 0),
                        "skinny": (96, 512, 128, 16)}[kind]
        a, b = t16(M, K), t16(N, K); bias = np.zeros(N, dtype=np.float32); out = torch.zeros(M * N, dtype=torch.float16)
        call = lambda: lib.pce_selftest_gemm(ctx, a.data_ptr(), b.data_ptr(), bias.ctypes.data, M, N, K, epi, 1, 0, out.data_ptr())
    rc = call()
    open(os.environ["PROBE_FLAG"], "w").close()
    print(tag, "neighbour", kind, "rc", rc, flush=True)
    n = 0; t0 = time.time()
    while True:
        for _ in range(200): call()
        n += 200
        if n % 20000 == 0: print(tag, kind, f"{(time.time() - t0) / n * 1e6:.0f} us per call", flush=True)
if os.environ.get("PROBE_SHIFT_MB") and tag == "p0":           # child 0 only: another device allocation FIRST, so that none of its buffers sits at the
    import ctypes                                              # virtual address the same buffer has in the neighbour processes
    _hip = ctypes.CDLL("libamdhip64.so"); _p = ctypes.c_void_p()
    assert _hip.hipMalloc(ctypes.byref(_p), ctypes.c_size_t(int(os.environ["PROBE_SHIFT_MB"]) << 20)) == 0
    print(tag, "shifted by", os.environ["PROBE_SHIFT_MB"], "MB at", hex(_p.value), flush=True)
eng = pkg.ProsodyEngine(0)
if tag.startswith("idle"):                                    # a context that exists and does nothing (the pytest parent of the world-2 test)
    eng.upload(clips, 16000); eng.logmel_run(80)
    time.sleep(3600)
eng.upload(clips, 16000)
eng.whisper_load(edims, WW.pack(We, edims)); eng.whisper_decoder_load(tdims, WW.pack_decoder(Wd, tdims))
V = tdims["n_vocab"]; eot, tsb = 300, 310
mask = DEC.vocab_mask(V, [eot + 1, eot + 2], [5, eot], tsb - 1)
prompts = [[301, 302, 303]] * len(clips)
h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
ref, bad = None, {}
sl_all = eng.whole_clip_slices(); pparams = pkg.PitchParams.praat(150.0, 600.0)
eng.logmel_run(80); eng.whisper_encode_run()
t0 = time.time()
for it in range(iters):
    out = {}
    if "mel" in stages:
        eng.logmel_run(80)
        mels = [eng.logmel_fetch(i) for i in range(len(clips))]
        for i in range(len(clips)): out[f"mel{i}"] = h(mels[i])
        if it == 0: ref_mels = mels
        elif os.environ.get("PROBE_DETAIL"):
            for i in range(len(clips)):
                d = np.argwhere(mels[i] != ref_mels[i])
                if len(d):
                    fr = np.unique(d[:, 1]); bd = np.unique(d[:, 0])
                    # does the wrong frame hold ANOTHER frame's values (a mixed-up buffer) or values no frame has (a wrong transform)?
                    twins = []
                    for f in fr[:4]:
                        for j in range(len(clips)):
                            eq = np.where((ref_mels[j] == mels[i][:, f:f + 1]).sum(axis=0) >= 40)[0]
                            twins += [(int(f), j, int(e)) for e in eq[:3]]
                    print(tag, "it", it, "clip", i, "differing cells", len(d), "frames", fr[:6], "..", fr[-3:], "n_frames", len(fr), "bands", len(bd),
                          "max abs diff", float(np.max(np.abs(mels[i] - ref_mels[i]))), "content frames", len(clips[i]) // 160,
                          "frame(s) equal to a reference frame (wrong frame, clip, frame):", twins or "none", flush=True)
                    # the cells of the first wrong frame, and two readings of them: the value normalised TWICE (a stale read / a lost raw write:
                    # (max(v, floor) + 4) / 4 applied to last iteration's result) or NOT normalised (a lost in-place write of k_logmel_norm)
                    f = int(fr[0]); bands = d[d[:, 1] == f][:, 0]
                    floor_n = (4.0 * float(ref_mels[i].max()) - 4.0) - 8.0
                    w, r = mels[i][bands, f].astype(np.float64), ref_mels[i][bands, f].astype(np.float64)
                    twice = (np.maximum(r, floor_n) + 4.0) / 4.0
                    raw = 4.0 * r - 4.0
                    print(tag, "   frame", f, "bands", bands.tolist(), "wrong", np.round(w, 4).tolist(), "right", np.round(r, 4).tolist(),
                          "| the SAME bands of the frames before:", {-k: np.round(ref_mels[i][bands, f - k], 4).tolist() for k in (1, 2, 4, 64) if f - k >= 0},
                          "| of the frames after:", {k: np.round(ref_mels[i][bands, f + k], 4).tolist() for k in (1, 64) if f + k < 3000}, flush=True)
    if "c2" in stages:                                        # the prosody kernels on the same clips (energy, LUFS, F0, STFT-dB, VAD energies)
        out["energy"] = h(eng.energy(sl_all, 500))
        lu, st = eng.lufs(sl_all); out["lufs"] = h(lu)
        pi = eng.pitch(sl_all, pparams); out["f0"] = h(pi["f0"]); out["summary"] = h(pi["summary"])
        eng.stft_db_run(1024, 256); out["stft"] = h(np.stack([eng.stft_db_fetch(i)[:, :600] for i in range(len(clips))]))
        eng.frame_energy_run(800, 800, requantize=True); out["vad"] = h(np.concatenate([eng.frame_energy_fetch(i)[0] for i in range(len(clips))]))
    if "enc" in stages:
        eng.whisper_encode_run()
        for i in range(len(clips)): out[f"enc{i}"] = h(eng.whisper_encode_fetch(i))
    if "dec" in stages:
        toks, lps, steps = eng.whisper_decode_loop(prompts, 3, eot, tsb, mask, 12, 50)
        out["tok"] = h(toks); out["lp"] = h(lps)
    if "align" in stages:
        al = eng.whisper_align([p + [7, 9, 11, 13, 15, 17, eot] for p in prompts], [len(c) // 160 for c in clips], 3, want_cost=True)
        for i, a in enumerate(al): out[f"cost{i}"] = h(a["cost"]); out[f"path{i}"] = h(a["time_indices"])
    if ref is None: ref = out
    for k in out:
        if out[k] != ref[k]: bad[k] = bad.get(k, 0) + 1
print(tag, "iterations", iters, "stages", ",".join(sorted(stages)), f"{time.time() - t0:.1f} s", "HSA_CU_MASK", os.environ.get("HSA_CU_MASK"),
      "stages that ever differed from iteration 0:", bad or "none", flush=True)
eng.close()
