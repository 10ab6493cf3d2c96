#!/usr/bin/env python3
"""bench.py -- throughput of the prosody hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): per GPU a batch of 256 synthetic 10 s 16 kHz
mono clips, resident in HBM as int16 before the timed region.  One step = one pass of the
hot path over the batch:
    k_energy (gate/peak/RMS integers)  +  BS.1770 LUFS  +  Praat-AC F0 (150-600 Hz) with path
    finding and voiced median  +  STFT-dB 1024/256,
followed by the fetch of the per-utterance statistics (a few KB) and, for N > 1, the single
all-gather of those statistics (RCCL).  The 329 MB STFT-dB result stays in HBM.

    python bench.py --gpus N --steps K --warmup W
prints ONE JSON line on rank 0.  `value` = audio seconds processed by all ranks / wall time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (never the 2:1-sparsity figure)


def load_pmc_traffic():
    """HBM bytes per launch from rocprofv3 PMC passes of this same command (FETCH_SIZE x2 for the
    gfx950 wide-read under-count + WRITE_SIZE, KB -> B; tools/pmc_traffic.py), committed under profiles/."""
    path = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)["bytes_per_launch"]
        if "k_stft_raw" in t and "k_stft_db" not in t:      # the dB pass is timed as `k_stft_db` whichever kernel implements it
            t["k_stft_db"] = t["k_stft_raw"]
        return t
    except Exception:
        return None


def _cpu_one(args):
    c, rate = args
    from oracle import oracle as O
    x = c.astype(np.float64)
    O.gate_check(c)
    O.rms_db_int16_wrapped(c)
    O.lufs_c(x, rate)
    f0 = O.pitch_ac(x / 32768.0, 1.0 / rate, 0.5 / rate, O.praat_params(150.0, 600.0))["f0"]
    v = f0[f0 > 0]
    _ = float(np.median(v)) if v.size else 0.0
    O.stft_db(c.astype(np.float32) / 32768.0)
    return len(c)


def _cpu_one_reference_shaped(args):
    """The same arithmetic called the way the reference calls it: every measurement takes a PATH and decodes the
    file again (Code/audioPipeline.py:319,327,340,360; Code/Aligners/use_whisper_timestamped.py:130,199,583)."""
    path, rate = args
    import wave
    from oracle import oracle as O

    def decode():
        with wave.open(path, "rb") as w:
            return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")

    O.gate_check(decode()); O.gate_check(decode())                  # the noise gate runs twice per file
    O.rms_db_int16_wrapped(decode())
    O.lufs_c(decode().astype(np.float64), rate)
    n = len(decode())                                                # get_duration
    f0 = O.pitch_ac(decode().astype(np.float64) / 32768.0, 1.0 / rate, 0.5 / rate, O.praat_params(150.0, 600.0))["f0"]
    v = f0[f0 > 0]
    _ = float(np.median(v)) if v.size else 0.0
    O.stft_db(decode().astype(np.float32) / 32768.0)
    return n


def load_pmc_valu(kernel):
    """Mean SQ_INSTS_VALU per launch of `kernel` from the committed rocprofv3 --pmc pass of this command."""
    import csv
    import re
    path = os.path.join(ROOT, "profiles", "r01", "pmc_sq_counter_collection.csv")
    try:
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                if r["Counter_Name"] == "SQ_INSTS_VALU" and re.search(r"\b" + kernel + r"\b", r["Kernel_Name"])]
        return sum(vals) / len(vals) if vals else None
    except Exception:
        return None


def cpu_baseline(clips, rate, budget_clips):
    """The CPU oracle ("port": C double-precision restatement) on a bounded sample: one thread (`value`, the
    contract's figure) and, as `all_cores`, utterance-parallel over the host's cores the way the reference runs
    one process per voice (config.yaml:58)."""
    sample = clips[:budget_clips]
    t0 = time.perf_counter()
    for c in sample:
        _cpu_one((c, rate))
    dt = time.perf_counter() - t0
    secs = sum(len(c) for c in sample) / rate
    out = {"value": secs / dt, "unit": "audio-seconds/sec", "cores": 1, "kind": "port",
           "sample": f"{len(sample)} of the same synthetic 10 s clips, oracle/pce_oracle.c + numpy, {dt:.1f} s of CPU time"}
    try:
        # worker PROCESSES (fork): called before this process touches the GPU, see main()
        import multiprocessing as mp
        workers = max(1, min(len(sample), os.cpu_count() or 1, 64))
        with mp.get_context("fork").Pool(workers) as pool:
            pool.map(_cpu_one, [(sample[0], rate)] * workers)      # start-up (library load) outside the timed part
            t0 = time.perf_counter()
            pool.map(_cpu_one, [(c, rate) for c in sample], chunksize=1)
            dtp = time.perf_counter() - t0
        out["all_cores"] = {"value": secs / dtp, "processes": workers, "host_cores": os.cpu_count(), "seconds": dtp}
    except Exception as e:                                         # the single-thread figure is the contract's
        out["all_cores"] = {"error": str(e)}
    try:
        # reference-shaped: paths in, seven decodes per file (SURVEY.md section 8d iii), one thread, 32 clips
        import tempfile
        import wave
        sub = sample[:32]
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for i, c in enumerate(sub):
                pth = os.path.join(td, f"clip{i}.wav")
                with wave.open(pth, "wb") as w:
                    w.setnchannels(1); w.setsampwidth(2); w.setframerate(rate); w.writeframes(c.astype("<i2").tobytes())
                paths.append(pth)
            t0 = time.perf_counter()
            for pth in paths:
                _cpu_one_reference_shaped((pth, rate))
            dtr = time.perf_counter() - t0
        out["reference_shaped"] = {"value": sum(len(c) for c in sub) / rate / dtr, "clips": len(sub), "decodes_per_file": 7, "cores": 1,
                                   "seconds": dtr}
    except Exception as e:
        out["reference_shaped"] = {"error": str(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clips", type=int, default=256, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--cpu-clips", type=int, default=256, help="clips timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--whisper-model", choices=["tiny", "base", "small", "medium"], default="small",
                    help="architecture of the c3 workload (BASELINE.json names small; the reference's config.yaml:15 default is medium)")
    ap.add_argument("--workload", choices=["c2", "c3"], default="c2",
                    help="c2 (default, BASELINE.json configs[1]): F0+energy+LUFS+STFT; c3: c2 + log-mel + Whisper-small encoder")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import prosody_control_french_tts_amd as pkg
    from prosody_control_french_tts_amd import shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rate = 16000
    n_samples = int(round(args.seconds * rate))
    # synthetic data of the workload's shape; rank r owns clips [r*clips, (r+1)*clips) (weak scaling)
    clips = synth.synth_batch(args.clips, args.seconds, rate, first=rank * args.clips)
    # CPU baseline first: its all-cores leg forks worker processes, which must happen before this process
    # initialises the GPU
    # (rank 0 of a single-GPU run only: the multi-GPU lines carry "cpu_baseline": null)
    cpu = cpu_baseline(clips, rate, args.cpu_clips) if (world == 1 and rank == 0 and args.cpu_clips > 0) else None
    if args.gpus > 1 or world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    eng = pkg.ProsodyEngine(local_rank)
    eng.upload(clips, rate)                      # inputs resident in HBM before the timed region
    sl = eng.whole_clip_slices()
    params = pkg.PitchParams.praat(float(os.environ.get("PCE_BENCH_FLOOR", "150")), 600.0)   # reference: floor 150 (Code/audioPipeline.py:329)
    off, _ = eng.pitch_plan(sl, params)
    n_pitch_frames = int(off[1] - off[0])
    n_stft_frames = 1 + n_samples // 256

    wdims = None
    if args.workload == "c3":
        from prosody_control_french_tts_amd import whisper_weights as WW
        wdims = WW.DIMS[args.whisper_model]
        eng.whisper_load(wdims, WW.pack(WW.synthetic_weights(wdims), wdims))       # random-init weights of the architecture
        tdims = WW.TEXT_DIMS[args.whisper_model]
        eng.whisper_decoder_load(tdims, WW.pack_decoder(WW.synthetic_decoder_weights(tdims), tdims))
        trng = np.random.default_rng(5)
        sot_len = 3                                    # <|sot|><|fr|><|transcribe|> ... <|eot|>: synthetic ids of a 10 s utterance's length
        align_tokens = [trng.integers(0, tdims["n_vocab"], size=int(trng.integers(24, 48))).tolist() for _ in range(args.clips)]
        align_frames = [n_samples // 160] * args.clips

    # One step = one pass of the hot path over the batch: launch() enqueues every kernel of the pass and the
    # asynchronous copy of the per-utterance statistics; finish() waits for that copy only, builds the 7-stat
    # record and all-gathers it.  Consecutive steps are software-pipelined (step i+1 is enqueued before step i's
    # record is read), so the device does not idle while the host unpacks: two statistic slots are in flight.
    def launch(slot):
        if wdims:
            eng.logmel_run(wdims["n_mels"])
            eng.whisper_encode_run()
            eng.whisper_align_run(align_tokens, align_frames, sot_len)   # teacher-forced decoder + cross-attention weights + DTW
        eng.energy_run(sl, 500)
        eng.lufs_run(sl)
        eng.pitch_run(sl, params)
        eng.stft_db_run(1024, 256)
        eng.stats_enqueue(slot)

    def finish(slot):
        r = eng.stats_wait(slot)
        en, lu, pi = r["energy"], r["lufs"][0], r["pitch"]
        # per-utterance record: [median F0, LUFS, rms, peak, silence ratio, duration, n_voiced]
        rec = np.stack([pi["median_f0"], lu, np.sqrt(en["sum_sq"] / np.maximum(en["n"], 1)), en["peak_abs"].astype(np.float64),
                        1.0 - en["n_loud"] / np.maximum(en["n"], 1), en["n"] / float(rate), pi["n_voiced"].astype(np.float64)], axis=1)
        return shard.allgather_records(rec)

    def run_steps(k):
        rec = None
        for i in range(k):
            launch(i & 1)
            if i > 0:
                rec = finish((i - 1) & 1)
        if k > 0:
            rec = finish((k - 1) & 1)
        return rec

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    run_steps(args.warmup)
    if not args.no_profile:
        eng.profile_enable(True)
        eng.profile_reset()
    fence()
    t0 = time.perf_counter()
    rec = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    prof = eng.profile() if not args.no_profile else {}
    eng.profile_enable(False)

    t_max = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())
    audio_seconds = args.clips * args.seconds * world * args.steps
    assert rec.shape == (args.clips * world, 7)

    if rank == 0:
        # per-kernel figures (HIP events on the engine's stream around every launch)
        kernels = []
        for name, p in prof.items():
            avg_ms = p["total_ms"] / p["launches"]
            kernels.append({"kernel": name, "avg_ms": avg_ms, "launches_per_step": p["launches"] / args.steps,
                            "ms_per_step": p["total_ms"] / args.steps})
        kernels.sort(key=lambda k: -k["ms_per_step"])
        kt = {k["kernel"]: k for k in kernels}
        # stages: a stage's algorithmic bytes are moved ONCE by its kernels together; the
        # intermediates between them (candidates, autocorrelations, chunk states) are not algorithmic traffic
        pcm = 2.0 * n_samples * args.clips
        f0_out = 8.0 * n_pitch_frames * args.clips
        stft_out = 513 * 4.0 * n_stft_frames * args.clips
        stages = [
            ("energy (R3/R7)", ["k_energy"], pcm, None),
            ("lufs (R4)", ["k_lufs_pass1", "k_lufs_scan", "k_lufs_pass2", "k_lufs_gate"], pcm, None),
            ("f0 (R1: Praat AC + path + median)", ["k_pitch_frames", "k_pitch_refine", "k_pitch_delta", "k_pitch_path", "k_pitch_median"],
             pcm + f0_out, None),
            ("stft-dB (R10)", ["k_stft_max", "k_stft_db", "k_stft_norm"], pcm + stft_out, None),
        ]
        if wdims:
            d, L = wdims["n_state"], wdims["n_layer"]
            flop = args.clips * (2.0 * 3000 * d * 240 + 2.0 * 1500 * d * 3 * d
                                 + L * (2.0 * 1500 * d * 3 * d + 4.0 * 1500 * 1500 * d + 2.0 * 1500 * d * d + 16.0 * 1500 * d * d))
            stages += [("log-mel (R8)", ["k_logmel"], pcm + 80 * 3000 * 4.0 * args.clips, None),
                       (f"whisper-{args.whisper_model} encoder (R8)", ["whisper_encoder"], None, flop)]
            # (the forced-alignment leg -- decoder over 24-48 tokens per clip, alignment heads, DTW -- is timed as `whisper_align` in `kernels`)
        # k_energy runs three times per step (gate, LUFS peak, pitch peak): split its time over the users
        rows = []
        for name, ks, nbytes, flops in stages:
            ms = sum(kt[k]["ms_per_step"] for k in ks if k in kt)
            if name.startswith("energy") and "k_energy" in kt:
                ms = kt["k_energy"]["avg_ms"]
            if ms <= 0:
                continue
            dom = max((k for k in ks if k in kt), key=lambda k: kt[k]["ms_per_step"])
            if flops is None:
                ach = nbytes / (ms * 1e-3) / 1e9
                rows.append({"stage": name, "kernels": ks, "dominant_kernel": dom, "ms_per_step": ms, "bound": "hbm",
                             "algorithmic_bytes": nbytes, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS})
            else:
                ach = flops / (ms * 1e-3) / 1e12
                rows.append({"stage": name, "kernels": ks, "dominant_kernel": dom, "ms_per_step": ms, "bound": "mfma",
                             "algorithmic_flops": flops, "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": ach / MFMA_BF16_PEAK_TFLOPS})
        rows.sort(key=lambda r: -r["ms_per_step"])
        traffic = load_pmc_traffic()
        roofline = None
        if rows:
            r0 = rows[0]
            t0 = traffic.get(r0["dominant_kernel"]) if traffic else None
            roofline = {"kernel": r0["dominant_kernel"], "stage": r0["stage"], "bound": r0["bound"], "achieved": r0["achieved"],
                        "peak": r0["peak"], "unit": r0["unit"], "frac": r0["frac"], "traffic": t0,
                        "note": "dominant stage by device time; `achieved` = the stage's algorithmic bytes (or flops) / the device time "
                                "of the stage's kernels.  The F0 stage is fp64-VALU bound (about 1e3 flop per algorithmic byte, "
                                "DESIGN.md section 3): its HBM fraction is small by construction; `stages` lists every stage.  "
                                "Issue-slot view of the dominant kernel (profiles/r01/pmc_sq_counter_collection.csv): "
                                "`valu_f64` prices k_pitch_refine's SQ_INSTS_VALU at the measured 4.6 cycles per fp64 wave-instruction "
                                "against the 1024 SIMDs' issue capacity over the kernel's duration.",
                        "valu_f64": {"kernel": "k_pitch_refine", "insts_valu_per_launch": load_pmc_valu("k_pitch_refine"), "cycles_per_inst": 4.6,
                                     "simds": 1024, "clock_ghz": 2.1,
                                     "issue_frac": ((load_pmc_valu("k_pitch_refine") or 0.0) * 4.6 / (1024 * 2.1e9)) / max(kt["k_pitch_refine"]["avg_ms"] * 1e-3, 1e-9)
                                     if "k_pitch_refine" in kt else None,
                                     "source": "SQ_INSTS_VALU from profiles/r01 (separate rocprofv3 --pmc pass of this command)"}}
        info = eng.device_info()
        print(json.dumps({
            "metric": "audio-seconds/sec prosody+align throughput, 16 kHz French",
            "value": audio_seconds / dt, "unit": "audio-seconds/sec (x real-time)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload.upper()}: {args.clips} synthetic {args.seconds:g} s 16 kHz mono clips per GPU, "
                                   "energy/gate + BS.1770 LUFS + Praat-AC F0 150-600 Hz (path finder, voiced median) + STFT-dB 1024/256; "
                                   + (f"+ log-mel + Whisper-{args.whisper_model} encoder + teacher-forced decoder / cross-attention DTW alignment (synthetic weights and token ids, bf16 MFMA)" if wdims else "Whisper-encoder alignment (C3) not included: --workload c3"),
                       "clips_per_gpu": args.clips, "clip_seconds": args.seconds, "sample_rate": rate,
                       "parallelism": f"utterance-sharded x{world}, one all-gather of 7 fp64 stats per clip"},
            "roofline": roofline, "stages": rows, "kernels": kernels, "pmc_traffic_bytes_per_launch": traffic, "cpu_baseline": cpu,
            "device": info["name"], "host_cores": os.cpu_count(),
        }))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
