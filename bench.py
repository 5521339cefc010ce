#!/usr/bin/env python3
"""Benchmark of the FunCodec encode+decode hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1 without WORLD_SIZE in the env: bench.py spawns its own N ranks, one per
                                                          GPU, through torch.distributed.run on a free port of 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one Speech2Token(run_mod="inference") pass (encode -> 32-stage RVQ -> decode) over this rank's utterances of
synthetic 10 s / 16 kHz audio on the 16k-nq32ds640 architecture:
  N = 1: BASELINE.json configs[1], 16 x 10 s in ONE engine call;
  N > 1: BASELINE.json configs[2], 128 utterances per GPU (1024 at N = 8) walked in micro-batches of 32 (every op of the path is
         per-utterance, results do not depend on the micro-batch), the int64 code indices all-gathered over RCCL inside the step.
`--workload freqcodec` is a SIDE measurement of the next scope row (BASELINE.json configs[3]: the STFT-domain FreqCodec recipe, 64 x 10 s
on one GPU); the contract metric stays the default.
Inputs are resident in HBM when timing starts.  The timed region runs WITHOUT the in-engine HIP-event brackets; the per-kernel
table (and with it `roofline`) comes from a second, separate pass of the same step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# utterances per engine call.  N = 1 is Config B itself (16 utterances, one call).  For N > 1 (Config C, 128 per GPU) the micro-batch
# is free (every op is per-utterance): 32 doubles the tile count of the layers at the bottleneck frame rate; the persistent LSTM
# (H = 1024 fills the chip with one 16-utterance tile) runs its second tile as a second launch.  Measured on one GPU at Config C's
# per-GPU rate (64 x 10 s): 62.7 ms in micro-batches of 32 = 10 205 audio-s/s, 63.3 ms in micro-batches of 16; FC_BENCH_MICRO overrides.
MICRO_BATCH = int(os.environ.get("FC_BENCH_MICRO", "0")) or (16 if int(os.environ.get("WORLD_SIZE", "1")) == 1 else 32)
SAMPLES = 160000
CONFIG = os.environ.get("FC_BENCH_CONFIG", "ds640")   # the contract metric is ds640; other recipes only for side measurements
PEAK_F32_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: fp32 matrix (= vector) peak
PEAK_HBM_TBS = 8.0
# clock the chip sustains inside these kernels (PMC GRBM_GUI_ACTIVE over the conv / LSTM kernels, profiles/r0*_pmc_layers.txt: 2.27 - 2.35 GHz
# against the 2.4 GHz the 157.3 TF peak is quoted at): the roof the silicon actually offers under this load.  Reported NEXT TO the
# nominal fraction, never instead of it.
SUSTAINED_GHZ, NOMINAL_GHZ = 2.31, 2.40
PEAK_F32_SUSTAINED = PEAK_F32_TFLOPS * SUSTAINED_GHZ / NOMINAL_GHZ
RIDGE = PEAK_F32_TFLOPS / PEAK_HBM_TBS      # FLOP per byte above which the fp32 roof is the tighter one
CONV_CLASSES = ("conv_", "reshead_", "gconv")   # kernel classes of the conv / transposed-conv layers (fc_engine_profile names)


def physical_cores() -> int:
    """Distinct (package, core) pairs of /proc/cpuinfo; falls back to os.cpu_count()."""
    try:
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return os.cpu_count() or 1


def is_freq() -> bool:
    return CONFIG.startswith(("freqmp", "tinyfreq"))


def synthetic_state(cfg, arch):
    from funcodec_amd.synth import make_freq_state_dict, make_state_dict
    return make_freq_state_dict(cfg, 0) if is_freq() else make_state_dict(arch, 0)


def cpu_baseline(utts: int = MICRO_BATCH):
    """the CPU PyTorch path on Config B's batch (16 x 10 s), 1 warm-up + median of 3, thread count stated.
    What runs is the oracle (ATen-CPU restatement of the reference path, pinned bit-exact against the real reference in the
    build container; the reference itself cannot travel to the GPU box) -> kind "port"."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from funcodec_amd.config import arch_from_config, recipe_config
    from funcodec_amd.synth import synthetic_audio
    cfg = recipe_config(CONFIG)
    sd = synthetic_state(cfg, arch_from_config(cfg))
    if is_freq():
        from freq_oracle import FreqOracle as Oracle
        utts = min(utts, 4)                           # bounded sample: the 2-D net costs ~10x the 1-D one per audio second on CPU
    else:
        from torch_oracle import Oracle
    orc = Oracle(cfg, {k: torch.from_numpy(v) for k, v in sd.items()})
    default_threads = torch.get_num_threads()
    cores = physical_cores()
    x = torch.from_numpy(synthetic_audio(utts, SAMPLES, 1234))
    t_all = time.perf_counter()
    # thread count: a short probe (2 utterances) over {physical cores, 32, 16}; oversubscribed ATen thread pools lose badly
    # on this path (round 1: 128 threads 4.0 audio-s/s, 32 threads 14.9), so the CPU side gets its best setting
    probe = {}
    for threads in sorted({min(cores, default_threads), min(32, default_threads), min(16, default_threads)}, reverse=True):
        torch.set_num_threads(threads)
        orc.inference(x[:1, :16000])
        t0 = time.perf_counter()
        orc.inference(x[:2])
        probe[threads] = time.perf_counter() - t0
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    orc.inference(x)                                  # warm-up on the full batch
    runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        orc.inference(x)
        runs.append(time.perf_counter() - t0)
    torch.set_num_threads(default_threads)
    med = statistics.median(runs)
    audio_s = utts * SAMPLES / 16000.0
    return {"value": round(audio_s / med, 3), "unit": "audio-s/s", "cores": best, "kind": "port",
            "physical_cores": cores, "logical_cpus": os.cpu_count(), "torch": torch.__version__,
            "port_over_reference": port_over_reference(),
            "runs_s": [round(r, 3) for r in runs], "median_s": round(med, 3),
            "probe_audio_s_per_s": {str(k): round(2 * SAMPLES / 16000.0 / v, 2) for k, v in probe.items()},
            "seconds": round(time.perf_counter() - t_all, 1),
            "sample": f"oracle/{'freq_oracle' if is_freq() else 'torch_oracle'}.py (same ATen CPU kernels as the reference's PyTorch path), "
                      + ("a bounded sample of the benchmark batch " if is_freq() else "the FULL benchmark batch ") +
                      f"({utts} x 10 s, {CONFIG}, n_q=32, run_mod=inference), 1 warm-up + median of 3 at {best} threads (best of a "
                      f"2-utterance probe over physical-core / 32 / 16 threads)"}


def port_over_reference():
    """The reference's own Speech2Token(device="cpu") cannot travel to the GPU box; the ratio port / reference (audio-s/s of
    oracle/torch_oracle.py over audio-s/s of the real reference, same 4 x 10 s batch, threads and box) was measured once in the build
    container by tools/port_over_reference.py and is carried here from the committed record (VERDICT r5 #6b)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_port_over_reference.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        return {"ratio": d["port_over_reference"], "reference_audio_s_per_s": d["reference_audio_s_per_s"],
                "port_audio_s_per_s": d["port_audio_s_per_s"], "threads": d["threads"], "utterances": d["utterances"],
                "where": "build container (the reference does not exist on the GPU box)", "source": os.path.basename(files[-1])}
    except (OSError, KeyError, ValueError):
        return None


def pmc_traffic(kernel: str, tag: str = ""):
    """HBM bytes per launch of `kernel` from the committed PMC passes (tools/collect_profiles.sh: FETCH_SIZE and WRITE_SIZE in
    separate rocprofv3 --pmc runs of this same benchmark).  Correction per MI355X_MICROARCH.md: FETCH_SIZE x 2 on gfx950
    (checked here on the 32->32 k=1 convs whose byte count is known: 0.320 GB reported for 0.656 GB read), WRITE_SIZE x 1
    (0.641 GB reported for 0.656 GB written).  None if the committed profile has no entry for this instantiation."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_hbm_traffic_pmc{tag}.json")))      # tag "_freqcodec": the FreqCodec side's passes
    if not files:
        return None
    try:
        base = kernel.split("<")[0]
        # the engine's class names carry the leading template arguments only (gconv2d_kernel<4, 2, 3, 3, 1, false>); rocprofv3 prints all of
        # them (..., false, 2, false>): a class matches the instantiations it is a prefix of
        stem = kernel[:-1] + "," if kernel.endswith(">") else None
        for k in json.load(open(files[-1]))["per_kernel"]:
            if (k["kernel"] == kernel or (stem and k["kernel"].startswith(stem)) or
                    (base == "lstm_persist_kernel" and k["kernel"].startswith(base))) and k["launches"]:
                return {"bytes_per_launch": round((2.0 * k["fetch_kb"] + k["write_kb"]) * 1024.0 / k["launches"]),
                        "fetch_x2_gb": round(2.0 * k["fetch_kb"] * 1024.0 / 1e9, 3), "write_gb": round(k["write_kb"] * 1024.0 / 1e9, 3),
                        "launches": k["launches"], "source": os.path.basename(files[-1])}
    except (OSError, KeyError, ValueError):
        pass
    return None


def pmc_step_traffic(tag: str = ""):
    """Whole-step HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE over every kernel of the last benchmark step) from the committed PMC passes."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_hbm_traffic_pmc{tag}.json")))
    try:
        t = json.load(open(files[-1]))["step_total"]
        return {"bytes_per_step": round((t["fetch_gb_x2"] + t["write_gb"]) * 1e9), "source": os.path.basename(files[-1])}
    except (IndexError, OSError, KeyError, ValueError):
        return None


def _sustained(obj):
    """Add the fraction of the roof at the SUSTAINED clock next to the nominal one (fp32 roofs only) and the measured / algorithmic
    traffic ratio."""
    if obj.get("unit") == "TFLOP/s" and obj.get("achieved"):
        obj["sustained_clock_ghz"] = SUSTAINED_GHZ
        obj["peak_at_sustained_clock"] = round(PEAK_F32_SUSTAINED, 1)
        obj["frac_at_sustained_clock"] = round(obj["achieved"] / PEAK_F32_SUSTAINED, 4)
    if obj.get("traffic") and obj.get("algorithmic_bytes_per_launch"):
        obj["traffic_over_algorithmic"] = round(obj["traffic"] / obj["algorithmic_bytes_per_launch"], 2)
    return obj


def laura_step_pmc_traffic():
    """HBM bytes per LauraTTS decoding step from the committed PMC table (tools/pmc_laura.sh -> profiles/r*_pmc_laura.txt): the step
    kernels' (fetch x 2 + write) x launches, divided by the number of sampler launches (one per step).  None without the profile."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_laura.txt")))
    if not files:
        return None
    tot, steps = 0.0, 0
    for ln in open(files[-1]):
        f = ln.split()
        if len(f) < 8 or not f[0].startswith("fc::laura::"):
            continue
        # columns: kernel(may hold a space after a comma) ... launches us GHz busy insts fetchMB writeMB
        try:
            launches, fetch_mb, write_mb = int(f[-7]), float(f[-2]), float(f[-1])
        except ValueError:
            continue
        if any(k in ln for k in ("step_persist_kernel", "gemv_kernel", "attn_step_kernel", "sample_kernel", "layernorm_rows_kernel")):
            tot += launches * (fetch_mb + write_mb) * 1e6
        if "sample_kernel" in ln:
            steps = launches
    return {"bytes_per_step": round(tot / steps), "source": os.path.basename(files[-1])} if steps else None


def kernel_rooflines(prof, prof_steps, pmc_tag: str = ""):
    """Per-kernel-class table of a fc_engine_profile pass + the two roofline objects (dominant MFMA-bound conv class against the
    fp32 matrix peak; the HBM-bound thin classes against 8 TB/s)."""
    out = {}
    kern = []
    for p in prof:
        if p["launches"] == 0:
            continue
        ms = p["total_ms"]
        tfl = p["flops"] / (ms * 1e-3) / 1e12 if ms > 0 else None
        gbs = p["bytes"] / (ms * 1e-3) / 1e9 if ms > 0 and p["bytes"] else None
        # which roof binds this class: algorithmic intensity against the ridge (157.3 TF / 8 TB/s = 19.7 FLOP/B)
        bound = None
        if p["bytes"] and p["kernel"].startswith(CONV_CLASSES):
            bound = "mfma" if p["flops"] / p["bytes"] >= RIDGE else "hbm"
        kern.append({"kernel": p["kernel"], "launches_per_step": p["launches"] // prof_steps,
                     "ms_per_step": round(ms / prof_steps, 3),
                     "avg_us_per_launch": round(ms * 1e3 / p["launches"], 2),
                     "tflops": round(tfl, 2) if tfl else None,
                     "alg_gbs": round(gbs, 1) if gbs else None,
                     "bound": bound,
                     "f32_frac": round(tfl / PEAK_F32_TFLOPS, 4) if tfl else None,
                     # a fraction above 1 can only come from an algorithmic byte count that is not one: the value stays visible and is
                     # flagged (nulling it hid accounting errors); the line's `accounting_errors` lists such classes
                     "hbm_frac": round(gbs / 1e3 / PEAK_HBM_TBS, 4) if gbs else None})
        if gbs and gbs / 1e3 > PEAK_HBM_TBS:
            kern[-1]["hbm_frac_invalid"] = True
    convs = [k for k in kern if k["kernel"].startswith(CONV_CLASSES)]
    mfma_bound = [k for k in convs if k["bound"] != "hbm"]
    if mfma_bound:
        dom = max(mfma_bound, key=lambda k: k["ms_per_step"])
        conv_ms = sum(k["ms_per_step"] for k in convs)
        conv_fl = sum(p["flops"] for p in prof if p["kernel"].startswith(CONV_CLASSES)) / prof_steps
        traffic = pmc_traffic(dom["kernel"], pmc_tag)
        out["roofline"] = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"],
                           "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(dom["tflops"] / PEAK_F32_TFLOPS, 4),
                           "traffic": (traffic or {}).get("bytes_per_launch"), "traffic_detail": traffic,
                           "algorithmic_bytes_per_launch": round(dom["alg_gbs"] * 1e9 * dom["avg_us_per_launch"] * 1e-6) if dom["alg_gbs"] else None,
                           "avg_us_per_launch": dom["avg_us_per_launch"],
                           "launches_per_step": dom["launches_per_step"],
                           "hbm_alg_gbs": dom["alg_gbs"],
                           "all_conv_instantiations": {"ms_per_step": round(conv_ms, 3),
                                                       "tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                                                       "frac": round(conv_fl / (conv_ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS, 4)},
                           "note": "fp32-input MFMA (exact fp32, peak = fp32 vector peak); achieved = algorithmic FLOPs of this kernel's "
                                   f"launches / sum of their HIP-event durations over a SEPARATE {prof_steps}-step pass (not the timed region)"}
    # `roofline` describes the DOMINANT kernel of the step = the class with the largest share of the step's kernel time, whatever it
    # is.  On Config B that is the persistent LSTM recurrence (latency-bound: priced against the fp32 matrix peak like the convs);
    # the dominant implicit-GEMM conv class keeps its own object (`roofline_conv`).
    top = max(kern, key=lambda k: k["ms_per_step"]) if kern else None
    if top is not None and "roofline" in out and top["kernel"] != out["roofline"]["kernel"] and top["tflops"]:
        out["roofline_conv"] = out["roofline"]
        ttr = pmc_traffic(top["kernel"], pmc_tag)
        out["roofline"] = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["tflops"], "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(top["tflops"] / PEAK_F32_TFLOPS, 4), "traffic": (ttr or {}).get("bytes_per_launch"),
                           "traffic_detail": ttr,
                           "algorithmic_bytes_per_launch": round(top["alg_gbs"] * 1e9 * top["avg_us_per_launch"] * 1e-6) if top["alg_gbs"] else None,
                           "avg_us_per_launch": top["avg_us_per_launch"], "launches_per_step": top["launches_per_step"],
                           "ms_per_step": top["ms_per_step"],
                           "share_of_kernel_time": round(top["ms_per_step"] / max(1e-9, sum(k["ms_per_step"] for k in kern)), 4),
                           "note": "top kernel class of the step by time.  lstm_persist_kernel: one launch = the whole 2-layer recurrence of a SLSTM "
                                   "block; algorithmic FLOPs = 2 * B * 4H * 3H per wavefront step; algorithmic bytes per SURVEY.md 8d = recurrent weights "
                                   "read ONCE + x-projection in + y out (the PMC traffic above that is the per-step hidden-state exchange); it is bound by that "
                                   "exchange (a grid-wide all-gather + barrier per step), not by the matrix pipe: frac is its distance from the fp32 MFMA roof"}
    # the LSTM recurrence and the dominant conv class take turns as the step's top class (2.5 vs 2.6 ms): the recurrence always gets its own
    # object as well, so that the line reads the same whichever is on top
    lst = next((k for k in kern if k["kernel"].startswith("lstm_persist_kernel") and k["tflops"]), None)
    if lst is not None and "roofline" in out and out["roofline"]["kernel"] != lst["kernel"]:
        ltr = pmc_traffic(lst["kernel"], pmc_tag)
        out["roofline_lstm"] = {"bound": "mfma", "kernel": lst["kernel"], "achieved": lst["tflops"], "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(lst["tflops"] / PEAK_F32_TFLOPS, 4), "traffic": (ltr or {}).get("bytes_per_launch"),
                                "algorithmic_bytes_per_launch": round(lst["alg_gbs"] * 1e9 * lst["avg_us_per_launch"] * 1e-6) if lst["alg_gbs"] else None,
                                "avg_us_per_launch": lst["avg_us_per_launch"], "launches_per_step": lst["launches_per_step"], "ms_per_step": lst["ms_per_step"],
                                "note": "latency-bound recurrence (a grid-wide all-gather + barrier per step): frac is its distance from the fp32 MFMA roof"}
    for key in ("roofline", "roofline_conv", "roofline_lstm"):
        if key in out:
            _sustained(out[key])
    if "roofline_conv" in out:
        # the driver's parsed record keeps `roofline` only: carry the dominant conv class inside it as well
        rc = out["roofline_conv"]
        out["roofline"]["conv_class"] = {k: rc.get(k) for k in ("kernel", "achieved", "frac", "frac_at_sustained_clock", "traffic",
                                                                "algorithmic_bytes_per_launch", "traffic_over_algorithmic",
                                                                "avg_us_per_launch", "launches_per_step")}
        out["roofline"]["conv_class"]["all_conv_instantiations"] = rc.get("all_conv_instantiations")
    hbm = [k for k in convs if k["bound"] == "hbm"]
    if hbm:   # the HBM-bound (thin, C <= 64) classes: north_star's roof
        hdom = max(hbm, key=lambda k: k["ms_per_step"])
        hb_ms = sum(k["ms_per_step"] for k in hbm)
        hb_by = sum(p["bytes"] for p in prof if any(p["kernel"] == k["kernel"] for k in hbm)) / prof_steps
        htr = pmc_traffic(hdom["kernel"], pmc_tag)
        out["roofline_hbm"] = {"bound": "hbm", "kernel": hdom["kernel"], "achieved": hdom["alg_gbs"], "peak": PEAK_HBM_TBS * 1e3,
                               "unit": "GB/s", "frac": hdom["hbm_frac"], "traffic": (htr or {}).get("bytes_per_launch"),
                               "avg_us_per_launch": hdom["avg_us_per_launch"], "launches_per_step": hdom["launches_per_step"],
                               "all_hbm_bound_conv_classes": {"ms_per_step": round(hb_ms, 3),
                                                              "alg_gbs": round(hb_by / (hb_ms * 1e-3) / 1e9, 1),
                                                              "frac": round(hb_by / (hb_ms * 1e-3) / 1e12 / PEAK_HBM_TBS, 4)}}
    out["kernels"] = kern
    out["accounting_errors"] = [k["kernel"] for k in kern if k.get("hbm_frac_invalid")]    # must be empty
    return out

def config_c_shard_side(eng, n_q: int, utts: int = 128, micro: int = 32, steps: int = 3, warmup: int = 1):
    """The per-rank work of BASELINE.json configs[2] on ONE GPU: 128 x 10 s utterances walked in micro-batches of 32 -- exactly what each
    rank of `--gpus N` (N > 1) does per step, minus the all_gather of the codes.  It is the same-shape base a future N-GPU value has to
    be divided by (the N = 1 contract line is Config B, 16 utterances in one call, a different shape)."""
    from funcodec_amd.synth import synthetic_audio
    old = eng.micro_batch
    eng.micro_batch = max(old, micro)
    wav = torch.from_numpy(synthetic_audio(utts, SAMPLES, 1234)).cuda()

    def step():
        parts = [eng.encode_decode(wav[i:i + micro], n_q, use_scale=True)["codes"] for i in range(0, utts, micro)]
        return torch.cat(parts, 1)

    eng.set_profiling(False)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        codes = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    eng.check_status()
    assert codes.shape[1] == utts
    work = eng.work(micro, SAMPLES, n_q)
    nmb = utts / micro
    out = {"workload": f"per-rank work of BASELINE.json configs[2] on one GPU: {utts} x 10 s (ds640, n_q=32, run_mod=inference) in micro-batches "
                       f"of {micro}; no gather (single rank)",
           "value": round(utts * SAMPLES / 16000.0 / dt, 1), "unit": "audio-s/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
           "ms_per_micro_batch": round(dt * 1e3 / nmb, 3),
           "whole_step": {"tflops": round(work["total_flops"] * nmb / dt / 1e12, 2),
                          "frac_of_f32_peak": round(work["total_flops"] * nmb / dt / 1e12 / PEAK_F32_TFLOPS, 4)},
           "use": "scaling base: efficiency(N) of a `--gpus N` line = value(N) / (N x this value)"}
    eng.micro_batch = old
    del wav, codes
    torch.cuda.empty_cache()
    return out


def freqcodec_side(config: str = "freqmpgr1", utts: int = 64, micro: int = 32, steps: int = 5, warmup: int = 2):
    """Side measurement of BASELINE.json configs[3] (STFT-domain FreqCodec, batch 64 x 10 s on one GPU): the recipe net with
    conv_group_ratio = tr_conv_group_ratio = 1 (the grouped, bandwidth-bound shape of the released "gr1" model)."""
    from funcodec_amd.config import arch_from_config, recipe_config
    from funcodec_amd.model import EncodecMI355X
    from funcodec_amd.synth import make_freq_state_dict, synthetic_audio
    cfg = recipe_config(config)
    arch = arch_from_config(cfg)
    model = EncodecMI355X(arch, "cuda:0")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in make_freq_state_dict(cfg, 0).items()})
    eng = model.engine
    eng.micro_batch = max(eng.micro_batch, micro)
    wav = torch.from_numpy(synthetic_audio(utts, SAMPLES, 1234)).cuda()
    n_q = arch.num_quantizers

    def step():
        for i in range(0, utts, micro):
            r = eng.encode_decode(wav[i:i + micro], n_q, use_scale=True)
        return r

    eng.set_profiling(False)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    eng.check_status()
    assert bool(torch.isfinite(r["recon"]).all())
    eng.set_profiling(True)
    psteps = max(1, min(3, steps))
    for _ in range(psteps):
        eng.encode_decode(wav[:micro], n_q, use_scale=True)
    prof = eng.read_profile()
    eng.set_profiling(False)
    work = eng.work(micro, SAMPLES, n_q)
    nmb = utts / micro
    tab = kernel_rooflines(prof, psteps, "_freqcodec" if config == "freqmpgr1" else "_none")
    out = {"workload": f"BASELINE.json configs[3] shape: FreqCodec mag_phase recipe + conv_group_ratio = tr_conv_group_ratio = 1 ({config}), "
                       f"{utts} x 10 s on one GPU in engine calls of {micro}, n_q=32, run_mod=inference",
           "value": round(utts * SAMPLES / 16000.0 / dt, 1), "unit": "audio-s/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
           "whole_step": {"tflops": round(work["total_flops"] * nmb / dt / 1e12, 2),
                          "alg_hbm_tbs": round(work["total_bytes"] * nmb / dt / 1e12, 3),
                          "frac_of_hbm_peak": round(work["total_bytes"] * nmb / dt / 1e12 / PEAK_HBM_TBS, 4)},
           "roofline_hbm": tab.get("roofline_hbm"), "roofline": tab.get("roofline"),
           "kernels": sorted(tab["kernels"], key=lambda k: -k["ms_per_step"])[:12]}
    ptr = pmc_step_traffic("_freqcodec") if config == "freqmpgr1" else None     # PMC passes of ONE engine call of `micro` utterances
    if ptr and micro == 32:
        out["whole_step"]["traffic"] = round(ptr["bytes_per_step"] * nmb)
        out["whole_step"]["traffic_over_algorithmic"] = round(ptr["bytes_per_step"] / work["total_bytes"], 2)
        out["whole_step"]["traffic_source"] = ptr["source"]
    del model, eng
    torch.cuda.empty_cache()
    return out


def laura_side(batch: int = 8, text_len: int = 100, prompt_frames: int = 75, new_frames: int = 250, steps: int = 3, warmup: int = 1,
               cpu_sample: bool = True):
    """Side measurement of BASELINE.json configs[4]: LauraTTS zero-shot generation, batch = 8 prompts on one GPU.  One step = the
    whole Text2Audio flow for the batch with device-resident inputs: text encoder -> autoregressive decode_codec (prompt tokens +
    `new_frames` new frames per prompt, top-k 25 sampling like the recipe's --sampling 25) -> fine codec predictor -> ds640 codec
    decoder (decode_emb).  Synthetic seeded checkpoints of the recipe's sizes (88 M-parameter LauraTTS over a phoneme token list,
    57.6 M ds640 codec)."""
    from funcodec_amd.config import arch_from_config, recipe_config
    from funcodec_amd.laura import LauraGenMI355X
    from funcodec_amd.laura_config import laura_recipe_config, laura_spec_from_config
    from funcodec_amd.model import EncodecMI355X
    from funcodec_amd.synth import make_laura_state_dict, make_state_dict, synthetic_text
    lcfg = laura_recipe_config("lauraphn")
    spec = laura_spec_from_config(lcfg)
    ccfg = recipe_config("ds640")
    arch = arch_from_config(ccfg)
    csd = make_state_dict(arch, 0)
    lsd = make_laura_state_dict(lcfg, 0)
    lsd["quantizer_codebook.embed"] = csd["quantizer.rq.model.embed"][: spec.num_quantizers].copy()
    # fixed work per step: a random-weight LM would sample <eos> at random frames; its logits are pushed out of reach so that every
    # prompt generates exactly `new_frames` frames (a trained model ends at <eos>; the per-frame cost is what is measured)
    for g_ in range(spec.predict_nq):
        lsd["codec_lm.decoder.bias"][g_ * (spec.codebook_size + 1) + spec.codebook_size] = -1e9
    codec = EncodecMI355X(arch, "cuda:0")
    codec.load_state_dict({k: torch.from_numpy(v) for k, v in csd.items()})
    m = LauraGenMI355X(spec, "cuda:0", max_positions=2048)
    m.load_state_dict(lsd)
    lens = [text_len - 3 * (i % 4) for i in range(batch)]                      # ragged texts
    ids = torch.from_numpy(synthetic_text(lcfg, batch, lens, 77)).cuda()
    g = torch.Generator().manual_seed(5)
    cont = torch.randint(0, spec.codebook_size, (batch, prompt_frames, spec.predict_nq), generator=g).cuda()
    cl = [prompt_frames] * batch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]

    def step(seed):
        ev[0].record()
        outs, _ = m.encode(ids, torch.tensor(lens))
        ev[1].record()
        tokens, out_lens = m.engine.decode_codec(outs, lens, new_frames, sampling=25, seed=seed, continual=cont, continual_lengths=cl)
        ev[2].record()
        emb = m.engine.codec_emb(outs, lens, tokens, out_lens)
        ev[3].record()
        wav = codec.engine.decode_emb(emb[:, prompt_frames:])                   # exclude_prompt: only the new frames are synthesised
        ev[4].record()
        return tokens, out_lens, wav

    for i in range(warmup):
        step(100 + i)
    torch.cuda.synchronize()
    phases = [0.0] * 4
    t0 = time.perf_counter()
    for i in range(steps):
        tokens, out_lens, wav = step(200 + i)
        torch.cuda.synchronize()
        for k in range(4):
            phases[k] += ev[k].elapsed_time(ev[k + 1])
    dt = (time.perf_counter() - t0) / steps
    assert bool(torch.isfinite(wav).all()) and all(v == prompt_frames + new_frames for v in out_lens), out_lens
    hop = 640
    gen_audio_s = batch * new_frames * hop / 16000.0
    lm, d, ff, V = spec.codec_lm, spec.codec_lm.d_model, spec.codec_lm.ff, spec.lm_vocab
    w_bytes = 4.0 * (lm.layers * (4 * d * d + 2 * d * ff) + V * d + d * spec.codebook_dim)          # LM weights streamed per decoding step
    kv_bytes = sum(4.0 * lm.layers * 2 * d * (l + 2 + prompt_frames + new_frames / 2.0) for l in lens)   # average KV-cache read per step
    ar_ms = phases[1] / steps
    ar_step_us = ar_ms * 1e3 / new_frames
    step_flops = 2.0 * batch * (lm.layers * (4 * d * d + 2 * d * ff) + V * d)
    out = {"workload": f"BASELINE.json configs[4]: LauraTTS zero-shot generation, batch {batch} prompts (texts of {min(lens)}-{max(lens)} phonemes, "
                       f"{prompt_frames}-frame prompt audio tokens), {new_frames} new frames (= {new_frames * hop / 16000.0:.0f} s) per prompt, top-k 25 "
                       "sampling on the device, KV cache; text encoder + LM + fine predictor + ds640 codec decoder, fp32",
           "value": round(gen_audio_s / dt, 1), "unit": "generated audio-s per wall-s",
           "tokens_per_s": round(batch * new_frames / dt, 1), "ms_per_step": round(dt * 1e3, 2), "steps": steps,
           "phases_ms": {"text_encoder": round(phases[0] / steps, 3), "decode_codec": round(ar_ms, 3),
                         "codec_emb": round(phases[2] / steps, 3), "codec_decoder": round(phases[3] / steps, 3)},
           "decode_step_us": round(ar_step_us, 2),
           "roofline": {"bound": "hbm", "kernel": "decoding step (one persistent launch: 61 phases of weight-streaming MFMA GEMV tiles + KV-cache "
                                                  "attention units handed over through arrival counters, + the sampler launch)",
                        "achieved": round((w_bytes + kv_bytes) / (ar_step_us * 1e-6) / 1e9, 1), "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s",
                        "frac": round((w_bytes + kv_bytes) / (ar_step_us * 1e-6) / 1e12 / PEAK_HBM_TBS, 4),
                        "traffic": (laura_step_pmc_traffic() or {}).get("bytes_per_step"), "traffic_detail": laura_step_pmc_traffic(),
                        "algorithmic_bytes_per_step": round(w_bytes + kv_bytes), "weights_bytes": round(w_bytes), "kv_bytes_avg": round(kv_bytes),
                        "traffic_over_algorithmic": round((laura_step_pmc_traffic() or {}).get("bytes_per_step", 0) / (w_bytes + kv_bytes), 2)
                        if laura_step_pmc_traffic() else None,
                        "step_tflops": round(step_flops / (ar_step_us * 1e-6) / 1e12, 3)}}
    if batch == 8:
        # the decoding step is latency-bound (a chain of 62 dependent kernels), so its time barely depends on the batch: the same flow at
        # the engine's maximum of 16 prompts per call, reported next to the contract's batch of 8
        lens16 = [text_len - 3 * (i % 4) for i in range(16)]
        ids16 = torch.from_numpy(synthetic_text(lcfg, 16, lens16, 78)).cuda()
        cont16 = torch.randint(0, spec.codebook_size, (16, prompt_frames, spec.predict_nq), generator=g).cuda()
        t16 = []
        for i in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            o16, _ = m.encode(ids16, torch.tensor(lens16))
            tk16, ol16 = m.engine.decode_codec(o16, lens16, new_frames, sampling=25, seed=300 + i, continual=cont16, continual_lengths=[prompt_frames] * 16)
            e16 = m.engine.codec_emb(o16, lens16, tk16, ol16)
            w16 = codec.engine.decode_emb(e16[:, prompt_frames:])
            torch.cuda.synchronize()
            t16.append(time.perf_counter() - t1)
        out["batch16"] = {"value": round(16 * new_frames * hop / 16000.0 / min(t16), 1), "unit": out["unit"], "ms_per_step": round(min(t16) * 1e3, 2),
                          "tokens_per_s": round(16 * new_frames / min(t16), 1)}
    if cpu_sample:
        # the reference's CPU path for the same prompts: no KV cache, batch 1 (oracle = ATen-CPU restatement, pinned bit-exact).
        # Bounded sample: ONE prompt, 6 new frames at the benchmark's prefix length; per-token cost grows with the prefix.
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from laura_oracle import LauraOracle
        orc = LauraOracle(lcfg, lsd)
        threads = min(16, torch.get_num_threads())
        old = torch.get_num_threads()
        torch.set_num_threads(threads)
        with torch.no_grad():
            emb0 = orc.token_embed(ids[:1, : lens[0]].cpu())
            to = orc.encode(emb0, [lens[0]])[0]
            t1 = time.perf_counter()
            orc.decode_codec(to, 6, sampling=False, continual=cont[0].cpu().tolist())
            cpu_dt = time.perf_counter() - t1
        torch.set_num_threads(old)
        out["cpu_baseline"] = {"value": round(6 / cpu_dt, 2), "unit": "tokens/s (one prompt)", "cores": threads, "kind": "port",
                               "sample": f"oracle/laura_oracle.py decode_codec, 1 prompt ({lens[0]} phonemes + {prompt_frames} prompt frames), 6 greedy "
                                         "frames, whole prefix re-scored per token like the reference (no KV cache)",
                               "gpu_tokens_per_s_per_prompt": round(new_frames / dt, 1)}
    del m, codec
    torch.cuda.empty_cache()
    return out


def transfer_times(wav_dev: torch.Tensor, codes_dev: torch.Tensor, reps: int = 5):
    """H2D of this rank's wav batch and D2H of its code indices through pinned host buffers (reported, never
    part of `value`: the C-ABI boundary takes device pointers)."""
    host_wav = torch.empty(wav_dev.shape, dtype=wav_dev.dtype).pin_memory()
    host_codes = torch.empty(codes_dev.shape, dtype=codes_dev.dtype).pin_memory()
    dst = torch.empty_like(wav_dev)
    out = {}
    for name, fn in (("h2d_ms", lambda: dst.copy_(host_wav, non_blocking=True)),
                     ("d2h_ms", lambda: host_codes.copy_(codes_dev, non_blocking=True))):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        out[name] = round(statistics.median(ts), 3)
    out["h2d_mb"] = round(wav_dev.numel() * 4 / 1e6, 2)
    out["d2h_mb"] = round(codes_dev.numel() * 8 / 1e6, 2)
    return out

def self_launch(n_ranks: int) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start N ranks of this same script under
    torch.distributed.run (one per GPU, rendezvous on a free port of 127.0.0.1) and pass their output through -- the reference's
    multi-GPU story is N self-launched processes as well (egs/LibriTTS/codec/encoding_decoding.sh:59-101,
    funcodec/bin/codec_inference.py:569-579).  Rank 0 prints the ONE JSON line."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL over xGMI needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def launcher_dry_run(args):
    """`--dry-run-launcher`: the N > 1 control flow of this script WITHOUT the engine (runs on a CPU-only box over gloo): ranks from the
    environment, utterance shards from shard_range, one gather of synthetic int64 codes (a pure function of the global utterance index,
    so rank 0 can check the gathered tensor), max-over-ranks timing, ONE JSON line from rank 0.  tests/test_bench_launcher.py runs it
    through the self-launch path."""
    import torch.distributed as dist
    from funcodec_amd.parallel import gather_codes, shard_range
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    per, n_q, tf = 3, 4, 5
    total = per * world + 1                                  # ragged: rank 0 holds one utterance more
    lo, hi = shard_range(total, rank, world)
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]

    def fake_codes(a, b):
        u = torch.arange(a, b, dtype=torch.int64)
        return (u[None, :, None] * 1000 + torch.arange(n_q)[:, None, None] * 10 + torch.arange(tf)[None, None, :]).contiguous()

    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps)):
        codes = gather_codes(fake_codes(lo, hi), dist if world > 1 else None, shard_sizes=sizes)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ok = bool(torch.equal(codes, fake_codes(0, total)))
    if rank == 0:
        print(json.dumps({"metric": "launcher dry run (no engine, gloo)", "n_gpus": world, "steps": args.steps, "gather_ok": ok,
                          "config": {"ranks_seen": dist.get_world_size() if world > 1 else 1, "global_utterances": total,
                                     "shard_sizes": sizes}, "seconds": round(dt, 4)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("launcher dry run: gathered codes differ from the expected tensor")


class _RehearsalEngine:
    """FC_BENCH_REHEARSAL=1: stands in for the engine so that the WHOLE main() flow of an N-rank run -- sharding, per-rank inputs, the step
    loop with its micro-batches, the gather of the codes, fences, max-over-ranks timing, the JSON line -- can be executed on a CPU-only box
    over gloo (tests/test_bench_launcher.py).  It computes nothing: codes are a function of the global utterance index, the line it
    produces says so in `metric` and `data`.  The N > 1 path has never run on hardware (no multi-GPU box has been available to any round);
    this is what keeps it from failing on a typo the first time it does."""
    micro_batch = 16

    def __init__(self, arch, first_utt):
        self.arch, self.first = arch, first_utt
        self.calls = 0

    def encode_decode(self, wav, n_q, use_scale=True):
        B, T = wav.shape
        tf = -(-T // self.arch.hop_length)
        u = self.first + self.calls_base + torch.arange(B, dtype=torch.int64)
        codes = (u[None, :, None] * 7 + torch.arange(n_q)[:, None, None] + torch.arange(tf)[None, None, :]) % 1024
        self.calls_base += B
        return dict(codes=codes.contiguous(), recon=torch.zeros(B, 1, T))

    calls_base = 0

    def new_step(self):
        self.calls_base = 0

    def set_profiling(self, on):
        pass

    def read_profile(self):
        return []

    def check_status(self):
        pass

    def work(self, B, T, n_q):
        return dict(total_flops=1.0, conv_flops=1.0, conv_bytes=1.0, total_bytes=1.0, total_launches=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-event-profile", action="store_true", help="skip the separate per-kernel HIP-event pass (no `roofline`)")
    ap.add_argument("--profile-steps", type=int, default=5, help="steps of the separate per-kernel pass")
    ap.add_argument("--workload", choices=("encodec", "freqcodec", "freqcodec_gr1", "laura"), default="encodec",
                    help="encodec = the contract metric (BASELINE.json configs[1] / [2]); freqcodec / freqcodec_gr1 = side measurement of "
                         "configs[3]; laura = side measurement of configs[4]")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[3] / configs[4] side measurements of the default run")
    ap.add_argument("--dry-run-launcher", action="store_true",
                    help="exercise rank launch + sharding + gather + the JSON line over gloo without the engine (CPU test of the N > 1 path)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:     # not under torchrun: spawn our own ranks
        raise SystemExit(self_launch(args.gpus))
    if args.gpus != int(os.environ.get("WORLD_SIZE", "1")):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: launch one rank per GPU (or drop WORLD_SIZE and "
                         "let bench.py spawn them)")
    if args.dry_run_launcher:
        launcher_dry_run(args)
        return
    global CONFIG, MICRO_BATCH
    if args.workload in ("laura", "freqcodec_gr1"):      # side measurements on their own (one JSON line)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
        torch.cuda.set_device(0)
        res = laura_side(steps=max(1, args.steps), warmup=max(1, args.warmup), cpu_sample=not args.no_cpu_baseline) if args.workload == "laura" \
            else freqcodec_side(steps=max(1, args.steps), warmup=max(1, args.warmup))
        print(json.dumps(res), flush=True)
        return
    if args.workload == "freqcodec" and not is_freq():
        CONFIG = "freqmp"
    if is_freq() and not os.environ.get("FC_BENCH_MICRO"):
        MICRO_BATCH = 32                                   # configs[3]: batch 64 = two engine calls of 32 (the persistent LSTM holds <= 32 utterances)
    if is_freq() and not os.environ.get("FC_BENCH_UTTS"):
        os.environ["FC_BENCH_UTTS"] = "64"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rehearsal = bool(int(os.environ.get("FC_BENCH_REHEARSAL", "0")))      # CPU rehearsal of the control flow (see _RehearsalEngine): never a measurement
    if not rehearsal and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    dev = torch.device("cpu") if rehearsal else torch.device("cuda", local_rank)
    if not rehearsal:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from funcodec_amd.config import arch_from_config, recipe_config
    from funcodec_amd.model import EncodecMI355X
    from funcodec_amd.parallel import gather_codes, shard_range
    from funcodec_amd.synth import synthetic_audio

    cfg = recipe_config(CONFIG)
    arch = arch_from_config(cfg)
    # N = 1: Config B (16 utterances, one engine call).  N > 1: Config C (128 utterances per GPU, micro-batches of 32).
    utts_per_gpu = int(os.environ.get("FC_BENCH_UTTS", MICRO_BATCH if world == 1 else 128))     # freqcodec: 64 (configs[3])
    total_utts = utts_per_gpu * world
    lo, hi = shard_range(total_utts, rank, world)
    shard_sizes = [shard_range(total_utts, r, world)[1] - shard_range(total_utts, r, world)[0] for r in range(world)]
    if rehearsal:
        model, eng = None, _RehearsalEngine(arch, lo)
    else:
        model = EncodecMI355X(arch, f"cuda:{local_rank}")
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state(cfg, arch).items()})
        eng = model.engine
    eng.micro_batch = max(eng.micro_batch, MICRO_BATCH)      # one engine call per bench micro-batch
    n_samples = int(os.environ.get("FC_BENCH_SAMPLES", SAMPLES)) if rehearsal else SAMPLES
    if world == 1:
        wav = torch.from_numpy(synthetic_audio(hi - lo, n_samples, 1234)).to(dev)
    else:   # per-rank seeds: no rank generates the whole 1024-utterance set
        wav = torch.from_numpy(synthetic_audio(hi - lo, n_samples, 1234 + rank)).to(dev)
    n_q = arch.num_quantizers

    def step():
        parts, r = [], None
        if rehearsal:
            eng.new_step()
        for i in range(0, wav.shape[0], MICRO_BATCH):      # one engine call per micro-batch; outputs as the reference returns them
            r = eng.encode_decode(wav[i:i + MICRO_BATCH], n_q, use_scale=True)
            parts.append(r["codes"])
        codes = parts[0] if len(parts) == 1 else torch.cat(parts, 1)
        if world > 1:
            codes = gather_codes(codes, dist, shard_sizes=shard_sizes)
        return r, codes

    def fence():
        if world > 1:
            dist.barrier()
        if not rehearsal:
            torch.cuda.synchronize()

    eng.set_profiling(False)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r, codes = step()
    fence()
    dt = time.perf_counter() - t0
    eng.check_status()
    diag = None
    if world > 1:
        # ---- self-diagnosing N > 1 line: every rank's own wall time for the K steps, and a separate gather-only loop --
        # efficiency then reads as load balance (rank spread) vs the collective (gather_ms), not as one opaque number
        mine = torch.tensor([dt], dtype=torch.float64, device=dev)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in per_rank]
        local_codes = codes[:, lo:hi].contiguous()
        for _ in range(2):
            gather_codes(local_codes, dist, shard_sizes=shard_sizes)
        fence()
        tg = time.perf_counter()
        g_reps = max(3, args.steps)
        for _ in range(g_reps):
            gather_codes(local_codes, dist, shard_sizes=shard_sizes)
        fence()
        tg = torch.tensor([(time.perf_counter() - tg) / g_reps], dtype=torch.float64, device=dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        try:
            ver = torch.cuda.nccl.version() if not rehearsal else None      # on ROCm builds this is RCCL's version triple
            ver = ".".join(str(v) for v in ver) if ver else None
        except Exception:                                                    # noqa: BLE001 -- a version string must never fail the bench
            ver = None
        diag = {"rank_ms_per_step": [round(x, 3) for x in per_rank_ms], "rank_ms_per_step_min": round(min(per_rank_ms), 3),
                "rank_ms_per_step_max": round(max(per_rank_ms), 3),
                "gather_ms": round(float(tg.item()) * 1e3, 3), "gather_bytes_per_rank": int(local_codes.numel() * 8),
                "gather_note": "one all_gather_into_tensor of this rank's int64 codes per step, timed alone (barrier + sync on both sides, max over "
                               "ranks); inside the timed step it runs behind the last micro-batch's kernels on the same stream",
                "backend": dist.get_backend(), "rccl_version": ver}
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(r["recon"]).all())
    assert codes.shape[1] == total_utts
    if world > 1:
        assert dist.get_world_size() == world == args.gpus, (dist.get_world_size(), world, args.gpus)     # ranks_seen

    # ---- separate pass: per-kernel-class HIP-event durations on the engine's stream (NOT inside the timed region above)
    prof, prof_steps = [], 0
    if rank == 0 and not args.no_event_profile:
        prof_steps = max(1, min(args.profile_steps, args.steps))
        eng.set_profiling(True)
        fence_local = (lambda: None) if rehearsal else torch.cuda.synchronize
        fence_local()
        for _ in range(prof_steps):
            for i in range(0, min(wav.shape[0], MICRO_BATCH), MICRO_BATCH):     # one micro-batch per profiled step
                eng.encode_decode(wav[i:i + MICRO_BATCH], n_q, use_scale=True)
        prof = eng.read_profile()
        eng.set_profiling(False)

    if rank == 0:
        audio_s = total_utts * n_samples / 16000.0 * args.steps
        work = eng.work(MICRO_BATCH, n_samples, n_q)
        nmb = utts_per_gpu / MICRO_BATCH                  # engine calls per step per GPU
        step_s = dt / args.steps
        out = {
            "metric": "audio-seconds encoded+decoded per wall-sec, 16k-nq32ds640" if CONFIG == "ds640" else
                      f"audio-seconds encoded+decoded per wall-sec, recipe {CONFIG} (side measurement, not the contract metric)",
            "value": round(audio_s / dt, 2), "unit": "audio-s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (("BASELINE.json configs[3] shape: freqcodec mag_phase 16k recipe (egs/LibriTTS/codec/conf/"
                                    "freqcodec_mag_phase_16k_n32_600k_step.yaml, 16.2M + codebooks, synthetic seeded checkpoint; the released "
                                    "gr1 config.yaml is not in the reference tree), STFT -> 2-D SEANet -> RVQ -> 2-D SEANet -> iSTFT, ") if is_freq() else
                                   (("BASELINE.json configs[1]: " if world == 1 else "BASELINE.json configs[2]: ") +
                                    "encodec 16k-nq32ds640 (57.6M, synthetic seeded checkpoint), ")) +
                                   f"run_mod=inference (encode + 32-stage RVQ "
                                   f"+ decode), {utts_per_gpu} x 10 s utterances per GPU in micro-batches of {MICRO_BATCH}, n_q=32",
                       "utterances_per_gpu": utts_per_gpu, "samples_per_utterance": n_samples, "micro_batch": MICRO_BATCH,
                       "global_utterances": total_utts,
                       "ranks_seen": dist.get_world_size() if world > 1 else 1,
                       "parallelism": f"utterance-sharded x{world}, all_gather(codes) over RCCL" if world > 1 else "single GPU"},
            "ms_per_micro_batch": round(step_s * 1e3 / nmb, 3),
            "algorithmic_per_micro_batch": {"tflop": round(work["total_flops"] / 1e12, 4),
                                            "conv_tflop": round(work["conv_flops"] / 1e12, 4),
                                            "conv_gb": round(work["conv_bytes"] / 1e9, 3),
                                            "launches": work["total_launches"]},
            "whole_step": {"tflops": round(work["total_flops"] * nmb / step_s / 1e12, 2),
                           "frac_of_f32_peak": round(work["total_flops"] * nmb / step_s / 1e12 / PEAK_F32_TFLOPS, 4),
                           "sustained_clock_ghz": SUSTAINED_GHZ,
                           "frac_of_f32_peak_at_sustained_clock": round(work["total_flops"] * nmb / step_s / 1e12 / PEAK_F32_SUSTAINED, 4),
                           "alg_hbm_tbs": round(work["total_bytes"] * nmb / step_s / 1e12, 3),
                           "frac_of_hbm_peak": round(work["total_bytes"] * nmb / step_s / 1e12 / PEAK_HBM_TBS, 4),
                           "algorithmic_bytes": round(work["total_bytes"] * nmb),
                           "traffic": (pmc_step_traffic() or {}).get("bytes_per_step") if (world == 1 and CONFIG == "ds640") else None,
                           "traffic_over_algorithmic": round(pmc_step_traffic()["bytes_per_step"] / (work["total_bytes"] * nmb), 2)
                           if (world == 1 and CONFIG == "ds640" and pmc_step_traffic()) else None},
            "timed_region": "in-engine HIP-event brackets OFF; the per-kernel table below is a separate pass",
        }
        if rehearsal:
            out["metric"] = "REHEARSAL of the control flow with a stand-in engine (FC_BENCH_REHEARSAL=1): not a measurement"
            out["data"] = "none (stand-in engine)"
            out["value"] = 0.0
            exp_u = torch.arange(total_utts, dtype=torch.int64)
            tf = codes.shape[2]
            expect = (exp_u[None, :, None] * 7 + torch.arange(n_q)[:, None, None] + torch.arange(tf)[None, None, :]) % 1024
            out["gather_ok"] = bool(torch.equal(codes, expect))
        else:
            out["transfers"] = transfer_times(wav, r["codes"])
        if prof:
            out.update(kernel_rooflines(prof, prof_steps))
            if "roofline" in out:
                out["roofline"]["whole_step"] = out["whole_step"]
        if world > 1:
            out["multi_gpu"] = diag
            out["config"]["scaling_base"] = ("efficiency is to be computed against secondary.config_c_shard_b128.value of the --gpus 1 line (same "
                                             "per-rank shape: 128 x 10 s in micro-batches of 32), NOT against the N = 1 contract value (Config B, "
                                             "16 utterances in one call)")
        if world == 1 and not args.no_cpu_baseline and not rehearsal:
            out["cpu_baseline"] = cpu_baseline()
        if world == 1 and CONFIG == "ds640" and not args.no_secondary and not rehearsal:
            # the next scope rows, measured AFTER the headline timing so that they cannot disturb it
            sec = {}
            try:
                sec["config_c_shard_b128"] = config_c_shard_side(eng, n_q)
            except Exception as ex:
                sec["config_c_shard_b128"] = {"error": f"{type(ex).__name__}: {ex}"}
            del model, eng
            torch.cuda.empty_cache()
            # freqcodec_gr1rel_b64: the candidate for the RELEASED gr1 architecture (n_filters 8, one LSTM layer: the README's 0.52 M parameters;
            # DESIGN.md) timed next to the recipe-sized gr1 net; its H = 128 one-layer LSTM runs on the per-step launches
            for key, fn in (("freqcodec_gr1_b64", lambda: freqcodec_side()),
                            ("freqcodec_gr1rel_b64", lambda: freqcodec_side(config="freqmpgr1rel", steps=3, warmup=1)),
                            ("laura_tts_b8", lambda: laura_side(cpu_sample=not args.no_cpu_baseline))):
                try:
                    sec[key] = fn()
                except Exception as ex:      # a side measurement must never take the contract line down with it
                    sec[key] = {"error": f"{type(ex).__name__}: {ex}"}
            out["secondary"] = sec
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
