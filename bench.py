#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on the MI355X engine.

metric : denoise-loop latents/s (512px, 50-step DDIM, batch 4 per GPU) = images / wall time of
         [denoise loop sd_unified_pipeline.py:465-507 + VAE decode :511-523], text encoding excluded.
step   : one full pass of the hot path over one batch: 50 UNet forwards at CFG batch 8 + scheduler
         steps + one VAE decode of 4 latents, driven through StableDiffusionUnifiedPipeline.__call__.
inputs : synthetic: latents N(0,1) [4,4,64,64] seed 0, text embeddings N(0,1)
         [4,77,768] x2 seed 1, weights N(0,1/fan_in) seed 2, all resident in HBM before timing.
N > 1  : one process per GPU (torchrun), weak scaling: every rank denoises its own 4 latents; rank 0
         broadcasts the text embeddings and images are all-gathered (RCCL over xGMI).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
measured live with HIP events on the launch stream) and `cpu_baseline` (the fp32 CPU oracle timed
on a bounded sample of the same workload; N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _imports():
    """Project imports happen in the worker ranks only: the launcher parent (`--gpus N` without a
    torchrun environment) must never touch the GPU, it only starts the ranks and relays rank 0."""
    global torch, _lib, config, sdd, weights, HipAutoencoderKL, HipUNet2DConditionModel
    global SDModelWrapper, StableDiffusionUnifiedPipeline, DDIMScheduler
    import torch
    from stablediffusion_amd import _lib, config, distributed as sdd, weights
    from stablediffusion_amd.models import HipAutoencoderKL, HipUNet2DConditionModel
    from stablediffusion_amd.pipeline import SDModelWrapper, StableDiffusionUnifiedPipeline
    from stablediffusion_amd.schedulers import DDIMScheduler


MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md:43
HBM_PEAK_GBS = 8000.0       # spec; 6.29 TB/s achievable (MI355X_MICROARCH.md:36)
# algorithmic TFLOP per unit (2*MAC of conv / linear / QK^T / PV only)
UNET_TFLOP_PER_SAMPLE = {"sd15": {32: 0.1803, 64: 0.8032, 96: 2.148, 128: 4.674}, "sdxl": {128: 6.761}}
VAE_TFLOP_PER_IMAGE = {32: 0.622, 64: 2.515, 96: 5.754, 128: 10.470}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_models(device, preset="sd15", seed=2):
    ucfg, vcfg = (f() for f in config.PRESETS[preset])
    t0 = time.time()
    usd = weights.synth_state_dict(weights.unet_manifest(ucfg), seed=seed, dtype=torch.float16)
    vsd = weights.synth_state_dict(weights.vae_manifest(vcfg), seed=seed + 1, dtype=torch.float16)
    unet = HipUNet2DConditionModel(ucfg, device).load_state_dict(usd)
    vae = HipAutoencoderKL(vcfg, device).load_state_dict(vsd)
    log(f"[bench] weights synthesised + packed in {time.time() - t0:.1f}s; "
        f"unet {unet.memory()[0] / 1e9:.2f} GB, vae {vae.memory()[0] / 1e9:.2f} GB packed")
    return ucfg, vcfg, usd, vsd, unet, vae


def live_roofline(lib, unet, vae, B, lat_hw, ehs, device, added=None):
    """One profiled UNet forward (CFG batch 2B) + one decode: per-kernel HIP-event times."""
    x = torch.randn(2 * B, 4, lat_hw, lat_hw, device=device, dtype=torch.float16)
    z = torch.randn(B, 4, lat_hw, lat_hw, device=device, dtype=torch.float16)
    t = torch.tensor(501.0)
    unet(x, t, ehs, added_cond_kwargs=added)           # warm
    torch.cuda.synchronize()
    lib.sd_prof_enable(1)
    unet(x, t, ehs, added_cond_kwargs=added)
    ents = (_lib.SdProfEntry * 64)()
    n = C.c_int()
    _lib.check(lib.sd_prof_collect(ents, 64, C.byref(n)), "sd_prof_collect")
    unet_rows = [(e.kernel.decode(), e.flops, e.bytes, e.ms, e.launches) for e in ents[: n.value]]
    vae.decode(z)
    _lib.check(lib.sd_prof_collect(ents, 64, C.byref(n)), "sd_prof_collect")
    vae_rows = [(e.kernel.decode(), e.flops, e.bytes, e.ms, e.launches) for e in ents[: n.value]]
    lib.sd_prof_enable(0)
    return unet_rows, vae_rows


def _profile_file(stem: str):
    """Newest committed profile of that kind: profiles/r03_<stem>, else an earlier round's."""
    for tag in ("r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{tag}_{stem}")
        if os.path.exists(path):
            return path
    return None


def pmc_traffic_for(function: str):
    """HBM bytes per launch of the roofline kernel function from the committed rocprofv3 --pmc passes
    (profiles/rNN_pmc_hbm_traffic_per_launch.json: FETCH_SIZE x2 per MI355X_MICROARCH.md + WRITE_SIZE, separate
    passes; PMC passes cannot run inside the timed bench), launch-weighted over its template instantiations."""
    path = _profile_file("pmc_hbm_traffic_per_launch.json")
    if not path:
        return None
    data = json.load(open(path))
    rows = [e for k, e in data.items() if isinstance(e, dict) and k.split("<")[0] == function]
    n = sum(e["launches"] for e in rows)
    if not n:
        return None
    mb = sum((e["fetch_MB_per_launch"] + e["write_MB_per_launch"]) * e["launches"] for e in rows) / n
    return {"hbm_MB_per_launch": round(mb, 2),
            "source": f"{os.path.relpath(path, ROOT)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE "
                      f"doubled for gfx950; launch-weighted average over every instantiation of {function} in that run)"}


def rocprof_avg_us_for(function: str):
    """Average launch duration of the kernel function in the committed rocprofv3 --kernel-trace --stats summary
    of this same command (profiles/rNN_bench_kernel_stats.csv), over all its instantiations.  The live
    HIP-event bracket reads ~10 % longer than the trace: an event between two launches keeps the next kernel's
    ramp from overlapping the previous kernel's tail, which is how the kernels run in the timed loop."""
    path = _profile_file("bench_kernel_stats.csv")
    if not path:
        return None
    import csv
    tot = n = 0.0
    for row in csv.DictReader(open(path)):
        if function in row["Name"]:
            tot += float(row["TotalDurationNs"])
            n += float(row["Calls"])
    return round(tot / n / 1e3, 2) if n else None


KERNEL_FAMILIES = (("conv/GEMM", ("igemm2_kernel", "conv3x3_halo_kernel", "igemm_kernel", "wsgemm_kernel", "geglu_persist_kernel",
                                  "ffn_fused_kernel", "conv_head_kernel", "conv_tail_kernel")), ("attention", ("attn_kernel",)),
                   ("norms", ("groupnorm", "layernorm", "row_stats", "gn_cat_finalize_kernel")))


def kernel_function(row_name: str) -> str:
    """`igemm2_kernel<256,160,...>` -> `igemm2_kernel`: template instantiations are one kernel function."""
    return row_name.split("<")[0].split("(")[0]


def roofline_report(urows, vrows):
    """`roofline` = the kernel FUNCTION with the largest summed time in one UNet forward (all its template
    instantiations together), next to the per-family aggregates, the whole-forward figure and
    the single best instantiation.  achieved = sum of algorithmic FLOPs of its launches / sum of their
    HIP-event times on the launch stream; peak = 2.5 PFLOP/s dense fp16 (MI355X_MICROARCH.md:43)."""
    def agg(rows, pred):
        sel = [r for r in rows if pred(r[0])]
        return (sum(r[1] for r in sel), sum(r[2] for r in sel), sum(r[3] for r in sel), sum(r[4] for r in sel))

    out = {}
    funcs = sorted({kernel_function(r[0]) for r in urows}, key=lambda f: -agg(urows, lambda n: kernel_function(n) == f)[2])
    top = funcs[0]
    flops, nbytes, ms, launches = agg(urows, lambda n: kernel_function(n) == top)
    fam_rows = []
    for fam, names in KERNEL_FAMILIES:
        f, b, m, l = agg(urows, lambda n: kernel_function(n) in names or any(n.startswith(x) for x in names))
        if l:
            fam_rows.append({"family": fam, "ms": round(m, 3), "launches": l,
                             "tflops": round(f / (m / 1e3) / 1e12, 1) if f > 0 else None,
                             "mfma_frac": round(f / (m / 1e3) / 1e12 / MFMA_PEAK_TFLOPS, 4) if f > 0 else None,
                             "gbs": round(b / (m / 1e3) / 1e9, 1)})
    tf, tb, tm, tl = agg(urows, lambda n: True)
    best = max((r for r in urows if r[1] > 0 and r[3] > 0), key=lambda r: r[1] / r[3])
    if flops > 0:
        ach = flops / (ms / 1e3) / 1e12
        out["roofline"] = {"kernel": top, "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic_for(top),
                           "launches_per_unet_forward": launches,
                           "algorithmic_MB_per_launch": round(nbytes / launches / 1e6, 2),
                           "avg_launch_us": round(ms / launches * 1e3, 2),
                           "frac_clock": "hip_event_bracket (avg_launch_us, measured live in this run; reads ~10 % "
                                         "longer than the kernel trace because an event between two launches keeps the "
                                         "next kernel's ramp from overlapping the previous tail)",
                           "rocprofv3_avg_launch_us": rocprof_avg_us_for(top),
                           "algorithmic_gflop_per_launch": round(flops / launches / 1e9, 3),
                           "scope": "all template instantiations of the function, one UNet forward (CFG batch)"}
        rp = out["roofline"]["rocprofv3_avg_launch_us"]
        # the same fraction on the committed rocprofv3 kernel-trace clock (same command, profiles/)
        out["roofline"]["frac_rocprofv3_clock"] = round(flops / launches / (rp * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4) if rp else None
    else:
        ach = nbytes / (ms / 1e3) / 1e9
        out["roofline"] = {"kernel": top, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "launches_per_unet_forward": launches,
                           "avg_launch_us": round(ms / launches * 1e3, 2)}
    out["roofline"]["families"] = fam_rows
    out["roofline"]["whole_forward"] = {"ms_sum_of_launches": round(tm, 3), "launches": tl,
                                        "mfma_frac": round(tf / (tm / 1e3) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
    out["roofline"]["best_instantiation"] = {"kernel": best[0], "launches": best[4],
                                             "tflops": round(best[1] / (best[3] / 1e3) / 1e12, 1),
                                             "mfma_frac": round(best[1] / (best[3] / 1e3) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
    out["unet_forward_profiled_sum_ms"] = round(tm, 3)
    fmt = lambda rows: [{"kernel": k, "ms": round(m, 3), "launches": l,
                         "tflops": round(f / (m / 1e3) / 1e12, 1) if f > 0 and m > 0 else None,
                         "gbs": round(b / (m / 1e3) / 1e9, 1) if m > 0 else None}
                        for k, f, b, m, l in sorted(rows, key=lambda r: -r[3])]
    out["kernels_unet_forward"] = fmt(urows)
    out["kernels_vae_decode"] = fmt(vrows)
    return out


def box_probe(lib, stream=None):
    """Fixed, model-independent probe of the box, run in-process BEFORE and outside the timed region
    (VERDICT r2 item 2): ~40 ms of back-to-back v_mfma_f32_16x16x32_f16 on register operands (dense fp16
    TFLOP/s the box sustains, random operands) and a 16-byte-per-lane copy of 512 MiB -> 512 MiB (GB/s read +
    write, beyond the Infinity Cache).  `value / box_probe.mfma_tflops` compares rounds across boxes."""
    tf, gb, d1, d2 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    _lib.check(lib.sd_probe_mfma(300000, C.byref(tf), None), "sd_probe_mfma")
    _lib.check(lib.sd_probe_copy(512 << 20, 5, C.byref(gb), None), "sd_probe_copy")
    # the operand path of the GEMM kernels (round 3): every CU streaming through the LDS-DMA path, a region all blocks
    # share (weights, L2-resident) and a region per block that overflows L2 into the Infinity Cache (activations)
    _lib.check(lib.sd_probe_lds_dma(1 << 20, 64, 4, 1, C.byref(d1), None), "sd_probe_lds_dma")
    _lib.check(lib.sd_probe_lds_dma(1 << 20, 64, 4, 0, C.byref(d2), None), "sd_probe_lds_dma")
    return {"mfma_tflops": round(tf.value, 1), "hbm_gbs": round(gb.value, 1),
            "l2_to_lds_shared_gbs": round(d1.value, 1), "l2_to_lds_own_gbs": round(d2.value, 1),
            "what": "sd_probe_mfma: 300000 x 16 v_mfma_f32_16x16x32_f16 per wave, 4 waves per CU, random operands; "
                    "sd_probe_copy: 5 x (512 MiB -> 512 MiB) float4 copy, read + write bytes; sd_probe_lds_dma: every CU "
                    "streaming 1 MiB regions 64 times through buffer_load ... lds, 4 pieces in flight per wave, one region "
                    "for all blocks (shared) / one per block (own)"}


def usable_cores() -> int:
    """Host cores this process may really use: cgroup quota, then affinity, capped at the GPU
    box's per-GPU CPU share (16) so torch does not oversubscribe a 256-core host."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return int(os.environ.get("SD_BENCH_CPU_THREADS", min(n, 16)))


def cpu_baseline(ucfg, vcfg, usd, vsd, steps, lat_hw, ehs_len):
    """fp32 CPU oracle ("port": the build's restatement of the reference's diffusers CPU float32 path) on a
    bounded sample, about 10-15 s of CPU work on the box's cores:
      * BASELINE.json config C1 run whole, as SURVEY.md section 8(d) specifies: SD1.5 256 px (32x32 latents),
        10-step DDIM, batch 1, CFG on, decode included -- reported under `c1_measured`;
      * for `value` (same unit and workload as `metric`): ONE UNet forward at CFG batch 2 and ONE VAE decode at
        the benchmark's own latent size, measured directly (no FLOP-ratio extrapolation), times the step count."""
    from oracle import pipeline_ref, unet_ref, vae_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        uw = {k: v.float() for k, v in usd.items()}
        vw = {k: v.float() for k, v in vsd.items()}
        lat = torch.randn(1, 4, 32, 32, generator=g)
        emb2 = torch.randn(2, ehs_len, ucfg.cross_attention_dim, generator=g)
        t0 = time.time()
        img, _ = pipeline_ref.txt2img_ref(ucfg, uw, vcfg, vw, lat, emb2, steps=10, guidance_scale=5.0)
        t_c1 = time.time() - t0
        assert img.shape == (1, 3, 256, 256) and bool(torch.isfinite(img).all())
        x = torch.randn(2, 4, lat_hw, lat_hw, generator=g)
        t0 = time.time()
        unet_ref.unet_forward(ucfg, uw, x, torch.tensor(501.0), emb2)
        t_unet = time.time() - t0
        z = torch.randn(1, 4, lat_hw, lat_hw, generator=g)
        t0 = time.time()
        vae_ref.vae_decode(vcfg, vw, z)
        t_vae = time.time() - t0
    per_latent = steps * t_unet + t_vae
    return {
        "value": 1.0 / per_latent, "unit": "latents/s", "cores": cores, "kind": "port",
        "sample": (f"oracle (fp32 torch CPU restatement of the reference's diffusers path), {cores} threads: "
                   f"1 UNet forward at CFG batch 2 on {lat_hw}x{lat_hw} latents = {t_unet:.2f}s (x {steps} steps) + "
                   f"1 VAE decode of a {lat_hw}x{lat_hw} latent = {t_vae:.2f}s, both measured at the benchmark's size"),
        "c1_measured": {"workload": "BASELINE.json C1: SD1.5 256x256, 10-step DDIM, batch 1, CFG on, decode included",
                        "seconds": round(t_c1, 2), "latents_per_s": round(1.0 / t_c1, 4)},
    }


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` outside torchrun: start N fresh worker processes (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relay rank 0's JSON line, and return non-zero if
    any rank fails.  The parent never initialises the GPU (`torch.cuda.device_count()` does not, on
    this image) and never replaces a GPU process; it refuses when fewer than N devices are visible
    unless SD_DIST_BACKEND=gloo asks for a rehearsal with several ranks per device."""
    import subprocess
    rehearse = "--rehearse" in argv
    if not rehearse and os.environ.get("SD_DIST_BACKEND") != "gloo":
        import torch
        have = torch.cuda.device_count()
        if have < n:
            log(f"[bench] --gpus {n} requested but only {have} GPU(s) visible: refusing to run "
                f"(set SD_DIST_BACKEND=gloo to rehearse {n} ranks on fewer devices)")
            return 2
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=300))
        except subprocess.TimeoutExpired:      # rank 0 is gone: a straggler can only be stuck
            p.kill()
            rcs.append(p.wait())
    if any(rcs):
        log(f"[bench] rank exit codes {rcs}: failing")
        if out0:
            log(out0)
        return 1
    sys.stdout.write(out0)
    sys.stdout.flush()
    return 0


def rehearse_ranks():
    """--rehearse: the ranks only rendezvous over gloo and count each other (no GPU, no engine):
    covers the launcher, the environment plumbing and the relay on a CPU-only box."""
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    import torch
    t = torch.tensor([1.0])
    dist.all_reduce(t)
    dist.barrier()
    if os.environ.get("SD_BENCH_REHEARSE_FAIL_RANK") == str(rank):
        sys.exit(3)
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": dist.get_world_size(), "ranks_counted": int(t.item())}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed passes of the whole hot path")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--denoise-steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4, help="latents per GPU")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--guidance", type=float, default=5.0)
    ap.add_argument("--preset", default="sd15", choices=["sd15", "sdxl"],
                    help="sd15 = BASELINE.json metric (C2); sdxl with --res 1024 --batch 2 --denoise-steps 30 "
                         "--scheduler 'DPM++ 2M' = config C4")
    ap.add_argument("--scheduler", default="DDIM", choices=["DDIM", "DPM++ 2M", "euler"])
    ap.add_argument("--graph", action="store_true", help="replay the UNet forward from a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large", action="store_true",
                    help="skip the extra 4x128x128-latent (1024 px) pass reported under `extra`")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="ranks only rendezvous over gloo and count each other (launcher self-test, no GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.rehearse:
        return rehearse_ranks()
    _imports()
    if sdd.env_world()[2] != args.gpus:
        world = sdd.env_world()[2]
        # under torchrun the environment is the truth; a mismatch is a mis-launch, not a fallback
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a mislabelled run")
        sys.exit(2)
    rank, world = sdd.init()
    n_gpus = sdd.world_size()       # the number of ranks the collectives really see
    assert n_gpus == world, (n_gpus, world)
    _lib.require_gpu()
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    lib = _lib.load()

    ucfg, vcfg, usd, vsd, unet, vae = build_models(device, args.preset)
    if args.graph:
        unet.use_graph(True)
    model = SDModelWrapper(base=unet, vae=vae, scheduler=DDIMScheduler(), device=str(device),
                           model_type=args.preset)
    model.set_scheduler(args.scheduler)
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device=str(device))

    probe = box_probe(lib) if rank == 0 else None
    B = args.batch
    lat_hw = args.res // 8
    total = B * n_gpus
    # full-batch synthetic inputs from single seeded CPU generators, then resident in HBM
    lat_full = torch.randn(total, 4, lat_hw, lat_hw, generator=torch.Generator().manual_seed(0)).half().to(device)
    ge = torch.Generator().manual_seed(1)
    pe_full = torch.randn(total, 77, ucfg.cross_attention_dim, generator=ge).half().to(device)
    ne_full = torch.randn(total, 77, ucfg.cross_attention_dim, generator=ge).half().to(device)
    pooled = npooled = None
    if args.preset == "sdxl":
        pdim = ucfg.projection_class_embeddings_input_dim - 6 * ucfg.addition_time_embed_dim
        pooled = torch.randn(total, pdim, generator=ge).half().to(device)
        npooled = torch.randn(total, pdim, generator=ge).half().to(device)
    if rank != 0:   # non-root ranks receive the embeddings through the broadcast
        pe_full.zero_(); ne_full.zero_()
        if pooled is not None:
            pooled.zero_(); npooled.zero_()

    # the text embeddings of the run: ONE RCCL broadcast from rank 0 (once per run), then every
    # pass works on this rank's slice; the all-gather of the decoded images is part of every pass
    sdd.broadcast_tensors([pe_full, ne_full] + [t for t in (pooled, npooled) if t is not None])

    def one_pass():
        return sdd.sharded_txt2img(pipe, model, lat_full, pe_full, ne_full, rank, n_gpus, pooled, npooled, broadcast=False,
                                   num_inference_steps=args.denoise_steps, guidance_scale=args.guidance,
                                   height=args.res, width=args.res)

    for _ in range(args.warmup):
        imgs = one_pass()
    sdd.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        imgs = one_pass()
    torch.cuda.synchronize(); sdd.barrier()
    dt = sdd.max_over_ranks(time.perf_counter() - t0, device)
    assert imgs.shape == (total, 3, args.res, args.res), imgs.shape
    finite = bool(torch.isfinite(imgs.float()).all().item())

    ms_per_step = dt / args.steps * 1e3
    value = total * args.steps / dt

    # north_star also asks for the 4x128x128-latent input: one warm-up + one timed pass of the same
    # loop at 2x the resolution (reported under `extra`, never mixed into `value`)
    large = None
    if not args.no_large and args.preset == "sd15" and args.res == 512:
        lres, lhw = 2 * args.res, 2 * lat_hw
        lat_big = torch.randn(total, 4, lhw, lhw, generator=torch.Generator().manual_seed(0)).half().to(device)

        def big_pass():
            return sdd.sharded_txt2img(pipe, model, lat_big, pe_full, ne_full, rank, n_gpus, pooled, npooled, broadcast=False,
                                       num_inference_steps=args.denoise_steps, guidance_scale=args.guidance,
                                       height=lres, width=lres)
        big_pass()
        sdd.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        big = big_pass()
        torch.cuda.synchronize(); sdd.barrier()
        dt_big = sdd.max_over_ranks(time.perf_counter() - t1, device)
        u128, v128 = UNET_TFLOP_PER_SAMPLE["sd15"].get(lhw), VAE_TFLOP_PER_IMAGE.get(lhw)
        large = {"workload": f"SD1.5 {lres}x{lres} ({B}x4x{lhw}x{lhw} latents/GPU), {args.denoise_steps}-step "
                             f"{args.scheduler}, CFG on", "latents_per_s": round(total / dt_big, 4),
                 "ms_per_pass": round(dt_big * 1e3, 2), "passes_timed": 1,
                 "outputs_finite": bool(torch.isfinite(big.float()).all().item())}
        if u128 and v128:
            tf = args.denoise_steps * 2 * B * u128 + B * v128
            large["whole_path_mfma_frac"] = round(tf / dt_big / MFMA_PEAK_TFLOPS, 4)
        del big, lat_big

    # UNet-only / VAE-only times (per GPU), same shapes as the loop
    x8 = torch.randn(2 * B, 4, lat_hw, lat_hw, device=device, dtype=torch.float16)
    e8 = torch.cat([ne_full[:B], pe_full[:B]])
    tt = torch.tensor(501.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    added = None
    if args.preset == "sdxl":
        added = {"text_embeds": torch.cat([npooled[:B], pooled[:B]]),
                 "time_ids": torch.tensor([[args.res, args.res, 0, 0, args.res, args.res]] * (2 * B), dtype=torch.float32)}
    unet(x8, tt, e8, added_cond_kwargs=added); vae.decode(x8[:B])
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(10):
        unet(x8, tt, e8, added_cond_kwargs=added)
    ev[1].record(); ev[2].record()
    for _ in range(3):
        vae.decode(x8[:B])
    ev[3].record(); torch.cuda.synchronize()
    unet_ms = ev[0].elapsed_time(ev[1]) / 10
    vae_ms = ev[2].elapsed_time(ev[3]) / 3

    result = None
    if rank == 0:
        u_tf = UNET_TFLOP_PER_SAMPLE[args.preset].get(lat_hw)
        v_tf = VAE_TFLOP_PER_IMAGE.get(lat_hw)
        result = {
            "metric": f"denoise-loop latents/s ({args.res}px, {args.denoise_steps}-step {args.scheduler}, batch {B})",
            "value": round(value, 4), "unit": "latents/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{'SD1.5' if args.preset == 'sd15' else 'SDXL-base'} {args.res}x{args.res}, "
                                   f"{args.denoise_steps}-step {args.scheduler}, "
                                   f"batch {B}/GPU, CFG on (UNet batch {2 * B}), UNet + VAE decode HIP kernels",
                       "global_batch": total, "parallelism": f"dp{n_gpus}", "guidance_scale": args.guidance},
            "unet_forward_ms": round(unet_ms, 3), "vae_decode_ms": round(vae_ms, 3),
            "outputs_finite": finite, "unet_hipgraph": bool(args.graph), "box_probe": probe,
            "extra": {"latents_4x128x128": large},
        }
        if probe:
            result["value_per_probe_pflops"] = round(value / (probe["mfma_tflops"] / 1e3), 4)
        if u_tf and v_tf:
            tflop = args.denoise_steps * 2 * B * u_tf + B * v_tf     # per GPU per pass
            result["whole_path_mfma_frac"] = round(tflop / (ms_per_step / 1e3) / MFMA_PEAK_TFLOPS, 4)
            result["unet_forward_mfma_frac"] = round(2 * B * u_tf / (unet_ms / 1e3) / MFMA_PEAK_TFLOPS, 4)
        if not args.no_roofline:
            urows, vrows = live_roofline(lib, unet, vae, B, lat_hw, e8, device, added)
            result.update(roofline_report(urows, vrows))
        if n_gpus == 1 and not args.no_cpu_baseline and args.preset == "sd15":
            del unet, vae, model
            torch.cuda.empty_cache()
            result["cpu_baseline"] = cpu_baseline(ucfg, vcfg, usd, vsd, args.denoise_steps, lat_hw, 77)
    sdd.barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
