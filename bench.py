#!/usr/bin/env python
"""Benchmark of the PhysicEdit denoising hot path on MI355X.

One "step" = ONE EDITED IMAGE through the whole hot path (BASELINE.json configs[1]):
    VAE-encode the 1024x1024 edit image  ->  40 flow-match steps x {posi forward (T=512), nega forward
    (T=272), CFG 4.0 combine, Euler update} on the 60-layer Qwen-Image DiT with a merged rank-128 LoRA
    and the visual-thinking adapter on 64 special tokens  ->  VAE-decode to 1024x1024.
Synthetic inputs and random-init weights of the real architecture (no network).  Inputs are resident
in HBM when the timed region starts.  N > 1: one process per GPU (torchrun), weights replicated,
images sharded over ranks (weak scaling, no data-path collective; an RCCL all-gather of the decoded
latents closes the batch, as the north star specifies).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the bf16 MFMA GEMM):
algorithmic FLOPs / HIP-event time of sampled launches inside the timed region.  `cpu_baseline` times
the oracle (CPU restatement, validated bit-exact against the reference) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2516.6   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense)
PEAK_FP8_TFLOPS = 5033.2    # MI355X dense fp8 MFMA peak (MI355X_MICROARCH.md: ~5 PF dense, block-scaled K=128 forms)


def flops_forward(S_img, T, layers=60):
    """SURVEY.md section 8(a) / BASELINE.md section 3 FLOP model."""
    S = S_img + T
    return layers * (226_492_416 * S + 12_288 * S * S + 226_492_416) + 786_432 * S_img + 22_020_096 * T + 58e6


def flops_image(height, width, steps, T_pos, T_neg, cfg, layers=60, edit_hw=(1024, 1024)):
    S0 = (height // 16) * (width // 16)
    Se = (edit_hw[0] // 16) * (edit_hw[1] // 16)
    f = steps * (flops_forward(S0 + Se, T_pos, layers) + (flops_forward(S0 + Se, T_neg, layers) if cfg != 1.0 else 0))
    f += steps * (2 if cfg != 1.0 else 1) * 19.73e9                       # adapter heads
    vae = (2.44e12 + 0.41e12) * (edit_hw[0] * edit_hw[1]) / (1024 * 1024) + (4.30e12) * (height * width) / (1024 * 1024) \
        + 0.41e12 * ((height * width) / (1024 * 1024)) ** 2
    return f + vae


def flops_trimmed_last_block(height, width, steps, T_pos, T_neg, cfg, edit_hw=(1024, 1024)):
    """FLOPs per image that the library does NOT execute (dit.hip, round 6): of the last block only the S0 noise rows' post-attention
    work survives the slice behind it (qwen_image_physical.py:1398-1402), so the other rows' out-projection + MLP (169,869,312 FLOPs per
    row) and attention queries (12,288 S per row) are not launched.  The algorithmic count keeps them."""
    S0 = (height // 16) * (width // 16)
    Se = (edit_hw[0] // 16) * (edit_hw[1] // 16)
    f = 0.0
    for T in ((T_pos, T_neg) if cfg != 1.0 else (T_pos,)):
        S = S0 + Se + T
        f += (S - S0) * (169_869_312 + 12_288 * S)
    return steps * f


def host_threads_per_rank(world, cpus=None):
    """the ranks of one node share its host: the CPU-side parts of the model build (LoRA / adapter / VAE tensors are generated with torch
    CPU ops) must not oversubscribe it N-fold, and more than 32 threads do not help a generator-bound build"""
    cpus = cpus or os.cpu_count() or 8
    return max(1, min(32, cpus // max(world, 1)))


def launcher_command(n_gpus, argv, port, script=None):
    """the command `python bench.py --gpus N` turns itself into: one rank per GPU under torch.distributed.run on 127.0.0.1 -- the
    driver's own command line"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1, help="timed images per rank")
    ap.add_argument("--warmup", type=int, default=1, help="untimed images per rank")
    ap.add_argument("--layers", type=int, default=60)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--inference-steps", type=int, default=40)
    ap.add_argument("--cfg", type=float, default=4.0)
    ap.add_argument("--t-pos", type=int, default=512)
    ap.add_argument("--t-neg", type=int, default=272)
    ap.add_argument("--lora-rank", type=int, default=128)
    ap.add_argument("--single-stream", action="store_true",
                    help="run the posi and the nega forward of a step one after the other on one HIP stream.  Default since round 3: "
                         "two streams (DenoiseLoop(dual_stream=True): second workspace on the same weights; the branches are "
                         "independent until the CFG combine and each fills the CUs the other leaves idle at the end of a kernel: "
                         "-2.6 %% time per image on the same box, bit-identical images)")
    ap.add_argument("--dual-stream", action="store_true", help="(default; kept for command lines written before round 3)")
    ap.add_argument("--fp8", action="store_true",
                    help="BASELINE configs[2]: DiT stored in e4m3 + enable_dit_fp8_computation (every DiT Linear runs "
                         "fp8_linear); NOT the headline configuration, reported with dtype fp8")
    ap.add_argument("--fp8-attention", action="store_true",
                    help="with --fp8: also enable_fp8_attention=True (e4m3 attention, pe_flash_attn_fp8) in the timed loop -- the third "
                         "`secondary` line of the default run as the primary workload (profiling); reported in config.workload")
    ap.add_argument("--attn-variant", type=int, default=5, choices=[0, 3, 4, 5, 6, 7],
                    help="flash-attention kernel: 5 = library default since round 4 (4 waves x 64 rows, one wave per SIMD, lazy running max; "
                         "the softmax scale is folded into Q by the QKV epilogue and the max enters through the MFMA C operand: "
                         "profiles/r04_attention_notes.md); 4 = round 3's default (same schedule, scale and max applied per score); 6 / 3 = "
                         "5 / 4 with the textbook max update; 0 = 8 waves x 32 rows (the round-1/2 default).  Values other than 5 are "
                         "A/B knobs, reported in config.attn_variant")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (backend nccl = RCCL) and run the batch-closing all-gather, the barriers and "
                         "the max-over-ranks all-reduce even at --gpus 1, so that the only thing a 1-GPU box leaves unexecuted of "
                         "the N > 1 path is the rank count (tests/test_gpu_parallel.py)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block (after the headline timed region, N = 1 headline runs also time 2 images each of "
                         "the 1328x1328 / 50-step geometry of configs[4] on this one GPU and of configs[2], the e4m3 Linears)")
    ap.add_argument("--no-probes", action="store_true",
                    help="skip the matrix-pipe / GEMM-mix probes after the timed region (profiling runs: keeps them out of the trace)")
    ap.add_argument("--no-prologue", action="store_true",
                    help="skip the `prologue` block (N = 1 headline runs time the text-encoder prologue at its real size after the timed "
                         "region: prefill seconds, captured decode tokens/s and achieved HBM TB/s; SURVEY.md section 8(d))")
    ap.add_argument("--no-self-check", action="store_true",
                    help="skip the determinism self-check (N = 1 headline runs re-run the first timed image on ONE stream with GEMM "
                         "schedule 15 -- one tile per work-group -- and require bit-identical final latents and pixels)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    args = ap.parse_args()
    args.dual_stream = not args.single_stream

    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ and not args.cpu_baseline_only:
        # `python bench.py --gpus N` without a launcher: become the launcher.  One process per GPU under
        # torch.distributed.run on 127.0.0.1 (the same command line the driver uses); rank 0 prints the JSON line.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = launcher_command(args.gpus, sys.argv[1:], port)
        print(f"[bench] spawning {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` without a launcher: it spawns the ranks itself)")
    if not args.cpu_baseline_only and torch.cuda.device_count() < world:
        raise SystemExit(f"[bench] --gpus {world} but only {torch.cuda.device_count()} visible GPU(s)")
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)))
        return

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm
    if world > 1:
        torch.set_num_threads(host_threads_per_rank(world))

    from physicedit_amd import synth, ops
    from physicedit_amd._lib import lib
    from physicedit_amd.dit import QwenImageDiTEngine
    from physicedit_amd.pipeline import DenoiseLoop
    from physicedit_amd.vae import QwenImageVAE, preprocess_image
    import ctypes as C

    BF = torch.bfloat16
    H, W = args.height, args.width
    t0 = time.time()
    # ---- model: random-init weights of the real architecture, generated on the device
    sd = synth.make_state_dict_device(synth.dit_layout(args.layers), 1234, dev)
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    eng = QwenImageDiTEngine(sd, ad, device=dev)
    del sd
    torch.cuda.empty_cache()
    if args.lora_rank > 0:
        # PhysicEdit LoRA, merged at load like validate.py does (hotload=False), 12 targets x layers.  The tensors are generated
        # with torch CPU ops: at N > 1 the ranks take turns two at a time (rank r starts when rank r - 2 has generated its set), so
        # that eight processes do not all hammer the host's memory system in the same seconds
        if dist is not None and world > 2:
            for turn in range(rank // 2):
                dist.barrier()
        n = 0
        for i in range(args.layers):
            lora = {k.replace("transformer_blocks.0.", f"transformer_blocks.{i}."): v.to(dev)
                    for k, v in synth.make_lora(4321 + i, 1, args.lora_rank).items()}
            n += eng.load_lora(lora)
        assert n == 12 * args.layers
        if dist is not None and world > 2:
            for turn in range(rank // 2, (world - 1) // 2):
                dist.barrier()
    if args.fp8:
        eng.enable_fp8_computation()
        torch.cuda.empty_cache()
    vae = QwenImageVAE(synth.make_state_dict(synth.vae_layout(), 77), device=dev)
    loop = DenoiseLoop(eng, dual_stream=args.dual_stream)
    if args.attn_variant != 5:
        from physicedit_amd._lib import lib
        assert lib().pe_debug_set(b"attn_variant", args.attn_variant) == 0
    torch.cuda.synchronize()
    ready_s = time.time() - t0
    # every rank says when its model stood (stderr): a slow or failed rank of an N > 1 run is visible from the tail alone
    print(f"[bench] rank {rank}/{world} (cuda:{local_rank}, {torch.get_num_threads()} host threads): model ready in {ready_s:.1f}s "
          f"({args.layers} layers, {torch.cuda.memory_allocated()/2**30:.1f} GiB resident)", file=sys.stderr, flush=True)
    ready_all = [ready_s]
    if dist is not None:
        ready_all = [None] * world
        dist.all_gather_object(ready_all, ready_s)
        if rank == 0:
            print(f"[bench] all {dist.get_world_size()} RCCL ranks ready: " + " ".join(f"r{i}={t:.1f}s" for i, t in enumerate(ready_all)),
                  file=sys.stderr, flush=True)

    # ---- inputs (resident in HBM before the timed region)
    n_img = args.warmup + args.steps
    edit_img = preprocess_image(synth.make_edit_image_u8(1024, 1024, 0), dev)       # edit image auto-resized to ~1024^2 area
    pe_p0 = synth.make_prompt_emb(7, args.t_pos).to(dev)
    pe_n0 = synth.make_prompt_emb(8, args.t_neg).to(dev)
    mask_p = synth.make_special_token_mask(args.t_pos)
    mask_n = synth.make_special_token_mask(args.t_neg)
    # work units = images; unit u belongs to rank u % world and its noise seed is u (not the rank), so
    # any world size edits the same set of images (parallel.shard_units)
    from physicedit_amd import parallel
    warm_units = [10_000 + rank * 100 + i for i in range(args.warmup)]
    units = parallel.shard_units(args.steps * world, rank, world)
    noises = {u: synth.make_noise(u, H, W).to(dev) for u in warm_units + units}
    results = []

    def one_image(i, loop_=None):
        edit_latents = vae.encode(edit_img)
        lat = (loop_ or loop)(noises[i], pe_p0.clone(), pe_n0.clone(), mask_p, mask_n, H, W,
                              num_inference_steps=args.inference_steps, cfg_scale=args.cfg, edit_latents=edit_latents,
                              enable_fp8_attention=args.fp8_attention)
        img = vae.decode(lat)
        return lat, img

    def barrier():
        if dist is not None:
            dist.barrier()

    for u in warm_units:
        one_image(u)
    torch.cuda.synchronize()
    barrier()
    # sample every 16th GEMM launch: ~1.2k event pairs per image
    lib().pe_profile_enable(16384, 16)
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for u in units:
        results.append(one_image(u))
    if dist is not None:
        # close the batch INSIDE the timed region: ONE RCCL all-gather of the final latents over xGMI
        # (512 KiB per image)
        gathered = parallel.gather_units([r[0] for r in results], args.steps * world, always_collective=True)
        assert len(gathered) == args.steps * world
    torch.cuda.synchronize()
    barrier()
    t_end = time.perf_counter()
    elapsed = t_end - t_start
    per_rank = [elapsed]
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, elapsed)
        elapsed = float(tmax.item())
    ok = all(torch.isfinite(r[1].float()).all().item() for r in results)

    def read_prof():
        out_ = {}
        for kind, name in ((0, "gemm"), (1, "attn"), (2, "row"), (3, "conv")):
            seen, sampled, ms, work = C.c_longlong(), C.c_longlong(), C.c_double(), C.c_double()
            lib().pe_profile_read(kind, C.byref(seen), C.byref(sampled), C.byref(ms), C.byref(work))
            out_[name] = dict(launches=seen.value, sampled=sampled.value, ms=ms.value, work=work.value)
        lib().pe_profile_disable()
        return out_

    prof = read_prof()
    # The bench proves its own outputs (outside the timed region, N = 1 headline runs): the first timed image again on ONE stream
    # with GEMM schedule 15 (one tile per work-group: no persistent work-groups, no cross-tile prefetch).  Every schedule performs
    # the same arithmetic in the same order, so the final latents and the decoded pixels must be BIT-IDENTICAL to the timed
    # two-stream / schedule-17 result; a race in the persistent GEMM, the attention ring or the two-stream loop shows up here at
    # 60 layers x 40 steps.  A mismatch fails the run.
    determinism = None
    if rank == 0 and world == 1 and not args.no_self_check and args.steps > 0:
        t_chk = time.perf_counter()
        assert lib().pe_debug_set(b"gemm_variant", 15) == 0
        try:
            lat1, img1 = one_image(units[0], DenoiseLoop(eng, dual_stream=False))
            torch.cuda.synchronize()
        finally:
            lib().pe_debug_set(b"gemm_variant", 0)
        same_lat = bool(torch.equal(lat1, results[0][0]))
        same_img = bool(torch.equal(img1, results[0][1]))
        determinism = {"bit_identical": same_lat and same_img, "latents_equal": same_lat, "pixels_equal": same_img,
                       "what": "timed image 0 (two streams, GEMM schedule 21) vs the same unit on one stream with GEMM schedule 15, "
                               f"{args.layers} layers x {args.inference_steps} steps", "seconds": round(time.perf_counter() - t_chk, 2)}
        if not (same_lat and same_img):
            d = (lat1.float() - results[0][0].float()).abs()
            determinism["max_abs_latent_diff"] = float(d.max().item())
            determinism["latent_elements_differing"] = int((d > 0).sum().item())
        del lat1, img1
    # diagnostic, OUTSIDE the timed region: the same kernels with the chip to themselves (one positive forward on
    # one stream, every launch sampled).  With two streams the timed-region launch durations include sharing the
    # CUs with the sibling branch's kernel, so they understate what a kernel achieves alone.
    prof_excl = None
    if rank == 0:
        lib().pe_profile_enable(8192, 1)
        ed = [vae.encode(edit_img)]
        torch.cuda.synchronize()
        eng.forward(noises[units[0]], torch.tensor([500.0]).to(BF), pe_p0.clone(), None, ed, step=0, enable_fp8_attention=args.fp8_attention)     # the step's positive ...
        eng.forward(noises[units[0]], torch.tensor([500.0]).to(BF), pe_n0.clone(), None, ed, step=0, enable_fp8_attention=args.fp8_attention)     # ... and negative forward
        torch.cuda.synchronize()
        prof_excl = read_prof()

    if rank == 0:
        images = args.steps * world
        value = images / elapsed
        fl = flops_image(H, W, args.inference_steps, args.t_pos, args.t_neg, args.cfg, args.layers)
        g = prof["gemm"]
        timed_tf = (g["work"] / (g["ms"] * 1e-3) / 1e12) if g["ms"] > 0 else 0.0
        # Two streams: the event pair around a launch in the timed region also brackets whatever the sibling stream runs on the same
        # CUs meanwhile, so "work / duration" there is not the kernel's rate.  The roofline then comes from the same process's
        # single-stream CFG pair of forwards right after the timed region (same kernels, shapes and operands, EVERY launch sampled);
        # the timed-region figures stay in `roofline.timed_region_two_streams`.  `--single-stream` samples inside the timed region.
        excl_ok = prof_excl is not None and prof_excl["gemm"]["ms"] > 0
        use_excl = args.dual_stream and args.cfg != 1.0 and excl_ok
        gsrc = prof_excl["gemm"] if use_excl else g
        psrc = prof_excl if use_excl else prof
        achieved = (gsrc["work"] / (gsrc["ms"] * 1e-3) / 1e12) if gsrc["ms"] > 0 else 0.0
        traffic, mfma_busy, traffic_src = pmc_traffic(args)
        headline = (H, W, args.inference_steps, args.cfg, args.layers) == (1024, 1024, 40, 4.0, 60)
        cfg_label = f"configs[{2 if args.fp8 else 1}]" if headline else (
            "configs[4] geometry on one GPU" if (H, W, args.inference_steps, args.layers) == (1328, 1328, 50, 60) else "non-headline geometry")
        peak = PEAK_FP8_TFLOPS if args.fp8 else PEAK_BF16_TFLOPS
        gemm_name = "gemm_fp8_kernel (e4m3 operands, all epilogues)" if args.fp8 else "gemm_bf16_kernel (all epilogues)"
        out = {
            "metric": "edited images/sec @1024px, 40-step flow-match",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp8_e4m3" if args.fp8 else "bf16", "data": "synthetic",
            "config": {"workload": f"{cfg_label}{(' (DiT Linears in e4m3: fp8_linear; ' + ('e4m3 attention: enable_fp8_attention; ' if args.fp8_attention else 'attention, ') + 'norms, adapter, VAE bf16)') if args.fp8 else ''}: {H}x{W} edit, {args.inference_steps} steps, CFG {args.cfg}, "
                                   f"{args.layers}-layer Qwen-Image DiT + merged rank-{args.lora_rank} LoRA + "
                                   f"visual-thinking adapter (64 special tokens), T_pos={args.t_pos} T_neg={args.t_neg}, "
                                   f"VAE encode(1024x1024 edit image)+decode included",
                       "images_per_rank": args.steps, "parallelism": f"dp{world} (images sharded, weights replicated)",
                       "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
                       "attn_variant": args.attn_variant,
                       "batch_closing_collective": ("one RCCL all_gather of the final latents, inside the timed region" if dist is not None else None),
                       "per_rank_elapsed_s": [round(float(t), 4) for t in per_rank],
                       "per_rank_model_ready_s": [round(float(t), 1) for t in ready_all],
                       "host_threads_per_rank": torch.get_num_threads(),
                       "finite_outputs": ok},
            "determinism": (determinism or {}).get("bit_identical"),
            "self_check": determinism,
            "whole_path": {"algorithmic_pflop_per_image": fl / 1e15,
                           "executed_pflop_per_image": (fl - flops_trimmed_last_block(H, W, args.inference_steps, args.t_pos, args.t_neg, args.cfg)
                                                        * (1 if args.layers > 0 else 0)) / 1e15,
                           "achieved_tflops_per_gpu": fl * value / world / 1e12,
                           "frac_of_bf16_mfma_peak": fl * value / world / 1e12 / PEAK_BF16_TFLOPS,
                           "frac_of_operand_dtype_mfma_peak": fl * value / world / 1e12 / peak,
                           "note": ("e4m3 Linears run on the 2x-rate block-scaled MFMA (peak %.0f TF/s), attention stays bf16: the "
                                    "bf16 fraction can exceed what a bf16 path could reach" % PEAK_FP8_TFLOPS) if args.fp8 else None},
            "roofline": {"kernel": gemm_name, "bound": "mfma", "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "mfma_busy": mfma_busy, "traffic_source": traffic_src,
                         "launches_in_timed_region": g["launches"], "launches_sampled": gsrc["sampled"],
                         "avg_launch_ms": gsrc["ms"] / max(gsrc["sampled"], 1),
                         "avg_algorithmic_gflop_per_launch": gsrc["work"] / max(gsrc["sampled"], 1) / 1e9,
                         "concurrent_streams": 2 if args.dual_stream else 1,
                         "sampled_in": ("one single-stream CFG pair of forwards in this process right after the timed region, every "
                                        "launch (the timed region runs the pair on two streams: a launch's event-to-event time there "
                                        "includes the sibling stream's kernels)") if use_excl else "the timed region (one stream)",
                         "timed_region_two_streams": ({"work_over_event_time_tflops": timed_tf, "launches_sampled": g["sampled"],
                                                       "avg_event_ms": g["ms"] / max(g["sampled"], 1)} if use_excl else None)},
            "roofline_exclusive": None if not prof_excl or prof_excl["gemm"]["ms"] <= 0 else {
                "what": "the GEMM launches of one untimed CFG pair of forwards alone on the chip (single stream, all launches sampled)",
                "achieved": prof_excl["gemm"]["work"] / (prof_excl["gemm"]["ms"] * 1e-3) / 1e12, "peak": peak,
                "unit": "TFLOP/s", "frac": prof_excl["gemm"]["work"] / (prof_excl["gemm"]["ms"] * 1e-3) / 1e12 / peak,
                "flash_attn_tflops": (prof_excl["attn"]["work"] / (prof_excl["attn"]["ms"] * 1e-3) / 1e12) if prof_excl["attn"]["ms"] > 0 else None},
            "other_kernels": {        # sampled like `roofline` (two streams: on the single-stream CFG pair after the timed region)
                "flash_attn": {"achieved_tflops": (psrc["attn"]["work"] / (psrc["attn"]["ms"] * 1e-3) / 1e12) if psrc["attn"]["ms"] > 0 else None,
                               "frac_of_bf16_mfma_peak": (psrc["attn"]["work"] / (psrc["attn"]["ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS) if psrc["attn"]["ms"] > 0 else None,
                               "avg_launch_ms": psrc["attn"]["ms"] / max(psrc["attn"]["sampled"], 1)},
                "row_kernels(ln_modulate,quantize_rows)": {"achieved_GBps": (psrc["row"]["work"] / (psrc["row"]["ms"] * 1e-3) / 1e9) if psrc["row"]["ms"] > 0 else None},
                "vae_conv": {"achieved_tflops": (prof["conv"]["work"] / (prof["conv"]["ms"] * 1e-3) / 1e12) if prof["conv"]["ms"] > 0 else None},
            },
        }
        if not args.fp8 and not args.no_probes:
            ceil = mfma_power_ceiling(dev)
            out["roofline"]["power_limited_ceiling"] = ceil
            if ceil["random_normal_operands"] > 0:
                out["roofline"]["frac_of_power_limited_ceiling"] = achieved / ceil["random_normal_operands"]
            att = attainable_ceiling(dev)
            out["roofline"]["attainable_ceiling"] = att
            if att.get("mfma_lds_reads_dma_stream", 0) > 0:
                out["roofline"]["frac_of_attainable_ceiling"] = achieved / att["mfma_lds_reads_dma_stream"]
            fa = out["other_kernels"]["flash_attn"]
            if args.attn_variant >= 5 and fa.get("achieved_tflops"):
                ac = attention_ceiling(dev, (H // 16) * (W // 16) + 4096 + args.t_pos)
                fa["attainable_ceiling"] = ac
                if ac.get("tflops", 0) > 0:
                    fa["frac_of_attainable_ceiling"] = fa["achieved_tflops"] / ac["tflops"]
        if world == 1 and headline and not args.fp8 and not args.no_secondary:
            out["secondary"] = secondary_configs(args, dev, eng, vae, edit_img, pe_p0, pe_n0, mask_p, mask_n, read_prof)
        if world == 1 and headline and not args.fp8 and not args.no_prologue:
            out["prologue"] = prologue_block(dev)
        if not args.no_cpu_baseline and world == 1:     # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
        if determinism is not None and not determinism["bit_identical"]:
            print("[bench] SELF-CHECK FAILED: the single-stream / schedule-15 re-run of timed image 0 is not bit-identical "
                  f"({determinism})", file=sys.stderr)
            if dist is not None:
                dist.destroy_process_group()
            sys.exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def secondary_configs(args, dev, eng, vae, edit_img, pe_p0, pe_n0, mask_p, mask_n, read_prof, n_images=2):
    """The other single-GPU configurations of BASELINE.json, timed in the SAME process after the headline region (so the
    driver's run carries them): (a) configs[4]'s geometry -- 1328 x 1328 target (83 x 83 noise tokens), 50 steps, same 1024 x 1024
    edit image, S = 11497 -- on this one GPU; (b) configs[2] -- every DiT Linear as fp8_linear on e4m3 weights.  Each: one
    2-step warm-up image (workspace, RoPE tables), then `n_images` full images between synchronisations; GEMM rate from HIP
    events of the launches inside that region.  (b) converts the weights in place, so it runs last."""
    import torch
    from physicedit_amd import synth
    from physicedit_amd._lib import lib
    from physicedit_amd.pipeline import DenoiseLoop
    res = {}

    def run(label, H, W, steps, peak, dtype, fp8_attention=False):
        loop = DenoiseLoop(eng, dual_stream=args.dual_stream)
        noises = [synth.make_noise(5000 + i, H, W).to(dev) for i in range(n_images + 1)]

        def image(i, n_steps):
            lat = loop(noises[i], pe_p0.clone(), pe_n0.clone(), mask_p, mask_n, H, W, num_inference_steps=n_steps,
                       cfg_scale=args.cfg, edit_latents=vae.encode(edit_img), enable_fp8_attention=fp8_attention)
            return vae.decode(lat)
        image(0, 2)
        torch.cuda.synchronize()
        lib().pe_profile_enable(8192, 32)
        t0 = time.perf_counter()
        imgs = [image(1 + i, steps) for i in range(n_images)]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_images
        prof = read_prof()
        sampled_in = "the timed images (one stream)"
        if args.dual_stream and args.cfg != 1.0:
            # as for the headline: with two streams a launch's event time includes the sibling stream's kernels, so the rates come
            # from one single-stream CFG pair of forwards of this geometry right after the timed images, every launch sampled
            lib().pe_profile_enable(8192, 1)
            ed = [vae.encode(edit_img)]
            t500 = torch.tensor([500.0]).to(torch.bfloat16)
            torch.cuda.synchronize()
            eng.forward(noises[0], t500, pe_p0.clone(), None, ed, step=0, enable_fp8_attention=fp8_attention)
            eng.forward(noises[0], t500, pe_n0.clone(), None, ed, step=0, enable_fp8_attention=fp8_attention)
            torch.cuda.synchronize()
            prof = read_prof()
            sampled_in = "one single-stream CFG pair of forwards right after the timed images, every launch"
        g = prof["gemm"]
        fl = flops_image(H, W, steps, args.t_pos, args.t_neg, args.cfg, args.layers)
        gemm_tf = g["work"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else None
        res[label] = {"ms_per_image": dt * 1e3, "images_per_s": 1.0 / dt, "images_timed": n_images, "dtype": dtype,
                      "geometry": f"{H}x{W}, {steps} steps, CFG {args.cfg}, T_pos={args.t_pos} T_neg={args.t_neg}",
                      "algorithmic_pflop_per_image": fl / 1e15, "achieved_tflops": fl / dt / 1e12,
                      "frac_of_bf16_mfma_peak": fl / dt / 1e12 / PEAK_BF16_TFLOPS,
                      "roofline": {"kernel": "gemm (all epilogues)", "bound": "mfma", "achieved": gemm_tf, "peak": peak, "unit": "TFLOP/s",
                                   "frac": (gemm_tf / peak) if gemm_tf else None, "launches_sampled": g["sampled"], "sampled_in": sampled_in},
                      "flash_attn_tflops": (prof["attn"]["work"] / (prof["attn"]["ms"] * 1e-3) / 1e12) if prof["attn"]["ms"] > 0 else None,
                      "finite_outputs": all(torch.isfinite(x.float()).all().item() for x in imgs)}

    run("configs[4] geometry on one GPU (1328x1328, 50 steps, bf16)", 1328, 1328, 50, PEAK_BF16_TFLOPS, "bf16")
    eng.enable_fp8_computation()
    torch.cuda.empty_cache()
    run("configs[2] (DiT Linears in e4m3: fp8_linear; attention, norms, adapter, VAE bf16)", args.height, args.width,
        args.inference_steps, PEAK_FP8_TFLOPS, "fp8_e4m3")
    # the same with enable_fp8_attention=True: q / k / v scaled by their global std and cast to e4m3, both attention matmuls on e4m3
    # operands (pe_flash_attn_fp8; `flash_attn_tflops` then covers the statistics, the quantisation pass and the kernel together)
    run("configs[2] + enable_fp8_attention (e4m3 Linears AND e4m3 attention)", args.height, args.width, args.inference_steps,
        PEAK_FP8_TFLOPS, "fp8_e4m3", fp8_attention=True)
    return res


def prologue_block(dev):
    """"report prologue (text encoder) separately".  The Qwen2.5-VL-7B text encoder at its real shape with
    random weights (tools/prologue_time.py, quick form): the two embedding passes of an edit, the prefill of the physical-text
    generation, and the captured greedy decode step -- tokens/s and the HBM rate of the weights it streams per token."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import prologue_time
        t0 = time.perf_counter()
        r = prologue_time.measure(decode_tokens=128, quick=True)
        g = r.get("generate_captured_decode_step") or {}
        return {"what": "text-encoder prologue (Qwen2.5-VL-7B shape, random weights, synthetic byte-level tokenizer), timed after the "
                        "headline region in this process; NOT part of `value` (the reference computes it once per prompt, outside the "
                        "denoising loop: qwen_image_physical.py:859-873, 943-967)",
                "embed_positive_seconds": r["embed_positive"]["seconds"], "embed_positive_tokens": r["embed_positive"]["tokens_after_drop"],
                "embed_negative_seconds": r["embed_negative"]["seconds"], "embed_negative_tokens": r["embed_negative"]["tokens_after_drop"],
                "generate_prompt_tokens": r["prompt_tokens"], "prefill_plus_1_token_seconds": g.get("prefill_plus_1_token_seconds"),
                "decode_tokens_per_second": g.get("decode_tokens_per_second"), "decode_tokens_timed": r["new_tokens_timed"],
                "weight_bytes_streamed_per_token": g.get("weight_bytes_streamed_per_token"), "achieved_TBps": g.get("achieved_TBps"),
                "frac_of_hbm_peak": g.get("frac_of_hbm_peak_8TBps"), "hbm_peak_TBps": 8.0,
                "seconds_for_1000_new_tokens_extrapolated": g.get("extrapolated_seconds_for_1000_new_tokens"),
                "token_ids_sha1": g.get("token_ids_sha1"), "block_seconds": round(time.perf_counter() - t0, 1)}
    except Exception as e:      # the block is a report, not the metric: never lose the line to it
        return {"error": f"{type(e).__name__}: {e}"}


def mfma_power_ceiling(dev):
    """The library's matrix-pipe probe (pe_mfma_probe: bf16 MFMAs only, no LDS / memory traffic), timed here after the timed region
    on N(0,1) bf16 fragments (~0.3 s sustained) and on zeros: `peak` in `roofline` is the nominal 2.4 GHz figure, which this chip
    only holds on zero operands; on random operands its power limit caps the matrix pipe itself at the rate reported here."""
    import ctypes
    import torch
    from physicedit_amd._lib import check, lib, stream_ptr
    BF = torch.bfloat16
    res = {}
    out = torch.empty(512 * 512, dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    for name in ("random_normal", "zeros"):
        frags = (torch.randn(8 << 20, generator=g, device=dev).to(BF) if name == "random_normal"
                 else torch.zeros(8 << 20, dtype=BF, device=dev))
        fl = ctypes.c_double(0.0)
        best = 0.0
        for iters in (2000, 120000, 120000):          # warm-up, then two sustained launches
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib().pe_mfma_probe(frags.data_ptr(), out.data_ptr(), 512, iters, ctypes.byref(fl), stream_ptr()), "pe_mfma_probe")
            e1.record()
            torch.cuda.synchronize()
            if iters > 2000:
                best = max(best, fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        res[name] = best
    return {"what": "pe_mfma_probe: v_mfma_f32_16x16x32_bf16 only (the shape the GEMM runs on since round 5; 32x32x16 sustains 14 - 16 % less), 8 waves/CU, operands in registers, measured in this process "
                    "after the timed region", "unit": "TFLOP/s", "random_normal_operands": res["random_normal"],
            "zero_operands": res["zeros"]}


def attainable_ceiling(dev):
    """How much of the distance to the nominal peak is the schedule's, and how much belongs to the tiling itself: pe_gemm_mix_probe
    issues the bf16 GEMM main loop's per-MFMA instruction mix -- per K tile and wave 32 MFMAs, 24 ds_read_b128 fragment reads, 8 LDS-DMA
    pieces from an L2-resident stream -- free-running (no barriers, no waits for arriving data, no epilogue), on N(0,1) operands, after
    the timed region.  Plus the library GEMM itself with its fixed costs amortised away: one round of 256 tiles with K = 32768."""
    import ctypes
    import torch
    from physicedit_amd import ops
    from physicedit_amd._lib import check, lib, stream_ptr
    BF = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(2)
    src = torch.randn(8 << 20, generator=g, device=dev).to(BF)            # 16 MiB = 8 windows of 2 MiB
    outb = torch.empty(256 * 512, dtype=torch.float32, device=dev)
    res = {"what": "pe_gemm_mix_probe (8 waves / CU, 160 KiB LDS, N(0,1) bf16, L2-resident operand stream, no barriers / epilogue); "
                   "gemm_main_loop_only = pe_gemm_bf16 at 4096 x 4096 x 32768 (one round of tiles, 512 K tiles each)", "unit": "TFLOP/s"}
    for mode, name in ((0, "mfma_only"), (1, "mfma_lds_reads"), (2, "mfma_lds_reads_dma_stream")):
        fl = ctypes.c_double(0.0)
        best = 0.0
        for iters in (2000, 150000, 150000):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib().pe_gemm_mix_probe(mode, src.data_ptr(), src.numel() * 2, outb.data_ptr(), 256, iters, ctypes.byref(fl),
                                          stream_ptr()), "pe_gemm_mix_probe")
            e1.record()
            torch.cuda.synchronize()
            if iters > 2000:
                best = max(best, fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        res[name] = best
    M = N = 4096
    K = 32768
    x = torch.randn((M, K), generator=g, device=dev).to(BF)
    w = (torch.randn((N, K), generator=g, device=dev) * K ** -0.5).to(BF)
    o = torch.empty((M, N), dtype=BF, device=dev)
    ops.gemm(x, w, None, out=o)
    best = 0.0
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(60):
            ops.gemm(x, w, None, out=o)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 60 * 2.0 * M * N * K / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    res["gemm_main_loop_only"] = best
    return res


def attention_ceiling(dev, S):
    """pe_attn_mix_probe: the default flash-attention kernel's own schedule without its softmax (per KV tile and wave 64 MFMAs, 32 LDS
    fragment reads, 8 LDS-DMA pieces, one barrier) on N(0,1) Q / K / Vt at the headline sequence length, after the timed region."""
    import ctypes
    import torch
    from physicedit_amd import ops
    from physicedit_amd._lib import check, lib, stream_ptr
    BF = torch.bfloat16
    H = 24
    g = torch.Generator(device=dev).manual_seed(3)
    sp = ops.s_pad_of(S)
    q = torch.randn((H, sp, 128), generator=g, device=dev).to(BF)
    k = torch.randn((H, sp, 128), generator=g, device=dev).to(BF)
    vt = torch.randn((H, 128, sp), generator=g, device=dev).to(BF)
    out = torch.empty((S, H * 128), dtype=BF, device=dev)
    fl = ctypes.c_double(0.0)

    def run(n):
        for _ in range(n):
            check(lib().pe_attn_mix_probe(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), H, S, sp, H * 128, ctypes.byref(fl),
                                          stream_ptr()), "pe_attn_mix_probe")
    run(3)
    best = 0.0
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(20)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 20 * fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return {"what": f"pe_attn_mix_probe at S = {S}, H = 24: flash_attn_w4_kernel's schedule (variant 5) without the softmax instruction stream, "
                    "N(0,1) operands, whole (head, q-block) items only", "unit": "TFLOP/s", "tflops": best}


def pmc_traffic(args):
    """HBM-side (L2-miss) bytes per launch and matrix-pipe utilisation of the block GEMMs, launch-weighted over QKV / out-proj /
    MLP-up / MLP-down.  PMC counters cannot be collected from inside this process: they come from the committed rocprofv3 --pmc
    passes of tools/pmc_collect.sh (profiles/r06_pmc.json, which records its own command line).  They are per-launch properties
    of (kernel, shape), so they are reported ONLY when this run launches the same kernels on the same shapes -- same layer count,
    geometry, prompt lengths, operand dtype and stream count as the recorded command -- and are null otherwise."""
    path = os.path.join(ROOT, "profiles", "r02_pmc_fp8.json" if args.fp8 else "r06_pmc.json")
    for older in ("r05_pmc.json", "r04_pmc.json"):
        if not os.path.exists(path) and not args.fp8:
            path = os.path.join(ROOT, "profiles", older)
    if not os.path.exists(path):
        return None, None, None
    rec = json.load(open(path))
    c = rec.get("config", {})
    same = (c.get("layers") == args.layers and c.get("height") == args.height and c.get("width") == args.width and
            c.get("t_pos") == args.t_pos and c.get("t_neg") == args.t_neg and bool(c.get("fp8")) == bool(args.fp8))
    # (the passes are collected on one stream; `traffic_source` says so when this run times two)
    if not same:
        return None, None, f"profiles/{os.path.basename(path)} was collected for another configuration ({rec.get('command')}): not reported"
    # the block GEMMs only (QKV 1224, MLP-up 1632, out-proj / MLP-down 408 work-groups at this geometry): the hoisted modulation
    # GEMMs of pe_dit_prepare (M = steps) share the kernel name but are not what `roofline` is about
    # (since round 4 all four run the persistent schedule on one work-group per CU; the gate + residual instantiation serves out-proj AND
    # MLP-down, so its row has no single algorithmic size: the block GEMMs are recognised by their launch count instead)
    rows = [r for r in rec["kernels"] if "gemm_bf16_kernel" in r["kernel"] and (r.get("algorithmic_gflop_per_launch") or (r.get("launches", 0) >= 300 and r.get("workgroups", 0) >= 256))]
    n = sum(r["launches"] for r in rows)
    if not n:
        return None, None, None
    traffic = sum(r["traffic_bytes_per_launch"] * r["launches"] for r in rows) / n
    busy = [r for r in rows if r.get("mfma_busy") is not None]
    mfma = sum(r["mfma_busy"] * r["launches"] for r in busy) / sum(r["launches"] for r in busy) if busy else None
    note = " -- counters collected with ONE stream on the chip; the timed region of this run interleaves two, whose kernels share each L2 " \
           "and the Infinity Cache, so its per-launch L2-miss bytes can differ" if args.dual_stream else ""
    return traffic, mfma, f"profiles/{os.path.basename(path)} ({rec.get('command')}){note}"


def cpu_baseline(args):
    """The oracle (kind "port": CPU restatement, bit-exact vs the reference on tests/golden) timed on the host cores on
    a BOUNDED sample of the same workload, extrapolated with the layer / step counts:
        t_image = steps * layers * (t_block(T_pos) + t_block(T_neg)) + t_vae_enc + t_vae_dec
    Sample (about a minute of CPU work in total):
      1. one full-width DiT block at the full configs[1] sequence (S_img = 8192, T = T_pos; T_neg scaled by tokens), timed once per
         torch thread count in {8, 16, 32, 64, 128} (stopping past the knee: 256 threads oversubscribe oneDNN, round 1 measured
         3-4x slower than 8 vCPUs that way); the best count's time is the sample, and that count is used for everything below;
      3. VAE encode + decode TIMED at the real image size (one pass each, ~10 s apiece at 1024^2 on 32 threads);
      4. configs[0] (c1) end to end: 2-layer DiT, 512 x 512, 4 steps, CFG off, T = 128, edit image NOT auto-resized
         (S_img = 2048): reported as its own number, not part of the extrapolation."""
    import torch
    import oracle.physicedit_oracle as O      # measured as the CPU baseline; never on the product path
    from physicedit_amd import synth
    BF = torch.bfloat16
    cores = os.cpu_count() or 1
    sd = synth.make_state_dict(synth.dit_block_layout(0), 1234)
    g = torch.Generator().manual_seed(0)
    temb = (torch.randn((1, 3072), generator=g) * 0.5).to(BF)

    def time_block(S_img_, T_, shapes):
        image = torch.randn((1, S_img_, 3072), generator=g).to(BF)
        text = torch.randn((1, T_, 3072), generator=g).to(BF)
        rope = O.rope_tables(shapes, T_)
        t0 = time.perf_counter()
        with torch.no_grad():
            O.block_forward(sd, 0, image, text, temb, rope)
        return time.perf_counter() - t0

    # thread sweep AT THE SHAPE THAT IS REPORTED (round 3 picked the count on a reduced block): the full-width block at the full
    # configs[1] sequence, one run per count after a small warm-up of the thread pool; counts whose first quarter already takes
    # longer than the best full run so far are not run to the end on the reduced block first
    S_img = (args.height // 16) * (args.width // 16) + 4096
    full_shapes = [(1, args.height // 16, args.width // 16), (1, 64, 64)]
    sweep = {}
    for nt in sorted({n for n in (8, 16, 32, 64, 128) if n <= cores}):
        torch.set_num_threads(nt)
        time_block(512, 64, [(1, 16, 16), (1, 16, 16)])            # warm the thread pool
        sweep[nt] = time_block(S_img, args.t_pos, full_shapes)
        if len(sweep) >= 3 and sweep[nt] > 2.5 * min(sweep.values()):
            break                                                  # past the knee: more threads only oversubscribe oneDNN
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t_blk = {args.t_pos: sweep[best]}
    t_blk[args.t_neg] = t_blk[args.t_pos] * (S_img + args.t_neg) / (S_img + args.t_pos)
    S = S_img + args.t_pos
    blk_flops = 226_492_416 * S + 12_288 * S * S + 226_492_416
    cpu_rate = blk_flops / t_blk[args.t_pos]
    # VAE: timed ONCE at the real size (round 5; rounds 1-4 scaled a 256^2 run by pixels, which under-prices the mid-block attention:
    # its work grows with the square of the pixel count), in the oracle's 2-D form of the causal convolutions (one frame: the last
    # temporal tap; pinned to the literal conv3d within 2 ulp by tests/test_oracle_golden.py)
    vs = synth.make_state_dict(synth.vae_layout(), 77)
    x = O.preprocess_image(synth.make_edit_image_u8(args.height, args.width, 0))
    O.VAE_CONV_MODE = "2d"
    try:
        with torch.no_grad():
            t0 = time.perf_counter(); z = O.vae_encode(vs, x); t_enc = time.perf_counter() - t0
            t0 = time.perf_counter(); O.vae_decode(vs, z); t_dec = time.perf_counter() - t0
    finally:
        O.VAE_CONV_MODE = "3d"
    # c1 end to end (loop only, as SURVEY 8d defines the timed region; the VAE is priced above)
    sd2 = synth.make_state_dict(synth.dit_layout(2), 1234)
    noise = synth.make_noise(0, 512, 512)
    edit = torch.randn((1, 16, 64, 64), generator=g).to(BF)
    pe = synth.make_prompt_emb(7, 128)
    with torch.no_grad():
        t0 = time.perf_counter()
        O.denoise_loop(sd2, None, noise, pe, None, None, None, 512, 512, 4, cfg_scale=1.0, edit_latents=edit)
        t_c1 = time.perf_counter() - t0
    per_image = args.inference_steps * args.layers * (t_blk[args.t_pos] + (t_blk[args.t_neg] if args.cfg != 1.0 else 0)) + t_enc + t_dec
    return {"value": 1.0 / per_image, "unit": "images/s", "cores": best, "host_logical_cpus": cores, "kind": "port",
            "sample": f"1 DiT block fwd at full shape S_img={S_img}, T={args.t_pos}, torch threads swept at that shape "
                      f"{{{', '.join(f'{k}: {v:.2f}s' for k, v in sweep.items())}}} -> {best} threads: {t_blk[args.t_pos]:.2f}s "
                      f"(T={args.t_neg} scaled by tokens: {t_blk[args.t_neg]:.2f}s) = {cpu_rate/1e12:.3f} TFLOP/s; "
                      f"VAE timed at {args.height}x{args.width} (2-D form of the causal convolutions): enc {t_enc:.1f}s dec {t_dec:.1f}s; "
                      f"extrapolated x{args.inference_steps} steps x{args.layers} layers",
            "c1_end_to_end_seconds": t_c1,
            "c1_config": "configs[0]: 2-layer DiT, 512x512 + 512x512 edit latents (S_img 2048), T=128, 4 steps, CFG off (loop only)",
            "extrapolated_seconds_per_image": per_image}


if __name__ == "__main__":
    main()
