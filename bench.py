#!/usr/bin/env python
"""bench.py — glyph-crops/s end-to-end (encode + L2-normalise + inner-product top-k) on MI355X.

Contract (see the task statement): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON
line on rank 0.  A *step* is one pass of the hot path over one batch of synthetic crops:
BASELINE.json configs[1] — ViT-S/16 encoder, bf16 MFMA operands, 1024 x 3x224x224 fp32 crops per GPU
already resident in HBM, 10,000-row fp32 glyph index, k=10.  For N>1 (one process per GPU; launched by
torch.distributed.run, or by this script itself when WORLD_SIZE is not set) the default is BASELINE
configs[2]: the SAME 1024 crops sharded over the ranks (``--scaling strong``: rank r encodes rows
shard_bounds(1024, r, N), weights and index replicated) and one RCCL all_gather assembles the [1024, 10]
ids on every rank; the weak-scaling figure (1024 crops per rank) is measured right after and reported
under "weak".  ``--scaling weak`` makes the weak figure the headline instead.

Extra objects on the line:
  roofline      the dominant kernel class, timed live with HIP events recorded on the launch stream
                INSIDE the timed region by the library's own profiler (effocr_encoder_profile_*):
                achieved = algorithmic FLOPs per launch / mean launch duration, vs the dense bf16
                MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md).
  cpu_baseline  rank 0, N=1 only: the CPU restatement of the reference path (oracle/: plain-torch
                fp32 ViT-S + F.normalize + Q@X^T/top-k, i.e. PyTorch-CPU + IndexFlatIP semantics; timm
                and faiss are tried first and their absence is recorded) on a bounded sample of the same workload.
  small_batch   N=1: the reference drivers' real call sizes — 64-crop batches device-resident, and through
                EffRecognizer.run(numpy) (pinned staging + per-call streams; PCIe-inclusive, 1 and 4 caller threads),
                plus the k-NN alone at B in {1, 16, 64} against a 1M-row index where HBM is the roof.
  precision     N=1: the encoder's three operand modes on the same crops — max-abs error of the embedding relative to the library's exact
                fp32 mode (itself 1.4e-6 from the CPU oracle, tests/test_gpu_encoder.py) and the step rate of each 16-bit mode.  north_star
                asks 1e-3: fp16 meets it at the bf16 rate; bf16 (the BASELINE dtype, the headline) does not (6e-3) and is reported as such.
  c3_shard_proxy  N=1: the per-rank workloads of BASELINE configs[2] — 128 / 256 / 512 crops per call (N = 8 / 4 / 2 ranks), same step.
  c4            N=1: BASELINE configs[3] — ViT-B/16 + 1M x 768 index: crops/s, per-linear TFLOP/s, k-NN time.
  knn_roofline  N=1: the k-NN where HBM is the roof (1 / 16 queries against 1M x 384 fp32): achieved GB/s vs 8 TB/s (north_star's second target).
  roofline_fp16 N=1: the dominant kernel's record in the engines' default operand type (fp16: meets the 1e-3 tolerance).
  crops_16bit   N=1: the same step with the crops handed over in the encoder's operand type (SURVEY f-2; bit-identical embeddings).
  c5            N=1: BASELINE configs[4] on one GPU — 4096 x 256 text-line images through the YOLOv5s localizer (letterbox,
                fp32-MFMA convolutions, NMS on the device), the boxes cropped on the device, ViT-S/16 + k-NN: lines/s, stage times.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK = {"bf16": 2.5e15, "fp16": 2.5e15, "fp32": 157.3e12}      # dense, MI355X_MICROARCH.md
FLOP_PER_CROP = {"vit_small_patch16_224": 9.197e9, "vit_base_patch16_224": 35.13e9}
# last block, tokens 1..196: attn.proj (2 D^2) + MLP (4 D H) per token — dead work for a class-token embedding (see "flop_per_crop")
PRUNED_FLOP_PER_CROP = {"vit_small_patch16_224": 196 * (2.0 * 384 * 384 + 4.0 * 384 * 1536),
                        "vit_base_patch16_224": 196 * (2.0 * 768 * 768 + 4.0 * 768 * 3072)}


def measured_traffic(kernel_class):
    """(HBM bytes per launch of ``kernel_class``, source file) from the newest committed rocprofv3 PMC summary
    (profiles/rNN_traffic.json: (2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes) or (None, None).  NOT measured in this run: PMC
    counters need their own rocprofv3 passes (tools/prof.sh); the line names the file so that nobody reads the number as live."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")) if "_c4_" not in os.path.basename(f))
    if not files:
        return None, None
    try:
        k = json.load(open(files[-1]))["kernels"].get(kernel_class)
        return (None, None) if k is None else (float(k["hbm_bytes_per_launch"]), "profiles/" + os.path.basename(files[-1]))
    except Exception:
        return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arch", default="vit_small_patch16_224")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--batch", type=int, default=1024, help="crops per GPU per step")
    ap.add_argument("--index-rows", type=int, default=10000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--chunk", type=int, default=0, help="encoder sub-batch (crops); 0 = library default")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=INT", help="library option (set_option) for A/B runs, e.g. tail_split=0")
    ap.add_argument("--panel-rows", type=int, default=0, help="row-panel height 64|128 (0 = library default)")
    ap.add_argument("--no-panel", action="store_true", help="A/B: K-streaming GEMM + standalone LayerNorm path")
    ap.add_argument("--scaling", default="auto", choices=["auto", "strong", "weak"],
                    help="N>1: strong = --batch crops in total, sharded (BASELINE configs[2]); weak = --batch crops per rank; auto = strong for N>1")
    ap.add_argument("--no-extras", action="store_true", help="skip the small_batch and c4 objects (N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--breakdown", action="store_true", help="also print a per-kernel-class table to stderr")
    ap.add_argument("--verify", action="store_true", help="N>1: check the all-gathered ids against rank 0 encoding every slice itself (exit 1 on mismatch)")
    return ap.parse_args()


def cpu_baseline(arch, sd, index_cpu, k, target_s):
    """PyTorch-CPU fp32 restatement of the reference path on a bounded sample; returns dict."""
    from oracle.encoders_ref import encoder_forward, l2_normalize
    probe = []
    for mod in ("timm", "faiss"):                        # use the real libraries when the box has them
        try:
            __import__(mod)
            probe.append(f"{mod} importable (not used: the oracle port is the committed baseline)")
        except Exception as e:
            probe.append(f"{mod} absent ({type(e).__name__})")
    ncpu = os.cpu_count() or 1
    g = torch.Generator().manual_seed(1234)

    def run(n):
        x = torch.randn(n, 3, 224, 224, generator=g)
        t0 = time.perf_counter()
        emb = l2_normalize(encoder_forward(arch, sd, x))
        s = emb @ index_cpu.T
        torch.topk(s, k, dim=1)
        return time.perf_counter() - t0

    # torch's intra-op pool with one thread per logical CPU of a 256-thread host is pathologically
    # slow (0.9 crops/s measured); probe a few pool sizes on 16 crops and keep the fastest
    best_t, best_n = None, None
    for nt in sorted({min(ncpu, c) for c in (16, 32, 64, 128, ncpu)}):
        torch.set_num_threads(nt)
        run(4)
        t = run(16)
        if best_t is None or t < best_t:
            best_t, best_n = t, nt
        if t > 3 * best_t:
            break
    torch.set_num_threads(best_n)
    rate = 16 / best_t
    n = int(max(32, min(1024, rate * target_s)))
    n = (n // 32) * 32
    t = run(n)
    c1 = cpu_baseline_c1(best_n)
    return {"value": round(n / t, 2), "unit": "glyph-crops/s", "cores": best_n, "host_logical_cpus": ncpu, "kind": "port", "c1": c1,
            "sample": f"{n} of the 1024 crops of one step (same shapes, fp32, torch {torch.__version__} CPU with "
                      f"{best_n} intra-op threads = fastest of the probed pool sizes, oracle/encoders_ref.py + normalize "
                      f"+ Q@X^T top-{k} over the full {index_cpu.shape[0]}-row index), {t:.1f} s; the container's CPU quota, not the host's "
                      f"{ncpu} logical CPUs, bounds this baseline; " + "; ".join(probe)}


def cpu_baseline_c1(nthreads):
    """BASELINE configs[0] IS the reference's CPU configuration: timm resnet18 on 64 crops of 32x32, 96-glyph IndexFlatIP, k=10
    ("C1 in full").  The same port (oracle/encoders_ref.py resnet18 + normalize + Q@X^T top-k), whole calls of 64 crops,
    a few hundred of them; thread counts 1 and ``nthreads`` (64 tiny crops do not feed a large pool), the faster one reported."""
    from oracle.encoders_ref import encoder_forward, l2_normalize
    from effocr_amd.weights import init_state_dict
    sd = init_state_dict("resnet18", seed=0, img_size=32)
    g = torch.Generator().manual_seed(13)
    index = torch.nn.functional.normalize(torch.randn(96, 512, generator=g), dim=1)
    x = torch.randn(64, 3, 32, 32, generator=g)

    def call():
        emb = l2_normalize(encoder_forward("resnet18", sd, x))
        torch.topk(emb @ index.T, 10, dim=1)

    best = None
    for nt in sorted({1, 4, min(16, nthreads)}):
        torch.set_num_threads(nt)
        for _ in range(3):
            call()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 1.5:
            call()
            n += 1
        rate = 64 * n / (time.perf_counter() - t0)
        if best is None or rate > best[0]:
            best = (rate, nt, n)
    torch.set_num_threads(nthreads)
    return {"workload": "BASELINE configs[0]: resnet18, 64 x 3x32x32 crops per call, 96 x 512 IndexFlatIP, k=10 (PyTorch-CPU port)",
            "value": round(best[0], 1), "unit": "glyph-crops/s", "cores": best[1], "ms_per_call": round(64e3 / best[0], 3),
            "sample": f"{best[2]} calls of 64 crops"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if "WORLD_SIZE" not in os.environ and a.gpus > 1:
            # plain ``python bench.py --gpus N``: become the launcher (one rank per GPU over RCCL)
            import socket
            import subprocess
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            raise SystemExit(subprocess.call(cmd))
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    # EFFOCR_BENCH_BACKEND=gloo: debugging aid only — lets the N>1 control flow run on a box with fewer GPUs than
    # ranks (ranks share devices; RCCL refuses duplicate devices).  The driver's runs use the default, "nccl" = RCCL.
    backend = os.environ.get("EFFOCR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from effocr_amd.encoders import HipEncoder
    from effocr_amd.knn import FaissKNN, IndexFlatIP
    from effocr_amd.weights import init_state_dict
    from effocr_amd.dist import all_gather_rows

    # ---- synthetic workload (seeded; identical weights/index on every rank, per-rank crops)
    sd = init_state_dict(a.arch, seed=0, img_size=224)
    enc = HipEncoder(a.arch, sd, img_size=224, precision=a.precision, device=dev)
    D = enc.embed_dim
    if a.chunk:
        enc.set_chunk(a.chunk)
    if os.environ.get("EFFOCR_DEBUG"):
        enc.set_option("debug", int(os.environ["EFFOCR_DEBUG"]))
    if a.panel_rows:
        enc.set_option("panel_rows", a.panel_rows)
    for kv in a.opt:
        name, _, val = kv.partition("=")
        enc.set_option(name, int(val))
    if os.environ.get("EFFOCR_NO_BLOCKED"):
        enc.set_option("use_blocked", 0)
    if os.environ.get("EFFOCR_NO_GEMM2"):
        enc.set_option("use_gemm2", 0)
    if a.no_panel:
        enc.set_option("use_panel", 0)
    gi = torch.Generator().manual_seed(0)
    index_cpu = torch.nn.functional.normalize(torch.randn(a.index_rows, D, generator=gi), dim=1)
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False, device=dev)
    knn.train(index_cpu)
    from effocr_amd.dist import shard_bounds
    scaling = a.scaling if a.scaling != "auto" else ("strong" if world > 1 else "weak")
    # strong scaling (BASELINE configs[2]): THE SAME 1024 crops on every rank (rank-independent seed), rank r encodes rows
    # shard_bounds(1024, r, N) of them; weak scaling: 1024 crops of its own per rank (seed + rank)
    gx = torch.Generator(device=dev).manual_seed(1000)
    x_same = torch.randn(a.batch, 3, 224, 224, generator=gx, device=dev)     # resident in HBM
    lo, hi = shard_bounds(a.batch, rank, world)
    x_shard = x_same[lo:hi]                                                  # strong: this rank's slice of the 1024 crops
    if world > 1:
        gw = torch.Generator(device=dev).manual_seed(1000 + rank)
        x_full = torch.randn(a.batch, 3, 224, 224, generator=gw, device=dev)
    else:
        x_full = x_same

    def make_step(x, n_total):
        def step():
            emb = enc.forward(x, normalize=True)
            d, i = knn(emb, k=a.k)
            if world > 1:
                i = all_gather_rows(i, n_total)
            return i
        return step

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    from effocr_amd import _lib as L_
    clk = torch.zeros(8192, dtype=torch.int64, device=dev)    # two samples x 2048 CU keys x (shader ticks, 100 MHz ticks)
    clock_ghz = [None]

    def timed(step, steps):
        fence()
        t0 = time.perf_counter()
        L_.check(L_.lib().effocr_clock_sample(L_.ptr(clk), L_.current_stream(dev)), "clock_sample")          # (inside the fences: two 1024-wave launches)
        for _ in range(steps):
            out = step()
        L_.check(L_.lib().effocr_clock_sample(L_.ptr(clk[4096:]), L_.current_stream(dev)), "clock_sample")
        fence()
        dt = time.perf_counter() - t0
        c = clk.cpu().view(2, 2048, 2)                     # s_memtime is a per-CU counter: CU by CU
        ok = (c[0, :, 1] > 0) & (c[1, :, 1] > c[0, :, 1])
        if bool(ok.any()):
            d = c[1, ok] - c[0, ok]
            keep = d[:, 1] <= d[:, 1].min() * 5 // 4 + 1000   # (a CU one sample missed keeps an older pair)
            st, rt = int(d[keep, 0].sum()), int(d[keep, 1].sum())
        else:
            st = rt = 0
        if rt > 0:
            clock_ghz[0] = round(st / rt * 0.1, 3)
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item()), out

    if scaling == "strong":
        step, n_total = make_step(x_shard, a.batch), a.batch
    else:
        step, n_total = make_step(x_full, a.batch * world), a.batch * world
    for _ in range(a.warmup):
        step()
    # pick the dominant kernel class with one fully-profiled step (untimed)
    enc.profile_begin()
    step()
    table = enc.profile_collect()
    dom = max(table, key=lambda n: table[n]["ms"]) if table else None
    if a.breakdown and rank == 0:
        tot = sum(v["ms"] for v in table.values())
        for n, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
            tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0
            print(f"  {n:18s} {v['ms']:8.3f} ms {100 * v['ms'] / tot:5.1f}%  x{v['launches']:3d}  {tf:8.1f} TFLOP/s  {v.get('shader_ghz', 0.0):5.2f} GHz", file=sys.stderr)
        print(f"  {'sum of kernels':18s} {tot:8.3f} ms", file=sys.stderr)

    # ---- timed region: exactly K steps between barrier+synchronize fences; only the dominant class
    # carries event pairs (2 events per launch of that class), everything else runs untouched
    if dom:
        enc.profile_begin(only=dom)
    dt, out = timed(step, a.steps)
    clock_main = clock_ghz[0]
    prof = enc.profile_collect() if dom else {}
    assert tuple(out.shape) == (n_total, a.k)
    other = None
    if world > 1:                                          # the other scaling mode, same run, reported as an extra key
        if scaling == "strong":
            ostep, on = make_step(x_full, a.batch * world), a.batch * world
        else:
            ostep, on = make_step(x_shard, a.batch), a.batch
        for _ in range(max(1, a.warmup)):
            ostep()
        odt, _ = timed(ostep, a.steps)
        other = {"scaling": "weak" if scaling == "strong" else "strong", "value": round(on * a.steps / odt, 1), "unit": "glyph-crops/s",
                 "ms_per_step": round(1e3 * odt / a.steps, 3), "global_batch": on}

    verify = None
    if a.verify and world > 1:
        # the gathered ids of the STRONG step against rank 0 alone: rank 0 encodes every rank's slice by itself, at that slice's own call
        # size (kernel selection follows the call size, so the per-slice results are reproducible bit for bit), in rank order
        sstep = step if scaling == "strong" else make_step(x_shard, a.batch)
        got = sstep()
        torch.cuda.synchronize(dev)
        if rank == 0:
            parts = []
            for r in range(world):
                l, h = shard_bounds(a.batch, r, world)
                parts.append(knn(enc.forward(x_same[l:h], normalize=True), k=a.k)[1])
            want = torch.cat(parts)
            whole = knn(enc.forward(x_same, normalize=True), k=a.k)[1]
            verify = {"gathered_ids_equal_rank0_per_slice": bool(torch.equal(got, want)), "rows": int(got.shape[0]),
                      "top1_agreement_with_one_1024_crop_call": round(float((got[:, 0] == whole[:, 0]).float().mean().item()), 4)}
        flag = torch.tensor([1 if (verify is None or verify["gathered_ids_equal_rank0_per_slice"]) else 0], device=dev)
        dist.broadcast(flag, src=0)
        verify_ok = bool(flag.item())
    else:
        verify_ok = True

    if rank == 0:
        value = n_total * a.steps / dt
        per_rank = f"{hi - lo} of {a.batch} crops per GPU (rows shard_bounds(B, rank, N))" if scaling == "strong" else f"{a.batch} crops per GPU"
        line = {
            "metric": "glyph-crops/sec end-to-end (encode+kNN), 224x224 bs=1024",
            "value": round(value, 1), "unit": "glyph-crops/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{2 if (world > 1 and scaling == 'strong') else 1}]: {a.arch} encoder ({a.precision} MFMA operands, fp32 accumulate/"
                                   f"residual/LN/softmax), {a.batch}x3x224x224 fp32 crops {'in total' if scaling == 'strong' else 'per GPU'} resident in HBM, "
                                   f"{a.index_rows}-row fp32 IndexFlatIP, k={a.k}; seeded random-init weights",
                       "crops_per_gpu": (hi - lo) if scaling == "strong" else a.batch, "global_batch": n_total, "index_rows": a.index_rows, "k": a.k,
                       "parallelism": f"{per_rank}, weights+index replicated"
                                      + (", all_gather(ids) over RCCL" if world > 1 else "")},
        }
        if other:
            line[other["scaling"]] = other
        if verify is not None:
            line["verify"] = verify
        line["shader_clock_GHz_in_timed_region"] = clock_main     # s_memtime / s_memrealtime over the K timed steps (power-managed: 2.4 GHz max)
        if dom and dom in prof and prof[dom]["launches"]:
            p = prof[dom]
            sec = p["ms"] * 1e-3 / p["launches"]
            fl = p["flops"] / p["launches"]
            peak = MFMA_PEAK[a.precision]
            line["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(fl / sec / 1e12, 2),
                                "peak": peak / 1e12, "unit": "TFLOP/s", "frac": round(fl / sec / peak, 4),
                                "flops_per_launch": fl, "avg_launch_us": round(sec * 1e6, 2), "launches": p["launches"], "traffic": None}
            # the chip is power-limited: the shader clock this kernel class ran at in the fully profiled (untimed) step — two clock
            # samples around every launch — and the fraction of the MFMA rate AT THAT CLOCK (the 2.5 PFLOP/s peak is quoted at 2.4 GHz)
            ghz = table.get(dom, {}).get("shader_ghz", 0.0)
            if ghz > 0:
                line["roofline"]["shader_clock_GHz"] = round(ghz, 3)
                line["roofline"]["frac_at_shader_clock"] = round(fl / sec / (peak * ghz / 2.4), 4)
            if a.arch == "vit_small_patch16_224" and a.batch == 1024 and world == 1:
                tr, src = measured_traffic(dom)
                line["roofline"]["traffic"] = tr
                line["roofline"]["traffic_source"] = (src + " (rocprofv3 PMC passes of the same command, committed; not re-measured in this run)") if src else None
        if a.arch in FLOP_PER_CROP:
            # FLOPs the GPU actually executes per crop: the library runs the last block's attn.proj + MLP only on the class-token row of
            # every image (the only row that reaches the embedding; same result as the reference, which computes and discards the other
            # 196 rows).  The fraction of peak below prices EXECUTED work; the model's nominal FLOPs are reported beside it.
            pruned = PRUNED_FLOP_PER_CROP.get(a.arch, 0.0) if ("proj_mlp_cls" in table or "cls_fc1_gelu" in table) else 0.0
            if pruned and "qkv_attn_fused" in table:        # ... and its q projection + attention rows (per-image kernel, class-token variant)
                pruned += (197 - 32) * (2.0 * 384 * 384 + 4.0 * 197 * 384)      # (it computes one 32-token tile per image)
            line["encoder_mfma_frac_end_to_end"] = round(value / world * (FLOP_PER_CROP[a.arch] - pruned) / MFMA_PEAK[a.precision], 4)
            line["flop_per_crop"] = {"model": FLOP_PER_CROP[a.arch], "executed": FLOP_PER_CROP[a.arch] - pruned,
                                     "encoder_mfma_frac_at_model_flops": round(value / world * FLOP_PER_CROP[a.arch] / MFMA_PEAK[a.precision], 4)}
        if world == 1 and not a.no_extras:
            try:
                line["precision"] = precision_extras(a, enc, knn, sd, dev, x_full, dom)
                # the tolerance-meeting mode's roofline record at the top level (the driver keeps top-level roofline objects)
                if "roofline" in line["precision"].get("fp16", {}):
                    line["roofline_fp16"] = dict(line["precision"]["fp16"]["roofline"], dtype="fp16", note="the engines' default operand type (meets north_star's 1e-3); same kernels, f16 MFMA operands")
                line["crops_16bit"] = crops16_extras(a, enc, knn, dev, x_full)
                line["c3_shard_proxy"] = shard_proxy_extras(a, enc, knn, dev)
                line["small_batch"] = small_batch_extras(a, enc, knn, sd, dev)
                # north_star's second target (">= 60 % of HBM peak on the k-NN kernel") where HBM is the roof: B = 1 / 16 against 1M x D fp32
                line["knn_roofline"] = line["small_batch"]["knn_roofline"]
                if a.arch == "vit_small_patch16_224":
                    del enc, x_full, x_shard, x_same
                    torch.cuda.empty_cache()
                    line["c1"] = c1_extras(a, dev)
                    line["c4"] = c4_extras(a, dev)
                    torch.cuda.empty_cache()
                    line["c5"] = c5_extras(a, dev)
            except Exception as e:                          # extras never take the headline down
                line["extras_error"] = f"{type(e).__name__}: {e}"
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.arch, sd, index_cpu, a.k, a.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not verify_ok:
        raise SystemExit("bench.py --verify: the gathered ids differ from rank 0's own per-slice results")


def _time_gpu(fn, dev, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / iters


def precision_extras(a, enc, knn, sd, dev, x, dom=None):
    """Embedding error of every operand mode against the library's exact-fp32 mode on 64 of the step's crops, and the step rate of the
    16-bit modes on the whole batch (same kernels, other MFMA operand type)."""
    from effocr_amd.encoders import HipEncoder
    from effocr_amd.encoders import DEFAULT_PRECISION
    out = {"tolerance_north_star": 1e-3, "engines_default_precision": DEFAULT_PRECISION,
           "reference": "the library's fp32 mode (v_mfma_f32_32x32x2_f32, exact fp32; 1.4e-6 from oracle A in tests/test_gpu_encoder.py), 64 crops; "
                        "rel_err = max|e - e_ref| / max|e_ref| over L2-normalised embeddings (max-norm); rel_l2_worst_row = max over crops of |e - e_ref|_2 / |e_ref|_2; "
                        "elementwise_rel_worst = max |e - e_ref| / |e_ref| over the elements that are at least 1 % of their row's largest"}
    xs = x[:64].contiguous()
    ref = HipEncoder(a.arch, sd, img_size=224, precision="fp32", device=dev).forward(xs, normalize=True)
    top1_ref = knn(ref, k=1)[1]
    for prec in ("bf16", "fp16"):
        e = enc if prec == a.precision else HipEncoder(a.arch, sd, img_size=224, precision=prec, device=dev)
        emb = e.forward(xs, normalize=True)
        rel = ((emb - ref).abs().max() / ref.abs().max()).item()
        row_l2 = ((emb - ref).norm(dim=1) / ref.norm(dim=1)).max().item()                       # worst row, relative L2
        big = ref.abs() >= 0.01 * ref.abs().amax(dim=1, keepdim=True)                            # element-wise, where an element is >= 1 % of its row's largest
        elem = (((emb - ref).abs() / ref.abs().clamp_min(1e-30))[big]).max().item()
        t = _time_gpu(lambda: knn(e.forward(x, normalize=True), k=a.k), dev, 10, warm=3)
        out[prec] = {"rel_err": float(f"{rel:.3e}"), "rel_l2_worst_row": float(f"{row_l2:.3e}"), "elementwise_rel_worst": float(f"{elem:.3e}"), "meets_tolerance": bool(rel <= 1e-3), "crops_per_s": round(x.shape[0] / t, 1),
                     "ms_per_step": round(1e3 * t, 3), "top1_identical_to_fp32_mode": bool(torch.equal(knn(emb, k=1)[1], top1_ref))}
        if dom:                                            # the dominant kernel's roofline record in THIS mode (in-stream events, 5 steps)
            e.profile_begin(only=dom)
            for _ in range(5):
                e.forward(x, normalize=True)
            p = e.profile_collect().get(dom)
            if p and p["launches"]:
                sec, fl = p["ms"] * 1e-3 / p["launches"], p["flops"] / p["launches"]
                out[prec]["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(fl / sec / 1e12, 2), "peak": MFMA_PEAK[prec] / 1e12,
                                         "unit": "TFLOP/s", "frac": round(fl / sec / MFMA_PEAK[prec], 4), "avg_launch_us": round(sec * 1e6, 2),
                                         "launches": p["launches"]}
        if e is not enc:
            del e
    torch.cuda.empty_cache()
    return out


def crops16_extras(a, enc, knn, dev, x):
    """SURVEY f-2's hand-off: the same step with the crops ALREADY in the encoder's operand type (what effocr_crop_transform_batch_ex
    writes for run_effocr) — bit-identical embeddings (asserted), half the input bytes.  Not the headline: the reference's engine
    interface takes float32 crops (onnx_engines/recognizer_engine.py:23-27)."""
    if enc.crop_dtype == torch.float32:
        return None
    x16 = x.to(enc.crop_dtype)
    same = bool(torch.equal(enc.forward(x16, normalize=True), enc.forward(x, normalize=True)))
    t = _time_gpu(lambda: knn(enc.forward(x16, normalize=True), k=a.k), dev, 10, warm=3)
    enc.profile_begin(only="patch_embed_fused")
    for _ in range(5):
        enc.forward(x16, normalize=True)
    p16 = enc.profile_collect().get("patch_embed_fused")
    enc.profile_begin(only="patch_embed_fused")
    for _ in range(5):
        enc.forward(x, normalize=True)
    p32 = enc.profile_collect().get("patch_embed_fused")
    out = {"crop_dtype": str(enc.crop_dtype).replace("torch.", ""), "embeddings_bit_identical_to_fp32_crops": same,
           "crops_per_s": round(x.shape[0] / t, 1), "ms_per_step": round(1e3 * t, 3)}
    if p16 and p32 and p16["launches"] and p32["launches"]:
        B = x.shape[0]
        by16, by32 = B * 3 * 224 * 224 * 2.0 + B * 196 * enc.embed_dim * 4.0, B * 3 * 224 * 224 * 4.0 + B * 196 * enc.embed_dim * 4.0
        u16, u32_ = 1e3 * p16["ms"] / p16["launches"], 1e3 * p32["ms"] / p32["launches"]
        out["patch_embed_fused_us"] = {"16bit_crops": round(u16, 1), "fp32_crops": round(u32_, 1),
                                       "hbm_frac_16bit_crops": round(by16 / (u16 * 1e-6) / 8.0e12, 4), "hbm_frac_fp32_crops": round(by32 / (u32_ * 1e-6) / 8.0e12, 4)}
    return out


def shard_proxy_extras(a, enc, knn, dev):
    """BASELINE configs[2] shards the 1024 crops over N ranks: what ONE rank then runs per step (encode + normalise + k-NN on its slice),
    timed on this GPU.  value(N) / (N * rate) bounds the strong-scaling efficiency from above (the all_gather of ids adds microseconds)."""
    out = {}
    full = None
    for n_ranks, B in ((1, 1024), (2, 512), (4, 256), (8, 128)):
        if B > a.batch:
            continue
        x = torch.randn(B, 3, 224, 224, device=dev)
        t = _time_gpu(lambda: knn(enc.forward(x, normalize=True), k=a.k), dev, max(30, 8192 // B), warm=5)   # (>= 0.1 s per size: ten-call samples moved by 6 % between runs on a power-managed clock)
        full = full or B / t
        out[f"ranks{n_ranks}_crops{B}"] = {"ms_per_step": round(1e3 * t, 3), "crops_per_s_per_gpu": round(B / t, 1),
                                           "fraction_of_the_1024_crop_rate": round(B / t / full, 4)}
    return out


def small_batch_extras(a, enc, knn, sd, dev):
    """The reference drivers' own call sizes (infer_effocr.py:313: one text line, tens of crops; infer_effocr_onnx_multi.py:157:
    literal 64) and the k-NN in the regime where HBM is the roof (B <= 16 against a large index)."""
    import threading
    import numpy as np
    from effocr_amd.knn import IndexFlatIP
    from effocr_amd.recognizer_engine import EffRecognizer
    out = {}
    x64 = torch.randn(64, 3, 224, 224, device=dev)
    t = _time_gpu(lambda: knn(enc.forward(x64, normalize=True), k=a.k), dev, 30)
    out["b64_device_resident"] = {"crops_per_s": round(64 / t, 1), "ms_per_call": round(1e3 * t, 3),
                                  "note": "calls of 37..83 crops run the fused proj+MLP on 64-token wave-pair panels (no partial sums in HBM, no reduction launch; option mlp_pair); <= 36 crops: the pair panels' hidden chunks over 6 / 3 / 2 workgroups + the reduction launch (option pair_parts)"}
    # the torch driver's call: ONE text line = B characters (infer_effocr.py:313-319, k = 10): per-call latency at 1 / 8 / 16 / 32 crops, the
    # encoder's launch count (in-library profiler) and the floor of the launch chain = the 1-crop call (every kernel at its minimum duration)
    per_line = {}
    for B in (1, 8, 16, 32):
        xb_ = torch.randn(B, 3, 224, 224, device=dev)
        tb = _time_gpu(lambda: knn(enc.forward(xb_, normalize=True), k=a.k), dev, 100, warm=5)
        enc.profile_begin()
        enc.forward(xb_, normalize=True)
        tab = enc.profile_collect()
        per_line[f"b{B}"] = {"ms_per_call": round(1e3 * tb, 3), "crops_per_s": round(B / tb, 1), "encoder_launches": int(sum(v["launches"] for v in tab.values())),
                             "encoder_kernel_ms": round(sum(v["ms"] for v in tab.values()), 3)}
    per_line["floor"] = ("b1 = the launch chain with every kernel at its minimum duration (one image: 2 panels, 6 head workgroups); a B-crop call "
                         "cannot be faster than it without fewer or shorter launches")
    out["per_line_calls"] = per_line
    # the reference's ONNX driver runs its 64-crop batches from N threads sharing ONE engine (infer_effocr_onnx_multi.py:207-223,350-364):
    # the same device-resident call from 2 / 4 Python threads, each on its own HIP stream (per-stream workspaces; a 64-crop kernel fills
    # at most 3/4 of the CUs, two streams' kernels overlap on the device)
    for nthr in (2, 4):
        calls = 40
        streams = [torch.cuda.Stream(device=dev) for _ in range(nthr)]
        xs = [torch.randn(64, 3, 224, 224, device=dev) for _ in range(nthr)]

        def run(i, n):
            with torch.cuda.stream(streams[i]):
                for _ in range(n):
                    knn(enc.forward(xs[i], normalize=True), k=a.k)
                streams[i].synchronize()
        for i in range(nthr):
            run(i, 2)
        torch.cuda.synchronize(dev)
        ths = [threading.Thread(target=run, args=(i, calls)) for i in range(nthr)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        torch.cuda.synchronize(dev)
        tt = time.perf_counter() - t0
        out[f"b64_device_resident_{nthr}_streams"] = {"crops_per_s": round(64 * calls * nthr / tt, 1), "ms_per_call_per_stream": round(1e3 * tt / calls, 3)}
    eng = EffRecognizer(sd, arch=a.arch, precision=a.precision, device=dev)
    rng_ = np.random.default_rng(0)
    batches = [rng_.standard_normal((64, 3, 224, 224), dtype=np.float32) for _ in range(4)]     # distinct arrays, like create_batches' list
    n_calls = 96

    def worker(i, n):
        for j in range(n):
            eng.run(batches[(i + j) % 4])

    def threads(nthr, n):
        ths = [threading.Thread(target=worker, args=(i, n)) for i in range(nthr)]
        t0_ = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        return time.perf_counter() - t0_
    threads(4, 3)                                          # every lane (stream, workspace) used once before anything is timed
    t1 = threads(1, n_calls)
    t4 = threads(4, n_calls // 4)
    out["b64_effrecognizer_run_numpy"] = {"crops_per_s_1_thread": round(64 * n_calls / t1, 1), "crops_per_s_4_threads": round(64 * n_calls / t4, 1),
                                          "calls": n_calls,
                                          "note": "pageable numpy in / numpy out per call: 38.5 MB by the runtime's pageable host->device copy + D2H of the embeddings, PCIe-inclusive; never the headline value"}
    del eng
    # k-NN alone, HBM-bound regime: 1M x D fp32 index (1.5 GB at D=384), exact kernel and the screened path
    D = enc.embed_dim
    N = 1_000_000
    idx = IndexFlatIP(D, device=dev, screen=False)
    g = torch.Generator(device=dev).manual_seed(7)
    xb = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
    idx.add(xb)
    idx_s = IndexFlatIP(D, device=dev, screen=True)
    idx_s._xb = idx._xb
    rows = {}
    for B in (1, 16, 64, 1024):
        q = torch.nn.functional.normalize(xb[:B] + 0.1 * torch.randn(B, D, generator=g, device=dev), dim=1)
        te = _time_gpu(lambda: idx.search_device(q, a.k), dev, 10)
        ts = _time_gpu(lambda: idx_s.search_device(q, a.k), dev, 10)
        by = N * D * 4.0
        rows[f"B{B}"] = {"exact_ms": round(1e3 * te, 3), "exact_hbm_frac": round(by / te / 8.0e12, 4),
                         "exact_mfma_fp32_frac": round(2.0 * B * N * D / te / 157.3e12, 4),
                         "screened_ms": round(1e3 * ts, 3), "screened_hbm_frac_bf16_one_scan": round(N * D * 2.0 / ts / 8.0e12, 4)}
    out["knn_1M_rows"] = {"index": f"{N} x {D} fp32 ({N * D * 4 / 1e9:.2f} GB)", "k": a.k, "hbm_peak_GBps": 8000, **rows}
    out["knn_roofline"] = knn_stream_roofline(idx, a.k, dev)
    return out


def knn_stream_roofline(idx, k, dev, batches=(1, 16)):
    """Roofline-shaped records of the HBM-bound k-NN regime (B <= 16 against a large index; north_star's ">= 60 % of HBM peak on
    the k-NN kernel"): the exact streaming search (`knn_stream_kernel`, 16-query tile + its merge) over an index resident in HBM, timed with
    events ON THE LAUNCH STREAM (torch's current stream is the stream the library launches on) around each of `iters` searches.
    achieved = algorithmic bytes (index once: N * D * 4; + queries + results) / mean search time; peak 8 TB/s."""
    was = idx.screen
    idx.screen = False
    N, D = idx.ntotal, idx.d
    g = torch.Generator(device=dev).manual_seed(5)
    out = []
    for B in batches:
        q = torch.nn.functional.normalize(torch.randn(B, D, generator=g, device=dev), dim=1)
        for _ in range(3):
            idx.search_device(q, k)
        iters = 20
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for e0, e1 in evs:
            e0.record()
            idx.search_device(q, k)
            e1.record()
        torch.cuda.synchronize(dev)
        us = sum(e0.elapsed_time(e1) for e0, e1 in evs) / iters * 1e3
        by = N * D * 4.0 + B * D * 4.0 + B * k * 12.0
        out.append({"bound": "hbm", "kernel": "knn_stream_kernel<16-query tile> + merge (exact fp32 search)", "queries": B, "index": f"{N} x {D} fp32",
                    "bytes_per_launch": by, "avg_launch_us": round(us, 1), "launches": iters, "achieved": round(by / us / 1e3, 1), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(by / (us * 1e-6) / 8.0e12, 4)})
    idx.screen = was
    return out


RESNET18_FLOP_32 = 74.0e6    # 2 x MACs of timm resnet18 (no fc) at a 32x32 input: conv1 4.8 + layer1 18.9 + layers 2-4 3 x 16.8 MFLOP


def c1_extras(a, dev):
    """BASELINE configs[0] on the GPU: timm resnet18 (fp32 MFMA implicit-GEMM conv) over 32x32 crops + a 96-glyph index, at the
    reference's own call size (64 crops) and at 1024."""
    from effocr_amd.encoders import HipEncoder
    from effocr_amd.knn import FaissKNN, IndexFlatIP
    from effocr_amd.weights import init_state_dict
    arch = "resnet18"
    enc = HipEncoder(arch, init_state_dict(arch, seed=0, img_size=32), img_size=32, precision="fp32", device=dev)
    g = torch.Generator(device=dev).manual_seed(13)
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False, device=dev)
    knn.index = IndexFlatIP(512, device=dev)
    knn.index.add(torch.nn.functional.normalize(torch.randn(96, 512, generator=g, device=dev), dim=1))
    out = {"workload": "BASELINE configs[0] shapes on one GPU: resnet18 (fp32 operands, v_mfma_f32_32x32x2), 3x32x32 crops resident in HBM, "
                       "96 x 512 fp32 IndexFlatIP, k=10"}
    for B in (8, 16, 32, 64, 1024):
        x = torch.randn(B, 3, 32, 32, generator=g, device=dev)
        step = lambda: knn(enc.forward(x, normalize=True), k=10)
        t = _time_gpu(step, dev, 20 if B >= 64 else 100, warm=3)
        out[f"B{B}"] = {"crops_per_s": round(B / t, 1), "ms_per_call": round(1e3 * t, 3),
                        "encoder_mfma_fp32_frac": round(B * RESNET18_FLOP_32 / t / MFMA_PEAK["fp32"], 4)}
    return out


def c4_extras(a, dev):
    """BASELINE configs[3]: ViT-B/16 encoder + 1M x 768 glyph index on one GPU."""
    from effocr_amd.encoders import HipEncoder
    from effocr_amd.knn import FaissKNN, IndexFlatIP
    from effocr_amd.weights import init_state_dict
    arch = "vit_base_patch16_224"
    enc = HipEncoder(arch, init_state_dict(arch, seed=0, img_size=224), img_size=224, precision=a.precision, device=dev)
    N, D = 1_000_000, 768
    g = torch.Generator(device=dev).manual_seed(11)
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False, device=dev)
    knn.index = IndexFlatIP(D, device=dev)
    for _ in range(4):                                     # 4 x 250k rows: bounded transient memory
        knn.index.add(torch.nn.functional.normalize(torch.randn(N // 4, D, generator=g, device=dev), dim=1))
    x = torch.randn(1024, 3, 224, 224, generator=g, device=dev)
    step = lambda: knn(enc.forward(x, normalize=True), k=a.k)
    for _ in range(2):
        step()
    enc.profile_begin()
    emb = enc.forward(x, normalize=True)
    table = enc.profile_collect()
    t = _time_gpu(step, dev, 5, warm=0)
    tk = _time_gpu(lambda: knn(emb, k=a.k), dev, 5, warm=1)
    te = t - tk
    lin = {n: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) for n, v in table.items() if v["flops"] > 0 and v["ms"] > 0}
    return {"workload": f"BASELINE configs[3]: {arch} ({a.precision}), 1024x3x224x224 crops resident in HBM, {N}x{D} fp32 IndexFlatIP (screened search), k={a.k}",
            "value": round(1024 / t, 1), "unit": "glyph-crops/s", "ms_per_step": round(1e3 * t, 3),
            "encoder_ms": round(1e3 * te, 3),
            "encoder_mfma_frac": round(1024 * (FLOP_PER_CROP[arch] - (PRUNED_FLOP_PER_CROP[arch] if "cls_fc1_gelu" in table else 0.0)) / te / MFMA_PEAK[a.precision], 4),
            "encoder_mfma_frac_at_model_flops": round(1024 * FLOP_PER_CROP[arch] / te / MFMA_PEAK[a.precision], 4),
            "kernel_TFLOPs": lin, "kernel_shader_clock_GHz": {n: round(v.get("shader_ghz", 0.0), 3) for n, v in table.items() if v["flops"] > 0 and v["ms"] > 0},
            "knn_ms": round(1e3 * tk, 3),
            # the screened search is ONE bf16 scan of the index (1.536 GB, 1.573 TFLOP: the Q-stationary kernel over the blocked copy, block
            # maxima out) + threshold collect + two-stage re-rank
            "knn_hbm_frac_bf16_one_scan": round(N * D * 2.0 / tk / 8.0e12, 4), "knn_mfma_bf16_frac": round(2.0 * 1024 * N * D / tk / 2.5e15, 4),
            "knn_roofline": knn_stream_roofline(knn.index, a.k, dev)}


YOLO_STRIDE = {0: 2, 1: 4, 2: 4, 3: 8, 4: 8, 5: 16, 6: 16, 7: 32, 8: 32, 9: 32, 10: 32, 13: 16, 14: 16, 17: 8, 18: 16, 20: 16, 21: 32, 23: 32}


def yolov5s_flops(nc, h, w):
    """2 * MACs of every convolution of the layer table at an h x w input."""
    from effocr_amd.localizer_engine import yolov5s_param_shapes
    fl = 0.0
    for k, shp in yolov5s_param_shapes(nc).items():
        if not (k.endswith("conv.weight") or (k.startswith("model.24.m.") and k.endswith(".weight"))):
            continue
        parts = k.split(".")
        st = (8, 16, 32)[int(parts[3])] if parts[1] == "24" else YOLO_STRIDE[int(parts[1])]
        n = 1
        for v in shp:
            n *= v
        fl += 2.0 * n * (h // st) * (w // st)
    return fl


def c5_extras(a, dev):
    """BASELINE configs[4] (full pipeline) on ONE GPU through the PRODUCT function effocr_amd.pipeline.run_effocr
    (infer_effocr_onnx_multi.py:227-397): 16 synthetic 4096 x 256 uint8 text-line images per call -> EffLocalizer (device letterbox to
    640 x 640, YOLOv5s, device NMS) -> character boxes parsed / scaled / double-clipped on the device -> one batched crop-transform
    launch -> ViT-S/16 + k-NN (k = 1) -> line strings (lang "jp": characters joined, no en_postprocess — the localizer is random-init,
    there are no word boxes to space by).  Seeded random localizer weights with the Detect biases raised so that every line yields boxes.
    max_det is the reference's default 1000 (localizer_engine.py:62), i.e. the product default — round 3 measured max_det=64.
    Median of 7 calls."""
    import numpy as np
    from effocr_amd.knn import FaissKNN, IndexFlatIP
    from effocr_amd.localizer_engine import EffLocalizer, init_yolov5s_state_dict
    from effocr_amd.pipeline import run_effocr
    from effocr_amd.recognizer_engine import EffRecognizer
    from effocr_amd.transforms import PairedTransform
    from effocr_amd.weights import init_state_dict
    nc = 2
    sd = init_yolov5s_state_dict(nc, seed=0)
    for l in range(3):
        b = sd[f"model.24.m.{l}.bias"].view(3, nc + 5)
        b[:, 4] += 5.5
        b[:, 5] += 2.5
    loc = EffLocalizer(sd, iou_thresh=0.05, conf_thresh=0.5, device=dev)
    arch = "vit_small_patch16_224"
    rec = EffRecognizer(init_state_dict(arch, seed=0, img_size=224), arch=arch, precision=a.precision, device=dev)
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False, device=dev)
    knn.train(torch.nn.functional.normalize(torch.randn(a.index_rows, rec._eng_net.embed_dim, generator=torch.Generator().manual_seed(0)), dim=1))
    chars = [chr(0x4E00 + i) for i in range(a.index_rows)]
    tf = PairedTransform(size=224, device=dev)
    rng = np.random.default_rng(0)
    nl = 16
    lines = [(rng.integers(0, 256, (256, 4096, 3)) // 32 * 32).astype(np.uint8) for _ in range(nl)]

    dev_lines = [torch.from_numpy(im).to(dev) for im in lines]

    def call(inp=None):
        t0 = time.perf_counter()
        res, _ = run_effocr(lines if inp is None else inp, loc, rec, tf, "jp", knn_func=knn, candidate_chars=chars)
        return time.perf_counter() - t0, res

    def loc_only():
        t0 = time.perf_counter()
        rows, counts = loc.run_device(lines)
        counts.cpu()
        return time.perf_counter() - t0

    # the reference hands run_effocr ALL line images of a job in one list (infer_effocr_onnx_multi.py:227): 64 lines per call = 4 chunks of 16,
    # the upload of chunk i+1 prefetched on a side stream under the kernels of chunk i (run_effocr, round 6)
    lines64 = lines + [(rng.integers(0, 256, (256, 4096, 3)) // 32 * 32).astype(np.uint8) for _ in range(48)]
    call(lines64)
    t64 = sorted(call(lines64)[0] for _ in range(5))[2]
    for _ in range(2):
        _, res = call()
    ts = sorted(call()[0] for _ in range(7))
    tl = sorted(loc_only() for _ in range(7))
    t, tlm = ts[len(ts) // 2], tl[len(tl) // 2]
    nb = sum(len(v) for v in res.values()) / nl
    call(dev_lines)
    td = sorted(call(dev_lines)[0] for _ in range(7))[3]           # line images already in HBM (uint8): no PCIe leg
    loc._eng_net.set_option("bf16_operands", 1)                    # + the bf16-operand localizer convolutions (EffLocalizer(precision="bf16"))
    call(dev_lines)
    td16 = sorted(call(dev_lines)[0] for _ in range(7))[3]
    loc._eng_net.set_option("bf16_operands", 0)
    # the localizer network alone, batched, device-resident input
    x = torch.rand(16, 3, 640, 640, device=dev)
    tn = _time_gpu(lambda: loc._eng_net.forward(x), dev, 5)
    loc._eng_net.set_option("bf16_operands", 1)            # the optional bf16-operand convolutions (EffLocalizer(precision="bf16"))
    tn16 = _time_gpu(lambda: loc._eng_net.forward(x), dev, 5)
    loc._eng_net.set_option("bf16_operands", 0)
    fl = yolov5s_flops(nc, 640, 640)
    return {"workload": "BASELINE configs[4] on 1 GPU, product function run_effocr: 64 x 4096x256 uint8 text-line images per call (chunks of 16 lines, next chunk's upload prefetched; the 16-line-call figure of rounds 3-5 beside it) -> YOLOv5s localizer "
                        f"(640x640 letterbox, fp32 MFMA, device NMS, max_det 1000 = the default) -> device box parsing + ONE crop-transform launch -> {arch} ({a.precision}) "
                        f"-> {a.index_rows}-row IndexFlatIP, k=1 -> strings; host uint8 images in, strings out (PCIe-inclusive); seeded random weights",
            "lines_per_s": round(64 / t64, 2), "lines_per_call": 64, "ms_per_64_line_call_median_of_5": round(1e3 * t64, 3),
            "lines_per_s_16_line_calls": round(nl / t, 2), "ms_per_call_median_of_7": round(1e3 * t, 3), "ms_per_call_min": round(1e3 * ts[0], 3),
            "chars_per_line": round(nb, 1),
            "lines_per_s_images_resident_in_hbm": round(nl / td, 2), "lines_per_s_images_resident_bf16_localizer": round(nl / td16, 2),
            "localizer_ms_per_line": round(1e3 * tlm / nl, 3), "rest_ms_per_line": round(1e3 * (t - tlm) / nl, 3),
            "localizer_network_images_per_s_batch16": round(16 / tn, 1), "localizer_network_ms_per_image": round(1e3 * tn / 16, 3),
            "localizer_GFLOP_per_image": round(fl / 1e9, 2), "localizer_mfma_fp32_frac": round(16 * fl / tn / 157.3e12, 4),
            "localizer_network_ms_per_image_bf16_operands": round(1e3 * tn16 / 16, 3)}


if __name__ == "__main__":
    main()
