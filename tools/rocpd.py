#!/usr/bin/env python
"""Summarise rocprofv3 rocpd databases (gpurun_out/prof_<tag>/) into text files under profiles/.

usage: python tools/rocpd.py <tag>        e.g. r01  -> profiles/r01_kernel_stats.txt, r01_pmc.txt, r01_traffic.json
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read stream
(MI355X_MICROARCH.md, HBM section), so reads = 2 * FETCH_SIZE * 1024 bytes.
"""
import json, os, re, sqlite3, sys

CLASSES = [  # (substring of the mangled/demangled kernel name, readable class)
    ("qkvattn_kernel", "qkv_attn_fused"), ("patch_embed_kernel", "patch_embed_fused"), ("knn_stream_kernel", "knn_stream"), ("knn_rerank", "knn_rerank"), ("knn_prep", "knn_prep"),
    ("knn_qs_kernel", "knn_qs_screen"), ("knn_pool_bound", "knn_pool_bound"), ("knn_pool_collect", "knn_pool_collect"), ("knn_pool_rerank", "knn_pool_rerank"),
    ("convert_bf16_blocked", "convert_bf16_blocked"), ("crop_transform_kernel", "crop_transform"),
    ("layernorm_blocked_kernel", "layernorm_blocked"), ("conv_igemm", "conv_igemm"),
    # mlp_fused_kernel<E, D, H, TNCW, PROJ>: whole panels + the split parts of the tail panels (TNCW chunks each) in one launch
    ("mlp_fused_kernelIDF16bLi384ELi1536ELi0ELb1", "proj_mlp_main"), ("mlp_fused_kernelIDF16bLi384ELi1536ELi3ELb1", "proj_mlp_main"),
    ("mlp_fused_kernelIDF16bLi384ELi1536ELi2ELb1", "proj_mlp_main"), ("mlp_pair_kernel", "proj_mlp_pair"),
    ("mlp_fused_kernelIDF16bLi384ELi1536ELi6ELb1", "proj_mlp_main"), ("mlp_fused_kernelIDF16bLi384ELi1536ELi0ELb0", "mlp_fused_main"),
    ("mlp_fused_kernelIDF16bLi384ELi1536ELi3ELb0", "mlp_fused_main"), ("mlp_fused_kernelIDF16bLi384ELi1536ELi6ELb0", "mlp_fused_main"),
    ("mlp_reduce_kernel", "mlp_fused_reduce"), ("gather_cls", "gather_cls"),
    ("layernorm_blocked", "layernorm_blocked"),
    ("panel_gemm_kernelIDF16bLi384ELi1ELi1", "panel_ln_fc1_gelu"), ("panel_gemm_kernelIDF16bLi384ELi1ELi0", "panel_ln_qkv"),
    ("panel_gemm_kernelIDF16bLi384ELi0ELi2", "panel_proj_resid"), ("gemm3_kernelIDF16bLi3ELi4ELi2", "gemm_fc2_resid"),
    ("gemm3_kernel<", "gemm_fc2_resid_tail"), ("gemm2_kernelIDF16bLi2", "gemm_fc2_resid"),
    ("gemm2_kernelIDF16bLi3", "gemm_patch_embed"), ("gemm_nt_kernelIDF16bLi2", "gemm_fc2_resid"),
    ("gemm_nt_kernelIDF16bLi3", "gemm_patch_embed"), ("attn_mfma_kernel", "attention"), ("knn_partial", "knn_partial"),
    ("knn_merge", "knn_merge"), ("im2col16", "im2col_patch16"), ("cls_norm", "final_cls_norm"), ("layernorm_kernel", "layernorm"),
]

def cls(name):
    for sub, c in CLASSES:
        if sub in name:
            return c
    if "gemm2_kernel<" in name or "gemm_nt_kernel<" in name:
        return "gemm_fc2_resid"       # rocprof half-demangles exactly one instantiation: the bf16 RESID one (fc2)
    return re.sub(r"_ZN6effocr12_GLOBAL__N_1\d*", "", name)[:48]

def stats(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    out.write(f"{'class':20s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}  kernel symbol\n")
    for n, cnt, s, a, mn, mx in rows:
        out.write(f"{cls(n):20s} {cnt:6d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}  {n[:110]}\n")

def pmc_by_kernel(db):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by kernel_name, counter_name, dispatch_id").fetchall()
    by = {}
    for n, cn, did, v in rows:
        by.setdefault(cls(n), {}).setdefault(cn, []).append(v)
    return {k: {cn: sum(v) / len(v) for cn, v in d.items()} for k, d in by.items()}

def main(tag):
    src = f"gpurun_out/prof_{tag}"
    if not os.path.isdir(src):
        raise SystemExit(f"{src} not found (run tools/prof.sh {tag} on the GPU box first)")
    os.makedirs("profiles", exist_ok=True)
    with open(f"profiles/{tag}_kernel_stats.txt", "w") as f:
        cmdline = open(f"{src}/cmd.txt").read().strip() if os.path.exists(f"{src}/cmd.txt") else "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
        f.write(f"# rocprofv3 --kernel-trace --stats -- {cmdline}   (1x MI355X, tag {tag})\n")
        f.write("# durations from the rocpd 'kernels' view; 5 timed + 2 warm-up + 1 profiled step => 8 forwards x 12 blocks = 96 launches per block kernel\n")
        stats(f"{src}/stats/stats_results.db", f)
    allc = {}
    for p in ("pmc1", "pmc2", "pmc3", "pmc4"):
        db = f"{src}/{p}/{p}_results.db"
        if os.path.exists(db):
            for k, d in pmc_by_kernel(db).items():
                allc.setdefault(k, {}).update(d)
    ctrs = sorted({c for d in allc.values() for c in d})
    with open(f"profiles/{tag}_pmc.txt", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --pmc <set> (one pass per set, no other trace domain), per-dispatch averages, tag {tag}\n")
        f.write("# SQ_* cycle counters are quad-cycles summed over waves/SEs; SQ_VALU_MFMA_BUSY_CYCLES = 32 x N_mfma; FETCH/WRITE_SIZE in KB\n")
        for k, d in sorted(allc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
            if not any(s in k for s in ("panel", "gemm", "attention", "knn", "mlp", "qkv", "layernorm")):
                continue
            f.write(f"\n[{k}]\n")
            for c in ctrs:
                if c in d:
                    f.write(f"  {c:28s} {d[c]:16.6g}\n")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_INSTS_MFMA" in d and d["SQ_INSTS_MFMA"]:
                f.write(f"  {'valu_per_mfma':28s} {d.get('SQ_INSTS_VALU', 0) / d['SQ_INSTS_MFMA']:16.2f}\n")
            if "SQ_WAIT_ANY" in d and d.get("SQ_WAVE_CYCLES"):
                f.write(f"  {'wait_any_frac':28s} {d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:16.3f}\n")
                f.write(f"  {'wait_inst_any_frac':28s} {d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES']:16.3f}\n")
    traffic = {}
    for k, d in allc.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            traffic[k] = {"read_bytes": 2 * d["FETCH_SIZE"] * 1024, "write_bytes": d["WRITE_SIZE"] * 1024,
                          "hbm_bytes_per_launch": 2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024}
    for main, agg in (("proj_mlp_main", "proj_mlp_fused"), ("mlp_fused_main", "mlp_fused")):
        parts = [traffic[k] for k in (main, "mlp_fused_reduce") if k in traffic]
        if main in traffic:   # bench.py times the launches of the fused MLP (main + reduction) as ONE class: same aggregate here
            traffic[agg] = {key: sum(p[key] for p in parts) for key in parts[0]}
    with open(f"profiles/{tag}_traffic.json", "w") as f:
        json.dump({"note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE half-count correction, MI355X_MICROARCH.md)",
                   "kernels": traffic}, f, indent=1)
    for fn in ("bench.json", "bench_breakdown.txt"):
        if os.path.exists(f"{src}/{fn}"):
            txt = open(f"{src}/{fn}").read()
            open(f"profiles/{tag}_{fn}", "w").write("\n".join(l for l in txt.splitlines() if "amdgpu.ids" not in l) + "\n")
    print(open(f"profiles/{tag}_kernel_stats.txt").read())

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
