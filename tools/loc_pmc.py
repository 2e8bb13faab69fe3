#!/usr/bin/env python
"""PMC summary of the localizer's kernels from a rocprofv3 --pmc pass over tools/loc_trace.py (GPU box):
   cd /tmp && rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \\
       -d $GRAFT_REPO_ROOT/gpurun_out/locpmc -o p -- python $GRAFT_REPO_ROOT/tools/loc_trace.py
   python tools/loc_pmc.py gpurun_out/locpmc/p_results.db
Per kernel symbol (averages over its dispatches): duration, MFMA instructions, the matrix-pipe time they stand for and its share of the duration,
LDS bank-conflict cycles / LDS-active cycles."""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by kernel_name, counter_name, dispatch_id").fetchall()
dur = dict(c.execute("select dispatch_id, duration from kernels").fetchall())
short = lambda n: re.sub(r"\(anonymous namespace\)::|effocr::|void ", "", n).split("(")[0][:44]
by, dd = {}, {}
for n, cn, did, v in rows:
    by.setdefault(short(n), {}).setdefault(cn, []).append(v)
    dd.setdefault(short(n), {})[did] = dur.get(did, 0)
GHZ, SIMDS = 2.1, 1024                                   # sustained shader clock of the fp32 MFMA kernels; 256 CUs x 4 SIMDs
print(f"# v_mfma_f32_32x32x2_f32 occupies its SIMD's matrix pipe for 64 cycles: pipe_us = MFMA insts x 64 / {SIMDS} SIMDs / {GHZ} GHz")
print(f"{'kernel':46s} {'disp':>5s} {'avg_us':>8s} {'MFMA insts':>11s} {'pipe_us':>8s} {'pipe/dur':>9s} {'lds_conflict/lds_active':>24s}")
for k, d in sorted(by.items(), key=lambda kv: -sum(dd[kv[0]].values())):
    g = lambda n: sum(d.get(n, [0])) / max(len(d.get(n, [1])), 1)
    us = sum(dd[k].values()) / max(len(dd[k]), 1) / 1e3
    pipe = g("SQ_INSTS_MFMA") * 64 / SIMDS / (GHZ * 1e3)
    lc, la = g("SQ_LDS_BANK_CONFLICT"), g("SQ_LDS_IDX_ACTIVE")
    print(f"{k:46s} {len(dd[k]):5d} {us:8.1f} {g('SQ_INSTS_MFMA'):11.3e} {pipe:8.1f} {pipe / us if us else 0:9.3f} {lc / la if la else 0:24.3f}")
