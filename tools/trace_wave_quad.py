#!/usr/bin/env python3
"""Phase timeline of the wave-tile and four-wave passes of ONE transform (s_memtime stamps, one row per WAVE):

    python tools/build_variant.py --tag _trace2 --units wave_f64,wave_f32,quad_f64,quad_f32 -- -DPHAST_TRACE
    PHASTFT_HIP_LIB=phastft_amd/lib/libphastft_hip_trace2.so python tools/trace_wave_quad.py [f64|f32] [log_n]

Per pass and stamp: when the stamp is reached since the FIRST wave of the chip entered the kernel (min / median / max over the
waves, in us at the s_memtime rate measured against HIP events) and the median time a wave spends between consecutive stamps.
Stamps that drain the wave's memory counters (loads back, stores retired) make the traced kernel a little slower than the
product's; where the time goes is what this is for.  wave tiles: 0 entry, 1 stagger over, 2 loads issued, 3 loads back,
4 arithmetic done, 5 transposed (first pass), 6 stores issued, 7 stores retired.  four-wave pass: 0 entry, 1 loads issued,
2 tables staged + barrier, 3 loads back, 4 pre-twiddle + steps 1-2, 5 exchange written, 6 barrier, 7 exchange read,
8 steps 3-4, 9 stores issued, 10 stores retired."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import phastft_amd as P
from phastft_amd import _lib

lib = _lib.lib()
dt = sys.argv[1] if len(sys.argv) > 1 else "f64"
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << log_n
tdt = torch.float64 if dt == "f64" else torch.float32
pl = (P.PlannerDit64 if dt == "f64" else P.PlannerDit32)(n)
fft = P.fft_64_dit_with_planner if dt == "f64" else P.fft_32_dit_with_planner
ring = 48
re = torch.empty(n * ring, dtype=tdt, device="cuda"); im = torch.empty_like(re)
P.fill_uniform(re, im, n)
for i in range(4):
    fft(re[i * n:(i + 1) * n], im[i * n:(i + 1) * n], P.Direction.Forward, pl)
torch.cuda.synchronize()
# ticks per us: s_memtime around a known wait
reps = []
trace = torch.zeros(3 * 4096 * 16, dtype=torch.int64, device="cuda")
acc = None
for rep in range(8):
    trace.zero_()
    lib.phast_debug_set_trace(C.c_void_p(trace.data_ptr()))
    j = 8 + rep
    fft(re[j * n:(j + 1) * n], im[j * n:(j + 1) * n], P.Direction.Forward, pl)
    torch.cuda.synchronize()
    lib.phast_debug_set_trace(C.c_void_p(0))
    reps.append(trace.cpu().numpy().reshape(3, 4096, 16).astype(np.float64))
print(dt, f"2^{log_n}", pl.describe_call())
if os.environ.get("PHAST_TRACE_DUMP"):
    np.save(os.environ["PHAST_TRACE_DUMP"], np.stack(reps)[:, :, :1100, :].astype(np.int64))
# ticks per us: the traced kernels' spans in ticks against the same kernels' HIP-event durations (Planner::time_passes)
acc = None
for i in range(16):
    ms = pl.time_passes(re[(24 + i) * n:(25 + i) * n], im[(24 + i) * n:(25 + i) * n], n, reps=1)
    acc = ms if acc is None else [x + y for x, y in zip(acc, ms)]
kern_us = [1e3 * x / 16 for x in acc]
spans = []
for p in range(len(kern_us)):
    v = []
    for t in reps:
        tp = t[p]; tp = tp[tp[:, 0] != 0]
        if len(tp): v.append(tp[tp != 0].max() - tp[:, 0].min())
    spans.append(float(np.median(v)) if v else 0.0)
TICK_US = float(np.median([sp / (us - 0.6) for sp, us in zip(spans, kern_us) if sp > 0]))   # (~0.6 us of an event pair is launch + drain outside the waves)
print("kernel us (events):", [round(x, 2) for x in kern_us], " spans (ticks):", [round(x) for x in spans], f" -> {TICK_US:.1f} ticks per us")
for p in range(3):
    rows = []
    for t in reps:
        tp = t[p]; tp = tp[tp[:, 0] != 0]
        if len(tp): rows.append(tp)
    if not rows: continue
    nst = int((rows[0][0] != 0).sum())
    rel = np.concatenate([(tp[:, :nst] - tp[:, 0].min()) for tp in rows]) / TICK_US
    print(f" pass {p}: {len(rows[0])} waves x {len(rows)} runs; kernel span (first entry -> last stamp) median "
          f"{np.median([(tp[:, :nst].max() - tp[:, 0].min()) / TICK_US for tp in rows]):.2f} us")
    for s in range(nst):
        d = rel[:, s] - rel[:, s - 1] if s else rel[:, 0]
        print(f"   stamp {s:2d}: reached at min {rel[:, s].min():6.2f}  med {np.median(rel[:, s]):6.2f}  max {rel[:, s].max():6.2f} us"
              f"   | since previous stamp: med {np.median(d):5.2f}  p90 {np.percentile(d, 90):5.2f} us")
