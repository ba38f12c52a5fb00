"""Per-call time of the host-slice API at small sizes (with a planner): A/B of PHAST_ZERO_COPY inside one gpurun call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, phastft_amd as P
for L in [int(a) for a in sys.argv[1:]] or (4, 8, 10, 12, 13, 14):
    n = 1 << L
    for name, dt, Pl, fn in (("f64", np.float64, P.PlannerDit64, P.fft_64_dit_with_planner), ("f32", np.float32, P.PlannerDit32, P.fft_32_dit_with_planner)):
        pl = Pl(n)
        rng = np.random.default_rng(L)
        re, im = rng.uniform(-1, 1, n).astype(dt), rng.uniform(-1, 1, n).astype(dt)
        ref = np.fft.fft(re.astype(np.float64) + 1j * im.astype(np.float64))
        a, b = re.copy(), im.copy()
        fn(a, b, P.Direction.Forward, pl)
        err = np.sqrt(np.sum(np.abs(a + 1j * b - ref) ** 2) / np.sum(np.abs(ref) ** 2))
        assert err < (1e-13 if dt == np.float64 else 1e-5), err
        for _ in range(20): fn(a, b, P.Direction.Forward, pl)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(200): fn(a, b, P.Direction.Forward, pl)
            best = min(best, (time.perf_counter() - t0) / 200)
        print(f"ZERO_COPY={os.environ.get('PHAST_ZERO_COPY','1')} {name} 2^{L}: {1e6*best:.1f} us per host-slice call")
