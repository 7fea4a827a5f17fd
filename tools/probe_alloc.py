"""Where does the non-kernel time of a bench step go?  Times table create / free and a stats round trip."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kat_amd
eng = kat_amd.Engine(0)
for slots in (1 << 20, 1_600_000_000, 4_770_000_000):
    for rep in range(2):
        t0 = time.perf_counter(); t = eng.table(27, True, size_hint=slots); eng.sync(); t1 = time.perf_counter()
        t.stats(want_total=False); t2 = time.perf_counter()
        t.free(); eng.sync(); t3 = time.perf_counter()
        print("slots %d: create+memset %.1f ms, stats %.3f ms, free %.1f ms" % (slots, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
