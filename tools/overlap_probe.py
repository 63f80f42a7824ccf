"""Development probe: do two independent wavefronts on two streams hide each other's launch drains?
Renders P passes with one context, then P/2 passes each with two contexts driven from two host threads."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Context, scenes  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
s = scenes.sponza_class()


def mk():
    c = Context(s, 1920, 1080, max_depth=8, seed=0, batch=batch)
    c.set_camera(s.camera)
    c.render(batch)
    c.synchronize()
    return c


a = mk()
t0 = time.perf_counter(); a.render(P); a.synchronize(); t1 = time.perf_counter()
print(f"one context: {(t1 - t0) / P * 1e3:.3f} ms/pass")
b = mk()
for rep in range(2):
    th = [threading.Thread(target=lambda c=c: (c.render(P // 2), c.synchronize())) for c in (a, b)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    t1 = time.perf_counter()
    print(f"two contexts, two threads: {(t1 - t0) / P * 1e3:.3f} ms/pass")
t0 = time.perf_counter(); a.render(P); a.synchronize(); t1 = time.perf_counter()
print(f"one context again: {(t1 - t0) / P * 1e3:.3f} ms/pass")
