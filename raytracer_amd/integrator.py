"""Python mirror of the reference's Integrator protocol
(src/cpu/Integrator.hpp:14-30) over the C-ABI: same method names and call
order as the app uses (src/app/App.cpp:40-43,92-93,113-121)."""
from __future__ import annotations

import threading
import time

import numpy as np

from .api import Context
from .scenes import Camera, Scene


class HipPathTracer:
    tileSize = 64  # src/cpu/Integrator.hpp:56

    def __init__(self, scene: Scene, width: int, height: int, max_depth: int = 4, seed: int = 0, device: int = 0,
                 passes_per_update: int = 1):
        # one pass per update() is the reference's protocol; more passes per update are traced as one wavefront
        self._passes = max(1, int(passes_per_update))
        self._ctx = Context(scene, width, height, max_depth=max_depth, seed=seed, device=device, batch=self._passes)
        self._width, self._height = width, height
        self._frame = np.zeros((height, width, 4), np.float32)
        self._texture = np.zeros((height, width, 4), np.float32)
        self._isFinish = True
        self._isRun = True
        self._camera = None
        self._thread = None
        self._frames = 0
        self.error = None

    # src/cpu/Integrator.cpp:69-79.  The thread leaves _isFinish set whichever way it ends.
    def run(self):
        def loop():
            try:
                while self._isRun:
                    if not self._isFinish:
                        try:
                            self.update()
                        except Exception as e:  # surface the library's error to the caller's thread
                            self.error = e
                            self._isRun = False
                        self._isFinish = True
                    else:
                        time.sleep(0)
            finally:
                self._isFinish = True
        if self._thread is None:
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()

    def isUpdateFinish(self) -> bool:
        return self._isFinish

    # src/cpu/Integrator.cpp:85-92; does nothing after an error or after finish()
    def nextFrame(self, camera: Camera):
        if self.error is not None or not self._isRun:
            return
        r, f = camera.right.astype(np.float32), camera.forward.astype(np.float32)
        c = np.array([r[1] * f[2] - f[1] * r[2], r[2] * f[0] - f[2] * r[0], r[0] * f[1] - f[0] * r[1]], np.float32)
        up = c * (np.float32(1.0) / np.sqrt(np.float32(c[0] * c[0] + c[1] * c[1] + c[2] * c[2])))
        self._camera = Camera(camera.pos.astype(np.float32), r, up.astype(np.float32), f)
        self._isFinish = False

    # src/cpu/Integrator.cpp:94-97: stops the loop and joins the thread (never waits on _isFinish, which a thread that
    # has already left its loop would not set)
    def finish(self):
        self._isRun = False
        if self._thread is not None:
            self._thread.join()
        self._isFinish = True

    def getTexture(self) -> np.ndarray:
        return self._texture

    # src/cpu/Integrator.cpp:21-24
    def sync(self):
        np.copyto(self._texture, self._frame)

    # src/cpu/Integrator.cpp:26-67
    def update(self):
        self._ctx.set_camera(self._camera)
        self._ctx.render(self._passes)
        self._frame = self._ctx.read_film()
        self._frames += self._passes

    @property
    def frames(self) -> int:
        return self._frames

    def close(self):
        self.finish()
        self._ctx.close()
