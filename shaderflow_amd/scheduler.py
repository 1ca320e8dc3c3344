"""
Frame scheduling.

The reference drives `ShaderScene.next(dt)` from a task scheduler (shaderflow/scheduler.py). What the export path
observes of it is the `dt` arithmetic of a FREEWHEEL task — virtual time that never sleeps:

    last_call = -period ; next_call = 0
    every call:  now = next_call ; dt = now - last_call ; last_call = now
                 while next_call <= now: next_call += period

`next_call` accumulates `period` in float64, so `dt` is not exactly 1/fps: at 60 fps it takes 12 distinct values
around 0.01666…, and those exact values feed `scene.time`, the DynamicNumber integrators and the PCM chunk
arithmetic. `freewheel_clock()` reproduces them (pinned bit for bit by tests/golden/clock.npz).

`SchedulerTask` / `Scheduler` keep the reference's surface (new/once/next/clear, `.fps`, `.period`, realtime sleeping)
for code written against it.
"""
from __future__ import annotations

import inspect
import time as _time
from contextlib import nullcontext
from typing import Any, Callable, Optional


def precise_sleep(seconds: float, *, error: float = 0.001) -> None:
    """Sleep most of the interval, spin the last millisecond"""
    deadline = _time.monotonic() + seconds
    if seconds > error:
        _time.sleep(seconds - error)
    while _time.monotonic() < deadline:
        pass


class SchedulerTask:
    """One periodic (or one-shot) callable. If the callable takes `dt`, it receives the time since its last call."""

    def __init__(self, task: Callable, args: Optional[list] = None, kwargs: Optional[dict] = None, *,
                 frequency: float = 60.0, frameskip: bool = True, freewheel: bool = False, precise: bool = False,
                 once: bool = False, enabled: bool = True, context: Any = None,
                 started: Optional[float] = None, next_call: Optional[float] = None, last_call: Optional[float] = None):
        self.task, self.args, self.kwargs = task, list(args or []), dict(kwargs or {})
        self.frequency, self.frameskip, self.freewheel, self.precise = frequency, frameskip, freewheel, precise
        self.once, self.enabled, self.context, self.output = once, enabled, context or nullcontext(), None
        self._wants_dt = "dt" in inspect.signature(task).parameters
        self.started = 0 if freewheel else (_time.monotonic() if started is None else started)
        self.last_call = (last_call or self.started) - self.period
        self.next_call = (next_call or self.started)

    # rate ---------------------------------------------------------------------------------------------------

    @property
    def fps(self) -> float:
        return self.frequency

    @fps.setter
    def fps(self, value: float) -> None:
        self.frequency = value

    @property
    def period(self) -> float:
        return 1.0/self.frequency

    @period.setter
    def period(self, seconds: float) -> None:
        self.frequency = 1/seconds

    # ordering: one-shot tasks first, then by due time ---------------------------------------------------------

    def _key(self) -> tuple:
        return (not self.once, self.next_call)

    def __lt__(self, other: "SchedulerTask") -> bool:
        return self._key() < other._key()

    @property
    def should_delete(self) -> bool:
        return self.once and not self.enabled

    @property
    def should_live(self) -> bool:
        return not self.should_delete

    # run ----------------------------------------------------------------------------------------------------

    def _wait_until_due(self, block: bool) -> bool:
        """Realtime only: False when the task is not due and we must not block"""
        remaining = max(0, self.next_call - _time.monotonic())
        if remaining and not block:
            return False
        (precise_sleep if self.precise else _time.sleep)(remaining)
        return True

    def next(self, block: bool = True) -> "SchedulerTask":
        if not self.freewheel and not self._wait_until_due(block):
            return self
        now = self.next_call if self.freewheel else _time.monotonic()
        if self._wants_dt:
            elapsed = now - self.last_call
            self.kwargs["dt"] = elapsed if self.frameskip else min(elapsed, self.period)
        self.last_call = now
        with self.context:
            self.output = self.task(*self.args, **self.kwargs)
        while self.next_call <= now:
            self.next_call += self.period
        self.enabled = not self.once
        return self


class Scheduler:
    Task = SchedulerTask

    def __init__(self):
        self.tasks: list[SchedulerTask] = []

    def add(self, task: SchedulerTask) -> SchedulerTask:
        self.tasks.append(task)
        return task

    def new(self, task: Callable, **options) -> SchedulerTask:
        return self.add(SchedulerTask(task, **options))

    def once(self, task: Callable, **options) -> SchedulerTask:
        return self.add(SchedulerTask(task, once=True, **options))

    def delete(self, task: SchedulerTask) -> None:
        self.tasks.remove(task)

    def clear(self) -> None:
        self.tasks.clear()

    @property
    def enabled_tasks(self):
        return (task for task in self.tasks if task.enabled)

    @property
    def next_task(self) -> Optional[SchedulerTask]:
        return min(self.enabled_tasks, default=None)

    def _sanitize(self) -> None:
        self.tasks = [task for task in self.tasks if task.should_live]

    def next(self, block: bool = True) -> Optional[SchedulerTask]:
        task = self.next_task
        if task is None:
            return None
        try:
            return task.next(block=block)
        finally:
            if task.should_delete:
                self._sanitize()

    def all_once(self) -> None:
        for task in [task for task in self.tasks if task.once]:
            task.next()
        self._sanitize()


def freewheel_clock_by_task(fps: float, frames: int, speed: float = 1.0):
    """(time, dt, rdt) that the modules of each frame SEE in a freewheel export, as lists of python floats: a SchedulerTask stepped
    frame by frame, exactly as the scene does. The scene stores them after the frame has run (reference scene.py:475-479), so frame
    0 sees zeros. Kept as the definition (and the checker of freewheel_clock in tests/test_host.py): 20 µs per frame."""
    ticks: list[float] = []
    task = SchedulerTask(lambda dt=0.0: ticks.append(dt), frequency=fps, freewheel=True)
    seen_time, seen_dt, seen_rdt = [], [], []
    now, dt, rdt = 0.0, 0.0, 0.0
    for _ in range(frames):
        seen_time.append(now); seen_dt.append(dt); seen_rdt.append(rdt)
        task.next()
        task.fps = fps                      # the scene re-assigns the rate every frame
        rdt = ticks[-1]
        dt = rdt*speed
        now += dt
    return seen_time, seen_dt, seen_rdt


_clock_cache: dict = {}


def freewheel_clock(fps: float, frames: int, speed: float = 1.0):
    """The same three lists from SchedulerTask.next's float64 operations written out (freewheel, frameskip: `now = next_call`,
    `dt = now - last_call`, `next_call += period` until it has passed `now`) — same operations in the same order, so the same bits —
    without an object, a lambda and a context manager per frame (a fifth of the time). The
    sequence of a longer export starts with that of a shorter one, so the longest one computed per (fps, speed) is kept."""
    key = (float(fps), float(speed))
    cached = _clock_cache.get(key)
    if cached is None or len(cached[0]) < frames:
        period = 1.0/fps
        last_call, next_call = 0 - period, 0                  # SchedulerTask.__init__ with freewheel: started = 0
        seen_time, seen_dt, seen_rdt = [], [], []
        now, dt, rdt = 0.0, 0.0, 0.0
        for _ in range(frames):
            seen_time.append(now); seen_dt.append(dt); seen_rdt.append(rdt)
            call = next_call                                  # SchedulerTask.next
            rdt = call - last_call
            last_call = call
            while next_call <= call:
                next_call += period
            dt = rdt*speed
            now += dt
        cached = _clock_cache[key] = (seen_time, seen_dt, seen_rdt)
    return cached[0][:frames], cached[1][:frames], cached[2][:frames]
