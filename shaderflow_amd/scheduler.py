"""
Task scheduler (reference: shaderflow/scheduler.py). Only what the headless path observes is kept: a task's
`dt` arithmetic in freewheel mode (scheduler.py:87-89,134-173) — `started = 0`, `last_call = -period`,
`next_call = 0`, then `now = next_call; dt = now - last_call; next_call += period` — bit for bit, because
the accumulated float64 `dt` is what the DynamicNumber integrators and the PCM chunk arithmetic see.
Realtime sleeping is kept for completeness.
"""
from __future__ import annotations

import contextlib
import inspect
import time
from collections import deque
from typing import Any, Callable, Iterable, Optional

from attrs import Factory, define, field


def precise_sleep(sleep: float, *, error: float = 0.001) -> None:
    start = time.monotonic()
    if (ahead := max(0, sleep - error)):
        time.sleep(ahead)
    else:
        return
    while (time.monotonic() - start) < sleep:
        pass


@define(eq=False)
class SchedulerTask:
    task: Callable
    args: list = field(factory=list, repr=False)
    kwargs: dict = field(factory=dict, repr=False)
    output: Any = field(default=None, repr=False)
    context: Any = Factory(contextlib.nullcontext)
    enabled: bool = True
    once: bool = False
    frequency: float = 60.0
    frameskip: bool = True
    freewheel: bool = False
    precise: bool = False
    started: float = Factory(time.monotonic)
    next_call: float = None
    last_call: float = None
    _dt: bool = False

    def __attrs_post_init__(self):
        self._dt = ("dt" in inspect.signature(self.task).parameters)
        if self.freewheel:
            self.started = 0
        self.last_call = (self.last_call or self.started) - self.period
        self.next_call = (self.next_call or self.started)

    def __hash__(self) -> int:
        return id(self)

    @property
    def fps(self) -> float:
        return self.frequency

    @fps.setter
    def fps(self, value: float):
        self.frequency = value

    @property
    def period(self) -> float:
        return (1.0/self.frequency)

    @period.setter
    def period(self, value: float):
        self.frequency = (1/value)

    @property
    def should_delete(self) -> bool:
        return (self.once and (not self.enabled))

    @property
    def should_live(self) -> bool:
        return (not self.should_delete)

    def __lt__(self, other) -> bool:
        if (self.once and not other.once):
            return True
        return (self.next_call < other.next_call)

    def __gt__(self, other) -> bool:
        if (not self.once and other.once):
            return True
        return (self.next_call > other.next_call)

    def next(self, block: bool = True):
        if (not self.freewheel):
            wait = max(0, (self.next_call - time.monotonic()))
            if (not block) and (wait > 0):
                return self
            (precise_sleep if self.precise else time.sleep)(wait)

        now = (self.next_call if self.freewheel else time.monotonic())

        if (self._dt):
            self.kwargs["dt"] = (now - self.last_call)
            if (not self.frameskip):
                self.kwargs["dt"] = min(self.kwargs["dt"], self.period)

        self.last_call = now

        with self.context:
            self.output = self.task(*self.args, **self.kwargs)

        while (self.next_call <= now):
            self.next_call += self.period

        self.enabled = (not self.once)
        return self


@define
class Scheduler:
    Task = SchedulerTask
    tasks: deque = Factory(deque)

    def add(self, task: SchedulerTask) -> SchedulerTask:
        self.tasks.append(task)
        return task

    def new(self, task: Callable, **options) -> SchedulerTask:
        return self.add(SchedulerTask(task=task, **options))

    def once(self, task: Callable, **options) -> SchedulerTask:
        return self.add(SchedulerTask(task=task, **options, once=True))

    def delete(self, task: SchedulerTask) -> None:
        self.tasks.remove(task)

    def clear(self) -> None:
        self.tasks.clear()

    @property
    def enabled_tasks(self) -> Iterable[SchedulerTask]:
        for task in self.tasks:
            if task.enabled:
                yield task

    @property
    def next_task(self) -> Optional[SchedulerTask]:
        return min(self.enabled_tasks, default=None)

    def _sanitize(self) -> None:
        alive = [task for task in self.tasks if task.should_live]
        self.tasks.clear()
        self.tasks.extend(alive)

    def next(self, block=True) -> Optional[SchedulerTask]:
        if (task := self.next_task) is None:
            return None
        try:
            return task.next(block=block)
        finally:
            if task.should_delete:
                self._sanitize()

    def all_once(self) -> None:
        for task in list(self.tasks):
            if task.once:
                task.next()
        self._sanitize()


def freewheel_clock(fps: float, frames: int, speed: float = 1.0):
    """The (time, dt, rdt) each frame's modules see in a freewheel export: scheduler.py:152-173 feeding
    scene.py:475-479 (values are stored AFTER the frame ran, so frame 0 sees zeros). Lists of python floats."""
    ticks: list[float] = []
    task = SchedulerTask(task=lambda dt=0.0: ticks.append(dt), frequency=fps, freewheel=True, precise=True)
    times, dts, rdts = [], [], []
    time_, dt, rdt = 0.0, 0.0, 0.0
    for _ in range(frames):
        times.append(time_); dts.append(dt); rdts.append(rdt)
        task.next()
        task.fps = fps
        dt = ticks[-1]*speed
        rdt = ticks[-1]
        time_ += dt
    return times, dts, rdts
