"""
Second-order dynamics used to smooth spectrogram bins, loudness and camera parameters.

Host-side numpy implementation of the reference's `DynamicNumber` / `ShaderDynamics`
(shaderflow/dynamics.py:77-305): same fields, same update rule, same dtype behaviour (a float32 state stays
float32 because the python scalars dt/k1/k2/k3 are "weak" under numpy's promotion rules). The per-frame loop
uses this class directly; the batched export path runs the SAME recurrence on the device
(csrc/audio_kernels.hpp k_dynamics_scan) from the coefficients `coefficients()` returns, so that the branch
selection and the libm calls (exp/cos/cosh) stay on the host, exactly where the reference evaluates them.
"""
from __future__ import annotations

import math
from collections.abc import Iterable
from copy import deepcopy
from numbers import Number
from typing import Optional

import numpy as np
from attrs import define, field

from shaderflow_amd.module import ShaderModule
from shaderflow_amd.variable import ShaderVariable, Uniform


def dynamics_coefficients(frequency: float, zeta: float, response: float, dt: float) -> tuple[float, float, float, int]:
    """(k1, k2, k3, branch) the update uses for a step of `dt` (dynamics.py:172-195, 231-242).
    branch 0: k2 clamped for stability; branch 1: pole matching for fast systems."""
    radians = math.tau*frequency
    k3 = (response*zeta)/(math.tau*frequency)
    if (radians*dt < zeta):
        k1 = zeta/(math.pi*frequency)
        k2 = max(k1*dt, 1.0/(radians*radians), 0.5*(k1 + dt)*dt)
        return k1, k2, k3, 0
    damping = radians*(abs(zeta*zeta - 1.0))**0.5
    t1 = math.exp(-1*zeta*radians*dt)
    a1 = 2*t1*(math.cos if zeta <= 1 else math.cosh)(damping*dt)
    t2 = 1/(1 + t1*t1 - a1)*dt
    return t2*(1 - t1*t1), t2*dt, k3, 1


class _NumberLike(Number):
    """A DynamicNumber can be used where a number is expected: arithmetic acts on `.value` (reference: the
    NumberDunder base, dynamics.py:22-73). Operators are generated from the `operator` module."""

    def __float__(self): return float(self.value)
    def __int__(self): return int(self.value)
    def __str__(self): return str(self.value)
    def __hash__(self): return id(self)


def _install_operators() -> None:
    import operator
    for name in ("mul", "add", "sub", "truediv", "floordiv", "mod", "pow"):
        apply = getattr(operator, name)
        setattr(_NumberLike, f"__{name}__", lambda self, other, _f=apply: _f(self.value, other))
        # the reference reflects `other ∘ x` onto `x ∘ other` (dynamics.py:36-73); kept, also for non-commutative ones
        setattr(_NumberLike, f"__r{name}__", lambda self, other, _f=apply: _f(self.value, other))


_install_operators()


@define(slots=False, eq=False)
class DynamicNumber(_NumberLike):

    def _as_array(self, value) -> np.ndarray:
        if isinstance(value, np.ndarray):
            return value
        dtype = getattr(value, "dtype", self.dtype)          # a numpy scalar keeps ITS dtype (dynamics.py:94-101)
        if dtype == "quaternion":
            return value
        return np.array(value, dtype=dtype)

    def _as_array_setattr(self, attribute, value) -> np.ndarray:
        return self._as_array(value)

    value: np.ndarray = field(default=0, on_setattr=_as_array_setattr)
    target: np.ndarray = field(default=0, on_setattr=_as_array_setattr)
    dtype: np.dtype = field(default=np.float64)
    initial: np.ndarray = field(default=None)

    frequency: float = 1.0
    zeta: float = 1.0
    response: float = 0.0
    precision: float = 1e-6
    integral: np.ndarray = 0.0
    integrate: bool = False
    derivative: np.ndarray = 0.0
    acceleration: np.ndarray = 0.0
    previous: np.ndarray = 0.0

    def __attrs_post_init__(self):
        self.set(self.target or self.value)

    def set(self, value, *, instant: bool = True) -> None:
        value = self._as_array(value)
        self.value = deepcopy(value) if (instant) else self.value
        self.target = deepcopy(value)
        self.initial = deepcopy(value)
        self.previous = deepcopy(value) if (instant) else self.previous
        zeros = np.zeros_like(value)
        self.integral = deepcopy(zeros)
        self.derivative = deepcopy(zeros)
        self.acceleration = deepcopy(zeros)

    def reset(self, instant: bool = False):
        self.set(self.initial, instant=instant)

    # coefficient accessors kept for API parity (dynamics.py:172-195)
    @property
    def radians(self) -> float:
        return math.tau*self.frequency

    @property
    def k1(self) -> float:
        return self.zeta/(math.pi*self.frequency)

    @property
    def k2(self) -> float:
        return 1.0/(self.radians*self.radians)

    @property
    def k3(self) -> float:
        return (self.response*self.zeta)/(math.tau*self.frequency)

    @property
    def damping(self) -> float:
        return self.radians*(abs(self.zeta*self.zeta - 1.0))**0.5

    def coefficients(self, dt: float) -> tuple[float, float, float, int]:
        return dynamics_coefficients(self.frequency, self.zeta, self.response, dt)

    def next(self, target=None, dt: float = 1.0):
        """One semi-implicit Euler step towards `target` (dynamics.py:197-250)"""
        if (not dt):
            return self.value

        if (target is not None):
            self.target = self._as_array(target)
            if (self.target.shape != self.value.shape):
                self.set(target)

        # Within precision of the target: frozen, only the integral keeps running
        if (np.abs(self.target - self.value).max() < self.precision):
            if (self.integrate):
                self.integral += (self.value*dt)
            return self.value

        velocity = (self.target - self.previous)/dt
        self.previous = self.target
        k1, k2, k3, _ = self.coefficients(dt)

        self.value += (self.derivative*dt)
        self.acceleration = (self.target + k3*velocity - self.value - k1*self.derivative)/k2
        self.derivative += (self.acceleration*dt)
        if (self.integrate):
            self.integral += (self.value*dt)
        return self.value

    @staticmethod
    def extract(*objects) -> tuple:
        return tuple(obj.value if isinstance(obj, DynamicNumber) else obj for obj in objects)


@define(eq=False, slots=False)
class ShaderDynamics(ShaderModule, DynamicNumber):
    """A DynamicNumber that lives in a scene and exports `<name>`, `<name>Integral`, `<name>Derivative`
    uniforms (dynamics.py:260-305)"""
    name: str = "iShaderDynamics"
    real: bool = False
    primary: bool = True
    differentiate: bool = False

    def build(self) -> None:
        DynamicNumber.__attrs_post_init__(self)

    def setup(self) -> None:
        self.reset(instant=self.scene.freewheel)

    def update(self) -> None:
        self.next(dt=abs(self.scene.rdt if self.real else self.scene.dt))

    @property
    def type(self) -> Optional[str]:
        if not (shape := self.value.shape):
            return "float"
        return {1: "float", 2: "vec2", 3: "vec3", 4: "vec4"}.get(shape[0])

    def pipeline_token(self):
        # the bytes of what pipeline() exports: a frozen system (dynamics.py:222-225) is skipped, a moving one walked
        value = self.value
        if value.__class__ is not np.ndarray or type(self).pipeline is not ShaderDynamics.pipeline:
            return None                                              # (a subclass with a pipeline() of its own is walked every frame)
        token = value.tobytes()
        if self.integrate:
            token += np.asarray(self.integral).tobytes()
        if self.differentiate:
            token += np.asarray(self.derivative).tobytes()
        return (self.name, self.type, self.primary, self.integrate, self.differentiate, token)

    def pipeline(self) -> Iterable[ShaderVariable]:
        if (not self.type):
            return None
        if (self.primary):
            yield Uniform(self.type, f"{self.name}", self.value)
        if (self.integrate):
            yield Uniform(self.type, f"{self.name}Integral", self.integral)
        if (self.differentiate):
            yield Uniform(self.type, f"{self.name}Derivative", self.derivative)
