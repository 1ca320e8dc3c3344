"""
ShaderCamera as a uniform source for the headless path (reference: shaderflow/camera.py:132-447).

The fragments read the camera through `GetCamera(iCamera)` (camera.glsl:132-155): five direct uniforms
(`iCameraMode/Projection/Right/Upward/Forward`, camera.py:196-201) plus nine ShaderDynamics children
(`iCameraPosition, Separation, Rotation(primary=False), Zenith, Zoom, Isometric, FocalLength, Orbital, Dolly`,
camera.py:147-185). All of that is kept, with the same names, defaults, frequencies and creation order — the
order fixes where the children sit in `scene.modules`. Keyboard/mouse driven motion (camera.py:240-355) is
interactive-only and not part of the export path; `move/rotate/look/align` remain for scenes that animate the
camera from `update()`. Rotations use a small (w, x, y, z) quaternion helper instead of numpy-quaternion.
"""
from __future__ import annotations

import math
from collections.abc import Iterable
from enum import Enum

import numpy as np
from attrs import define, field

from shaderflow_amd.dynamics import DynamicNumber, ShaderDynamics
from shaderflow_amd.module import ShaderModule
from shaderflow_amd.variable import ShaderVariable, Uniform


class GlobalBasis:
    Origin = np.array((0, 0, 0), dtype=np.float64)
    Null = np.array((0, 0, 0), dtype=np.float64)
    Up = np.array((0, 1, 0), dtype=np.float64)
    Down = np.array((0, -1, 0), dtype=np.float64)
    Left = np.array((-1, 0, 0), dtype=np.float64)
    Right = np.array((1, 0, 0), dtype=np.float64)
    Forward = np.array((0, 0, 1), dtype=np.float64)
    Backward = np.array((0, 0, -1), dtype=np.float64)


class CameraProjection(Enum):
    Perspective = 0
    Stereoscopic = 1
    Equirectangular = 2

    @classmethod
    def _missing_(cls, value):
        names = {"perspective": 0, "default": 0, "stereoscopic": 1, "stereo": 1, "vr": 1, "sbs": 1,
                 "spherical": 2, "equirectangular": 2, "360": 2}
        if value in names:
            return cls(names[value])
        raise ValueError(f"{value} is not a valid {cls.__name__}")


class CameraMode(Enum):
    FreeCamera = 0
    Camera2D = 1
    Spherical = 2

    @classmethod
    def _missing_(cls, value):
        names = {"free": 0, "freecamera": 0, "2d": 1, "plane": 1, "flat": 1, "spherical": 2, "aligned": 2}
        if value in names:
            return cls(names[value])
        raise ValueError(f"{value} is not a valid {cls.__name__}")


class Algebra:
    """Quaternions as float64 arrays (w, x, y, z)"""

    @staticmethod
    def quaternion(axis: np.ndarray, degrees: float) -> np.ndarray:
        theta = math.radians(degrees/2)
        return np.array((math.cos(theta), *(math.sin(theta)*np.asarray(axis, np.float64))), np.float64)

    @staticmethod
    def multiply(a: np.ndarray, b: np.ndarray) -> np.ndarray:
        aw, ax, ay, az = a
        bw, bx, by, bz = b
        return np.array((
            aw*bw - ax*bx - ay*by - az*bz,
            aw*bx + ax*bw + ay*bz - az*by,
            aw*by - ax*bz + ay*bw + az*bx,
            aw*bz + ax*by - ay*bx + az*bw,
        ), np.float64)

    @staticmethod
    def conjugate(q: np.ndarray) -> np.ndarray:
        return np.array((q[0], -q[1], -q[2], -q[3]), np.float64)

    @staticmethod
    def rotate_vector(vector: np.ndarray, R: np.ndarray) -> np.ndarray:
        R = np.asarray(R, np.float64)
        pure = np.array((0.0, *np.asarray(vector, np.float64)), np.float64)
        return Algebra.multiply(Algebra.multiply(R, pure), Algebra.conjugate(R))[1:]

    @staticmethod
    def angle(A, B) -> float:
        A, B = DynamicNumber.extract(A, B)
        if not (LA := np.linalg.norm(A)):
            return 0.0
        if not (LB := np.linalg.norm(B)):
            return 0.0
        return float(np.degrees(np.arccos(np.clip(np.dot(A, B)/(LA*LB), -1, 1))))

    @staticmethod
    def unit_vector(vector: np.ndarray) -> np.ndarray:
        if (magnitude := np.linalg.norm(vector)):
            return (vector/magnitude)
        return vector


@define(slots=False, eq=False)
class ShaderCamera(ShaderModule):
    name: str = "iCamera"
    mode: CameraMode = field(default=CameraMode.Camera2D, converter=CameraMode)
    projection: CameraProjection = field(default=CameraProjection.Perspective, converter=CameraProjection)
    separation: ShaderDynamics = None
    rotation: ShaderDynamics = None
    position: ShaderDynamics = None
    zenith: ShaderDynamics = None
    zoom: ShaderDynamics = None
    isometric: ShaderDynamics = None
    focus: ShaderDynamics = None
    orbital: ShaderDynamics = None
    dolly: ShaderDynamics = None

    def build(self):
        def child(suffix: str, frequency: float, value, **kw) -> ShaderDynamics:
            return ShaderDynamics(scene=self.scene, name=f"{self.name}{suffix}", real=True,
                                  frequency=frequency, zeta=1, response=0, value=value, **kw)
        self.position = child("Position", 4, np.copy(GlobalBasis.Origin))
        self.separation = child("Separation", 0.5, 0.05)
        self.rotation = child("Rotation", 5, np.array((1.0, 0.0, 0.0, 0.0)), primary=False)
        self.zenith = child("Zenith", 1, np.copy(GlobalBasis.Up))
        self.zoom = child("Zoom", 3, 1)
        self.isometric = child("Isometric", 1, 0)
        self.focus = child("FocalLength", 1, 1)
        self.orbital = child("Orbital", 1, 0)
        self.dolly = child("Dolly", 1, 0)

    @property
    def fov(self) -> float:
        return 2.0*math.degrees(math.atan(float(self.zoom.value) - float(self.isometric.value)))

    @fov.setter
    def fov(self, value: float):
        self.zoom.target = math.tan(math.radians(value)/2.0) + float(self.isometric.value)

    def pipeline_token(self):
        # mode, projection and the quaternion the three basis vectors are rotated by (three numpy rotations per frame otherwise)
        rotation = self.rotation.value
        if type(self).pipeline is not ShaderCamera.pipeline:
            return None
        return (self.mode, self.projection, rotation.tobytes() if rotation.__class__ is np.ndarray else None) if rotation.__class__ is np.ndarray else None

    def pipeline(self) -> Iterable[ShaderVariable]:
        yield Uniform("int", f"{self.name}Mode", value=self.mode.value)
        yield Uniform("int", f"{self.name}Projection", value=self.projection.value)
        yield Uniform("vec3", f"{self.name}Right", value=self.right)
        yield Uniform("vec3", f"{self.name}Upward", value=self.up)
        yield Uniform("vec3", f"{self.name}Forward", value=self.forward)

    # actions ------------------------------------------------------------------------------------------

    def move(self, direction: np.ndarray, absolute: bool = False):
        self.position.target = self.position.target + (direction - (self.position.target*absolute))
        return self

    def rotate(self, direction: np.ndarray, degrees: float = 0.0):
        target = Algebra.multiply(Algebra.quaternion(direction, degrees), self.rotation.target)
        self.rotation.target = target/np.linalg.norm(target)
        return self

    def rotate2d(self, degrees: float = 0.0):
        """Turn the UP vector around FORWARD by a plane angle (camera.py:220-223)"""
        target = Algebra.rotate_vector(self.zenith.value, Algebra.quaternion(self.forward_target, degrees))
        return self.align(self.up_target, target)

    def apply_zoom(self, value: float) -> None:
        """Zooming in then out by the same amount returns to the same value (camera.py:280-285)"""
        if (value > 0):
            self.zoom.target *= (1 + value)
        else:
            self.zoom.target /= (1 - value)

    def update(self):
        """The headless part of camera.py:240-278: spherical mode keeps the horizon level; key and mouse motion is interactive"""
        if self.mode == CameraMode.Spherical:
            self.align(self.right_target, self.zenith.target, 90)

    def align(self, A, B, degrees: float = 0.0):
        A, B = DynamicNumber.extract(A, B)
        return self.rotate(Algebra.unit_vector(np.cross(A, B)), Algebra.angle(A, B) - degrees)

    def look(self, target: np.ndarray):
        return self.align(self.forward_target, target - self.position.target)

    # basis ----------------------------------------------------------------------------------------------

    def _rotated(self, basis: np.ndarray, target: bool = False) -> np.ndarray:
        return Algebra.rotate_vector(basis, self.rotation.target if target else self.rotation.value)

    @property
    def right(self): return self._rotated(GlobalBasis.Right)
    @property
    def right_target(self): return self._rotated(GlobalBasis.Right, True)
    @property
    def left(self): return (-1)*self.right
    @property
    def up(self): return self._rotated(GlobalBasis.Up)
    @property
    def up_target(self): return self._rotated(GlobalBasis.Up, True)
    @property
    def down(self): return (-1)*self.up
    @property
    def forward(self): return self._rotated(GlobalBasis.Forward)
    @property
    def forward_target(self): return self._rotated(GlobalBasis.Forward, True)
    @property
    def backward(self): return (-1)*self.forward

    @property
    def left_target(self): return (-1)*self.right_target
    @property
    def down_target(self): return (-1)*self.up_target
    @property
    def backward_target(self): return (-1)*self.forward_target

    @property
    def x(self) -> float: return self.position.value[0]
    @x.setter
    def x(self, value: float): self.position.target[0] = value
    @property
    def y(self) -> float: return self.position.value[1]
    @y.setter
    def y(self, value: float): self.position.target[1] = value
    @property
    def z(self) -> float: return self.position.value[2]
    @z.setter
    def z(self, value: float): self.position.target[2] = value
