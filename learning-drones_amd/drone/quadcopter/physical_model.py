"""QuadcopterPhysics: mirror of drone/quadcopter/physical_model.py:29-122 -- same constructor arguments and
attribute names, and calculate_accelerations() with the reference's signature, evaluated on the GPU through
ld_accelerations_* (float64 entry point: the reference's arithmetic including its float32 rounding points).

The reference's own fly() calls it once before its loop (flight/__init__.py:35-37) and Drone.measure calls it
every step (drone/drone.py:131-136); here the fused kernels contain that arithmetic, and this class exists
for callers that reach for ``drone.quadcopter.calculate_accelerations`` directly."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ...params import SimParams


class QuadcopterPhysics:
    def __init__(self, mass_center: float = 0.200, mass_motor: float = 0.050, radius_motor_center: float = 0.1,
                 I_x: float = 2160e-6, I_y: float = 2160e-6, I_z: float = 4210e-6, coef_force: float = 5,
                 coef_moment: float = 0.5, coef_wind: float = 0.1, gravity: float = 9.81, mass_payload: float = 0,
                 x_payload: float = 0, y_payload: float = 0, precision: str = "f64"):
        # defaults: drone/quadcopter/parameters.py:15-35 (what drone/drone.py:34-47 passes)
        self.m_c, self.m_m, self.m_p = mass_center, mass_motor, mass_payload
        self.r_m, self.c_f, self.c_m, self.c_w = radius_motor_center, coef_force, coef_moment, coef_wind
        self.params = SimParams.reference(mass_center=mass_center, mass_motor=mass_motor,
                                          radius_motor_center=radius_motor_center, I_x=I_x, I_y=I_y, I_z=I_z,
                                          coef_force=coef_force, coef_moment=coef_moment, coef_wind=coef_wind,
                                          gravity=gravity, mass_payload=mass_payload, x_payload=x_payload,
                                          y_payload=y_payload)
        self.G = self.params.gravity_force
        self.moments_payload = np.array(list(self.params.moments_payload), dtype=np.float64).reshape(3, 1)
        self.inverse_inertia_vector = np.array([[1 / I_x, 1 / I_y, 1 / I_z]]).T
        self.angle_control_to_thrust = np.array([[1, -1, -1], [-1, -1, 1], [-1, 1, -1], [1, 1, 1]])
        self.Rot = np.identity(3, dtype=np.float32)
        self.precision = precision

    def sim_params(self, wind=(0.0, 0.0, 0.0)) -> SimParams:
        """The uniform physics block the kernels take (struct ld_sim_params) with this wind vector."""
        p = SimParams.from_buffer_copy(self.params)
        p.wind[:] = [float(w) for w in np.asarray(wind, dtype=float).ravel()]
        return p

    def calculate_accelerations(self, rotation, wind_speed, thrust, lin_acc_drone_2_lab: bool = True):
        """rotation (3,1) = roll, pitch, yaw; wind_speed (3,1); thrust (4,1) -> (lin_acc (3,1), rot_acc (3,1)),
        physical_model.py:75-122.  lin_acc_drone_2_lab=True (what every caller in the reference passes): linear
        acceleration in the lab frame and Euler-angle accelerations; False: the pair in the drone's frame, before
        Rot^T and the Euler-rate matrix are applied (:114-120).  Either way the call leaves the rotation matrix in
        self.Rot (:90).  One launch of ld_accelerations_frames_*."""
        import torch
        from ... import _native
        _native.require_gpu()
        f64 = self.precision == "f64"
        dtype = torch.float64 if f64 else torch.float32
        dev = torch.device("cuda", torch.cuda.current_device())
        host = np.concatenate([np.asarray(rotation, dtype=np.float64).reshape(3), np.asarray(thrust, dtype=np.float64).reshape(4)])
        inp = torch.as_tensor(host).to(dev, dtype)                       # one upload: angle[3], thrust[4]
        out = torch.empty((15,), dtype=dtype, device=dev)                # one download: lin[3], rot[3], Rot[9]
        fn = _native.lib().ld_accelerations_frames_f64 if f64 else _native.lib().ld_accelerations_frames_f32
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        sim = self.sim_params(wind_speed)
        size = inp.element_size()
        _native.check(fn(C.byref(sim), C.c_void_p(inp.data_ptr()), C.c_void_p(inp.data_ptr() + 3 * size),
                         C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr() + 3 * size), C.c_void_p(out.data_ptr() + 6 * size),
                         1 if lin_acc_drone_2_lab else 0, 1, stream), "ld_accelerations_frames")
        res = out.double().cpu().numpy()
        self.Rot = res[6:15].reshape(3, 3).copy()
        lin = res[0:3].reshape(3, 1).copy()
        if not lin_acc_drone_2_lab:
            lin = lin.astype(np.float32)                                 # the reference's forces / m stays float32 (:98,:114)
        return lin, res[3:6].reshape(3, 1).copy()
