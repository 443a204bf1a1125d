"""Python host of the C ABI: one `HydroEngine` = one `hydro_t` handle on one GPU.

PyTorch is used for plumbing only - device buffers, the current HIP stream and
(in `distributed.py`) the process group.  Every numerical result comes from the
hand-written HIP kernels in `csrc/hydro_kernels.hip` through `libhydro.so`;
there is no CPU or eager-PyTorch fallback.

Layouts: "SoA" tensors are contiguous `(F, N)` float32 device tensors - row f is
field f (orders in include/hydro.h).
"""
from __future__ import annotations

import ctypes
import logging

import numpy as np
import torch

from . import _native as nat
from ._native import HydroError

log = logging.getLogger("silver2_isaacsim_amd")
_warp_mode_announced = False


def _as_soa_tensor(x, fields: int, device: torch.device) -> torch.Tensor:
    """(N,F) or (F,N) array-like -> contiguous (F,N) float32 tensor on `device`."""
    t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
    if t.ndim != 2:
        raise ValueError("expected a 2-D array")
    if t.shape[0] != fields and t.shape[1] == fields:
        t = t.t()
    elif t.shape[0] != fields:
        raise ValueError(f"expected {fields} fields, got shape {tuple(t.shape)}")
    return t.to(device=device, dtype=torch.float32).contiguous()


class HydroEngine:
    """Batched replacement for N per-prim calculator objects of the reference
    (`WarpHydrodynamicsWrapper`, warp_hydrodynamics_wrapper.py:5-132)."""

    def __init__(self, capacity: int, device: int | str | torch.device = 0,
                 water_density: float = 1025.0, gravity: float = 9.81):
        self._lib = nat.load()                       # raises if the HIP extension is missing
        dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if dev.type != "cuda":
            raise ValueError("HydroEngine runs on a GPU device only (no CPU path)")
        self.device = torch.device("cuda", dev.index if dev.index is not None else 0)
        self.capacity = int(capacity)
        self._h = ctypes.c_void_p()
        rc = self._lib.hydro_create(self.device.index, self.capacity, ctypes.byref(self._h))
        if rc != nat.HYDRO_OK:
            self._h = None
            raise HydroError(rc, f"hydro_create(device={self.device.index}, capacity={capacity}) failed")
        self.n = 0
        self.coeff_dtype = "f32"
        self.semantics = "numba"
        self._tables: dict = {}
        self.set_scene(water_density, gravity)

    # ------------------------------------------------------------------ utils
    def _check(self, rc: int) -> None:
        if rc != nat.HYDRO_OK:
            msg = self._lib.hydro_last_error(self._h)
            raise HydroError(rc, msg.decode() if msg else "")

    def _table(self, t: torch.Tensor, fields: int):
        """Pointer table for a contiguous (fields, N) float32 device tensor (cached)."""
        key = (t.data_ptr(), t.shape[1], fields)
        tab = self._tables.get(key)
        if tab is None:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.shape[0] != fields or t.device != self.device:
                raise ValueError(f"expected contiguous float32 ({fields},N) tensor on {self.device}")
            stride = t.shape[1] * 4
            base = t.data_ptr()
            tab = nat.pointer_table([base + f * stride for f in range(fields)])
            if len(self._tables) > 256:
                self._tables.clear()
            self._tables[key] = tab
        return tab

    def _stream(self, stream) -> ctypes.c_void_p:
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        return ctypes.c_void_p(int(stream.cuda_stream if hasattr(stream, "cuda_stream") else stream))

    # ------------------------------------------------------------ configuration
    def set_scene(self, water_density: float, gravity: float) -> None:
        self.water_density, self.gravity = float(water_density), float(gravity)
        self._check(self._lib.hydro_set_scene(self._h, self.water_density, self.gravity))

    def set_semantics(self, semantics: str = "numba") -> None:
        """'numba' (default, the parity target) or 'warp': follow the reference's Warp twin where the two
        calculators differ (added-mass rotation, centres of a dry body; include/hydro.h).
        PARITY UNPINNED for 'warp': that mode restates warp_hydrodynamics.py from its source text; the reference holds
        no outputs of its Warp calculator and `warp` is not importable where this package is built.  Said once per
        process through the package logger."""
        global _warp_mode_announced
        code = {"numba": nat.HYDRO_SEM_NUMBA, "warp": nat.HYDRO_SEM_WARP}.get(semantics)
        if code is None:
            raise ValueError("semantics must be 'numba' or 'warp'")
        if semantics == "warp" and not _warp_mode_announced:
            _warp_mode_announced = True
            log.warning("semantics='warp': restated from the source text of the reference's warp_hydrodynamics.py, no "
                        "reference outputs behind it (parity unpinned); 'numba' is the verified mode")
        self._check(self._lib.hydro_set_semantics(self._h, code))
        self.semantics = semantics

    def set_params(self, params, coeff_dtype: str = "f32") -> None:
        """Per-body constants, (N,11) or (11,N): dims(3), cd_lin, cd_ang, damp_lin, damp_ang,
        lift, am_lin, am_ang, mass.  coeff_dtype 'f16' stores the seven coefficients as half."""
        t = _as_soa_tensor(params, nat.PARAM_FIELDS, self.device)
        n = t.shape[1]
        fn = {"f32": self._lib.hydro_set_params_f32, "f16": self._lib.hydro_set_params_f16}[coeff_dtype]
        torch.cuda.current_stream(self.device).synchronize()      # the copy runs on the engine's stream
        self._check(fn(self._h, n, self._table(t, nat.PARAM_FIELDS), 1))
        self.n = n
        self.coeff_dtype = coeff_dtype

    def reserve_soa(self) -> None:
        """Make the engine's plain-SoA copies of the parameters / previous velocity now (82 B per body).  They are
        otherwise made by the first call of an entry that takes plain field pointers (`step_wrench`, `step_components`),
        which then allocates and synchronises once - call this first if that call is to be captured into a HIP graph."""
        self._check(self._lib.hydro_reserve_soa(self._h))

    def set_tuning(self, bodies_per_lane: int = 0, block_threads: int = 0, non_temporal: int = -1,
                   waves_per_simd: int = -1) -> None:
        self._check(self._lib.hydro_set_tuning(self._h, bodies_per_lane, block_threads, non_temporal, waves_per_simd))

    # ------------------------------------------------- previous-velocity state
    def reset_prev_velocity(self) -> None:
        self._check(self._lib.hydro_reset_prev_velocity(self._h))

    def get_prev_velocity(self) -> torch.Tensor:
        """(6, n) copy of the engine-owned previous velocity (checkpoint / resume).  The C call works on the engine's
        private stream: the step that last wrote the records ran on the caller's, so that one is drained first."""
        out = torch.empty((nat.PREV_FIELDS, self.n), dtype=torch.float32, device=self.device)
        torch.cuda.current_stream(self.device).synchronize()
        self._check(self._lib.hydro_get_prev_velocity(self._h, self.n, self._table(out, nat.PREV_FIELDS), 1))
        return out

    def set_prev_velocity(self, prev) -> None:
        t = _as_soa_tensor(prev, nat.PREV_FIELDS, self.device)
        torch.cuda.current_stream(self.device).synchronize()
        self._check(self._lib.hydro_set_prev_velocity(self._h, t.shape[1], self._table(t, nat.PREV_FIELDS), 1))

    # ----------------------------------------------------------------- hot path
    def step_wrench(self, state: torch.Tensor, dt: float, out: torch.Tensor | None = None,
                    prev: torch.Tensor | None = None, stream=None) -> torch.Tensor:
        """Fused wrench for state (13,N) -> out (6,N) [F|T].  With `prev` (6,N) the caller owns
        the previous velocity (buffer-swap mode, nothing else is written); without it the
        engine's own previous-velocity buffer is read and updated."""
        n = state.shape[1]
        if out is None:
            out = torch.empty((nat.WRENCH_FIELDS, n), dtype=torch.float32, device=self.device)
        st, ot = self._table(state, nat.STATE_FIELDS), self._table(out, nat.WRENCH_FIELDS)
        if prev is None:
            rc = self._lib.hydro_step_wrench(self._h, n, st, float(dt), ot, self._stream(stream))
        else:
            rc = self._lib.hydro_step_wrench_ext(self._h, n, st, self._table(prev, nat.PREV_FIELDS),
                                                 float(dt), ot, self._stream(stream))
        self._check(rc)
        return out

    # ------------------------------------------------- tiled SoA (native layout)
    @staticmethod
    def tiles(n: int) -> int:
        return (int(n) + nat.TILE - 1) // nat.TILE

    def alloc_tiled(self, fields: int, n: int) -> torch.Tensor:
        """Zeroed (tiles, fields, 64) float32 buffer: body i, field f at [i // 64, f, i % 64]."""
        return torch.zeros((self.tiles(n), fields, nat.TILE), dtype=torch.float32, device=self.device)

    def _check_tiled(self, t: torch.Tensor, fields: int, n: int) -> None:
        if (t.dtype != torch.float32 or t.device != self.device or not t.is_contiguous() or t.ndim != 3
                or t.shape[1] != fields or t.shape[2] != nat.TILE or t.shape[0] < self.tiles(n)):
            raise ValueError(f"expected contiguous float32 (>= {self.tiles(n)}, {fields}, {nat.TILE}) tensor on {self.device}")

    def to_tiled(self, soa: torch.Tensor, out: torch.Tensor | None = None, stream=None) -> torch.Tensor:
        """(F,N) plain SoA -> (tiles,F,64) tiled, on device (hydro_repack)."""
        f, n = soa.shape
        if out is None:
            out = self.alloc_tiled(f, n)
        self._check_tiled(out, f, n)
        self._check(self._lib.hydro_repack(self._h, n, f, self._table(soa, f), out.data_ptr(), f * nat.TILE, 1, self._stream(stream)))
        return out

    def from_tiled(self, tiled: torch.Tensor, n: int, out: torch.Tensor | None = None, stream=None) -> torch.Tensor:
        f = tiled.shape[1]
        self._check_tiled(tiled, f, n)
        if out is None:
            out = torch.empty((f, n), dtype=torch.float32, device=self.device)
        self._check(self._lib.hydro_repack(self._h, n, f, self._table(out, f), tiled.data_ptr(), f * nat.TILE, 0, self._stream(stream)))
        return out

    def _check_ke_out(self, ke_out: torch.Tensor) -> None:
        if ke_out.dtype != torch.float64 or ke_out.device != self.device or ke_out.numel() < 2 or not ke_out.is_contiguous():
            raise ValueError(f"ke_out: expected a contiguous float64 tensor of 2 elements on {self.device}")

    def step_wrench_tiled(self, state: torch.Tensor, n: int, dt: float, out: torch.Tensor | None = None,
                          prev: torch.Tensor | None = None, stream=None, ke_out: torch.Tensor | None = None,
                          rotational: bool = True) -> torch.Tensor:
        """Fused wrench on tiled buffers: state (tiles,13,64) -> out (tiles,6,64).
        prev: None = engine-owned previous velocity (read + updated); a (tiles,6,64) tensor; or a
        (tiles,13,64) STATE tensor whose velocity fields are used in place (ping-pong integrator:
        pass the previous step's state buffer - nothing is copied).
        ke_out (float64, 2 elements, on the device): the kernel also samples the kinetic energy of `state` - the
        bodies it holds in registers anyway, no second pass - into ke_out[0:2] = [translational, rotational]."""
        self._check_tiled(state, nat.STATE_FIELDS, n)
        if out is None:
            out = self.alloc_tiled(nat.WRENCH_FIELDS, n)
        self._check_tiled(out, nat.WRENCH_FIELDS, n)
        if prev is None:
            p_ptr, p_stride = None, 0
        elif prev.shape[1] == nat.STATE_FIELDS:
            self._check_tiled(prev, nat.STATE_FIELDS, n)
            p_ptr, p_stride = prev.data_ptr() + 7 * nat.TILE * 4, nat.STATE_FIELDS * nat.TILE
        else:
            self._check_tiled(prev, nat.PREV_FIELDS, n)
            p_ptr, p_stride = prev.data_ptr(), nat.PREV_FIELDS * nat.TILE
        if ke_out is None:
            self._check(self._lib.hydro_step_wrench_tiled(
                self._h, n, state.data_ptr(), nat.STATE_FIELDS * nat.TILE, p_ptr, p_stride, float(dt),
                out.data_ptr(), nat.WRENCH_FIELDS * nat.TILE, self._stream(stream)))
        else:
            self._check_ke_out(ke_out)
            self._check(self._lib.hydro_step_wrench_tiled_ke(
                self._h, n, state.data_ptr(), nat.STATE_FIELDS * nat.TILE, p_ptr, p_stride, float(dt),
                out.data_ptr(), nat.WRENCH_FIELDS * nat.TILE, int(bool(rotational)), ke_out.data_ptr(), self._stream(stream)))
        return out

    def prepare_step_wrench_tiled(self, state: torch.Tensor, n: int, dt: float, out: torch.Tensor | None = None,
                                  prev: torch.Tensor | None = None, stream=None, ke_out: torch.Tensor | None = None,
                                  rotational: bool = True):
        """Validate the arguments of `step_wrench_tiled` ONCE and return a zero-argument callable that issues
        that launch again (same buffers, same stream - the one current now if `stream` is None).  For step
        loops over small scenes, where the per-call Python work (shape checks, ctypes conversions: ~10 us)
        exceeds the kernel (~3 us at 4 096 bodies): the prepared call costs ~4 us of host time.  The callable
        returns `out`; errors raise HydroError as usual."""
        self._check_tiled(state, nat.STATE_FIELDS, n)
        if out is None:
            out = self.alloc_tiled(nat.WRENCH_FIELDS, n)
        self._check_tiled(out, nat.WRENCH_FIELDS, n)
        if prev is None:
            p_ptr, p_stride = None, 0
        elif prev.shape[1] == nat.STATE_FIELDS:
            self._check_tiled(prev, nat.STATE_FIELDS, n)
            p_ptr, p_stride = prev.data_ptr() + 7 * nat.TILE * 4, nat.STATE_FIELDS * nat.TILE
        else:
            self._check_tiled(prev, nat.PREV_FIELDS, n)
            p_ptr, p_stride = prev.data_ptr(), nat.PREV_FIELDS * nat.TILE
        args = (self._h, ctypes.c_int64(n), ctypes.c_void_p(state.data_ptr()), ctypes.c_int64(nat.STATE_FIELDS * nat.TILE),
                ctypes.c_void_p(p_ptr), ctypes.c_int64(p_stride), ctypes.c_double(float(dt)),
                ctypes.c_void_p(out.data_ptr()), ctypes.c_int64(nat.WRENCH_FIELDS * nat.TILE))
        if ke_out is None:
            fn = self._lib.hydro_step_wrench_tiled
            args = args + (self._stream(stream),)
        else:                                           # the variant that samples the kinetic energy of `state` on the way
            self._check_ke_out(ke_out)
            fn = self._lib.hydro_step_wrench_tiled_ke
            args = args + (ctypes.c_int(int(bool(rotational))), ctypes.c_void_p(ke_out.data_ptr()), self._stream(stream))
        keep = (state, prev, out, ke_out)               # the buffers must outlive the callable
        check = self._check

        def step():
            if self._h is None:                         # engine closed: the captured handle is gone
                raise HydroError(-5, "engine is closed")
            rc = fn(*args)
            if rc:
                check(rc)
            return keep[2]
        return step

    @staticmethod
    def prepare_step_wrench_tiled_batch(engines, states, dt: float, outs=None, prevs=None, ns=None, stream=None):
        """k independent scenes in ONE launch (hydro_step_wrench_tiled_batch): engines[i] steps states[i] (tiled
        (tiles,13,64)) into outs[i] ((tiles,6,64), allocated when None) with prevs[i] as the previous velocity (tiled
        6-field buffers or the velocity fields of a previous state buffer; None = every engine's own).  Same bits as k
        single calls; one ramp and drain for all of them.  Returns (step, outs): `step()` re-issues the launch on the
        stream current at the time of the call (arguments are validated here, once)."""
        k = len(engines)
        if not 1 <= k <= nat.BATCH_MAX or len(states) != k:
            raise ValueError(f"1 .. {nat.BATCH_MAX} scenes per launch, one state buffer each")
        ns = list(ns) if ns is not None else [e.n for e in engines]
        outs = list(outs) if outs is not None else [e.alloc_tiled(nat.WRENCH_FIELDS, n) for e, n in zip(engines, ns)]
        arr = (nat.Scene * k)()
        keep = []
        for i, (e, st, n, out) in enumerate(zip(engines, states, ns, outs)):
            e._check_tiled(st, nat.STATE_FIELDS, n)
            e._check_tiled(out, nat.WRENCH_FIELDS, n)
            sc = arr[i]
            sc.engine, sc.n = e._h, n
            sc.state, sc.state_tile_stride = st.data_ptr(), st.shape[1] * nat.TILE
            sc.wrench, sc.wrench_tile_stride = out.data_ptr(), out.shape[1] * nat.TILE
            if prevs is None:
                sc.prev, sc.prev_tile_stride = None, 0
            else:
                pv = prevs[i]
                if pv.shape[1] == nat.STATE_FIELDS:                 # a previous STATE buffer: its six velocity fields
                    e._check_tiled(pv, nat.STATE_FIELDS, n)
                    sc.prev, sc.prev_tile_stride = pv.data_ptr() + 7 * nat.TILE * 4, nat.STATE_FIELDS * nat.TILE
                else:
                    e._check_tiled(pv, nat.PREV_FIELDS, n)
                    sc.prev, sc.prev_tile_stride = pv.data_ptr(), nat.PREV_FIELDS * nat.TILE
                keep.append(pv)
            keep += [st, out]
        first = engines[0]
        fn, dtc = first._lib.hydro_step_wrench_tiled_batch, ctypes.c_double(dt)

        def step(stream=stream):
            rc = fn(k, arr, dtc, first._stream(stream))
            if rc:
                first._check(rc)
            return outs
        step._keep = (keep, arr)
        return step, outs

    @staticmethod
    def step_wrench_tiled_batch(engines, states, dt: float, outs=None, prevs=None, ns=None, stream=None):
        """One launch for k scenes; returns the list of wrench buffers.  See prepare_step_wrench_tiled_batch."""
        step, outs = HydroEngine.prepare_step_wrench_tiled_batch(engines, states, dt, outs, prevs, ns, stream)
        step()
        return outs

    def step_fused_tiled(self, state: torch.Tensor, prev_state: torch.Tensor, n: int, dt: float,
                         state_out: torch.Tensor | None = None, wrench: torch.Tensor | None = None,
                         implicit_drag: bool = False, stream=None, ke_out: torch.Tensor | None = None,
                         rotational: bool = True):
        """Wrench + integrator in one kernel.  `prev_state` (tiles,13,64) supplies the previous velocity;
        `state_out` defaults to `prev_state` itself (ping-pong: the old buffer receives the new state).
        ke_out (float64, 2 elements, device): also sample the kinetic energy of the NEW state into it."""
        self._check_tiled(state, nat.STATE_FIELDS, n)
        self._check_tiled(prev_state, nat.STATE_FIELDS, n)
        if state_out is None:
            state_out = prev_state
        self._check_tiled(state_out, nat.STATE_FIELDS, n)
        w_ptr, w_stride = None, 0
        if wrench is not None:
            self._check_tiled(wrench, nat.WRENCH_FIELDS, n)
            w_ptr, w_stride = wrench.data_ptr(), nat.WRENCH_FIELDS * nat.TILE
        st = nat.STATE_FIELDS * nat.TILE
        if ke_out is None:
            self._check(self._lib.hydro_step_fused_tiled(
                self._h, n, state.data_ptr(), st, prev_state.data_ptr() + 7 * nat.TILE * 4, st, float(dt),
                state_out.data_ptr(), st, w_ptr, w_stride, int(bool(implicit_drag)), self._stream(stream)))
        else:
            self._check_ke_out(ke_out)
            self._check(self._lib.hydro_step_fused_tiled_ke(
                self._h, n, state.data_ptr(), st, prev_state.data_ptr() + 7 * nat.TILE * 4, st, float(dt),
                state_out.data_ptr(), st, w_ptr, w_stride, int(bool(implicit_drag)), int(bool(rotational)),
                ke_out.data_ptr(), self._stream(stream)))
        return state_out

    def step_fused_tiled_multi(self, state: torch.Tensor, prev_state: torch.Tensor, n: int, dt: float, steps: int,
                               state_out: torch.Tensor | None = None, implicit_drag: bool = False, stream=None,
                               ke_out: torch.Tensor | None = None, rotational: bool = True):
        """`steps` closed-loop steps in one kernel, every body carried through them in registers (same bits as `steps`
        calls of step_fused_tiled).  Buffers as step_fused_tiled: `prev_state` supplies the previous velocity and, by
        default, receives the final state; the velocity fields of `state` receive the velocity of the step before the
        last one, so that after the call (state_out, state) are the (current, previous) pair of the next call.
        ke_out: also sample the kinetic energy of the final state."""
        self._check_tiled(state, nat.STATE_FIELDS, n)
        self._check_tiled(prev_state, nat.STATE_FIELDS, n)
        if state_out is None:
            state_out = prev_state
        self._check_tiled(state_out, nat.STATE_FIELDS, n)
        if ke_out is not None:
            self._check_ke_out(ke_out)
        st, vel = nat.STATE_FIELDS * nat.TILE, 7 * nat.TILE * 4
        self._check(self._lib.hydro_step_fused_tiled_multi(
            self._h, n, state.data_ptr(), st, prev_state.data_ptr() + vel, st, float(dt), int(steps),
            state_out.data_ptr(), st, state.data_ptr() + vel, st, int(bool(implicit_drag)), int(bool(rotational)),
            ke_out.data_ptr() if ke_out is not None else None, self._stream(stream)))
        return state_out

    def integrate_tiled(self, state_in: torch.Tensor, wrench: torch.Tensor, n: int, dt: float,
                        state_out: torch.Tensor | None = None, stream=None) -> torch.Tensor:
        if state_out is None:
            state_out = torch.zeros_like(state_in)
        for t, f in ((state_in, nat.STATE_FIELDS), (wrench, nat.WRENCH_FIELDS), (state_out, nat.STATE_FIELDS)):
            self._check_tiled(t, f, n)
        self._check(self._lib.hydro_integrate_tiled(
            self._h, n, state_in.data_ptr(), nat.STATE_FIELDS * nat.TILE, wrench.data_ptr(), nat.WRENCH_FIELDS * nat.TILE,
            float(dt), state_out.data_ptr(), nat.STATE_FIELDS * nat.TILE, self._stream(stream)))
        return state_out

    def pack_state_aos(self, positions: torch.Tensor, orientations: torch.Tensor, velocities: torch.Tensor,
                       out: torch.Tensor | None = None, quat_xyzw: bool = False, stream=None) -> torch.Tensor:
        """Simulator tensors (N,3), (N,4), (N,6) -> tiled state (tiles,13,64)."""
        n = positions.shape[0]
        if out is None:
            out = self.alloc_tiled(nat.STATE_FIELDS, n)
        self._check_tiled(out, nat.STATE_FIELDS, n)
        self._check(self._lib.hydro_pack_state_aos(
            self._h, n, positions.data_ptr(), orientations.data_ptr(), int(bool(quat_xyzw)), velocities.data_ptr(),
            out.data_ptr(), nat.STATE_FIELDS * nat.TILE, self._stream(stream)))
        return out

    def unpack_wrench_aos(self, wrench: torch.Tensor, n: int, forces: torch.Tensor | None = None,
                          torques: torch.Tensor | None = None, stream=None):
        self._check_tiled(wrench, nat.WRENCH_FIELDS, n)
        if forces is None:
            forces = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        if torques is None:
            torques = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        self._check(self._lib.hydro_unpack_wrench_aos(self._h, n, wrench.data_ptr(), nat.WRENCH_FIELDS * nat.TILE,
                                                      forces.data_ptr(), torques.data_ptr(), self._stream(stream)))
        return forces, torques

    def step_wrench_aos(self, positions: torch.Tensor, orientations: torch.Tensor, velocities: torch.Tensor,
                        dt: float, forces: torch.Tensor | None = None, torques: torch.Tensor | None = None,
                        quat_xyzw: bool = False, stream=None):
        """Fused wrench on the simulator's tensors: positions (N,3), orientations (N,4) (WXYZ as the
        simulator gives them, or XYZW with quat_xyzw=True), velocities (N,6) -> forces (N,3),
        torques (N,3).  Uses and updates the engine's previous-velocity state."""
        n = positions.shape[0]
        for t, w in ((positions, 3), (orientations, 4), (velocities, 6)):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != (n, w) or t.device != self.device:
                raise ValueError(f"expected contiguous float32 ({n},{w}) tensor on {self.device}")
        if forces is None:
            forces = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        if torques is None:
            torques = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        self._check(self._lib.hydro_step_wrench_aos(
            self._h, n, positions.data_ptr(), orientations.data_ptr(), int(bool(quat_xyzw)), velocities.data_ptr(), float(dt),
            forces.data_ptr(), torques.data_ptr(), self._stream(stream)))
        return forces, torques

    def prepare_step_wrench_aos(self, positions: torch.Tensor, orientations: torch.Tensor, velocities: torch.Tensor,
                                forces: torch.Tensor | None = None, torques: torch.Tensor | None = None,
                                quat_xyzw: bool = False):
        """Validate the arguments of `step_wrench_aos` ONCE and return `step(dt, stream=None) -> (forces, torques)`,
        which re-issues that launch on the same buffers (a simulator's tensor API hands out views of the same
        device buffers every physics step).  For the plugin path, where the per-call Python work of
        `step_wrench_aos` (five tensor checks, pointer conversions) is several times the kernel at 20 bodies.
        `stream`: a torch stream / raw handle; None = the stream current at the time of the call."""
        n = positions.shape[0]
        for t, w in ((positions, 3), (orientations, 4), (velocities, 6)):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != (n, w) or t.device != self.device:
                raise ValueError(f"expected contiguous float32 ({n},{w}) tensor on {self.device}")
        if forces is None:
            forces = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        if torques is None:
            torques = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        for t in (forces, torques):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != (n, 3) or t.device != self.device:
                raise ValueError(f"expected contiguous float32 ({n},3) output on {self.device}")
        fn, check = self._lib.hydro_step_wrench_aos, self._check
        head = (self._h, ctypes.c_int64(n), ctypes.c_void_p(positions.data_ptr()), ctypes.c_void_p(orientations.data_ptr()),
                ctypes.c_int(int(bool(quat_xyzw))), ctypes.c_void_p(velocities.data_ptr()))
        tail = (ctypes.c_void_p(forces.data_ptr()), ctypes.c_void_p(torques.data_ptr()))
        keep = (positions, orientations, velocities, forces, torques)          # the buffers must outlive the callable
        dev, cur = self.device, torch.cuda.current_stream
        # the current stream's raw handle without building a torch.cuda.Stream object (0.2 instead of 1 us per step)
        raw_current, dev_index = getattr(torch._C, "_cuda_getCurrentRawStream", None), self.device.index
        last = [None, None]                              # (dt, its c_double): a simulator steps with ONE dt

        def step(dt: float, stream=None):
            if self._h is None:
                raise HydroError(-5, "engine is closed")
            if stream is None:
                sp = raw_current(dev_index) if raw_current is not None else cur(dev).cuda_stream
            else:
                sp = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
            if dt != last[0]:
                last[0], last[1] = dt, ctypes.c_double(dt)
            rc = fn(*head, last[1], *tail, sp)
            if rc:
                check(rc)
            return keep[3], keep[4]
        return step

    def step_components(self, state: torch.Tensor, accel: torch.Tensor, out: torch.Tensor | None = None,
                        ratio: torch.Tensor | None = None, stream=None):
        """Component mode: state (13,N), accel (6,N) -> comps (24,N), ratio (N,)."""
        n = state.shape[1]
        if out is None:
            out = torch.empty((nat.COMP_FIELDS, n), dtype=torch.float32, device=self.device)
        if ratio is None:
            ratio = torch.empty((n,), dtype=torch.float32, device=self.device)
        self._check(self._lib.hydro_step_components(
            self._h, n, self._table(state, nat.STATE_FIELDS), self._table(accel, nat.PREV_FIELDS),
            self._table(out, nat.COMP_FIELDS), ratio.data_ptr(), self._stream(stream)))
        return out, ratio

    def step_components_aos(self, position, orientation_xyzw, linear_vel, angular_vel, linear_accel, angular_accel,
                            out: torch.Tensor, ratio: torch.Tensor | None = None, stream=None) -> torch.Tensor:
        """Component mode on the calculator's argument layout: (N,3)/(N,4) tensors in, out (8,N,3)."""
        n = position.shape[0]
        ins = (position, orientation_xyzw, linear_vel, angular_vel, linear_accel, angular_accel)
        for t, w in zip(ins, (3, 4, 3, 3, 3, 3)):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != (n, w) or t.device != self.device:
                raise ValueError(f"expected contiguous float32 ({n},{w}) tensor on {self.device}")
        if out.shape != (8, n, 3) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device:
            raise ValueError(f"expected contiguous float32 (8,{n},3) output on {self.device}")
        key = ("comp_aos", out.data_ptr(), n)
        tab = self._tables.get(key)
        if tab is None:
            tab = self._tables[key] = nat.pointer_table([out.data_ptr() + k * n * 12 for k in range(8)])
        self._check(self._lib.hydro_step_components_aos(
            self._h, n, *[t.data_ptr() for t in ins], tab, ratio.data_ptr() if ratio is not None else None,
            self._stream(stream)))
        return out

    def kinetic_energy(self, state: torch.Tensor, rotational: bool = False, out: torch.Tensor | None = None,
                       stream=None) -> torch.Tensor:
        """[sum 1/2 m v^2, sum 1/2 w.I.w] as a float64 device tensor of shape (2,)."""
        if out is None:
            out = torch.empty((2,), dtype=torch.float64, device=self.device)
        if state.ndim == 3:                       # tiled (tiles,13,64): n = bodies with parameters set
            self._check_tiled(state, nat.STATE_FIELDS, self.n)
            self._check(self._lib.hydro_kinetic_energy_tiled(self._h, self.n, state.data_ptr(), nat.STATE_FIELDS * nat.TILE,
                                                             int(bool(rotational)), out.data_ptr(), self._stream(stream)))
            return out
        self._check(self._lib.hydro_kinetic_energy(self._h, state.shape[1], self._table(state, nat.STATE_FIELDS),
                                                   int(bool(rotational)), out.data_ptr(), self._stream(stream)))
        return out

    def ke_allreduce(self, nccl_comm: int, ke: torch.Tensor, stream=None) -> torch.Tensor:
        """Sum the pair a kinetic-energy entry left in `ke` over the ranks of an RCCL communicator (raw ncclComm_t
        address), in place, on `stream` (hydro_ke_allreduce: the C-level route; the Python monitor of simulate.py uses
        torch.distributed for the same collective)."""
        self._check_ke_out(ke)
        self._check(self._lib.hydro_ke_allreduce(self._h, ctypes.c_void_p(nccl_comm), ke.data_ptr(), self._stream(stream)))
        return ke

    def ke_rearm(self, stream=None) -> None:
        """Zero the ticket counters of the kinetic-energy reduction on `stream` (hydro_ke_rearm): the recovery path after
        a launch that did not run to its end - such a launch leaves NaNs in its output, never a stale pair."""
        self._check(self._lib.hydro_ke_rearm(self._h, self._stream(stream)))

    @staticmethod
    def bind_rccl(library: "ctypes.CDLL | str | None") -> str:
        """Tell hydro_ke_allreduce which RCCL to call: the ctypes library object (or path) of the copy that made the
        communicators - a communicator belongs to ONE loaded copy.  None forgets the binding.  Returns
        hydro_rccl_origin()."""
        lib = nat.load()
        if library is None:
            lib.hydro_bind_rccl(None, None)
        else:
            rccl = ctypes.CDLL(library) if isinstance(library, str) else library
            err = ctypes.cast(rccl.ncclGetErrorString, ctypes.c_void_p) if hasattr(rccl, "ncclGetErrorString") else None
            lib.hydro_bind_rccl(ctypes.cast(rccl.ncclAllReduce, ctypes.c_void_p), err)
        return lib.hydro_rccl_origin().decode()

    def integrate(self, state_in: torch.Tensor, wrench: torch.Tensor, dt: float,
                  state_out: torch.Tensor | None = None, stream=None) -> torch.Tensor:
        if state_out is None:
            state_out = torch.empty_like(state_in)
        self._check(self._lib.hydro_integrate(
            self._h, state_in.shape[1], self._table(state_in, nat.STATE_FIELDS), self._table(wrench, nat.WRENCH_FIELDS),
            float(dt), self._table(state_out, nat.STATE_FIELDS), self._stream(stream)))
        return state_out

    # ----------------------------------------------------------------- lifetime
    def sync(self) -> None:
        self._check(self._lib.hydro_sync(self._h))
        torch.cuda.current_stream(self.device).synchronize()

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.hydro_destroy(self._h)
            self._h = None
            self._tables.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
