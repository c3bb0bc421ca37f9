"""On-device control-signal synthesis -- host side of `c3p_synth_signals`.

The reference builds each drive-line signal through the generator chain LO + AWG ->
DigitalToAnalog -> Mixer -> VoltsToHertz (c3/generator/generator.py:172-229 `generate_signals`,
devices.py:72-122,203-221,306-351,914-939,1073-1195; envelopes via Instruction.get_awg_signal
c3/signal/gates.py:341-370 and Envelope / EnvelopeDrag c3/signal/pulse.py:88-180).  Here a batch
is described by B x K x E envelope-parameter rows and the B x K x N samples are produced in HBM,
where `propagate_batch` consumes them without a host round trip.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

from . import _lib
from ._cache import TensorMemo
from ._lib import C3PropError
from .propagation import _Call, _is_torch, _ptr

# shape ids / parameter slots: include/c3prop.h (a test checks these against the header)
ENV_SHAPES = {
    "no_drive": 0,
    "rect": 1,
    "gaussian_nonorm": 2,
    "flattop": 3,
    "flattop_risefall": 4,
    "cosine": 5,
    "gaussian_sigma": 6,
    "gaussian": 7,
    "trapezoid": 8,
}
ENV_SLOTS = {
    "amp": 0,
    "xy_angle": 1,
    "freq_offset": 2,
    "delta": 3,
    "t_final": 4,
    "sigma": 5,
    "t_up": 6,
    "t_down": 7,
    "risefall": 8,
    "delay": 9,
    "flags": 10,
}
ENV_NPAR = 12
ENVF_T_BEFORE = 1
ENVF_DRAG = 2


def slice_num(t_start: float, t_end: float, resolution: float) -> int:
    """Device.calc_slice_num (devices.py:72-84)."""
    return int(np.abs(t_start - t_end) * resolution)


def pack_components(channels: Sequence[Sequence[Dict]], B: int = 1):
    """Envelope components per drive line -> (env_params [B,K,E,NPAR], env_shapes [K,E]).

    `channels[k]` lists the envelope components of line k as dicts keyed like the reference's
    `Envelope.params` (amp, xy_angle, freq_offset, delta, t_final, sigma, t_up, t_down, risefall)
    plus `shape` (a name from the envelope library), `use_t_before`, `drag`, `delay`.  A value may be
    a scalar (shared by the batch) or an array of B per-sample values.
    """
    K = len(channels)
    E = max(1, max(len(c) for c in channels))
    env = np.zeros((B, K, E, ENV_NPAR), dtype=np.float64)
    shapes = np.full((K, E), -1, dtype=np.int32)
    for k, comps in enumerate(channels):
        for e, comp in enumerate(comps):
            name = comp["shape"]
            if name not in ENV_SHAPES:
                raise C3PropError(f"C3:Error: envelope shape {name!r} is not available on the device (have {sorted(ENV_SHAPES)})")
            shapes[k, e] = ENV_SHAPES[name]
            for key, val in comp.items():
                if key in ("shape", "use_t_before", "drag"):
                    continue
                if key not in ENV_SLOTS:
                    raise C3PropError(f"C3:Error: unknown envelope parameter {key!r}")
                env[:, k, e, ENV_SLOTS[key]] = np.asarray(val, dtype=np.float64)
            fl = (ENVF_T_BEFORE if comp.get("use_t_before", False) else 0) | (ENVF_DRAG if comp.get("drag", False) else 0)
            env[:, k, e, ENV_SLOTS["flags"]] = float(fl)
            if name != "no_drive" and "t_final" not in comp:
                raise C3PropError("C3:Error: envelope component needs t_final")
    return env, shapes


_shapes_memo = TensorMemo()  # device shape tables: per tensor object (weak reference + version), never per address
_shapes_upload: Dict[tuple, object] = {}  # content-keyed uploads of host tables


def _shapes(call, env_shapes, K: int, E: int):
    """(host int32 [K,E], the array the library call takes): validated once per distinct table; on the device path the upload
    is cached -- an optimiser calls the synthesis and its vjp every iteration with the same shapes, and a pageable
    host-to-device copy per call is a synchronisation (and illegal inside a stream capture).  A device table is remembered
    per tensor OBJECT; the [K,E] check against the parameter rows runs on every call, hit or not."""
    on_dev = _is_torch(env_shapes) and env_shapes.is_cuda
    mkey = (str(env_shapes.dtype), str(call.dev)) if on_dev else None  # the upload lives on call.dev
    hit = _shapes_memo.get(env_shapes, mkey) if on_dev else None
    if hit is not None:
        if hit[0].shape != (K, E):
            raise C3PropError(f"C3:Error: env_shapes must be [{K},{E}], got {hit[0].shape}")
        return hit
    shapes_np = np.ascontiguousarray(np.asarray(env_shapes.cpu() if _is_torch(env_shapes) else env_shapes, dtype=np.int32))
    if shapes_np.shape != (K, E):
        raise C3PropError(f"C3:Error: env_shapes must be [{K},{E}], got {shapes_np.shape}")
    if shapes_np.max(initial=-1) >= len(ENV_SHAPES):
        raise C3PropError("C3:Error: env_shapes holds an unknown shape id")
    if not call.device:
        return shapes_np, shapes_np
    key2 = (shapes_np.tobytes(), shapes_np.shape, str(call.dev))
    shp = _shapes_upload.get(key2)
    if shp is None:
        if len(_shapes_upload) > 64:
            _shapes_upload.clear()
        shp = _shapes_upload[key2] = call.torch.as_tensor(shapes_np, device=call.dev)
    if on_dev:
        _shapes_memo.put(env_shapes, mkey, (shapes_np, shp))
    return shapes_np, shp


def synthesize_signals(env_params, env_shapes, carrier, t_start: float, t_end: float, awg_res: float, sim_res: float, *, want_iq: bool = False, device=None):
    """signals [B,K,N] (and optionally the AWG-resolution I/Q [B,K,2,Na]) from parameter rows.

    env_params [B,K,E,NPAR] f64, env_shapes [K,E] int32, carrier [B,K,2] f64 = (LO angular
    frequency, V_to_Hz).  numpy in -> numpy out (staged by the library); with `device` (or any torch
    CUDA input) the outputs are torch CUDA tensors that `propagate_batch` takes zero-copy.
    """
    if device is not None:
        import torch

        env_params = torch.as_tensor(np.asarray(env_params, dtype=np.float64) if not _is_torch(env_params) else env_params, device=device)
    call = _Call(env_params, carrier)
    env = call.f64(env_params)
    if env.ndim != 4 or env.shape[-1] != ENV_NPAR:
        raise C3PropError(f"C3:Error: env_params must be [B,K,E,{ENV_NPAR}], got {tuple(env.shape)}")
    B, K, E = (int(x) for x in env.shape[:3])
    shapes_np, shp = _shapes(call, env_shapes, K, E)
    car = call.f64(carrier)
    if tuple(car.shape) != (B, K, 2):
        raise C3PropError(f"C3:Error: carrier must be [{B},{K},2], got {tuple(car.shape)}")
    N, Na = slice_num(t_start, t_end, sim_res), slice_num(t_start, t_end, awg_res)
    if N <= 0 or Na <= 1:
        raise C3PropError(f"C3:Error: empty time grid (N={N}, AWG samples={Na})")
    if call.device:
        sig = call.torch.empty((B, K, N), dtype=call.torch.float64, device=call.dev)
        iq = call.torch.empty((B, K, 2, Na), dtype=call.torch.float64, device=call.dev) if want_iq else None
    else:
        sig = np.empty((B, K, N), dtype=np.float64)
        iq = np.empty((B, K, 2, Na), dtype=np.float64) if want_iq else None
    _lib.check(
        _lib.load().c3p_synth_signals(
            _ptr(env), _ptr(shp), _ptr(car), float(t_start), float(t_end), float(awg_res), float(sim_res), B, K, E, call.flags, _ptr(iq), _ptr(sig), call.stream
        )
    )
    return (sig, iq) if want_iq else sig


def synthesize_signals_vjp(env_params, env_shapes, carrier, t_start: float, t_end: float, awg_res: float, sim_res: float, grad_signals):
    """(grad_env [B,K,E,NPAR], grad_carrier [B,K,2]) from d loss / d signals [B,K,N].

    Differentiates amp, xy_angle, freq_offset, delta (their ENV_SLOTS) and the carrier pair; the
    reference gets the same numbers from the tape that also covers the propagation
    (optimizers/optimizer.py:206-216, gates.py:341-370).  Chain with `propagate_batch_vjp`.
    """
    call = _Call(env_params, carrier, grad_signals)
    env = call.f64(env_params)
    if env.ndim != 4 or env.shape[-1] != ENV_NPAR:
        raise C3PropError(f"C3:Error: env_params must be [B,K,E,{ENV_NPAR}], got {tuple(env.shape)}")
    B, K, E = (int(x) for x in env.shape[:3])
    shapes_np, shp = _shapes(call, env_shapes, K, E)
    car = call.f64(carrier)
    if tuple(car.shape) != (B, K, 2):
        raise C3PropError(f"C3:Error: carrier must be [{B},{K},2], got {tuple(car.shape)}")
    N, Na = slice_num(t_start, t_end, sim_res), slice_num(t_start, t_end, awg_res)
    if N <= 0 or Na <= 1:
        raise C3PropError(f"C3:Error: empty time grid (N={N}, AWG samples={Na})")
    gs = call.f64(grad_signals)
    if tuple(gs.shape) != (B, K, N):
        raise C3PropError(f"C3:Error: grad_signals must be [{B},{K},{N}], got {tuple(gs.shape)}")
    if call.device:
        t = call.torch
        genv = t.empty((B, K, E, ENV_NPAR), dtype=t.float64, device=call.dev)
        gcar = t.empty((B, K, 2), dtype=t.float64, device=call.dev)
    else:
        genv = np.empty((B, K, E, ENV_NPAR), dtype=np.float64)
        gcar = np.empty((B, K, 2), dtype=np.float64)
    _lib.check(
        _lib.load().c3p_synth_signals_vjp(
            _ptr(env), _ptr(shp), _ptr(car), float(t_start), float(t_end), float(awg_res), float(sim_res), B, K, E, call.flags, _ptr(gs), _ptr(genv), _ptr(gcar), call.stream
        )
    )
    return genv, gcar


def create_ts(t_start: float, t_end: float, resolution: float) -> np.ndarray:
    """Centred sample times (devices.py:86-122)."""
    num = slice_num(t_start, t_end, resolution)
    dt = 1.0 / resolution
    return np.linspace(t_start + dt / 2, t_end - dt / 2, num)


def generate_signals(channels: Dict[str, Dict], t_start: float, t_end: float, awg_res: float, sim_res: float) -> Dict[str, Dict[str, np.ndarray]]:
    """`Generator.generate_signals(instr)`-shaped result (generator.py:172-229) for one instruction.

    `channels[name] = {"components": [envelope dicts], "lo_freq": w [rad/s], "v_to_hz": f}`;
    returns `{name: {"values": [N], "ts": [N]}}` -- what `pwc` reads (propagation.py:289-294).
    """
    names = list(channels)
    env, shapes = pack_components([channels[n]["components"] for n in names], B=1)
    carrier = np.array([[[channels[n]["lo_freq"], channels[n].get("v_to_hz", 1.0)] for n in names]], dtype=np.float64)
    sig = synthesize_signals(env, shapes, carrier, t_start, t_end, awg_res, sim_res)
    ts = create_ts(t_start, t_end, sim_res)
    return {n: {"values": np.asarray(sig[0, k]), "ts": ts} for k, n in enumerate(names)}
