"""Signal synthesis: oracle pinned to the reference's stored signals, device kernels
against the oracle."""
import os
import re

import numpy as np
import pytest

from c3_amd import signals as sg
from oracle import c3_oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWO_PI = 2 * np.pi


def two_qubit_d1_component():
    """test/conftest.py:345-385: the rx90p[0] gaussian on line d1."""
    T = 7e-9
    shift = (20e6) ** 2 / (5.6e9 - 5e9)
    return dict(shape=o.ENV_GAUSSIAN_NONORM, amp=359e-3, t_final=T, sigma=T / 4, xy_angle=0.0, freq_offset=(-50e6 - shift) * TWO_PI, use_t_before=True)


def test_oracle_reproduces_two_qubit_signal(golden_dir):
    """test/test_two_qubits.py:22-33 compares generate_signals(rx90p[0])['d1'] with the pickle."""
    g = np.load(golden_dir + "/two_qubit.npz")
    r = o.generate_signal([two_qubit_d1_component()], 5.05e9 * TWO_PI, 1e9 * TWO_PI, 0.0, 7e-9, 2e9, 100e9)
    assert np.abs(r["ts"] - g["ts"]).max() == 0.0
    assert np.abs(r["values"] - g["sig_d1"]).max() < 1e-14 * np.abs(g["sig_d1"]).max()
    r2 = o.generate_signal([dict(shape=o.ENV_NO_DRIVE, amp=0.0, t_final=7e-9)], 5.65e9 * TWO_PI, 1e9 * TWO_PI, 0.0, 7e-9, 2e9, 100e9)
    assert np.abs(r2["values"] - g["sig_d2"]).max() == 0.0


def tunable_coupler_flux_component():
    """test/test_tunable_coupler.py:232-262 (xy_angle sign flipped at :283-288)."""
    T = 100e-9
    return dict(shape=o.ENV_FLATTOP, amp=1.0, t_final=T, t_up=5e-9, t_down=T - 5e-9, risefall=5e-9, freq_offset=0.0, xy_angle=0.3590456701578104)


def test_oracle_reproduces_tunable_coupler_awg(golden_dir):
    """test/test_tunable_coupler.py:429-445: AWG inphase / quadrature of the flux line (flattop)."""
    z = np.load(golden_dir + "/tunable_coupler.npz")
    ts = o.create_ts(0.0, 100e-9, 2.4e9)
    I, Q = o.awg_iq([tunable_coupler_flux_component()], ts, 0.0)
    assert np.abs(ts - z["tc_awg_ts"]).max() < 1e-22
    assert np.abs(I - z["tc_awg_I"]).max() < 1e-15 and np.abs(Q - z["tc_awg_Q"]).max() < 1e-15


@pytest.mark.parametrize("shape", [o.ENV_GAUSSIAN_NONORM, o.ENV_FLATTOP, o.ENV_FLATTOP_RISEFALL, o.ENV_COSINE, o.ENV_GAUSSIAN_SIGMA, o.ENV_GAUSSIAN, o.ENV_TRAPEZOID])
def test_oracle_shape_derivative(shape):
    p = dict(t_final=20e-9, sigma=4e-9, t_up=3e-9, t_down=16e-9, risefall=2e-9)
    t = np.linspace(0.5e-9, 19.5e-9, 41)
    h = 1e-14
    fd = (o.envelope_shape(shape, t + h, p) - o.envelope_shape(shape, t - h, p)) / (2 * h)
    an = o.envelope_shape_der(shape, t, p)
    assert np.abs(fd - an).max() < 2e-3 * np.abs(an).max()


def test_dac_nearest_rule():
    x = np.arange(14.0)
    up = o.dac_nearest(x, 700)
    assert (up == np.repeat(x, 50)).all()
    assert (o.dac_nearest(np.arange(3.0), 7) == np.array([0, 0, 1, 1, 1, 2, 2])).all()


def test_slot_tables_match_header():
    text = open(os.path.join(ROOT, "include", "c3prop.h")).read()
    defs = {k: int(v) for k, v in re.findall(r"#define (C3P_ENVF?_[A-Z_]+) (\d+)", text)}
    for name, idx in sg.ENV_SHAPES.items():
        assert defs["C3P_ENV_" + name.upper()] == idx
    for name, idx in sg.ENV_SLOTS.items():
        assert defs["C3P_ENV_" + name.upper()] == idx
    assert defs["C3P_ENV_NPAR"] == sg.ENV_NPAR and defs["C3P_ENV_NSHAPES"] == len(sg.ENV_SHAPES)
    assert (defs["C3P_ENVF_T_BEFORE"], defs["C3P_ENVF_DRAG"]) == (sg.ENVF_T_BEFORE, sg.ENVF_DRAG)
    assert [getattr(o, "ENV_" + n.upper()) for n in sg.ENV_SHAPES] == list(sg.ENV_SHAPES.values())


def test_pack_components_errors():
    with pytest.raises(Exception, match="C3:Error"):
        sg.pack_components([[dict(shape="slepian", amp=1.0, t_final=1e-9)]])
    with pytest.raises(Exception, match="C3:Error"):
        sg.pack_components([[dict(shape="rect", amp=1.0, t_final=1e-9, wobble=2)]])
    env, shapes = sg.pack_components([[dict(shape="rect", amp=np.array([1.0, 2.0]), t_final=1e-9)], []], B=2)
    assert env.shape == (2, 2, 1, sg.ENV_NPAR) and shapes.tolist() == [[1], [-1]] and env[1, 0, 0, 0] == 2.0


# ---------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def prop(lib):
    from c3_amd import propagation, _lib

    _lib.require_gpu()
    return propagation



def _oracle_batch(channels_b, lo, v2hz, t0, t1, awg_res, sim_res):
    out, iq = [], []
    for b, channels in enumerate(channels_b):
        rows, iqs = [], []
        for k, comps in enumerate(channels):
            oc = [dict(c, shape=sg.ENV_SHAPES[c["shape"]]) for c in comps]
            r = o.generate_signal(oc, lo[b][k], v2hz[b][k], t0, t1, awg_res, sim_res)
            rows.append(r["values"])
            iqs.append(np.stack([r["inphase"], r["quadrature"]]))
        out.append(np.stack(rows))
        iq.append(np.stack(iqs))
    return np.stack(out), np.stack(iq)


def _random_problem(rng, B, T):
    names = ["gaussian_nonorm", "flattop", "flattop_risefall", "cosine", "rect", "no_drive"]
    K = 5
    # the shape table is shared by the batch; parameters vary per sample
    layout = [[names[0], names[1]], [names[2], names[3]], [names[4], names[5]], ["gaussian_sigma", "trapezoid"], ["gaussian", "gaussian_sigma"]]
    flags = [[(True, True), (False, True)], [(True, False), (False, True)], [(False, False), (False, False)],
             [(True, True), (False, True)], [(False, True), (True, False)]]
    channels_b = []
    for b in range(B):
        chans = []
        for k in range(K):
            comps = []
            for e, nm in enumerate(layout[k]):
                tb, dr = flags[k][e]
                comps.append(
                    dict(
                        shape=nm,
                        amp=rng.uniform(0.1, 0.6),
                        xy_angle=rng.uniform(-1, 3),
                        freq_offset=rng.uniform(-60e6, 60e6) * TWO_PI,
                        delta=rng.uniform(-1, 1),
                        t_final=T * rng.uniform(0.7, 1.0),
                        sigma=T * rng.uniform(0.15, 0.3),
                        t_up=T * 0.1,
                        t_down=T * rng.uniform(0.6, 0.8),
                        risefall=T * rng.uniform(0.05, 0.1),
                        delay=(T * 0.05 if e == 1 else 0.0),
                        use_t_before=tb,
                        drag=dr,
                    )
                )
            chans.append(comps)
        channels_b.append(chans)
    lo = rng.uniform(4.5e9, 6e9, size=(B, K)) * TWO_PI
    v2hz = rng.uniform(0.9e9, 1.1e9, size=(B, K)) * TWO_PI
    return channels_b, lo, v2hz


def _pack_batch(channels_b):
    B = len(channels_b)
    env0, shapes = sg.pack_components(channels_b[0], B=1)
    env = np.concatenate([sg.pack_components(c, B=1)[0] for c in channels_b], axis=0)
    assert env.shape[0] == B
    return env, shapes


@pytest.mark.gpu
@pytest.mark.parametrize("device_resident", [False, True])
def test_synthesis_vs_oracle(prop, device_resident):
    rng = np.random.default_rng(77)
    T, awg_res, sim_res = 12e-9, 2.4e9, 100e9
    channels_b, lo, v2hz = _random_problem(rng, 5, T)
    env, shapes = _pack_batch(channels_b)
    carrier = np.stack([lo, v2hz], axis=-1)
    want, want_iq = _oracle_batch(channels_b, lo, v2hz, 0.0, T, awg_res, sim_res)
    sig, iq = sg.synthesize_signals(env, shapes, carrier, 0.0, T, awg_res, sim_res, want_iq=True, device="cuda:0" if device_resident else None)
    if device_resident:
        assert sig.is_cuda
        sig, iq = sig.cpu().numpy(), iq.cpu().numpy()
    assert np.abs(iq - want_iq).max() < 1e-13 * np.abs(want_iq).max()
    assert np.abs(sig - want).max() < 1e-12 * np.abs(want).max()


def test_oracle_synthesis_vjp_matches_finite_differences():
    rng = np.random.default_rng(1)
    T = 12e-9
    comps = [
        dict(shape=o.ENV_GAUSSIAN_NONORM, amp=0.4, xy_angle=0.3, freq_offset=-50e6 * TWO_PI, delta=0.7, t_final=T, sigma=T / 4, use_t_before=True, drag=True),
        dict(shape=o.ENV_FLATTOP, amp=0.2, xy_angle=-0.5, freq_offset=20e6 * TWO_PI, delta=-0.4, t_final=T * 0.9, t_up=1e-9, t_down=8e-9, risefall=1e-9, delay=0.5e-9, drag=True),
    ]
    lo, v = 5.05e9 * TWO_PI, 1e9 * TWO_PI
    gs = rng.normal(size=int(T * 100e9))
    gr, gc = o.generate_signal_vjp(comps, lo, v, 0.0, T, 2.4e9, 100e9, gs)
    f = lambda cc, lo_=lo, v_=v: float(np.sum(gs * o.generate_signal(cc, lo_, v_, 0.0, T, 2.4e9, 100e9)["values"]))
    for e in range(2):
        for key, h in (("amp", 1e-6), ("xy_angle", 1e-6), ("freq_offset", 1e2), ("delta", 1e-6)):
            cp, cm = [dict(c) for c in comps], [dict(c) for c in comps]
            cp[e][key] += h
            cm[e][key] -= h
            fd = (f(cp) - f(cm)) / (2 * h)
            assert abs(fd - gr[e][key]) < 1e-7 * abs(fd)
    assert abs((f(comps, lo_=lo + 1e2) - f(comps, lo_=lo - 1e2)) / 2e2 - gc["lo_freq"]) < 1e-6 * abs(gc["lo_freq"])
    assert abs((f(comps, v_=v * (1 + 1e-6)) - f(comps, v_=v * (1 - 1e-6))) / (2e-6 * v) - gc["v_to_hz"]) < 1e-7 * abs(gc["v_to_hz"])


@pytest.mark.gpu
@pytest.mark.parametrize("device_resident", [False, True])
def test_synthesis_vjp_vs_oracle(prop, device_resident):
    rng = np.random.default_rng(78)
    T, awg_res, sim_res = 12e-9, 2.4e9, 100e9
    B = 4
    channels_b, lo, v2hz = _random_problem(rng, B, T)
    env, shapes = _pack_batch(channels_b)
    carrier = np.stack([lo, v2hz], axis=-1)
    N = sg.slice_num(0.0, T, sim_res)
    gs = rng.normal(size=(B, 5, N))
    if device_resident:
        import torch

        genv, gcar = sg.synthesize_signals_vjp(torch.as_tensor(env, device="cuda:0"), shapes, carrier, 0.0, T, awg_res, sim_res, torch.as_tensor(gs, device="cuda:0"))
        genv, gcar = genv.cpu().numpy(), gcar.cpu().numpy()
    else:
        genv, gcar = sg.synthesize_signals_vjp(env, shapes, carrier, 0.0, T, awg_res, sim_res, gs)
    for b in range(B):
        for k in range(5):
            oc = [dict(c, shape=sg.ENV_SHAPES[c["shape"]]) for c in channels_b[b][k]]
            want, wcar = o.generate_signal_vjp(oc, lo[b][k], v2hz[b][k], 0.0, T, awg_res, sim_res, gs[b, k])
            for e, wg in enumerate(want):
                for key in ("amp", "xy_angle", "freq_offset", "delta"):
                    scale = max(abs(wg[key]), 1e-12 * np.abs(gs).max() * v2hz[b][k])
                    assert abs(genv[b, k, e, sg.ENV_SLOTS[key]] - wg[key]) < 1e-10 * max(scale, max(abs(x) for x in wg.values()) * (1e-9 if key == "freq_offset" else 1.0)), (b, k, e, key)
            assert abs(gcar[b, k, 0] - wcar["lo_freq"]) < 1e-10 * abs(wcar["lo_freq"]) + 1e-20
            assert abs(gcar[b, k, 1] - wcar["v_to_hz"]) < 1e-10 * abs(wcar["v_to_hz"]) + 1e-20
    # untouched slots are zero
    assert np.all(genv[..., 4:] == 0.0)


@pytest.mark.gpu
def test_synthesis_golden_two_qubit(prop, golden_dir):
    g = np.load(golden_dir + "/two_qubit.npz")
    c = dict(two_qubit_d1_component(), shape="gaussian_nonorm")
    out = sg.generate_signals(
        {
            "d1": {"components": [c], "lo_freq": 5.05e9 * TWO_PI, "v_to_hz": 1e9 * TWO_PI},
            "d2": {"components": [dict(shape="no_drive", amp=0.0, t_final=7e-9)], "lo_freq": 5.65e9 * TWO_PI, "v_to_hz": 1e9 * TWO_PI},
        },
        0.0,
        7e-9,
        2e9,
        100e9,
    )
    assert np.abs(out["d1"]["ts"] - g["ts"]).max() == 0.0
    assert np.abs(out["d1"]["values"] - g["sig_d1"]).max() < 1e-12 * np.abs(g["sig_d1"]).max()
    assert np.abs(out["d2"]["values"]).max() == 0.0


@pytest.mark.gpu
def test_synthesis_feeds_propagator_zero_copy(prop, golden_dir):
    """Parameter rows -> signals in HBM -> propagators, against the reference's stored two-qubit gate
    (test/test_two_qubits.py:46-62) -- no sample array crosses PCIe."""
    import torch

    g = np.load(golden_dir + "/two_qubit.npz")
    c = dict(two_qubit_d1_component(), shape="gaussian_nonorm")
    env, shapes = sg.pack_components([[c], [dict(shape="no_drive", amp=0.0, t_final=7e-9)]], B=3)
    carrier = np.tile(np.array([[5.05e9 * TWO_PI, 1e9 * TWO_PI], [5.65e9 * TWO_PI, 1e9 * TWO_PI]]), (3, 1, 1))
    sig = sg.synthesize_signals(env, shapes, carrier, 0.0, 7e-9, 2e9, 100e9, device="cuda:0")
    dt = g["ts"][1] - g["ts"][0]
    r = prop.propagate_batch(torch.as_tensor(g["hdrift"], device="cuda:0"), torch.as_tensor(np.stack([g["hk_d1"], g["hk_d2"]]), device="cuda:0"), sig, dt)
    U = r["U"].cpu().numpy()
    for b in range(3):
        assert np.linalg.norm(U[b] - g["propagator"]) < 1e-11


@pytest.mark.gpu
def test_synthesis_errors(prop):
    env, shapes = sg.pack_components([[dict(shape="rect", amp=1.0, t_final=1e-9)]])
    with pytest.raises(Exception, match="C3:Error"):
        sg.synthesize_signals(env, shapes, np.zeros((1, 1, 2)), 0.0, 0.0, 2e9, 100e9)
    with pytest.raises(Exception, match="C3:Error"):
        sg.synthesize_signals(env, shapes, np.zeros((2, 1, 2)), 0.0, 1e-9, 2e9, 100e9)
