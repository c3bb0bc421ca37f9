"""Extract golden arrays from the reference's own test fixtures into small .npz files.

Run once in the build container (reads /root/reference/test/*.pickle, which never
travels to the GPU box):

    python tests/golden/make_golden.py

The pickles are data only.  The single non-numpy global they reference is
`tensorflow.python.framework.ops.convert_to_tensor(ndarray)`; TensorFlow is not
installed here, so the unpickler maps any `tensorflow.*` global to
`numpy.asarray`.  This decodes stored arrays; it does not execute reference code.

Outputs (committed):
  two_qubit.npz          <- test/two_qubit_data.pickle   (test_two_qubits.py:46-62,193-213)
  transmon_expanded.npz  <- test/transmon_expanded.pickle (test_transmon_expanded.py:252-280)
  tunable_coupler.npz    <- test/tunable_coupler_data.pickle (test_tunable_coupler.py:393-403)
  tf_utils.npz           <- test/test_tf_utils.pickle    (test_tf_utils.py:79-111)
"""
import os
import pickle

import numpy as np

REF = "/root/reference/test"
OUT = os.path.dirname(os.path.abspath(__file__))


class _DataUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("tensorflow"):
            return lambda a, *x, **k: np.asarray(a)
        return super().find_class(module, name)


def load(name):
    with open(os.path.join(REF, name), "rb") as f:
        return _DataUnpickler(f).load()


def main():
    d = load("two_qubit_data.pickle")
    np.savez_compressed(
        os.path.join(OUT, "two_qubit.npz"),
        sig_d1=np.asarray(d["signal"]["d1"]["values"]),
        sig_d2=np.asarray(d["signal"]["d2"]["values"]),
        ts_d1=np.asarray(d["signal"]["d1"]["ts"]),
        ts_d2=np.asarray(d["signal"]["d2"]["ts"]),
        ts=np.asarray(d["ts"]),
        hdrift=np.asarray(d["hdrift"]),
        hk_d1=np.asarray(d["hks"]["d1"]),
        hk_d2=np.asarray(d["hks"]["d2"]),
        propagator=np.asarray(d["propagator"]),
        lindblad_propagator=np.asarray(d["lindblad_propagator"]),
    )

    d = load("transmon_expanded.pickle")
    np.savez_compressed(
        os.path.join(OUT, "transmon_expanded.npz"),
        **{
            k: np.asarray(d[k])
            for k in (
                "hamiltonians_q1",
                "hamiltonians_q2",
                "partial_propagators_q1",
                "partial_propagators_q2",
                "propagators_q1",
                "propagators_q2",
            )
        },
        ts_q1=np.asarray(d["signal_q1"]["ts"]),
        ts_q2=np.asarray(d["signal_q2"]["ts"]),
        sig_q1=np.asarray(d["signal_q1"]["values"]),
        sig_q2=np.asarray(d["signal_q2"]["values"]),
    )

    d = load("tunable_coupler_data.pickle")
    np.savez_compressed(
        os.path.join(OUT, "tunable_coupler.npz"),
        # every 50th dU of 10 000 is stored by the reference (200); keep every 5th of those
        dUs=np.asarray(d["dUs"])[::5],
        dU_slice_index=np.arange(0, 10000, 50)[::5],
        tc_signal=np.asarray(d["tc_signal"]),
        tc_ts=np.asarray(d["tc_ts"]),
        # AWG-resolution I/Q of the flux line (test_tunable_coupler.py:429-445): pins the flattop envelope
        tc_awg_I=np.asarray(d["tc_awg_I"]),
        tc_awg_Q=np.asarray(d["tc_awg_Q"]),
        tc_awg_ts=np.asarray(d["tc_awg_ts"]),
    )

    d = load("test_tf_utils.pickle")
    out = {}
    for key in ("tf_kron", "tf_spre", "tf_spost", "Id_like", "tf_super"):
        for i, el in enumerate(d[key]):
            if key == "tf_kron":
                out[f"{key}_{i}_inA"] = np.asarray(el["in"][0])
                out[f"{key}_{i}_inB"] = np.asarray(el["in"][1])
            else:
                out[f"{key}_{i}_in"] = np.asarray(el["in"])
            out[f"{key}_{i}_desired"] = np.asarray(el["desired"])
    np.savez_compressed(os.path.join(OUT, "tf_utils.npz"), **out)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
