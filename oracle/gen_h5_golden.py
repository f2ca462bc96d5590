"""Writes tests/golden/h5py_written.h5 with REAL h5py (libhdf5) the way the reference does (zoo/util.py:108-111:
`h5py.File(path, "w")`, `file[key] = array`), as the fixture tests/test_h5io.py reads with emei_amd/h5io.py.

h5py is not installed for this image's /usr/bin/python3; the image's conda environment has it:
    /opt/conda/bin/python3.9 oracle/gen_h5_golden.py
(h5py 3.3.0 on libhdf5 1.10.6 when the committed fixture was made).  The arrays are functions of their shape only
(`expected()` below, imported by the test), so the fixture needs no companion file.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "h5py_written.h5")


def expected():
    n = 37
    ramp = lambda shape, dt: (np.arange(int(np.prod(shape)), dtype=np.float64).reshape(shape) * 0.25 - 3).astype(dt)
    d = {
        "observations": ramp((n, 4), np.float32), "next_observations": ramp((n, 4), np.float32) + 1, "actions": ramp((n, 1), np.float32),
        "rewards": ramp((n,), np.float32), "dones": (np.arange(n) % 5 == 0).astype(np.float32), "timeouts": (np.arange(n) % 7 == 0).astype(np.float32),
        "extra_f64": ramp((3, 2, 2), np.float64), "extra_i64": np.arange(-2, 9, dtype=np.int64), "extra_u8": np.arange(6, dtype=np.uint8).reshape(2, 3),
        "extra_scalar": np.float64(3.5), "extra_be": ramp((5,), ">f8"), "infos/episode": np.arange(4, dtype=np.int32),
    }
    return d


if __name__ == "__main__":
    import h5py

    with h5py.File(OUT, "w") as f:
        for k, v in expected().items():
            f[k] = v  # zoo/util.py:110
    print(OUT, os.path.getsize(OUT), "bytes; h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version, file=sys.stderr)
