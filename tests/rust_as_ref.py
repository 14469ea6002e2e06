"""Independent Python statement of Rust `as` between primitive numeric types (the Rust reference:
"Numeric cast" semantics) used to pin both the oracle and the HIP path.  Exact integer arithmetic; float rounding is
delegated to single IEEE conversions (np.longdouble holds every 64-bit integer exactly on x86-64)."""
import math

import numpy as np

INT_BITS = {"u1": (8, False), "i1": (8, True), "u2": (16, False), "i2": (16, True), "u4": (32, False), "i4": (32, True),
            "u8": (64, False), "i8": (64, True)}


def _key(dt):
    dt = np.dtype(dt)
    return dt.kind + str(dt.itemsize) if dt.kind in "ui" else None


def rust_as_scalar(v, from_dt, to_dt):
    from_dt, to_dt = np.dtype(from_dt), np.dtype(to_dt)
    if from_dt.kind in "ui":
        iv = int(v)
        if to_dt.kind in "ui":
            bits, signed = INT_BITS[_key(to_dt)]
            iv &= (1 << bits) - 1
            if signed and iv >= 1 << (bits - 1):
                iv -= 1 << bits
            return to_dt.type(iv)
        if to_dt == np.float64:
            return np.float64(float(iv))  # Python int -> float is correctly rounded
        assert np.finfo(np.longdouble).nmant >= 63, "needs x87 long double"
        return np.float32(np.longdouble(iv))  # exact in long double, then ONE rounding to f32
    # float source
    fv = from_dt.type(v)
    if to_dt.kind == "f":
        with np.errstate(over="ignore"):
            return to_dt.type(fv)
    bits, signed = INT_BITS[_key(to_dt)]
    lo, hi = (-(1 << (bits - 1)), (1 << (bits - 1)) - 1) if signed else (0, (1 << bits) - 1)
    if np.isnan(fv):
        return to_dt.type(0)
    if np.isinf(fv):
        return to_dt.type(hi if fv > 0 else lo)
    t = math.trunc(float(fv))  # exact: every finite float is an integer multiple of a power of two
    return to_dt.type(min(max(t, lo), hi))


def rust_as_array(arr, to_dt):
    arr = np.asarray(arr)
    out = np.empty(arr.shape, dtype=to_dt)
    flat_in, flat_out = arr.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = rust_as_scalar(flat_in[i], arr.dtype, to_dt)
    return out


def edge_values(dt):
    dt = np.dtype(dt)
    if dt.kind in "ui":
        info = np.iinfo(dt)
        vals = {info.min, info.max, 0, 1, info.max - 1, info.min + 1 if info.min < 0 else 2, 127, 128, 255, 256, 511, 512, 32767, 32768,
                65535, 65536, 16777216, 16777217, 16777219, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1, 2 ** 32, 2 ** 53, 2 ** 53 + 1, 2 ** 53 + 3,
                2 ** 63 - 1, 2 ** 63, 2 ** 64 - 1, 2 ** 64 - 1025, 2 ** 24 + 2 ** 0, -1, -2, -128, -129, -32768, -32769, -2 ** 31,
                -2 ** 31 - 1, -2 ** 53 - 1, -16777217}
        return np.array(sorted(v for v in vals if info.min <= v <= info.max), dtype=dt)
    f = [0.0, -0.0, 0.5, -0.5, 0.9, -0.9, 1.0, -1.0, 1.5, 2.5, -1.5, 127.0, 127.5, 128.0, -128.0, -128.5, -129.0, 255.0, 255.9, 256.0,
         32767.9, 32768.0, -32768.9, -32769.0, 65535.9, 65536.0, 2147483647.0, 2147483647.5, 2147483648.0, -2147483648.0, -2147483648.9,
         -2147483649.0, 4294967295.0, 4294967295.9, 4294967296.0, 9007199254740992.0, 9223372036854775807.0, 9223372036854775808.0,
         -9223372036854775808.0, -9223372036854777856.0, 18446744073709551615.0, 18446744073709551616.0, 1e30, -1e30, 1e39, -1e39, 1e-40,
         1e-46, 0.1, 16777217.0, 3.4028234663852886e38, 3.4028235677973366e38, 1.401298464324817e-45, 7.0e-46, float("inf"),
         float("-inf"), float("nan"), 1.7976931348623157e308, 5e-324]
    with np.errstate(over="ignore"):
        return np.array(f, dtype=dt)
