"""Host-side C++ parsers (kube_throttler_amd/host) cross-checked against the independent Python ones
(kube_throttler_amd/quantity.py) — two implementations of the same restated apimachinery grammar."""
import os
import subprocess
from fractions import Fraction

import pytest

from kube_throttler_amd.quantity import NANO, QuantityError, format_decimal_si, parse_quantity, parse_rfc3339

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "kube_throttler_amd", "host")


@pytest.fixture(scope="module")
def tool():
    from kube_throttler_amd import engine
    engine.build()
    subprocess.check_call(["make", "-C", HOST, "kt_host_tool"], stdout=subprocess.DEVNULL)
    return os.path.join(HOST, "kt_host_tool")


QUANTITIES = ["0", "1", "500m", "100m", "1.1", "900m", "50m", "1m", "1Gi", "512Mi", "64Mi", "16Gi", "1Ki", "0.5Ki",
              "1.5Gi", "2Ti", "3Pi", "1Ei", "1k", "5M", "2G", "1T", "1P", "1E", "100n", "1u", "1e3", "1E3", "2e-3", "1.5e2",
              "-1", "-2", "-500m", "+3", ".5", "5.", "0.000000001", "0.0000000001", "0.0000000015", "123456789.123456789",
              "9223372036854775807", "200m", "300m", "4000m", "1e-9", "1e-10"]
BAD = ["", "abc", "1Xi", "1.2.3", "1e", "--1", "1 Gi", "Ki", "1ki", "1KI"]


def test_known_values():
    """Known answers for the grammar (suffix families, binary SI, rounding up beyond nano)."""
    assert parse_quantity("1Gi") == 2 ** 30
    assert parse_quantity("500m") == Fraction(1, 2)
    assert parse_quantity("1.1") == Fraction(11, 10)
    assert parse_quantity("0.5Ki") == 512
    assert parse_quantity("1E") == 10 ** 18 and parse_quantity("1e3") == 1000 and parse_quantity("1E3") == 1000
    assert parse_quantity("0.0000000001") == NANO          # rounded UP to 1n
    assert parse_quantity("-0.0000000001") == -NANO        # away from zero
    assert format_decimal_si(parse_quantity("50m") * 20) == "1"       # test/integration/throttle_test.go:189
    assert format_decimal_si(parse_quantity("1m") * 100) == "100m"    # clusterthrottle_stress_test.go:36,81
    for b in BAD:
        with pytest.raises(QuantityError):
            parse_quantity(b)


def test_cpp_quantity_matches_python(tool):
    out = subprocess.check_output([tool, "quantity"] + QUANTITIES + BAD).decode().splitlines()
    assert len(out) == len(QUANTITIES) + len(BAD)
    for text, line in zip(QUANTITIES, out):
        want = parse_quantity(text) / NANO
        assert want.denominator == 1
        nano, canon = line.split()
        assert int(nano) == int(want), text
        if abs(want) < 10 ** 27:
            assert canon == format_decimal_si(parse_quantity(text)), text
    for text, line in zip(BAD, out[len(QUANTITIES):]):
        assert line.startswith("error:"), (text, line)


TIMES = ["2026-01-01T00:00:00Z", "2019-02-01T00:00:00+09:00", "2019-03-01T00:00:00+09:00", "2006-01-02T15:04:05Z",
         "2021-08-04T10:00:00Z", "2021-08-05T10:00:00Z", "2024-02-29T23:59:59.123456789-07:30", "1970-01-01T00:00:00Z",
         "0001-01-01T00:00:00Z", "2000-12-31T23:59:59.5Z"]
BAD_TIMES = ["error", "not-time", "2021-08-04", "2021-13-01T00:00:00Z", "2021-02-29T00:00:00Z", "2021-08-04T10:00:00",
             "2021-08-04 10:00:00Z"]


def test_cpp_rfc3339_matches_python(tool):
    out = subprocess.check_output([tool, "time"] + TIMES + BAD_TIMES).decode().splitlines()
    for text, line in zip(TIMES, out):
        s, ns = parse_rfc3339(text)
        assert line.split() == [str(s), str(ns)], text
    assert parse_rfc3339("0001-01-01T00:00:00Z") == (-62135596800, 0)   # Go's zero time
    for text, line in zip(BAD_TIMES, out[len(TIMES):]):
        assert line.startswith("error:"), text
        with pytest.raises(ValueError):
            parse_rfc3339(text)
