"""Exact parsing of Kubernetes ``resource.Quantity`` text and RFC3339 instants (host-side helpers).

The engine never sees strings: the host turns every quantity into an exact integer at the
dimension's fixed decimal scale.  Grammar and rounding are restated from
k8s.io/apimachinery v0.26.4 ``pkg/api/resource`` (not present under /root/reference; the CRD repeats
the grammar as a pattern in deploy/crd.yaml:420):

    quantity  ::= [+-]? digits [. digits]? suffix   |   [+-]? . digits suffix
    suffix    ::= Ki|Mi|Gi|Ti|Pi|Ei | n|u|m|""|k|M|G|T|P|E | (e|E)[+-]?digits

Values finer than 1e-9 are rounded away from zero to a multiple of 1e-9 at parse time.
"""
from __future__ import annotations

import datetime as _dt
import re
from fractions import Fraction

_BIN = {"Ki": 10, "Mi": 20, "Gi": 30, "Ti": 40, "Pi": 50, "Ei": 60}
_DEC = {"n": -9, "u": -6, "m": -3, "": 0, "k": 3, "M": 6, "G": 9, "T": 12, "P": 15, "E": 18}
_RE = re.compile(r"^([+-]?)(\d*)(?:\.(\d*))?((?:[KMGTPE]i)|[numkMGTPE]|(?:[eE][+-]?\d+))?$")
NANO = Fraction(1, 10**9)


class QuantityError(ValueError):
    pass


DECIMAL_SI, BINARY_SI, DECIMAL_EXPONENT = "DecimalSI", "BinarySI", "DecimalExponent"


def quantity_format(text) -> str:
    """The Format a Quantity text parses to (its suffix family); steers only the canonical string."""
    if isinstance(text, int):
        return DECIMAL_SI
    m = _RE.match(str(text).strip())
    suf = (m.group(4) or "") if m else ""
    if suf in _BIN:
        return BINARY_SI
    return DECIMAL_SI if suf in _DEC else DECIMAL_EXPONENT


def parse_quantity(text) -> Fraction:
    """Exact value of a Quantity string (int/float-free)."""
    if isinstance(text, int):
        return Fraction(text)
    s = str(text).strip()
    m = _RE.match(s)
    if not m or (m.group(2) == "" and not m.group(3)):
        raise QuantityError(f"quantities must match the regular expression: {text!r}")
    sign, ip, fp, suf = m.group(1), m.group(2) or "0", m.group(3) or "", m.group(4) or ""
    num = Fraction(int(ip + fp), 10 ** len(fp))
    if suf in _BIN:
        val = num * (1 << _BIN[suf])
    elif suf in _DEC:
        val = num * Fraction(10) ** _DEC[suf]
    else:
        val = num * Fraction(10) ** int(suf[1:])
    # round away from zero to nano precision
    q = val / NANO
    if q.denominator != 1:
        q = Fraction(-(-q.numerator // q.denominator))  # ceil of the magnitude (val >= 0 here)
    val = q * NANO
    return -val if sign == "-" else val


def min_scale(values) -> int:
    """Smallest decimal exponent s in [-9, 0] such that every value is an integer multiple of 10**s."""
    s = 0
    for v in values:
        while s > -9 and (v / Fraction(10) ** s).denominator != 1:
            s -= 1
    return s


def to_scaled(value: Fraction, scale: int) -> int:
    q = value / Fraction(10) ** scale
    if q.denominator != 1:
        raise QuantityError(f"{value} is not representable at scale 1e{scale}")
    return int(q)


_SUFFIX_DEC = [(18, "E"), (15, "P"), (12, "T"), (9, "G"), (6, "M"), (3, "k"), (0, ""), (-3, "m"), (-6, "u"), (-9, "n")]


def format_decimal_si(value: Fraction) -> str:
    """Canonical DecimalSI string (largest suffix keeping an integer mantissa), e.g. 20 x 50m -> "1"."""
    if value == 0:
        return "0"
    for exp, suf in _SUFFIX_DEC:
        q = value / Fraction(10) ** exp
        if q.denominator == 1:
            return f"{int(q)}{suf}"
    raise QuantityError("finer than nano")


def format_quantity(value: Fraction, fmt: str = DECIMAL_SI) -> str:
    """``Quantity.String()`` for a value carrying Format ``fmt`` (apimachinery quantity.go CanonicalizeBytes, restated):
    BinarySI is shown as DecimalSI when |value| < 1024 or the value is not an integer, else as mantissa x 1024^k with
    every factor of 1024 moved into the suffix; DecimalSI / DecimalExponent strip the mantissa's factors of ten and then
    lower the exponent to a multiple of three (``1500`` stays "1500", ``1100m`` stays "1100m", 20 x 50m is "1")."""
    if value == 0:
        return "0"
    if fmt == BINARY_SI and (abs(value) < 1024 or value.denominator != 1):
        fmt = DECIMAL_SI
    if fmt == BINARY_SI:
        v, e = int(value), 0
        while e < 6 and v % 1024 == 0:
            v //= 1024
            e += 1
        return f"{v}{['', 'Ki', 'Mi', 'Gi', 'Ti', 'Pi', 'Ei'][e]}"
    if fmt == DECIMAL_SI:
        return format_decimal_si(value)
    q = value / NANO
    if q.denominator != 1:
        raise QuantityError("finer than nano")
    v, e = int(q), -9
    while v % 10 == 0:
        v //= 10
        e += 1
    while e % 3:
        v *= 10
        e -= 1
    return str(v) if e == 0 else f"{v}e{e}"


def add_quantities(items):
    """Fold ``Quantity.Add`` over (value, format) pairs: a zero receiver takes the addend's format."""
    total, fmt = Fraction(0), DECIMAL_SI
    for v, f in items:
        if total == 0:
            fmt = f
        total += v
    return total, fmt


_RFC3339 = re.compile(
    r"^(\d{4})-(\d{2})-(\d{2})[Tt](\d{2}):(\d{2}):(\d{2})(?:[.,](\d+))?([Zz]|[+-]\d{2}:\d{2})$")


def parse_rfc3339(text: str):
    """Go ``time.Parse(time.RFC3339, text)`` -> (unix_seconds, nanoseconds); raises ValueError.

    The empty string is handled by the caller (it means Go's zero time, temporary_threshold_override.go:33-55).
    """
    m = _RFC3339.match(text)
    if not m:
        raise ValueError(f'parsing time "{text}" as "2006-01-02T15:04:05Z07:00"')
    y, mo, d, h, mi, sec = (int(m.group(i)) for i in range(1, 7))
    frac = (m.group(7) or "")[:9].ljust(9, "0")
    tz = m.group(8)
    if not (1 <= mo <= 12 and 0 <= h <= 23 and 0 <= mi <= 59 and 0 <= sec <= 59):
        raise ValueError(f'parsing time "{text}": out of range')
    try:
        days = (_dt.date(y, mo, d) - _dt.date(1970, 1, 1)).days if y >= 1 else None
    except ValueError as e:
        raise ValueError(f'parsing time "{text}": {e}') from None
    if days is None:
        # year 0000: proleptic Gregorian, 366 days before 0001-01-01
        days = (_dt.date(4, mo, d) - _dt.date(4, 1, 1)).days - 366 + (_dt.date(1, 1, 1) - _dt.date(1970, 1, 1)).days
    off = 0
    if tz not in ("Z", "z"):
        off = (int(tz[1:3]) * 3600 + int(tz[4:6]) * 60) * (1 if tz[0] == "+" else -1)
    return days * 86400 + h * 3600 + mi * 60 + sec - off, int(frac)
