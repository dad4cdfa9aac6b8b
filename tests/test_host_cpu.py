"""Host-side C++ parsers (kube_throttler_amd/host) cross-checked against the independent Python ones
(kube_throttler_amd/quantity.py) — two implementations of the same restated apimachinery grammar."""
import os
import subprocess
import sys
from fractions import Fraction

import pytest

from kube_throttler_amd.quantity import (BINARY_SI, DECIMAL_EXPONENT, DECIMAL_SI, NANO, QuantityError, add_quantities,
                                         format_decimal_si, format_quantity, parse_quantity, parse_rfc3339,
                                         quantity_format)

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
              "9223372036854775807", "200m", "300m", "4000m", "1e-9", "1e-10", "1024Mi", "1023Ki", "2048Ki", "0.0001Ki",
              "1500e0", "5e-1", "12e4", "1e0", "0Gi", "0e5", "-1Gi", "-1500m", "1000E", "1100m", "1000", "1500", "1000000",
              "-1.5Gi", "-0.5Ki", "3e-9", "25e-4"]
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


def test_canonical_strings():
    """Quantity.String() per Format — the text UpdateStatus persists (throttle_controller.go:157-175).  Answers follow
    apimachinery v0.26.4's documented canonical form (source not on disk: parity unpinned beyond the two reference
    expectations marked below)."""
    def canon(text):
        return format_quantity(parse_quantity(text), quantity_format(text))
    table = {
        "1000m": "1",            # throttle_test.go:189: 20 x 50m is reported as "1"
        "100m": "100m",          # clusterthrottle_stress_test.go:36,81
        "1100m": "1100m", "1.1": "1100m", "1500": "1500", "1000": "1k", "1000000": "1M", "0.5": "500m",
        "1Gi": "1Gi", "1024Mi": "1Gi", "1.5Gi": "1536Mi", "0.5Ki": "512", "1023Ki": "1023Ki", "1025Ki": "1025Ki",
        "2048Ki": "2Mi", "1048576Ki": "1Gi", "0.0001Ki": "102400u",
        "1e3": "1e3", "1E3": "1e3", "1500e0": "1500", "5e-1": "500e-3", "12e4": "120e3", "1e0": "1",
        "0": "0", "0Gi": "0", "0e5": "0", "-1Gi": "-1Gi", "-1500m": "-1500m", "1E": "1E",
    }
    for text, want in table.items():
        assert canon(text) == want, text
    assert quantity_format("1Gi") == BINARY_SI and quantity_format("1E") == DECIMAL_SI
    assert quantity_format("1E3") == DECIMAL_EXPONENT and quantity_format("7") == DECIMAL_SI
    # Quantity.Add: the sum keeps the format of the addend that first made it non-zero
    mem = [(parse_quantity(t), quantity_format(t)) for t in ("0", "512Mi", "1G", "512Mi")]
    total, fmt = add_quantities(mem)
    assert fmt == BINARY_SI and format_quantity(total, fmt) == str(2 ** 30 + 10 ** 9)   # no factor of 1024 left
    total, fmt = add_quantities([(parse_quantity("512Mi"), BINARY_SI)] * 2)
    assert format_quantity(total, fmt) == "1Gi"
    total, fmt = add_quantities([(parse_quantity("50m"), DECIMAL_SI)] * 20)
    assert format_quantity(total, fmt) == "1"


def test_canonical_strings_round_trip(tool):
    """String() then Parse gives the value back, in every format, in both implementations; and the C++ text equals the
    Python text (5000 random values: nano to exa, negative, binary multiples)."""
    import random
    r = random.Random(20260921)
    texts = []
    for _ in range(5000):
        kind = r.random()
        if kind < 0.35:
            t = f"{r.randint(0, 10 ** r.randint(1, 12))}{r.choice(['n', 'u', 'm', '', 'k', 'M', 'G'])}"
        elif kind < 0.6:
            t = f"{r.randint(0, 1 << r.randint(1, 20))}{r.choice(['Ki', 'Mi', 'Gi', 'Ti'])}"
        elif kind < 0.75:
            t = f"{r.randint(0, 10 ** 6)}.{r.randint(0, 999999):06d}{r.choice(['', 'k', 'Mi', 'm'])}"
        elif kind < 0.9:
            t = f"{r.randint(0, 10 ** 9)}e{r.randint(-9, 6)}"
        else:
            t = f"-{r.randint(1, 10 ** 9)}{r.choice(['m', '', 'Ki', 'e3'])}"
        texts.append(t)
    out = subprocess.check_output([tool, "quantity"] + texts).decode().splitlines()
    assert len(out) == len(texts)
    for text, line in zip(texts, out):
        v, fmt = parse_quantity(text), quantity_format(text)
        own = format_quantity(v, fmt)
        assert parse_quantity(own) == v, (text, own)
        if fmt != BINARY_SI or own.endswith("i"):
            assert quantity_format(own) in (fmt, DECIMAL_SI), (text, own)   # a bare number reads back as DecimalSI
        assert line.split()[2] == own, (text, line, own)


def test_cpp_quantity_matches_python(tool):
    out = subprocess.check_output([tool, "quantity"] + QUANTITIES + BAD).decode().splitlines()
    assert len(out) == len(QUANTITIES) + len(BAD)
    for text, line in zip(QUANTITIES, out):
        want = parse_quantity(text) / NANO
        assert want.denominator == 1
        nano, canon, own = line.split()
        assert int(nano) == int(want), text
        if abs(want) < 10 ** 27:
            assert canon == format_decimal_si(parse_quantity(text)), text
            assert own == format_quantity(parse_quantity(text), quantity_format(text)), text
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


def test_cpp_host_unit(tool):
    """Pure host helpers in C++ (no engine call): Quantity.Add format rule, Quantity.String(), and the
    Semantic.DeepEqual stand-in behind ThrottleStatus.needsUpdate (tests/cpp/host_unit_test.cpp)."""
    subprocess.check_call(["make", "-C", HOST, "host_unit_test"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(HOST, "host_unit_test")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_index_builder_against_cpu_scan_replay(tool):
    """kt_index.cpp (the selector index the scan kernels walk) on the CPU: random programs are compiled, every chunk
    image is decoded and the device scan is replayed on the host; matches / errors must equal the brute-force
    evaluation of the program, from one resident chunk down to a few words per chunk and at the size of
    BASELINE configs[4]'s shard (tests/cpp/index_sim_test.cpp)."""
    subprocess.check_call(["make", "-C", HOST, "index_sim_test"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(HOST, "index_sim_test")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "all expectations held" in out.stdout
    # The fingerprint of everything the device gets of all these indexes.  A faster index BUILD must leave it alone (that
    # is how the round-3 rewrite of the builder's phases was accepted, together with the fingerprints of the dumped
    # BASELINE programs); a change of the index LAYOUT moves it on purpose — then re-pin it here after the GPU parity
    # tests have passed on the new layout.
    assert "fingerprint of all indexes f610dd7cf5b008e7" in out.stdout, out.stdout[-400:]
    # the builder's phases on several host threads (only programs beyond 16k throttles split by themselves): parts built
    # side by side and joined must give the same indexes
    out3 = subprocess.run([os.path.join(HOST, "index_sim_test")], capture_output=True, text=True, env=dict(os.environ, KT_INDEX_THREADS="3"))
    assert out3.returncode == 0, out3.stderr[-2000:]
    assert "fingerprint of all indexes f610dd7cf5b008e7" in out3.stdout, out3.stdout[-400:]
    # the GROUPED plan of cut_chunks (chunks per group of namespaces, words copied between groups: KT_CUT_PLAN=grouped — measured
    # on the GPU in round 6 and not the default) must describe the same matches: every throttle reported once, nothing missed
    outg = subprocess.run([os.path.join(HOST, "index_sim_test")], capture_output=True, text=True, env=dict(os.environ, KT_CUT_PLAN="grouped"))
    assert outg.returncode == 0 and "all expectations held" in outg.stdout, outg.stdout[-400:] + outg.stderr[-2000:]
    assert "fingerprint of all indexes 25dee9f5734fec01" in outg.stdout, outg.stdout[-400:]


def test_anchor_split_against_brute_force(tool, tmp_path):
    """The selector program split by anchor atom (tools/study/kt_anchor.h — groundwork of the inverted scan,
    not wired into the engine): per throttle one copy per anchor value of its terms' `In` requirements, each copy vetoing
    the earlier anchors; one index per anchor from the SAME kt::build_index.  A pod walked through the sub-indexes of
    anchor 0 and of the pairs it carries must give exactly the brute-force result of the ORIGINAL program, every throttle
    reported once — random programs (all operators, slow shapes, > 64 terms, unconvertible terms), and the real program of
    a BASELINE configs[4] shard, where it also has to pay: < 25 word visits per pod where the classic index needs 75.
    Both as separate per-anchor indexes and as ONE concatenated chunked index over virtual namespaces."""
    subprocess.check_call(["make", "-C", HOST, "index_sim_test"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(HOST, "index_sim_test"), "--anchored"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "--anchored: all expectations held" in out.stdout
    dump = str(tmp_path / "cfg4.bin")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "dump_program.py"), "--config", "4", "--pods", "1024", dump],
                          stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(HOST, "index_sim_test"), "--anchored", dump], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-1000:]
    assert "--anchored: all expectations held" in out.stdout
    import re
    m = re.search(r"word visits per pod: classic ([0-9.]+), per-anchor ([0-9.]+)", out.stdout)
    assert m and float(m.group(1)) > 60 and float(m.group(2)) < 25, out.stdout[-600:]
    # the CONCATENATED form (build_anchored_index: every sub-index built on the classic index's atom numbering, chunk c
    # serving the virtual namespaces [c * n_ns, (c + 1) * n_ns) through BmChunk::ns_base / ns_cnt — what the device would
    # get): walked item by item it reports exactly what the classic index reports, in as few word visits
    m2 = re.search(r"concatenated .*? ([0-9.]+) word visits per pod", out.stdout)
    assert m2 and abs(float(m2.group(1)) - float(m.group(2))) < 1e-9, out.stdout[-800:]


def test_word_forms_of_the_configs4_program(tool, tmp_path):
    """Inside a class the index builder numbers the groups by form (kt_index.h: kNsWord*), so that most 64-bit words of a
    rich program hold no veto bit in any atom row — the scans then read half the bytes for them.  The CPU replay checks,
    for every pod and visited word of the random suite and of the real program of a BASELINE configs[4] shard, that a word
    flagged veto-free shows no veto bit in the pod's rows and that a word flagged need-3-free gives the same matches through
    the OR / XOR accumulation as through the counting form; here: the statistic the layout exists for."""
    import re
    subprocess.check_call(["make", "-C", HOST, "index_sim_test"], stdout=subprocess.DEVNULL)
    dump = str(tmp_path / "cfg4.bin")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "dump_program.py"), "--config", "4", "--pods", "1024", dump],
                          stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(HOST, "index_sim_test"), dump], capture_output=True, text=True, env=dict(os.environ, KT_SIM_WORD_FORMS="1"))
    assert out.returncode == 0, out.stderr[-2000:] + out.stdout[-1000:]
    m = re.search(r"visited words by form: ([0-9.]+) % hold a veto bit in some row, ([0-9.]+) % a term with three positive keys", out.stdout)
    assert m, out.stdout[-800:]
    assert float(m.group(1)) < 60.0 and float(m.group(2)) < 60.0, out.stdout[-800:]   # (every word was mixed before: 100 % / 99.9 %)
    v = re.search(r"([0-9.]+) word steps per pod", out.stdout)
    assert v and float(v.group(1)) < 80.0, out.stdout[-800:]                           # (the order inside a class costs no visits)


def test_label_key_value_validation(tool):
    """What makes LabelSelectorAsSelector fail on a key / value (validation.IsQualifiedName, IsValidLabelValue of
    apimachinery v0.26.4, restated — parity unpinned: no reference test feeds a malformed label): known answers, and the
    C++ mirror agrees with the Python translation layer on all of them."""
    from kube_throttler_amd.objects import _valid_label_key, _valid_label_value
    x63, x64 = "x" * 63, "x" * 64
    dns253 = ".".join(["a" * 61] * 4) + "." + "b" * 5          # 4*61 + 4 dots + 5 = 253
    cases = {   # text: (valid key, valid value)
        "app": (1, 1), "a": (1, 1), "A": (1, 1), "a.b": (1, 1), "a-b_c.d": (1, 1), "1": (1, 1), x63: (1, 1),
        "example.com/name": (1, 0), "k8s.io/a": (1, 0), "a.b.c/x-y": (1, 0), "kubernetes.io/metadata.name": (1, 0),
        dns253 + "/n": (1, 0), "a-b.c/n": (1, 0),
        "-a": (0, 0), "a-": (0, 0), "_a": (0, 0), "a_": (0, 0), ".a": (0, 0), "a b": (0, 0), x64: (0, 0),
        "a/b/c": (0, 0), "/name": (0, 0), "example.com/": (0, 0), "Example.com/name": (0, 0), "ex_ample.com/name": (0, 0),
        "-a.com/n": (0, 0), "a-.com/n": (0, 0), "a..b/n": (0, 0), ".a.com/n": (0, 0), "a.com./n": (0, 0),
        dns253 + "b/n": (0, 0), "a.com/" + x64: (0, 0), "a.com/-n": (0, 0),
    }
    texts = list(cases)
    out = subprocess.check_output([tool, "label"] + texts).decode().splitlines()
    assert len(out) == len(texts)
    for text, line in zip(texts, out):
        want = cases[text]
        assert (int(_valid_label_key(text)), int(_valid_label_value(text))) == want, text
        assert tuple(int(v) for v in line.split()) == want, (text, line)
    assert _valid_label_value("") and not _valid_label_key("")     # the empty VALUE is fine, the empty key is not


def test_scan_replay_on_the_headline_program(tool, tmp_path):
    """The CPU scan replay on the REAL selector program of BASELINE configs[2] (1k throttles) with a pod sample
    (tools/dump_program.py -> index_sim_test <file>): the index is one chunk in the simple {any} form and small enough
    for two workgroups per CU, every sampled pod's exact match bits equal brute force."""
    import re
    import sys
    subprocess.check_call(["make", "-C", HOST, "index_sim_test"], stdout=subprocess.DEVNULL)
    dump = tmp_path / "cfg2.bin"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "dump_program.py"), "--config", "2", "--pods", "2048", str(dump)],
                          stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(HOST, "index_sim_test"), str(dump)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "1000 throttles, 1000 terms" in out.stdout and "-> 1 chunks" in out.stdout and "simple {any}" in out.stdout
    matches = float(re.search(r"([\d.]+) matches per pod", out.stdout).group(1))
    assert 1.0 < matches < 10.0
    assert int(re.search(r"LDS: check (\d+) B", out.stdout).group(1)) <= 80 * 1024


def test_scan_replay_on_the_10k_throttle_program(tool, tmp_path):
    """Same replay on the program of BASELINE configs[4] (10k throttles, ~30k multi-requirement terms, 256 namespaces):
    dozens of LDS-sized chunks at the real budgets in the rich {any, veto} form (NotIn / Exists / DoesNotExist and up to
    three positive requirements decided by the bitmaps alone), the launchers' LDS sizing holds, sampled pods match brute
    force."""
    import re
    import sys
    subprocess.check_call(["make", "-C", HOST, "index_sim_test"], stdout=subprocess.DEVNULL)
    dump = tmp_path / "cfg4.bin"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "dump_program.py"), "--config", "4", "--pods", "512", str(dump)],
                          stdout=subprocess.DEVNULL)
    # (round 6: the global plan — consecutive word ranges — with the veto columns of veto-free words elided; and the GROUPED
    #  plan, KT_CUT_PLAN=grouped: chunks per group of namespaces, more of them and far fewer per namespace)
    for budget, plan, lo, hi, per_ns in (("147000", "global", 20, 45, 9.0), ("65000", "global", 50, 100, 14.0), ("147000", "grouped", 60, 200, 5.5)):
        # (the engine's figures for an 8-dimension engine: 816 bytes of check tables per word, 40-byte packed records)
        out = subprocess.run([os.path.join(HOST, "index_sim_test"), str(dump), budget], capture_output=True, text=True,
                             env=dict(os.environ, KT_CUT_PLAN=plan, KT_SIM_CHK_WORD="816", KT_SIM_PACKED="40"))
        assert out.returncode == 0, out.stdout + out.stderr
        chunks = int(re.search(r"-> (\d+) chunks", out.stdout).group(1))
        assert "10000 throttles" in out.stdout and lo <= chunks <= hi and "rich {any, veto}" in out.stdout, out.stdout
        assert plan + " plan" in out.stdout and float(re.search(r"chunks per namespace: ([\d.]+) on average", out.stdout).group(1)) <= per_ns, out.stdout
        assert " 0 slow confirmations" in out.stdout
        matches = float(re.search(r"([\d.]+) matches per pod", out.stdout).group(1))
        assert 50.0 < matches < 500.0
