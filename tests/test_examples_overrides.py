"""example/*-with-temporaryThresholdOverrides.yaml on the oracle (see tests/examples_overrides.py)."""
from examples_overrides import run_examples
from scenario_runner import OracleBackend


def test_override_examples_on_oracle(oracle_mod):
    run_examples(OracleBackend(oracle_mod))
