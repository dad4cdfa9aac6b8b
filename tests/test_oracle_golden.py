"""Pins the CPU oracle against the reference's integration scenarios."""
import pytest

from scenario_runner import OracleBackend, load_scenarios, run_scenario

SCENARIOS = load_scenarios()


@pytest.mark.parametrize("sc", SCENARIOS, ids=[s["name"] for s in SCENARIOS])
def test_oracle_scenario(sc, oracle_mod):
    run_scenario(sc, OracleBackend(oracle_mod))
