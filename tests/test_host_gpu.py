"""Runs the C++ plugin mirror's test driver (tests/cpp/host_plugin_test.cpp): NewPlugin / PreFilter /
Reserve / Unreserve / ReconcileAll over the C-ABI on the GPU, with the reference's scenarios and reason strings."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "kube_throttler_amd", "host")


def test_host_plugin_scenarios():
    exe = os.path.join(HOST, "host_plugin_test")
    # always through make: a binary older than its sources must not be what gets tested
    subprocess.check_call(["make", "-C", HOST, "host_plugin_test"], stdout=subprocess.DEVNULL)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all expectations held" in r.stdout


def test_host_plugin_extended():
    """Status write-back, the pod Update / Delete handlers, and unreserveAffectedPods by the reconciled throttle's own
    affected-pod list (kt_affected_pods: a pod relabelled after Reserve keeps its reservation)."""
    exe = os.path.join(HOST, "host_plugin_test")
    subprocess.check_call(["make", "-C", HOST, "host_plugin_test"], stdout=subprocess.DEVNULL)
    r = subprocess.run([exe, "extended"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all expectations held" in r.stdout
