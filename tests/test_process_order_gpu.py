"""The order in which a process first touches the device (rounds 4-5's teardown hang): each case in a process of its
own (tests/fresh_process_case.py), with the runtime's error log on - `hsa_amd_signal_async_handler() failed to set the handler!` is
what the runtime prints when a device is switched to interrupt-driven waits while streams with spin-wait signals exist, and what
ends in a hipHostFree / hipFree that never returns."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_case(case, extra_env=None):
    env = dict(os.environ, AMD_LOG_LEVEL="1")
    env.pop("HP_BLOCKING_SYNC", None)
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(HERE, "fresh_process_case.py"), case], env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-4000:]
    assert "failed to set the handler" not in p.stderr, p.stderr[-4000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.timeout(480)
@pytest.mark.parametrize("rep", range(3))
def test_stream_after_generic_compact_sessions_on_other_threads(rep):
    """The library is the first user of the device and its first entry is NOT one that asks for the device count: the wait mode is
    decided before the first stream all the same (1 = blocking), the stream's teardown returns, every block equals the oracle's."""
    r = run_case("lib-first")
    assert r["mode_before"] == -1 and r["mode_after_generic"] == 1 and r["mode"] == 1
    assert r["generic_ok"] and r["mismatches"] == 0 and r["blocks"] > 3


@pytest.mark.timeout(480)
def test_library_after_a_host_framework_keeps_the_devices_wait_mode():
    """torch got to the device first: the library must not switch the device's wait mode under the framework's streams (mode 0)."""
    pytest.importorskip("torch")
    r = run_case("framework-first")
    assert r["mode_after_generic"] == 0 and r["mode"] == 0
    assert r["generic_ok"] and r["mismatches"] == 0


@pytest.mark.timeout(480)
def test_blocking_sync_switch_off():
    r = run_case("lib-first", {"HP_BLOCKING_SYNC": "0"})
    assert r["mode"] == 0 and r["generic_ok"] and r["mismatches"] == 0


@pytest.mark.timeout(300)
def test_the_old_order_is_what_hung():
    """HP_DEBUG_LATE_WAIT_MODE=1 puts rounds 1-5's behaviour back (the flag set by the first hp_device_count(), on a device that has
    streams by then): the runtime then reports handlers it could not set and the stream's teardown does not return. Shows that
    the cases above exercise the sequence that hung; skipped if a later runtime refuses or survives the late switch."""
    env = dict(os.environ, AMD_LOG_LEVEL="1", HP_DEBUG_LATE_WAIT_MODE="1")
    env.pop("HP_BLOCKING_SYNC", None)
    try:
        p = subprocess.run([sys.executable, os.path.join(HERE, "fresh_process_case.py"), "lib-first"], env=env, capture_output=True, text=True, timeout=100)
    except subprocess.TimeoutExpired as e:
        err = e.stderr.decode() if isinstance(e.stderr, bytes) else (e.stderr or "")
        assert "failed to set the handler" in err
        return
    if "failed to set the handler" not in p.stderr:
        pytest.skip("this runtime takes hipDeviceScheduleBlockingSync on an active device without losing completions")
