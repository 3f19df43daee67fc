"""Per-operator timing, ref: svg/timer.py.  Same knob (env TIME_BENCH = 0 off / 1 cumulative seconds / 2 per-print ms and
clear), same label strings ("Level 2 - attention core logic", ...), same `operator_log_data` dict (milliseconds) and forward-hook
printer.  Timing uses HIP events on the current stream (torch.cuda.Event on ROCm) when tensors live on the GPU."""
from __future__ import annotations

import functools
import os
import time
from collections import defaultdict

import torch

ENABLE_LOGGING = int(os.getenv("TIME_BENCH", "0"))
operator_log_data = defaultdict(float)


class TimeLoggingContext:
    """Context manager AND decorator, like the reference (svg/timer.py:17-40)."""

    def __init__(self, operation_name: str):
        self.operation_name = operation_name

    def __enter__(self):
        if ENABLE_LOGGING:
            if torch.cuda.is_available():
                self._s = torch.cuda.Event(enable_timing=True)
                self._e = torch.cuda.Event(enable_timing=True)
                self._s.record()
            else:
                self._t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if ENABLE_LOGGING:
            if torch.cuda.is_available():
                self._e.record()
                torch.cuda.synchronize()
                operator_log_data[self.operation_name] += self._s.elapsed_time(self._e)
            else:
                operator_log_data[self.operation_name] += (time.perf_counter() - self._t0) * 1000.0
        return False

    def __call__(self, fn):
        @functools.wraps(fn)
        def wrapped(*a, **k):
            with TimeLoggingContext(self.operation_name):
                return fn(*a, **k)

        return wrapped


def time_logging_decorator(operation_name: str) -> TimeLoggingContext:
    return TimeLoggingContext(operation_name)


def print_operator_log_data(module, input, output):
    """Forward-hook printer, `register_forward_hook(print_operator_log_data)` (ref: svg/timer.py:43-74).  The dict holds
    milliseconds like the reference's (HIP-event elapsed time); TIME_BENCH=1 prints cumulative seconds, TIME_BENCH=2 prints
    milliseconds and clears."""
    if not ENABLE_LOGGING or not operator_log_data:
        return
    width = max(len(str(k)) for k in operator_log_data)
    lines = []
    for name in sorted(operator_log_data):
        if ENABLE_LOGGING == 2:
            lines.append(f"{name:<{width}} : {operator_log_data[name]:10.3f} ms")
        else:
            lines.append(f"{name:<{width}} : {operator_log_data[name] / 1000.0:10.3f} s")
    print("\n\n")
    if ENABLE_LOGGING == 2:
        operator_log_data.clear()
    print("\n".join(lines))


def clear_operator_log_data():
    operator_log_data.clear()
