"""`time_it`: wall-clock timing of a call, logged as the reference does
(slam/utils/time_utils.py:4-16: "Execution time : name() = ...sec" into timing.log).

The log file is opened on the first timed call, not at import; SLAM_TIMING_LOG names
it (default "timing.log" in the working directory, empty string = no file, the
records still go to the "slam_amd.timing" logger).  With `drain_gpu=True` the
device is synchronised before the clock is read, so that the figure is the time of
the queued GPU work and not of its submission."""
import functools
import logging
import os
import time

_logger = logging.getLogger("slam_amd.timing")
_file_ready = False


def _ensure_file():
    global _file_ready
    if _file_ready:
        return
    _file_ready = True
    path = os.environ.get("SLAM_TIMING_LOG", "timing.log")
    if path:
        handler = logging.FileHandler(path)
        handler.setFormatter(logging.Formatter("%(asctime)-15s %(threadName)s  %(message)s"))
        _logger.addHandler(handler)
    if _logger.level == logging.NOTSET:
        _logger.setLevel(logging.DEBUG)


def _drain_gpu():
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
    except ImportError:
        pass


def time_it(func=None, *, drain_gpu=False):
    """A decorator for timing the execution time of functions: `@time_it`, or
    `@time_it(drain_gpu=True)` for a call that queues GPU work."""
    def wrap(f):
        @functools.wraps(f)
        def decorator(*args, **kwargs):
            _ensure_file()
            start_time = time.time()
            result = f(*args, **kwargs)
            if drain_gpu:
                _drain_gpu()
            end_time = time.time()
            _logger.info("Execution time : {}() = {}sec".format(f.__name__, end_time - start_time))
            return result
        return decorator
    return wrap if func is None else wrap(func)
