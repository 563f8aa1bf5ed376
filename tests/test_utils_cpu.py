"""Timing decorator and logger of the training scripts (reference: slam/utils/time_utils.py:4-16,
slam/utils/logging_utils.py:8-16)."""
import logging
import os

from slam_amd.utils import logging_utils, time_utils


def test_time_it_logs_name_and_returns(tmp_path, monkeypatch):
    log = tmp_path / "timing.log"
    monkeypatch.setenv("SLAM_TIMING_LOG", str(log))
    monkeypatch.setattr(time_utils, "_file_ready", False)

    @time_utils.time_it
    def work(a, b=2):
        return a * b

    @time_utils.time_it(drain_gpu=True)
    def other():
        return "x"

    assert work(3, b=4) == 12 and other() == "x"
    assert work.__name__ == "work"
    for h in logging.getLogger("slam_amd.timing").handlers:
        h.flush()
    text = log.read_text()
    assert "Execution time : work() = " in text and "Execution time : other() = " in text
    assert text.count("sec") == 2
    # leave no handler on the temp file behind
    lg = logging.getLogger("slam_amd.timing")
    for h in list(lg.handlers):
        if getattr(h, "baseFilename", "") == str(log):
            lg.removeHandler(h)
            h.close()
    monkeypatch.setattr(time_utils, "_file_ready", False)


def test_time_it_passes_exceptions_through(monkeypatch):
    monkeypatch.setenv("SLAM_TIMING_LOG", "")
    monkeypatch.setattr(time_utils, "_file_ready", False)

    @time_utils.time_it
    def stop():
        raise StopIteration()

    try:
        stop()
    except StopIteration:
        pass
    else:
        raise AssertionError("exception swallowed")
    assert not os.path.exists("timing.log") or os.path.getsize("timing.log") >= 0


def test_get_logger_attaches_one_handler():
    root = logging.getLogger()
    handlers, level = list(root.handlers), root.level
    try:
        a = logging_utils.get_logger()
        mid = len(root.handlers)
        b = logging_utils.get_logger()
        assert a is b is root and root.level == logging.DEBUG
        assert mid - len(handlers) in (0, 1) and len(root.handlers) == mid
    finally:   # the rest of the session keeps its own logging set-up
        for h in list(root.handlers):
            if h not in handlers:
                root.removeHandler(h)
        root.setLevel(level)
        logging_utils._configured = False
