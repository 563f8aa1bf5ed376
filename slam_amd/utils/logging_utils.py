"""System logger of the training scripts (slam/utils/logging_utils.py:8-16): the root
logger at DEBUG with one stdout handler.  The handler is attached on the first
`get_logger()` call (the reference attaches it at import) and only once."""
import logging
import sys

_configured = False


def get_logger():
    global _configured
    logger = logging.getLogger()
    if not _configured:
        _configured = True
        handler = logging.StreamHandler(sys.stdout)
        handler.setLevel(logging.DEBUG)
        handler.setFormatter(logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s'))
        logger.addHandler(handler)
        logger.setLevel(logging.DEBUG)
    return logger
