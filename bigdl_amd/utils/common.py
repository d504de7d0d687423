"""Error helpers (reference: python/llm/src/ipex_llm/utils/common/log4Error.py:22-41)."""
import logging

logger = logging.getLogger("bigdl_amd")


def outputUserMessage(errMsg, fixMsg=None):
    logger.error(f"\n\n****************************Usage Error************************\n{errMsg}")
    if fixMsg:
        logger.error(f"\n\n**************************How to fix***********************\n{fixMsg}")
    logger.error("\n\n****************************Call Stack*************************")


def invalidInputError(condition, errMsg, fixMsg=None, cls=RuntimeError):
    if not condition:
        outputUserMessage(errMsg, fixMsg)
        raise cls(errMsg)


def invalidOperationError(condition, errMsg, fixMsg=None, cause=None):
    if not condition:
        outputUserMessage(errMsg, fixMsg)
        if cause:
            raise cause
        raise RuntimeError(errMsg)


class MuteHFLogger:
    """Context manager silencing one HF logger while weights are re-materialised."""

    def __init__(self, logger, speak_level=logging.ERROR):
        self.logger = logger
        self.speak_level = speak_level
        self.old_level = logger.level

    def __enter__(self):
        self.logger.setLevel(self.speak_level)

    def __exit__(self, exc_type, exc_value, traceback):
        self.logger.setLevel(self.old_level)
