"""Logging for vdetlib_amd.  The reference configures the ROOT logger at DEBUG on import
(utils/log.py:5-7); a library should not, so this mirror is quiet by default.  Callers that want the
reference's verbosity call ``enable_reference_logging()``.  ``logging`` is re-exported because the
reference's modules do ``from ..utils.log import logging``."""
import logging

logger = logging.getLogger("vdetlib_amd")
logger.addHandler(logging.NullHandler())

REFERENCE_FORMAT = '[%(asctime)s %(process)d %(filename)s:%(lineno)s %(levelname)s] %(message)s'


def enable_reference_logging(level=logging.DEBUG):
    logging.basicConfig(format=REFERENCE_FORMAT, level=level)
    logger.setLevel(level)
