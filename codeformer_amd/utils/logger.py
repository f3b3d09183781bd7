"""Root logger used by the arch constructors (reference: basicsr/utils/logger.py:105-142 get_root_logger)."""
import logging

_initialized = set()


def get_root_logger(logger_name='basicsr', log_level=logging.INFO, log_file=None):
    logger = logging.getLogger(logger_name)
    if logger_name in _initialized:
        return logger
    fmt = '%(asctime)s %(levelname)s: %(message)s'
    handler = logging.StreamHandler()
    handler.setFormatter(logging.Formatter(fmt))
    logger.addHandler(handler)
    logger.propagate = False
    logger.setLevel(log_level)
    if log_file is not None:
        fh = logging.FileHandler(log_file, 'w')
        fh.setFormatter(logging.Formatter(fmt))
        fh.setLevel(log_level)
        logger.addHandler(fh)
    _initialized.add(logger_name)
    return logger
