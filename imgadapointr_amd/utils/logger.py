"""Rank-0 logging helpers (get_root_logger / get_logger / print_log of utils/logger.py)."""
import logging

import torch.distributed as dist

_initialized = {}


def get_root_logger(log_file=None, log_level=logging.INFO, name='main'):
    return get_logger(name=name, log_file=log_file, log_level=log_level)


def get_logger(name, log_file=None, log_level=logging.INFO, file_mode='w'):
    logger = logging.getLogger(name)
    if name in _initialized:
        return logger
    for known in _initialized:
        if name.startswith(known):
            return logger
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    handlers = [logging.StreamHandler()]
    if rank == 0 and log_file is not None:
        handlers.append(logging.FileHandler(log_file, file_mode))
    fmt = logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s')
    for h in handlers:
        h.setFormatter(fmt)
        h.setLevel(log_level)
        logger.addHandler(h)
    logger.setLevel(log_level if rank == 0 else logging.ERROR)
    logger.propagate = False
    _initialized[name] = True
    return logger


def print_log(msg, logger=None, level=logging.INFO):
    if logger is None:
        print(msg)
    elif isinstance(logger, logging.Logger):
        logger.log(level, msg)
    elif logger == 'silent':
        pass
    elif isinstance(logger, str):
        get_logger(logger).log(level, msg)
    else:
        raise TypeError(f'logger should be a logging.Logger, a str, "silent" or None, got {type(logger)}')
