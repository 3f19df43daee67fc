"""ref: svg/logger.py — `logger` alias.  loguru when installed, stdlib logging otherwise (same call surface used here)."""
try:  # pragma: no cover
    from loguru import logger
except ImportError:  # pragma: no cover
    import logging

    logging.basicConfig(level=logging.INFO, format="%(asctime)s | %(levelname)s | %(message)s")
    logger = logging.getLogger("svg")
